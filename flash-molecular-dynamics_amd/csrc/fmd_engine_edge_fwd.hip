// Fused engine, forward pair kernel: filter network (models/mlp.py:41-57 as built at schnet.py:813-817) and the
// CFConv message passing (schnet.py:706-715) of one interaction block over the undirected pair list; the molecule's
// aggregate stays in LDS.  Replaces K3/K4/K5 + K9 of the reference (kernels/cfconv_kernels.py:469-540,657-720,
// 780-832; csr_kernels.py:625-724) on the whole-path route.
// This file is compiled twice: as itself it holds the fp32-grade kernels, whose pair arithmetic is SCALAR
// (FMD_SCALAR_PAIRS, fmd_engine.h: packed fp32 cannot issue beside the partner wave's MFMAs, and these kernels are three
// MFMAs per product); fmd_engine_edge_fwd_half.hip includes it with FMD_HALF_TU set and holds the W16A16 kernels, one
// MFMA per product, which are faster with the PACKED pair type (same-box A/B, profiles/r05_ab_packed_half.txt).
#ifndef FMD_HALF_TU
#define FMD_SCALAR_PAIRS 1
#endif
#include "fmd_engine.h"

namespace {
// Wave-specialised pair kernels: 8 waves per workgroup, two per SIMD (waves w and w + 4 share one).
//   waves 0-3 (consumers): own the 32-feature slices -- L1 MFMAs, x1 gathers, aggregate update in LDS
//   waves 4-7 (producers): rbf values on the VALU, L0 MFMAs, tanh, hidden tile -> LDS
// Per tile two workgroup barriers:  A(t) "hidden(t) visible", B(t) "hidden(t) consumed".
struct SegRegs {              // consumer state of one 32-entry tile, register r = 4q + e <-> entry 8q+4h+e
  uint32_t oi[4];             // centre row byte offset of segment q (+ this lane's column)
  uint32_t oj[16];            // neighbour row byte offsets
  float cr[16];               // C(d) per entry
  float xi[4], xj[16];        // gathered x1[centre], x1[neighbour] (this lane's feature)
};

// =================================================================================================
// fp32-grade forward pair kernel on the fp16 MFMA (v_mfma_f32_32x32x16_f16) with split operands:
// every operand x is carried as hi = f16(x), lo = f16((x - hi) * 2048) and a product a.b is
// a_hi.b_hi + (a_hi.b_lo + a_lo.b_hi) / 2048, accumulated in fp32 (two accumulators).  The dropped
// a_lo.b_lo term is 2^-22 relative: measured error of a 128-deep contraction 2e-7, the same as the
// fp32 MFMA, at 3/16 of its MFMA cycles.  Same wave specialisation and segment scheme as
// edge_fwd_ws_kernel; the hidden tile crosses LDS as two fp16 tiles (hi, lo).
// =================================================================================================
// FS ("feature split", molecules of 284..566 beads): the aggregate of the whole molecule does not fit the LDS at 128
// features per row, so each of the molecule's TWO workgroups keeps 64 features of it (blockIdx.x = which half),
// walks ALL tiles, stages the full hidden tile (the producers' work is done twice per molecule) and runs two
// consumer waves (the other two only keep the barriers).  The slab a workgroup writes holds its half and zeros.
// SH: the tile's Gaussians are computed once per workgroup (producer w: k-step w, while it would wait for barrier B),
// and reach all producers as split fragments in LDS (see the backward kernel); used when the region fits.
// STASH: the consumers also store the filter values W = filt C of every tile (a.wst, see EdgeArgs) for the backward
// kernel of the same block, which then neither recomputes layer 1 nor needs the hidden tile.
template <int NS0, bool HALF, bool FS, bool SH, bool STASH>  // NS0: k-steps of 16 covering the rbf index; HALF: W16A16
__global__ __launch_bounds__(512, 2) void edge_fwd_split_kernel(EdgeArgs a) {
  static_assert(!(HALF && STASH), "the filter stash is an fp32-grade form");
  static_assert(NS0 <= 4 || !SH, "NS0 = 8 runs in the per-wave Gaussian form");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  _Float16* hid_hi = reinterpret_cast<_Float16*>(smem);          // [32][LDH]
  _Float16* hid_lo = hid_hi + 32 * LDH;                          // [32][LDH]
  // DB (W16A16, whole-aggregate mode): the hidden tile is ONE fp16 plane, so the region holds two of them and the
  // tile is double-buffered -- the producers write tile t+S into the plane the consumers are not reading, and ONE barrier
  // per tile ("tile t+S visible, tile t consumed") replaces the pair A / B: with a third of the MFMA work the two
  // barriers and the serial put between them were a fifth of the tile.
  constexpr bool DB = HALF && !FS;
  // DBS (W16A16 with shared Gaussians): two fragment buffers of [nb][32] x 16 B (one fp16 record per half k-step) that
  // alternate like the hidden planes.  In this precision the SIMDs are issue-bound (per tile ~450 VALU, 64
  // transcendentals, 12 MFMA: the sum of both roles' streams IS the tile), and each producer forming ALL Gaussians of
  // its pairs was a quarter of that stream.
  constexpr bool DBS = DB && SH;
  float* frag_l = smem + TILE_FLOATS + 128;                      // SH: [nb][hi, lo][32] x 16 B  (DBS: 2 x [nb][32] x 16 B)
  float* agg = smem + TILE_FLOATS + 128 + (SH ? a.nb * 256 : 0); // [n][128]
  const int tid0 = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const bool producer = wave >= 4;   // wave-uniform
  const int w = wave & 3;            // feature slice
  // The barriers of the tile loop order LDS traffic only.  __syncthreads() also releases GLOBAL stores, i.e. waits for
  // vmcnt(0): with the stash stores in flight that is an HBM write round trip per tile (measured: +46 us per launch).
  auto tile_barrier = [&]() {
    if constexpr (STASH) lds_barrier();
    else __syncthreads();   // (LDS-only barriers in the other variants: same-box A/B, no difference)
  };
  int m_own, split;
  wg_to_mol_split(m_own, split);
  // This workgroup's work items (sched_body): runs of tiles of one molecule, each with its own aggregate and slab.
  // Slot mode (a.queue == null): the two records of this workgroup.  Queue mode (batches with more items than CUs, round
  // 5): the launch has one workgroup per CU; workgroup k starts with item k and draws further items from a device
  // counter -- an item's result lands in its own slab rows whoever computes it, so the schedule does not touch the bits.
  const int lid_wg = blockIdx.y * gridDim.x + blockIdx.x, n_wg = gridDim.x * gridDim.y;
  // (the next item's index crosses the workgroup through the first word of the tile region, which nothing touches between
  // the last tile of an item and the first barrier of the next: a static __shared__ word would put the kernel over the
  // 160 KB its dynamic region may already take)
  volatile int* const next_item_p = reinterpret_cast<volatile int*>(smem);
  FMD_STAMP_DECL
  // The body is instantiated once per wave role (generic lambda): a role's weight fragments are loaded ONCE per
  // workgroup, in front of the item loop, and stay in registers for all of its items -- with the roles as the outer
  // structure they are live in their own role only (declared in front of a loop that contains both roles they would be
  // live across both and spill: see the note on the thread index below).
  auto run = [&](auto role_c) {
  constexpr bool PRODUCER = decltype(role_c)::value;
  const int lane0 = tid0 & 63, h0 = lane0 >> 5;
  f16x8 a0h[NS0], a0l[NS0];                      // producers: filter layer 0 (split), centres, bias
  f32x4 cen[SH ? 1 : 2 * NS0], b0r[4];
  f16x8 b1h[8], b1l[8];                          // consumers: filter layer 1 slice (split)
  if constexpr (PRODUCER) {
#pragma unroll
    for (int s = 0; s < NS0; ++s) {
      a0h[s] = reinterpret_cast<const f16x8*>(a.q0h)[q0_rec<NS0>(w, s) * 64 + lane0];
      if (!HALF) a0l[s] = reinterpret_cast<const f16x8*>(a.q0l)[q0_rec<NS0>(w, s) * 64 + lane0];
      if (!SH) {
        cen[2 * s] = *reinterpret_cast<const f32x4*>(a.centers + 16 * s + 8 * h0);
        cen[2 * s + 1] = *reinterpret_cast<const f32x4*>(a.centers + 16 * s + 8 * h0 + 4);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)  // bias pre-multiplied by 2 log2(e): tanh(z) = 1 - 2 / (exp2(2 log2(e) z) + 1)
      b0r[q] = *reinterpret_cast<const f32x4*>(a.b0 + 32 * w + 8 * q + 4 * h0) * (2.0f * LOG2E);
  } else if (!(FS && w >= 2)) {
    const int cw0 = FS ? 2 * split + w : w;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      b1h[s] = reinterpret_cast<const f16x8*>(a.q1h)[(cw0 * 8 + s) * 64 + lane0];
      if (!HALF) b1l[s] = reinterpret_cast<const f16x8*>(a.q1l)[(cw0 * 8 + s) * 64 + lane0];
    }
  }
  const bool queued = !FS && a.queue != nullptr;   // uniform over the launch (the feature-split mode has no queue)
  const int* const wi_base = queued ? a.items : a.items + ((int64_t)m_own * gridDim.x + split) * (2 * ITEM_INTS);
  int q_item = lid_wg;
  for (int it = 0;; ++it) {
  if (it) {
    __syncthreads();  // the flush of the item before has read the aggregate (and next_item_s is written)
    if (queued) q_item = __builtin_amdgcn_readfirstlane(*next_item_p);
  }
  if (queued ? q_item >= a.n_items : it >= 2) break;   // uniform over the workgroup
  const int* wi = wi_base + (int64_t)(queued ? 2 * q_item : it) * ITEM_INTS;
  const int m = __builtin_amdgcn_readfirstlane(wi[0]);
  if (m < 0) break;
  // Every per-lane quantity of the body derives from this opaque copy of the thread index: as plain loop invariants
  // the body's address arithmetic would be hoisted in front of the item loop and stay live across BOTH roles
  // (60 spilled registers in the backward kernel, one of them reloaded per tile on the producers' critical path).
  int tid = tid0;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, h = lane >> 5, c32 = lane & 31;
  const int64_t a0 = grp_begin(a.mol, m);
  const int n = min((int)(grp_begin(a.mol, m + 1) - a0), a.max_beads);  // LDS is sized for max_beads (pairs_kernel flags a larger molecule)
  const int np = a.npairs[m];
  constexpr int AW = FS ? 64 : F;  // aggregate row width kept in LDS
  for (int i = tid * 4; i < n * AW; i += 2048) *reinterpret_cast<f32x4*>(agg + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t* poi = a.poi + (int64_t)m * a.cap;
  const uint32_t* poj = a.poj + (int64_t)m * a.cap;
  const float* pdist = a.pd + (int64_t)m * a.cap;
  const float* pcut = a.pc + (int64_t)m * a.cap;
  const int ntiles = (np + 31) >> 5;  // of the molecule: prefetches past the item's end stay inside its list
  const int T0 = __builtin_amdgcn_readfirstlane(wi[1]);     // the item's first tile,
  const int t_end = __builtin_amdgcn_readfirstlane(wi[2]);  // end,
  const int S = __builtin_amdgcn_readfirstlane(wi[3]);      // tile stride
  const int dst_row = __builtin_amdgcn_readfirstlane(wi[4]);
  const int last_base = (ntiles - 1) * 32;
  const int drop_l0 = FMD_DROP(a, 2), drop_l1 = FMD_DROP(a, 4);  // diagnostic build: two-product variants
  (void)drop_l0; (void)drop_l1;
  __syncthreads();
  FMD_STAMP_START
  if (T0 < t_end) {  // uniform over the workgroup
    if constexpr (PRODUCER) {
      // ------------------------------------------------------------------ producer waves
      const float gl = a.gamma * LOG2E;
      float d_n = pdist[min(T0 * 32, last_base) + c32], c_n = pcut[min(T0 * 32, last_base) + c32];
      f16x4 hh[4], hl[4];  // hidden slice of the staged tile, split, 4 features per register group
      float* const frag_b0 = frag_l;                 // DBS: the two fragment buffers ...
      float* const frag_b1 = frag_l + a.nb * 128;
      const float* frag_rd_buf = frag_b0;            // ... and the one compute() reads
      (void)frag_b1; (void)frag_rd_buf;
      auto compute = [&](int t_next) {
        const float d = d_n, C = c_n, C2 = c_n * (2.0f * LOG2E);
        const int nb = min(t_next * 32, last_base);
        d_n = pdist[nb + c32];
        c_n = pcut[nb + c32];
        f32x16 acc1 = zero16();  // C is a per-pair (per-lane) scalar: applied to the sums
        if (SH) {
          // the fragments of k-step s+1 are requested before the MFMAs of k-step s
          auto frag_rd = [&](int s_, int rec) {
            const int blk = min(2 * s_ + h, a.nb - 1);  // past the last Gaussian: any finite data (zero weights)
            if constexpr (DBS) return *reinterpret_cast<const f16x8*>(frag_rd_buf + (blk * 32 + c32) * 4);
            return *reinterpret_cast<const f16x8*>(frag_l + ((blk * 2 + rec) * 32 + c32) * 4);
          };
          f16x8 bh = frag_rd(0, 0), bl = bh;
          if (!HALF) bl = frag_rd(0, 1);
          static_for<0, NS0>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            f16x8 nh = bh, nl = bl;
            if constexpr (s + 1 < NS0) {
              nh = frag_rd(s + 1, 0);
              if (!HALF) nl = frag_rd(s + 1, 1);
            }
            FMD_MERGED(HALF, a0h[s], a0l[s], bh, bl, acc1, drop_l0);
            bh = nh; bl = nl;
            __builtin_amdgcn_sched_barrier(0);
          });
        } else {
#pragma unroll
          for (int s = 0; s < NS0; ++s) {
            f16x8 bh, bl, ph, pl;
            rbf_fragments<false, HALF>(d, gl, cen[SH ? 0 : 2 * s], cen[SH ? 0 : 2 * s + 1], bh, bl, ph, pl);
            FMD_MERGED(HALF, a0h[s], a0l[s], bh, bl, acc1, drop_l0);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint32_t th[2], tl[2];
#pragma unroll
          for (int ep = 0; ep < 2; ++ep) {
            const int r = 4 * q + 2 * ep;
            // z = C (W0 raw) + b0 ; the tile carries C tanh(z): the consumers' L1 then yields W = filt C
            const f32x2 g0 = f32x2{acc1[r], acc1[r + 1]};
            const f32x2 hv = tanh_from_scaled2(splat2(C2) * g0 + f32x2{b0r[q][2 * ep], b0r[q][2 * ep + 1]});
            if (HALF) { th[ep] = half_pair(splat2(C) * hv); tl[ep] = 0; }
            else split_pair_u(splat2(C) * hv, th[ep], tl[ep]);
          }
          __builtin_memcpy(&hh[q], th, 8);
          __builtin_memcpy(&hl[q], tl, 8);
        }
        // pin the computation in front of the barrier that follows
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(hh[q]), "+v"(hl[q]));
      };
      auto put_to = [&](_Float16* plane) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          *reinterpret_cast<f16x4*>(plane + c32 * LDH + 32 * w + 8 * q + 4 * h) = hh[q];
          if (!HALF) *reinterpret_cast<f16x4*>(hid_lo + c32 * LDH + 32 * w + 8 * q + 4 * h) = hl[q];
        }
      };
      auto put = [&]() { put_to(hid_hi); };
      // SH: the Gaussians of a tile are formed ONCE per workgroup, k-step w by producer wave w, as split fragments in
      // LDS.  The consumers' chain (L1 + aggregate update) is the longer one between A and B, so the producers form
      // the fragments of the tile after next in the time they would wait for B (d_n holds that tile's distances once
      // compute() has run), keep them in eight registers and write them next to the hidden tile after B.
      f16x8 fr_h, fr_l;
      f32x4 cs0, cs1;
      const bool shares = SH && w < NS0 && 2 * w + h < a.nb;
      if (shares) {
        cs0 = *reinterpret_cast<const f32x4*>(a.centers + 16 * w + 8 * h);
        cs1 = *reinterpret_cast<const f32x4*>(a.centers + 16 * w + 8 * h + 4);
      }
      auto gauss = [&]() {  // of the tile whose distances d_n holds
        if (shares) {
          f16x8 ph, pl;
          rbf_fragments<false, HALF>(d_n, gl, cs0, cs1, fr_h, fr_l, ph, pl);
          asm volatile("" : "+v"(fr_h), "+v"(fr_l));  // formed here, not sunk behind the barrier
        }
      };
      auto put_frags = [&]() {
        if (shares) {
          const int blk = 2 * w + h;
          *reinterpret_cast<f16x8*>(frag_l + ((blk * 2 + 0) * 32 + c32) * 4) = fr_h;
          if (!HALF) *reinterpret_cast<f16x8*>(frag_l + ((blk * 2 + 1) * 32 + c32) * 4) = fr_l;
        }
      };
      auto put_frags_to = [&](float* buf) {   // DBS: one record per half k-step
        if (shares) *reinterpret_cast<f16x8*>(buf + ((2 * w + h) * 32 + c32) * 4) = fr_h;
      };
      if constexpr (DBS) {
        gauss();                 // first tile (d_n as fetched above)
        put_frags_to(frag_b0);
        tile_barrier();          // its fragments visible
        frag_rd_buf = frag_b0;
        compute(T0 + S);         // stages the first tile; d_n <- second tile
        gauss();                 // second tile
        put_frags_to(frag_b1);
      } else {
      if (SH) {
        gauss();          // first tile (d_n as fetched above)
        put_frags();
        tile_barrier();  // its fragments visible
      }
      compute(T0 + S);
      if (SH) {
        gauss();          // second tile
        tile_barrier();  // the first tile's fragments consumed
        put_frags();
      }
      }
      put();
      tile_barrier();  // A(first)
      FMD_MARK(11)
      if constexpr (DB) {
        int par = 1;
        for (int t = T0; t < t_end; t += S, par ^= 1) {
          // DBS, step i (par = (i + 1) & 1): tile t+S is staged from the fragments in buffer `par` (written one step
          // ago), the fragments of t+2S go into the other buffer, whose last reader was the compute() of the step before
          if constexpr (DBS) frag_rd_buf = par ? frag_b1 : frag_b0;
          if (!FMD_DBG_SKIP(a, 1)) {
          compute(t + 2 * S);  // tile t+S, while the consumers work on tile t ...
          if constexpr (DBS) gauss();   // tile t+2S (d_n holds its distances now)
          }
          FMD_MARK(0)
          if (!FMD_DBG_SKIP(a, 1)) {
          put_to(par ? hid_lo : hid_hi);  // ... into the plane they are not reading
          if constexpr (DBS) put_frags_to(par ? frag_b0 : frag_b1);
          }
          FMD_MARK(2)
          tile_barrier();     // tile t+S visible, tile t consumed
          FMD_MARK(3)
        }
      } else {
      for (int t = T0; t < t_end; t += S) {
        if (!FMD_DBG_SKIP(a, 1)) compute(t + 2 * S);  // tile t+S, while the consumers work on tile t
        if (SH) gauss();     // tile t+2S
        FMD_MARK(0)
        tile_barrier();     // B(t)
        FMD_MARK(1)
        put();
        if (SH) put_frags();
        FMD_MARK(2)
        tile_barrier();     // A(t+S)
        FMD_MARK(3)
      }
      }
    } else {
      // ------------------------------------------------------------------ consumer waves
      if (FS && w >= 2) {  // no feature slice for this wave in FS mode: keep the barrier protocol
        tile_barrier();  // A(first)
        for (int t = T0; t < t_end; t += S) {
          tile_barrier();  // B(t)
          tile_barrier();  // A(t+S)
        }
      } else {
      const int cw = FS ? 2 * split + w : w;  // feature slice of this wave
      // Whole-aggregate mode: a row offset o = row * 512 + column carries the aggregate's LDS address as well, so it is
      // the LDS address of the aggregate element as it stands (no add per access: 20 per tile on the consumers' chain,
      // which is the longer one here) and, against a base moved back by the same amount, the offset of the x1 gather.
      typedef __attribute__((address_space(3))) float* lds_fp;
      const uint32_t agg_lds = FS ? 0u : (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(agg);
      const uint32_t colp = (uint32_t)(32 * cw + c32) * 4u;
      const uint32_t col4 = colp + agg_lds;
      // FS: byte offset inside the LDS aggregate of the row/column a global offset o = row * 512 + col4 names
      const int lds_adj = (32 * w + c32) * 4 - (int)(colp >> 1);
      const char* x1b = reinterpret_cast<const char*>(a.x1 + a0 * F) - agg_lds;
      char* aggb = reinterpret_cast<char*>(agg);
      auto agg_rd = [&](uint32_t o) -> float {
        if (FS) return *reinterpret_cast<const float*>(aggb + (uint32_t)((int)(o >> 1) + lds_adj));
        return *(lds_fp)(uintptr_t)o;
      };
      auto agg_wr = [&](uint32_t o, float v) {
        if (FS) *reinterpret_cast<float*>(aggb + (uint32_t)((int)(o >> 1) + lds_adj)) = v;
        else *(lds_fp)(uintptr_t)o = v;
      };
      float* const wst_m = STASH ? a.wst + (int64_t)m * a.cap * F : nullptr;  // this molecule's tiles of the stash
      (void)wst_m;
      SegRegs X;
      uint32_t oj_m[16], oi_m[4];
      auto fetch_meta = [&](int t) {
        const int base = min(t * 32, last_base);
        static_for<0, 4>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          const u32x4 v = *reinterpret_cast<const u32x4*>(poj + base + 8 * q + 4 * h);
          oj_m[4 * q] = v[0]; oj_m[4 * q + 1] = v[1]; oj_m[4 * q + 2] = v[2]; oj_m[4 * q + 3] = v[3];
          oi_m[q] = poi[base + 8 * q + 4 * h];
        });
      };
      auto fill = [&]() {  // X <- fetched tile: offsets and x1 gathers
        static_for<0, 4>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          X.oi[q] = oi_m[q] + col4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            X.oj[4 * q + e] = oj_m[4 * q + e] + col4;
          }
        });
        static_for<0, 4>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          X.xi[q] = *reinterpret_cast<const float*>(x1b + X.oi[q]);
        });
        static_for<0, 16>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          X.xj[r] = *reinterpret_cast<const float*>(x1b + X.oj[r]);
        });
      };
      // aggregate update: segment q has one centre and distinct neighbours -> plain read-modify-write
      auto message = [&](const f32x16& fl) {
        static_for<0, 4>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          float Wq[4], vj[4];
          float si = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            Wq[e] = fl[4 * q + e];  // already filt * C (the hidden tile carries the cutoff)
            vj[e] = agg_rd(X.oj[4 * q + e]);
            if (!HALF) si += X.xj[4 * q + e] * Wq[e];
          }
          const float cval = agg_rd(X.oi[q]);
          if constexpr (HALF) {  // the arithmetic of the tile loop's update (same bits wherever a tile is processed)
            const f32x2 w01 = f32x2{Wq[0], Wq[1]}, w23 = f32x2{Wq[2], Wq[3]};
            const f32x2 n01 = f32x2{vj[0], vj[1]} + splat2(X.xi[q]) * w01;
            const f32x2 n23 = f32x2{vj[2], vj[3]} + splat2(X.xi[q]) * w23;
            const f32x2 s2 = f32x2{X.xj[4 * q], X.xj[4 * q + 1]} * w01 + f32x2{X.xj[4 * q + 2], X.xj[4 * q + 3]} * w23;
            agg_wr(X.oj[4 * q], n01[0]);
            agg_wr(X.oj[4 * q + 1], n01[1]);
            agg_wr(X.oj[4 * q + 2], n23[0]);
            agg_wr(X.oj[4 * q + 3], n23[1]);
            si = s2[0] + s2[1];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              agg_wr(X.oj[4 * q + e], vj[e] + X.xi[q] * Wq[e]);
          }
          si = fmd_segment_sum(si, X.oi[q]);
          // both halves store the same value to the same address
          agg_wr(X.oi[q], cval + si);
        });
      };
      // The aggregate update of a tile is deferred by one tile: it does not depend on the hidden
      // tile, so it runs (with the refill of X) while the producers stage the next tile, instead of
      // between the two barriers where the producers only write LDS.
      if (DBS) {        // the producers' fragment hand-over (see there): one more barrier before the first tile ...
        tile_barrier();
      } else if (SH) {  // ... two in the two-barrier form
        tile_barrier();
        tile_barrier();
      }
      fetch_meta(T0);
      fill();            // X <- first tile (valid rows for the first, zero-weight, update)
      f32x16 fl_prev = zero16();
      tile_barrier();  // A(first)
      FMD_MARK(11)
      int par = 0;   // DB: plane that holds tile t
      for (int t = T0; t < t_end; t += S, par ^= 1) {
        // Order inside the A-B window: the update of tile t-S first, L1 of tile t second.  After barrier A the producer
        // that shares this SIMD starts with the 12 MFMAs of the next tile's layer 0; with L1 in front the two bursts
        // met on the matrix pipe.  The update (VALU / LDS / gathers) runs under that burst, and L1 under the producer's
        // tanh arithmetic.
        // aggregate update of tile t-S (X holds its gathers) interleaved with the refill of X for tile t, segment by
        // segment: the next segment's aggregate rows are requested, then the finished segment's registers take the new
        // offsets and its gathers are issued while that LDS read is in flight (a gather costs the CU's address unit
        // ~16 cycles, an LDS round trip ~100), then the next segment is finished, ...
        if (!FMD_DBG_SKIP(a, 0)) {
          float vj[4], cval;
          auto seg_reads = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
#pragma unroll
            for (int e = 0; e < 4; ++e) vj[e] = agg_rd(X.oj[4 * q + e]);
            cval = agg_rd(X.oi[q]);
          };
          auto seg_finish = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            float si = 0.f;
            if constexpr (HALF) {  // packed pairs (this translation unit's f32x2): two fmas per product pair, (0 + 2) + (1 + 3)
              const f32x2 w01 = f32x2{fl_prev[4 * q], fl_prev[4 * q + 1]}, w23 = f32x2{fl_prev[4 * q + 2], fl_prev[4 * q + 3]};
              const f32x2 n01 = f32x2{vj[0], vj[1]} + splat2(X.xi[q]) * w01;
              const f32x2 n23 = f32x2{vj[2], vj[3]} + splat2(X.xi[q]) * w23;
              const f32x2 s2 = f32x2{X.xj[4 * q], X.xj[4 * q + 1]} * w01 + f32x2{X.xj[4 * q + 2], X.xj[4 * q + 3]} * w23;
              agg_wr(X.oj[4 * q], n01[0]);
              agg_wr(X.oj[4 * q + 1], n01[1]);
              agg_wr(X.oj[4 * q + 2], n23[0]);
              agg_wr(X.oj[4 * q + 3], n23[1]);
              si = s2[0] + s2[1];
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float Wqe = fl_prev[4 * q + e];  // already filt * C (the hidden tile carries the cutoff)
                si += X.xj[4 * q + e] * Wqe;
                agg_wr(X.oj[4 * q + e], vj[e] + X.xi[q] * Wqe);
              }
            }
            si = fmd_segment_sum(si, X.oi[q]);
            // both halves store the same value to the same address
            agg_wr(X.oi[q], cval + si);
          };
          auto seg_refill = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            X.oi[q] = oi_m[q] + col4;
#pragma unroll
            for (int e = 0; e < 4; ++e) X.oj[4 * q + e] = oj_m[4 * q + e] + col4;
            X.xi[q] = *reinterpret_cast<const float*>(x1b + X.oi[q]);
#pragma unroll
            for (int e = 0; e < 4; ++e) X.xj[4 * q + e] = *reinterpret_cast<const float*>(x1b + X.oj[4 * q + e]);
          };
          seg_reads(std::integral_constant<int, 0>{});
          static_for<0, 4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            seg_finish(qc);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (q + 1 < 4) seg_reads(std::integral_constant<int, q + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            seg_refill(qc);
            __builtin_amdgcn_sched_barrier(0);
          });
        }
        FMD_MARK(0)
        if (DB) fetch_meta(t + S);  // (the list registers are free once X is refilled; there is no B-A window in this form)
        // L1: filt slice = hidden . W1^T slice (lane = feature, registers = pairs)
        f32x16 acc1 = zero16();
        if (!FMD_DBG_SKIP(a, 3)) {
          // fragments are read two k-steps ahead of their MFMAs; the fences keep the scheduler from
          // hoisting all sixteen 16-byte reads (64 registers) to the top of the loop
          // (W16A16: no low plane -- in the double-buffered form hid_lo is the OTHER tile, which the producers are writing)
          const _Float16* ph = (DB && par ? hid_lo : hid_hi) + c32 * LDH + 8 * h;
          const _Float16* pl = hid_lo + c32 * LDH + 8 * h;
          f16x8 ah0 = *reinterpret_cast<const f16x8*>(ph), ah1 = *reinterpret_cast<const f16x8*>(ph + 16);
          f16x8 al0 = ah0, al1 = ah1;
          if constexpr (!HALF) {
            al0 = *reinterpret_cast<const f16x8*>(pl);
            al1 = *reinterpret_cast<const f16x8*>(pl + 16);
          }
          static_for<0, 8>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            f16x8 ahn = ah1, aln = al1;
            if constexpr (s + 2 < 8) {
              ahn = *reinterpret_cast<const f16x8*>(ph + 16 * (s + 2));
              if constexpr (!HALF) aln = *reinterpret_cast<const f16x8*>(pl + 16 * (s + 2));
            }
            FMD_MERGED(HALF, ah0, al0, b1h[s], b1l[s], acc1, drop_l1);
            ah0 = ah1; al0 = al1; ah1 = ahn; al1 = aln;
            __builtin_amdgcn_sched_barrier(0);
          });
        }
        FMD_MARK(1)
#pragma unroll
        for (int r = 0; r < 16; ++r) fl_prev[r] = acc1[r];
        asm volatile("" : "+v"(fl_prev));  // L1 (and its LDS reads) complete before the barrier
        // STASH: registers 4q .. 4q+3 (entries 8q + 4h .. + 3 of this lane's feature) go out as one 16-byte store each.
        // They are issued BEHIND the next tile's list loads (below): the memory counter retires in order, so a wait for
        // loads issued after the stores would also wait for an HBM write round trip.
        auto stash_tile = [&]() {
          if constexpr (STASH) {
            float* wt = wst_m + (int64_t)t * WST_TILE_FLOATS + cw * 1024 + lane * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q)  // (streamed: read once, 0.7 ms later, by the backward kernel -- not worth a cache line)
              __builtin_nontemporal_store(f32x4{fl_prev[4 * q], fl_prev[4 * q + 1], fl_prev[4 * q + 2], fl_prev[4 * q + 3]},
                                          reinterpret_cast<f32x4*>(wt + q * 256));
          }
        };
        if (DB) stash_tile();
        FMD_MARK(2)
        tile_barrier();  // B(t): hidden tiles consumed (DB: and tile t+S visible)
        FMD_MARK(3)
        if (!DB) {
          fetch_meta(t + S);  // (in the window where the producers write the next tile: the consumers have nothing else there)
          stash_tile();
          FMD_MARK(4)
          tile_barrier();  // A(t+S)
          FMD_MARK(5)
        }
      }
      message(fl_prev);  // last tile
      }
    }
  }
  __syncthreads();
  float* out = a.slabs + (int64_t)dst_row * F;
  if (FS) {  // this half's 64 features, zeros in the other half (the node kernels add the slabs)
    for (int i = tid * 4; i < n * 64; i += 2048) {
      const int row = i >> 6, c = i & 63;
      *reinterpret_cast<f32x4*>(out + row * F + 64 * split + c) = *reinterpret_cast<const f32x4*>(agg + i);
      *reinterpret_cast<f32x4*>(out + row * F + 64 * (1 - split) + c) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  } else {
    for (int i = tid * 4; i < n * F; i += 2048)
      *reinterpret_cast<f32x4*>(out + i) = *reinterpret_cast<const f32x4*>(agg + i);
  }
  if (queued && tid0 == 0) *next_item_p = (int)atomicAdd(a.queue, 1) + n_wg;   // read behind the next item's first barrier
  }  // work items
  };  // run
  if (producer) run(std::true_type{});
  else run(std::false_type{});
  FMD_STAMP_FLUSH(a, wave)
}

}  // namespace

namespace fmd_eng {

// (HALF, FS, SH); the stash variants exist in fp32-grade precision only (no stash is ever allocated for W16A16)
#define FMD_FWD_F32(M_) M_(false, false, false) M_(false, true, false) M_(false, false, true)
#define FMD_FWD_HALF(M_) M_(true, false, false) M_(true, true, false) M_(true, false, true)
#ifdef FMD_HALF_TU   // this translation unit: the W16A16 kernels, reached through the *_half entry points
constexpr bool TU_HALF = true;
#define FMD_TU(x) x##_half
#else
constexpr bool TU_HALF = false;
#define FMD_TU(x) x
#endif
int setup_edge_fwd_attrs_half();
int launch_edge_fwd_half(const EdgeArgs& a, int num_rbf, bool half, int64_t n_mol, hipStream_t s);

// the instantiations of this translation unit for (NS0, STASH), handed to M_
#define FMD_FWD_MINE(M_)                                          \
  if constexpr (TU_HALF) {                                        \
    if constexpr (!STASH) {                                       \
      if constexpr (NS0 > 4) { M_(true, false, false) M_(true, true, false) }           \
      else { FMD_FWD_HALF(M_) }                                   \
    }                                                             \
  } else if constexpr (NS0 > 4) {                                 \
    M_(false, false, false) M_(false, true, false) /* 65..128 Gaussians: per-wave Gaussian form */ \
  } else {                                                        \
    FMD_FWD_F32(M_)                                               \
  }

template <int NS0, bool STASH>
static int set_fwd_attrs() {
#define FMD_ONE(H_, FS_, SH_)                                                                                  \
  if (hipFuncSetAttribute((const void*)edge_fwd_split_kernel<NS0, H_, FS_, SH_, STASH>,                        \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)MAX_LDS_BYTES) != hipSuccess)       \
    return FMD_ELAUNCH;
  FMD_FWD_MINE(FMD_ONE)
#undef FMD_ONE
  return FMD_OK;
}
int FMD_TU(setup_edge_fwd_attrs)() {
  int rc = set_fwd_attrs<2, false>();
  if (rc == FMD_OK) rc = set_fwd_attrs<2, true>();
  if (rc == FMD_OK) rc = set_fwd_attrs<4, false>();
  if (rc == FMD_OK) rc = set_fwd_attrs<4, true>();
  if (rc == FMD_OK) rc = set_fwd_attrs<8, false>();
  if (rc == FMD_OK) rc = set_fwd_attrs<8, true>();
#ifndef FMD_HALF_TU
  if (rc == FMD_OK) rc = setup_edge_fwd_attrs_half();
#endif
  return rc;
}

template <int NS0, bool STASH>
static int launch_fwd(const EdgeArgs& a, bool half, dim3 grid, hipStream_t s) {
  // [tile region][red 128][centres 64][bias 128][aggregate rows] (the backward kernel's carve: one size for both)
  const size_t lds = (size_t)(TILE_FLOATS + 128 + 128 + MAX_RBF + 128 + a.max_beads * (a.fs ? 64 : F)) * sizeof(float);
  // shared-Gaussian form when its fragment region fits next to the aggregate
  // W16A16: [tile region = two hidden planes][two fragment buffers of nb x 512 B][aggregate rows] (the same size)
  const size_t lds_sh = (size_t)(TILE_FLOATS + 128 + a.nb * 256 + a.max_beads * F) * sizeof(float);
  const bool sh = NS0 <= 4 && !a.fs && lds_sh <= MAX_LDS_BYTES;
  if (lds > MAX_LDS_BYTES) return FMD_EUNSUPPORTED;
#define FMD_ONE(H_, FS_, SH_)                                                                     \
  if (half == H_ && a.fs == FS_ && sh == SH_) {                                                   \
    edge_fwd_split_kernel<NS0, H_, FS_, SH_, STASH><<<grid, 512, (SH_) ? lds_sh : lds, s>>>(a);  \
    return hipGetLastError() == hipSuccess ? FMD_OK : FMD_ELAUNCH;                                \
  }
  FMD_FWD_MINE(FMD_ONE)
#undef FMD_ONE
  return FMD_EUNSUPPORTED;
}

int FMD_TU(launch_edge_fwd)(const EdgeArgs& a, int num_rbf, bool half, int64_t n_mol, hipStream_t s) {
#ifndef FMD_HALF_TU
  if (half) return launch_edge_fwd_half(a, num_rbf, half, n_mol, s);
#endif
  const dim3 grid((unsigned)a.n_split, (unsigned)n_mol);
  if (a.wst && !half)
    return num_rbf <= 32 ? launch_fwd<2, true>(a, half, grid, s)
           : num_rbf <= 64 ? launch_fwd<4, true>(a, half, grid, s) : launch_fwd<8, true>(a, half, grid, s);
  return num_rbf <= 32 ? launch_fwd<2, false>(a, half, grid, s)
         : num_rbf <= 64 ? launch_fwd<4, false>(a, half, grid, s) : launch_fwd<8, false>(a, half, grid, s);
}

}  // namespace fmd_eng
