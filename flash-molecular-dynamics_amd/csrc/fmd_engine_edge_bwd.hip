// Fused engine, backward pair kernel: gradient of one interaction block's filter network and message passing with
// respect to x1 (-> gx1 aggregate) and to the pair distance (-> dE/dd, including the d/dd of the cutoff that the
// reference's Triton path drops).  Replaces K6/K7 + K10/K2 (kernels/cfconv_kernels.py:973-1035,
// 1174-1208,178-258; csr_kernels.py:302-396) and the autograd chain through them on the whole-path route.
// Compiled twice, as the forward kernel's file: as itself the fp32-grade kernels with scalar pair arithmetic;
// included by fmd_engine_edge_bwd_half.hip with FMD_HALF_TU set, the W16A16 kernels with the packed pair type.
#ifndef FMD_HALF_TU
#define FMD_SCALAR_PAIRS 1
#endif
#include "fmd_engine.h"

namespace {
// =================================================================================================
// fp32-grade backward pair kernel, split-operand fp16 MFMA, wave specialised (see edge_fwd_split_kernel).
//   producers (waves 4-7): rbf and d(rbf)/dd, L0 and z' = W0 . rbf' (12 + 12 MFMA), tanh, hidden tiles
//                          -> LDS; later u = gW . W1 (24 MFMA) and the per-pair reduction to dE/dd
//   consumers (waves 0-3): L1 recomputed (24 MFMA), gathers of x1 / d(agg) rows, the gW tile -> LDS,
//                          gx1 aggregate update in LDS
// One LDS region carries the hidden tiles (fp16 hi, lo) and then the gW tile (u32 = hi | lo << 16).
// Barriers per tile: A hidden visible, B hidden consumed, C gW visible, D gW consumed + partial sums.
// =================================================================================================
struct SegRegsB {             // consumer state of one tile (register r = 4q + e <-> entry 8q+4h+e)
  uint32_t oi[4], oj[16];
  float xi[4], xj[16], gi[4], gj[16];
};

// gW hand-over.  The consumers hold gW with the feature on the lane and the tile's pairs in registers; the u GEMM
// contracts over features, i.e. over the lane index of that layout, so the tile crosses LDS transposed
// (cdna_hip_programming.md section 3, "an accumulator tile as the next MFMA's operand"): image [feature][pair], 64-byte
// rows of 32 halves, hi plane then lo plane (GW_PLANE bytes each) in the region of the hidden tiles.  A consumer lane
// stores its registers 4q .. 4q+3 (pairs 8q + 4h .. + 3 of its feature) as one 8-byte word, the producers read B
// fragments back with ds_read_b64_tr_b16.  The 8-byte units of a row are XOR-swizzled with (row >> 1) & 7 so that the
// 16 lanes of a store group (consecutive features, same unit) hit different banks; rows 4 apart keep the transposed
// reads conflict free as they are.
constexpr int GW_PLANE = 128 * 64;
__device__ __forceinline__ int gw_row_off(int feature, int unit) {
  return feature * 64 + 8 * (unit ^ ((feature >> 1) & 7));
}
typedef __fp16 fmd_h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) fmd_h4* gw_lds_ptr;  // LDS-space pointer: constant offsets fold into the read
__device__ __forceinline__ gw_lds_ptr gw_lds(const char* p) { return (gw_lds_ptr)(uintptr_t)(uint32_t)(uintptr_t)p; }
template <int BYTES>
__device__ __forceinline__ f16x8 gw_frag(gw_lds_ptr a1, gw_lds_ptr a2) {
  const fmd_h4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(a1 + BYTES / 8);
  const fmd_h4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16(a2 + BYTES / 8);
  f16x8 r;
  __builtin_memcpy(&r, &v0, 8);
  __builtin_memcpy(reinterpret_cast<char*>(&r) + 8, &v1, 8);
  return r;
}

// FS: feature-split mode for molecules of 284..566 beads, see edge_fwd_split_kernel (each of the molecule's two
// workgroups: all tiles, 64 features of gx1, its own dE/dd slab -- the u GEMM only sees its 64 features of gW).
// STASH (blocks with a gx1 aggregate): the filter values W = filt C come from the stash the forward kernel of this block
// wrote (EdgeArgs::wst), so the consumers run no layer 1, the producers hand over no hidden tile, and the tile region
// only carries the gW image: two barriers per tile (C: gW(t) visible; A: gW(t) consumed, partial sums visible) in the
// fp32-grade kernel as well.
template <int NS0, bool NEED_GX1, bool FIRST, bool SHARE, bool HALF, bool FS, bool STASH>
__global__ __launch_bounds__(512, 2) void edge_bwd_split_kernel(EdgeArgs a) {
  static_assert(!(HALF && STASH), "the filter stash is an fp32-grade form");
  // 65..128 Gaussians (NS0 = 8): every producer forms the Gaussians of its own pairs (the shared form deals ONE k-step to
  // each of four waves, and its centre-weighted operand would not fit the register file next to eight k-steps of W0)
  static_assert(NS0 <= 4 || !SHARE, "NS0 = 8 runs in the per-wave Gaussian form");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  _Float16* hid_hi = reinterpret_cast<_Float16*>(smem);          // [32][LDH]
  _Float16* hid_lo = hid_hi + 32 * LDH;                          // [32][LDH]
  // the gW tile reuses the two fp16 tiles (hi, lo), [pair][feature]
  float* red = smem + TILE_FLOATS + 128;                         // [4][32]
  float* cen_l = red + 128;                                      // [MAX_RBF] rbf centres
  float* b0_l = cen_l + MAX_RBF;                                 // [128] filter layer-0 bias
  // SHARE: the Gaussians of a tile are computed once, by the consumers (consumer w: k-step w), and cross
  // LDS as split fragments [half k-step][hi, lo][pair] x 16 B; the producers then run W0 and
  // Wc = W0 diag(centre - c0) on the same fragments (W0 (raw (d - centre)) = (d - c0) W0 raw - Wc raw).
  float* frag_l = b0_l + 128;                                    // [nb][2][32] x 16 B   (SHARE only)
  float* gx1 = frag_l + (SHARE ? a.nb * 256 : 0);                // [n][128]
  // PIPE (block 0: no gx1 aggregate, hence no hidden tile, no L1 and an almost empty LDS): the gW image, the shared
  // fragments and the partial sums are double-buffered, and ONE barrier per tile separates "the consumers hand over
  // gW(t+1) and the Gaussians of t+2" from "the producers stage k(t+1) and run the u GEMM of t" -- instead of the four
  // phases the single tile region forces on the other blocks (the producers there wait ~15 % of the tile).
  constexpr bool PIPE = !NEED_GX1 && SHARE && !FS;
  // TWOB (W16A16, whole-aggregate mode, blocks with a gx1 aggregate): the hidden tile and the gW image are ONE fp16
  // plane each (8.5 + 8 KB), so they lie side by side in the tile region instead of sharing it, and two of the four
  // barriers go: "hidden consumed" before the gW store (B) and "gW consumed" before the next hidden tile is put (D) only
  // existed because the stores overwrote what the other role was reading.  A: hidden(t) visible, gW(t-S) consumed,
  // partial sums visible.  C: gW(t) visible, hidden(t) consumed.
  constexpr bool STASHED = NEED_GX1 && STASH;   // no hidden tile, no layer 1
  constexpr bool TWOB = HALF && NEED_GX1 && SHARE && !FS && !STASHED;
  constexpr bool TWO = TWOB || STASHED;         // the two-barrier protocol (A, C)
  char* const gw_base = reinterpret_cast<char*>(smem) + (TWOB ? 32 * LDH * 2 : 0);
  float* frag_l2 = gx1;                                          // PIPE: second fragment buffer
  float* gw_img2f = frag_l2 + a.nb * 256;                        // PIPE: second gW image (2 planes, 16 KB)
  float* red2 = gw_img2f + 4096;                                 // PIPE: second [4][32] partial sums
  const int tid0 = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const bool producer = wave >= 4;   // wave-uniform
  const int w = wave & 3;            // feature slice
  // (an LDS-only barrier, as the stash variant of the forward kernel needs, makes no difference here: same-box A/B)
  auto tile_barrier = [&]() { __syncthreads(); };
  int m_own, split;
  wg_to_mol_split(m_own, split);
  // work items, slot or queue mode, and the per-role instantiation of the body with its weights loaded once per
  // workgroup: see the forward kernel
  const int lid_wg = blockIdx.y * gridDim.x + blockIdx.x, n_wg = gridDim.x * gridDim.y;
  // (the next item's index crosses the workgroup through the first word of the tile region, which nothing touches between
  // the last tile of an item and the first barrier of the next: a static __shared__ word would put the kernel over the
  // 160 KB its dynamic region may already take)
  volatile int* const next_item_p = reinterpret_cast<volatile int*>(smem);
  FMD_STAMP_DECL
  auto run = [&](auto role_c) {
  constexpr bool PRODUCER = decltype(role_c)::value;
  constexpr int NKU = FS ? 4 : 8;          // k-steps of the u GEMM: the features of gW this workgroup holds
  const int ku0 = FS ? 4 * split : 0;
  const int lane0 = tid0 & 63;
  // producers: filter layer 0 (and its centre-weighted twin), W1^T slice of the u GEMM
  f16x8 a0h[NS0], a0l[NS0], ach[SHARE ? NS0 : 1], acl[SHARE ? NS0 : 1], wth[8], wtl[8];
  f16x8 b1h[8], b1l[8];                    // consumers: filter layer 1 slice (recomputation forms only)
  if constexpr (PRODUCER) {
#pragma unroll
    for (int s = 0; s < NS0; ++s) {
      a0h[s] = reinterpret_cast<const f16x8*>(a.q0h)[q0_rec<NS0>(w, s) * 64 + lane0];
      if (!HALF) a0l[s] = reinterpret_cast<const f16x8*>(a.q0l)[q0_rec<NS0>(w, s) * 64 + lane0];
      if constexpr (SHARE && !HALF) {
        ach[s] = reinterpret_cast<const f16x8*>(a.q0ch)[q0_rec<NS0>(w, s) * 64 + lane0];
        acl[s] = reinterpret_cast<const f16x8*>(a.q0cl)[q0_rec<NS0>(w, s) * 64 + lane0];
      }
    }
#pragma unroll
    for (int s = 0; s < NKU; ++s) {
      wth[s] = reinterpret_cast<const f16x8*>(a.q1th)[(w * 8 + ku0 + s) * 64 + lane0];
      if (!HALF) wtl[s] = reinterpret_cast<const f16x8*>(a.q1tl)[(w * 8 + ku0 + s) * 64 + lane0];
    }
  } else if constexpr (NEED_GX1 && !(NEED_GX1 && STASH)) {
    const int cw0 = FS ? 2 * split + (w & 1) : w;
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
  constexpr int AW = FS ? 64 : F;  // gx1 row width kept in LDS
  if (NEED_GX1)
    for (int i = tid * 4; i < n * AW; i += 2048) *reinterpret_cast<f32x4*>(gx1 + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid < MAX_RBF) cen_l[tid] = a.centers[tid];
  if (tid < 128) b0_l[tid] = a.b0[tid] * (2.0f * LOG2E);  // pre-multiplied: tanh(z) = 1 - 2 / (exp2(2 log2(e) z) + 1)
  const uint32_t* poi = a.poi + (int64_t)m * a.cap;
  const uint32_t* poj = a.poj + (int64_t)m * a.cap;
  const float* pdist = a.pd + (int64_t)m * a.cap;
  const float* pcut = a.pc + (int64_t)m * a.cap;
  const float* pdcut = a.pdc + (int64_t)m * a.cap;
  float* pgd = a.pgd + (int64_t)m * a.cap + (FS ? split * a.pgd_stride : 0);  // FS: one dE/dd slab per half
  const int ntiles = (np + 31) >> 5;  // of the molecule
  const int T0 = __builtin_amdgcn_readfirstlane(wi[1]);
  const int t_end = __builtin_amdgcn_readfirstlane(wi[2]);
  const int S = __builtin_amdgcn_readfirstlane(wi[3]);
  const int dst_row = __builtin_amdgcn_readfirstlane(wi[4]);
  const int last_base = (ntiles - 1) * 32;
  const int drop_l0 = FMD_DROP(a, 2), drop_l1 = FMD_DROP(a, 4);  // diagnostic build: two-product variants
  (void)drop_l0; (void)drop_l1;
  __syncthreads();
  FMD_STAMP_START
  if (T0 < t_end) {  // uniform over the workgroup
    if constexpr (PRODUCER) {
      // ------------------------------------------------------------------ producer waves
      // gW hand-over image (see gw_row_off): this lane's two read addresses of k-step 0 in the 16-lane group scheme of
      // ds_read_b64_tr_b16 -- lane 4q+p of a group addresses row (feature) f0 + q, pairs 16 g + 4p .. + 3
      const int tg = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;
      const char* gw_img = gw_base;
      const gw_lds_ptr gw_a1 = gw_lds(gw_img + gw_row_off(16 * ku0 + 8 * h + tq, 4 * tg + tp));
      const gw_lds_ptr gw_a2 = gw_lds(gw_img + gw_row_off(16 * ku0 + 8 * h + 4 + tq, 4 * tg + tp));
      // meta of the tile to stage next (lane = pair view)
      float d_n, c_n, dc_n, gd_n = 0.f;
      auto fetch = [&](int t) {
        const int base = min(t * 32, last_base);
        d_n = pdist[base + c32];
        c_n = pcut[base + c32];
        dc_n = pdcut[base + c32];
        if (!FIRST) gd_n = pgd[base + c32];
      };
      // staged tile ("nx") and tile whose reduction is pending ("cu")
      f16x4 hh[4], hl[4];
      // k = d(message weight)/dd per unit u: C (1-h^2) z' + C' h   (one array: registers are tight)
      // two sets, staged ("nx") and pending ("cu"), that swap roles from tile to tile (the tile loop is unrolled by two:
      // copying nx -> cu cost 17 moves per tile on the producers' critical path)
      struct KSet { float k[16]; float gd; };
      KSet ks_a, ks_b;
      ks_a.gd = 0.f; ks_b.gd = 0.f;
      const float gl = a.gamma * LOG2E;
      const float* frag_cur = frag_l;  // fragment buffer compute() reads (PIPE alternates it)
      auto compute = [&](int t_fetch, KSet& nx) {
        const float d = d_n, C = c_n, dC = dc_n;
        nx.gd = gd_n;
        fetch(t_fetch);
        f32x16 z1 = zero16(), p1 = zero16();
        if constexpr (SHARE) {
          // shared fragments: p accumulates Gc = Wc raw.  The fragments of k-step s+1 are requested before the MFMAs
          // of k-step s (an LDS round trip per k-step would otherwise sit in front of every group of MFMAs).
          const float* fbuf = frag_cur;
          auto frag_rd = [&](int s_, int rec) {
            const int blk = min(2 * s_ + h, a.nb - 1);  // past the last Gaussian: any finite data (zero weights)
            return *reinterpret_cast<const f16x8*>(fbuf + ((blk * 2 + rec) * 32 + c32) * 4);
          };
          f16x8 bh = frag_rd(0, 0), bl = frag_rd(0, 1);
          static_for<0, NS0>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            f16x8 nh = bh, nl = bl;
            if constexpr (s + 1 < NS0) { nh = frag_rd(s + 1, 0); nl = frag_rd(s + 1, 1); }
            if constexpr (HALF) {  // W16A16: the second record holds f16(raw (d - centre)); no cancellation-prone Wc form
              z1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h[s], bh, z1, 0, 0, 0);
              p1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h[s], bl, p1, 0, 0, 0);
            } else {
              FMD_MERGED(false, a0h[s], a0l[s], bh, bl, z1, drop_l0);
              FMD_MERGED(false, ach[s], acl[s], bh, bl, p1, drop_l0);
            }
            bh = nh; bl = nl;
            __builtin_amdgcn_sched_barrier(0);
          });
        } else {
#pragma unroll
          for (int s = 0; s < NS0; ++s) {
            // z  = C G0 + b0,            G0 = W0 raw
            // z' = 2 gamma C G1 + C' G0, G1 = W0 (raw diff)
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(cen_l + 16 * s + 8 * h);
            const f32x4 c1 = *reinterpret_cast<const f32x4*>(cen_l + 16 * s + 8 * h + 4);
            f16x8 bh, bl, ph, pl;
            rbf_fragments<true, HALF>(d, gl, c0, c1, bh, bl, ph, pl);
            FMD_MERGED(HALF, a0h[s], a0l[s], bh, bl, z1, drop_l0);
            FMD_MERGED(HALF, a0h[s], a0l[s], ph, pl, p1, drop_l0);
            __builtin_amdgcn_sched_barrier(0);  // one k-step of temporaries live at a time
          }
        }
        // z' = dz/dd = 2 gamma C G1 + C' G0; in the shared-Gaussian form G1 = (d - c0) G0 - Gc, so
        // z' = (2 gamma C (d - c0) + C') G0 - 2 gamma C Gc
        const float g2c = 2.0f * a.gamma * C, C2 = C * (2.0f * LOG2E);
        const float za = (SHARE && !HALF) ? g2c * (d - a.c0) + dC : dC;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(b0_l + 32 * w + 8 * q + 4 * h);
          uint32_t th[2], tl[2];
#pragma unroll
          for (int ep = 0; ep < 2; ++ep) {
            const int r = 4 * q + 2 * ep;
            const f32x2 g0 = f32x2{z1[r], z1[r + 1]};
            f32x2 g1 = f32x2{p1[r], p1[r + 1]};
            const f32x2 zp = (SHARE && !HALF) ? splat2(za) * g0 - splat2(g2c) * g1
                                              : splat2(g2c) * g1 + splat2(za) * g0;
            const f32x2 hv = tanh_from_scaled2(splat2(C2) * g0 + f32x2{b4[2 * ep], b4[2 * ep + 1]});
            // the tile carries C h: L1 then yields W = filt C
            const f32x2 ch = splat2(C) * hv;
            if constexpr (!NEED_GX1 || STASHED) { th[ep] = 0; tl[ep] = 0; }  // no L1 (block 0, stash form): no hidden tile
            else if (HALF) { th[ep] = half_pair(ch); tl[ep] = 0; }
            else split_pair_u(ch, th[ep], tl[ep]);
            // k = d(C h)/dd = C (1 - h^2) z' + C' h = (C - (C h) h) z' + C' h
            const f32x2 kk = (splat2(C) - ch * hv) * zp + splat2(dC) * hv;
            nx.k[r] = kk[0];
            nx.k[r + 1] = kk[1];
          }
          __builtin_memcpy(&hh[q], th, 8);
          __builtin_memcpy(&hl[q], tl, 8);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(hh[q]), "+v"(hl[q]));
        // ... and the k array too: left alone, the compiler sinks its arithmetic behind the reduction
        // and carries both accumulators of G1 and the hidden values (48 registers) across it instead
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(nx.k[r]));
      };
      auto put = [&]() {
        if constexpr (NEED_GX1 && !STASHED) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<f16x4*>(hid_hi + c32 * LDH + 32 * w + 8 * q + 4 * h) = hh[q];
            if (!HALF) *reinterpret_cast<f16x4*>(hid_lo + c32 * LDH + 32 * w + 8 * q + 4 * h) = hl[q];
          }
        }
      };
      // u^T slice = W1^T slice . gW^T and the reduction against k: partial sums of this wave's 32 hidden features
      auto u_gemm_reduce = [&](gw_lds_ptr g1, gw_lds_ptr g2, const KSet& cu) -> float {
        f32x16 u1 = zero16(), u2 = zero16();
        f16x8 bh = gw_frag<0>(g1, g2), bl = bh;
        if (!HALF) bl = gw_frag<GW_PLANE>(g1, g2);
        static_for<0, NKU>([&](auto sc) {
          constexpr int s = decltype(sc)::value;
          f16x8 nh = bh, nl = bl;
          if constexpr (s + 1 < NKU) {
            nh = gw_frag<1024 * (s + 1)>(g1, g2);
            if (!HALF) nl = gw_frag<GW_PLANE + 1024 * (s + 1)>(g1, g2);
          }
          mfma_prod<HALF>(wth[s], wtl[s], bh, bl, u1, u2);
          bh = nh; bl = nl;
          __builtin_amdgcn_sched_barrier(0);
        });
        float ta = 0.f, tb = 0.f, tc = 0.f, td = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          ta = __builtin_fmaf(u1[r], cu.k[r], ta);
          tb = __builtin_fmaf(u1[r + 1], cu.k[r + 1], tb);
          if (!HALF) {
            tc = __builtin_fmaf(u2[r], cu.k[r], tc);
            td = __builtin_fmaf(u2[r + 1], cu.k[r + 1], td);
          }
        }
        return fmd_xor32_sum((ta + tb) + (tc + td) * (1.0f / 2048.0f));
      };
      fetch(T0);
      if constexpr (PIPE) {
        // step i (tile t): k(t+S) from the fragments the consumers wrote two steps ago, u GEMM of t from the image they
        // wrote in the step before, partial sums -> red[i & 1]; after the step's barrier wave 0 adds them up.
        const gw_lds_ptr gw_b1 = gw_a1 + (reinterpret_cast<const char*>(gw_img2f) - reinterpret_cast<const char*>(hid_hi)) / 8;
        const gw_lds_ptr gw_b2 = gw_a2 + (reinterpret_cast<const char*>(gw_img2f) - reinterpret_cast<const char*>(hid_hi)) / 8;
        tile_barrier();  // P0: fragments of the first two tiles and gW of the first tile visible
        frag_cur = frag_l;
        compute(T0 + S, ks_a);  // k of the first tile
        tile_barrier();  // P1: the first tile's fragments consumed (the consumers replace them in step 0)
        FMD_MARK(11)
        auto ptile = [&](int t, auto odd, KSet& cu, KSet& nx) {
          constexpr bool ODD = decltype(odd)::value;
          frag_cur = ODD ? frag_l : frag_l2;  // tile t+S
          if (!FMD_DBG_SKIP(a, 1)) compute(t + 2 * S, nx);
          FMD_MARK(0)
          const float tot = ODD ? u_gemm_reduce(gw_b1, gw_b2, cu) : u_gemm_reduce(gw_a1, gw_a2, cu);
          float* rbuf = ODD ? red2 : red;
          if (h == 0) rbuf[w * 32 + c32] = tot;
          FMD_MARK(3)
          tile_barrier();   // the step's only barrier
          FMD_MARK(4)
          if (w == 0 && h == 0) {
            const float sum = ((rbuf[c32] + rbuf[32 + c32]) + rbuf[64 + c32]) + rbuf[96 + c32];
            pgd[t * 32 + c32] = FIRST ? sum : cu.gd + sum;
          }
        };
        for (int t = T0; t < t_end; t += 2 * S) {
          ptile(t, std::false_type{}, ks_a, ks_b);
          if (t + S >= t_end) break;
          ptile(t + S, std::true_type{}, ks_b, ks_a);
        }
      } else {
      if (SHARE) tile_barrier();  // fragments of the first tile visible
      compute(T0 + S, ks_a);
      if (SHARE) tile_barrier();  // ... and consumed: the consumers write the second tile's
      put();
      tile_barrier();  // A(first)
      FMD_MARK(11)
      auto tile = [&](int t, KSet& cu, KSet& nx) {
        if (!FMD_DBG_SKIP(a, 1)) compute(t + 2 * S, nx);  // stages tile t+S while the consumers contract tile t
        FMD_MARK(0)
        if (!TWO) tile_barrier();     // B(t)
        FMD_MARK(1)
        tile_barrier();     // C(t): gW(t) visible (TWOB: and hidden(t) consumed)
        FMD_MARK(2)
        if (TWOB) put();     // hidden tile of t+S into its own plane, which the consumers have left
        // u^T slice = W1^T slice . gW^T  (lane = pair, registers = hidden features); dE/dd = C sum_h u (1-h^2) z' + C' sum_h u h
        const float tot = u_gemm_reduce(gw_a1, gw_a2, cu);
        if (h == 0) red[w * 32 + c32] = tot;
        FMD_MARK(3)
        if (!TWO) {
          tile_barrier();     // D(t): gW consumed, partial sums visible
          FMD_MARK(4)
          if (w == 0 && h == 0) {
            const float sum = ((red[c32] + red[32 + c32]) + red[64 + c32]) + red[96 + c32];
            pgd[t * 32 + c32] = FIRST ? sum : cu.gd + sum;
          }
          put();               // hidden tiles of t+S
        }
        FMD_MARK(5)
        tile_barrier();     // A(t+S) (two-barrier forms: also gW(t) consumed, partial sums visible)
        FMD_MARK(6)
        if (TWO && w == 0 && h == 0) {   // (red is next written after C(t+S), which this wave reaches after the read)
          const float sum = ((red[c32] + red[32 + c32]) + red[64 + c32]) + red[96 + c32];
          pgd[t * 32 + c32] = FIRST ? sum : cu.gd + sum;
        }
      };
      for (int t = T0; t < t_end; t += 2 * S) {
        tile(t, ks_a, ks_b);
        if (t + S >= t_end) break;
        tile(t + S, ks_b, ks_a);
      }
      }  // !PIPE
    } else {
      // ------------------------------------------------------------------ consumer waves
      const bool slice_owner = !FS || w < 2;  // FS: waves 2, 3 only share Gaussians and keep the barriers
      const int cw = FS ? 2 * split + (w & 1) : w;
      // STASHED: this lane's filter values of a tile, as the forward kernel's consumers stored them
      const float* const wst_m = STASHED ? a.wst + (int64_t)m * a.cap * F + cw * 1024 + lane * 4 : nullptr;
      f32x16 fl_next = zero16();
      auto fetch_w = [&](int t) {
        if constexpr (STASHED) {
          const float* wt = wst_m + (int64_t)min(t, ntiles - 1) * WST_TILE_FLOATS;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wt + q * 256));
            fl_next[4 * q] = v[0]; fl_next[4 * q + 1] = v[1]; fl_next[4 * q + 2] = v[2]; fl_next[4 * q + 3] = v[3];
          }
        }
      };
      const int col = 32 * cw + c32;
      // whole-aggregate mode: the row offsets carry the LDS address of the gx1 aggregate (see the forward kernel)
      typedef __attribute__((address_space(3))) float* lds_fp;
      const uint32_t agg_lds = (FS || !NEED_GX1) ? 0u : (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(gx1);
      const uint32_t colp = (uint32_t)col * 4u;
      const uint32_t col4 = colp + agg_lds;
      const int lds_adj = (32 * (w & 1) + c32) * 4 - (int)(colp >> 1);
      const char* x1b = reinterpret_cast<const char*>(a.x1 + a0 * F) - agg_lds;
      const char* gab = reinterpret_cast<const char*>(a.gagg + a0 * F) - agg_lds;
      char* gx1b = reinterpret_cast<char*>(gx1);
      auto agg_rd = [&](uint32_t o) -> float {
        if (FS) return *reinterpret_cast<const float*>(gx1b + (uint32_t)((int)(o >> 1) + lds_adj));
        return *(lds_fp)(uintptr_t)o;
      };
      auto agg_wr = [&](uint32_t o, float v) {
        if (FS) *reinterpret_cast<float*>(gx1b + (uint32_t)((int)(o >> 1) + lds_adj)) = v;
        else *(lds_fp)(uintptr_t)o = v;
      };
      SegRegsB X;
      uint32_t oj_m[16], oi_m[4];
      float d_m = 0.f;  // SHARE: distance of pair c32 of the fetched tile
      auto fetch_meta = [&](int t) {
        const int base = min(t * 32, last_base);
        static_for<0, 4>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          const u32x4 v = *reinterpret_cast<const u32x4*>(poj + base + 8 * q + 4 * h);
          oj_m[4 * q] = v[0]; oj_m[4 * q + 1] = v[1]; oj_m[4 * q + 2] = v[2]; oj_m[4 * q + 3] = v[3];
          oi_m[q] = poi[base + 8 * q + 4 * h];
        });
      };
      auto fill = [&]() {  // X <- fetched tile: offsets, gathers of x1 and d/d(agg) rows
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
          X.gi[q] = *reinterpret_cast<const float*>(gab + X.oi[q]);
        });
        static_for<0, 16>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          X.xj[r] = *reinterpret_cast<const float*>(x1b + X.oj[r]);
          X.gj[r] = *reinterpret_cast<const float*>(gab + X.oj[r]);
        });
      };
      // SHARE: Gaussians of one tile, k-step w of this wave, as split fragments in LDS
      const float gl = a.gamma * LOG2E;
      auto share_fragments_to = [&](float* fbuf) {  // of the tile whose distances were fetched last
        if (w < NS0 && 2 * w + h < a.nb) {
          const float d = d_m;
          const f32x4 c0 = *reinterpret_cast<const f32x4*>(cen_l + 16 * w + 8 * h);
          const f32x4 c1 = *reinterpret_cast<const f32x4*>(cen_l + 16 * w + 8 * h + 4);
          f16x8 bh, bl, ph, pl;
          rbf_fragments<HALF, HALF>(d, gl, c0, c1, bh, bl, ph, pl);
          const int blk = 2 * w + h;
          *reinterpret_cast<f16x8*>(fbuf + ((blk * 2 + 0) * 32 + c32) * 4) = bh;
          *reinterpret_cast<f16x8*>(fbuf + ((blk * 2 + 1) * 32 + c32) * 4) = HALF ? ph : bl;
        }
      };
      auto share_fragments = [&]() { share_fragments_to(frag_l); };
      // Schedule of one tile (consumer side), between A and B: gW from the rows gathered one tile ago, L1, then the
      // aggregate update interleaved with the gathers of the next tile's rows (they need ~2-3 thousand cycles to
      // arrive under load: issued here they land while the producers run the u GEMM and stage the next tile).
      auto fetch_d = [&](int t) {
        if (SHARE) d_m = pdist[min(t * 32, last_base) + c32];
      };
      // gW of the tile X holds.  It is formed right after the refill's gathers have landed -- between C and D of the
      // PREVIOUS tile, where the consumers would otherwise wait for the producers' u GEMM -- and stored after B.
      f16x2 gwh[8], gwl[8];
      auto form_gw = [&]() {
        // gW = x1[i] g[j] + x1[j] g[i], split two values at a time: gwh[p] = hi halves of entries 2p, 2p+1,
        // gwl[p] = their lo halves
#pragma unroll
        for (int p2 = 0; p2 < 8; ++p2) {
          const int r = 2 * p2;
          const f32x2 gw = f32x2{X.xi[r >> 2], X.xi[r >> 2]} * f32x2{X.gj[r], X.gj[r + 1]} +
                           f32x2{X.xj[r], X.xj[r + 1]} * f32x2{X.gi[r >> 2], X.gi[r >> 2]};
          uint32_t hw, lw = 0;
          if (HALF) hw = half_pair(gw);
          else split_pair(gw, hw, lw);
          __builtin_memcpy(&gwh[p2], &hw, 4);
          __builtin_memcpy(&gwl[p2], &lw, 4);
        }
#pragma unroll
        for (int p2 = 0; p2 < 8; ++p2) asm volatile("" : "+v"(gwh[p2]), "+v"(gwl[p2]));  // ahead of the MFMAs
      };
      auto gw_store = [&](char* img) {  // pairs 8q + 4h .. + 3 of this lane's feature: one 8-byte store per plane
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          char* dst = img + gw_row_off(col, 2 * q + h);
          uint32_t w0, w1;
          __builtin_memcpy(&w0, &gwh[2 * q], 4);
          __builtin_memcpy(&w1, &gwh[2 * q + 1], 4);
          *reinterpret_cast<uint2*>(dst) = uint2{w0, w1};
          if (!HALF) {
            __builtin_memcpy(&w0, &gwl[2 * q], 4);
            __builtin_memcpy(&w1, &gwl[2 * q + 1], 4);
            *reinterpret_cast<uint2*>(dst + GW_PLANE) = uint2{w0, w1};
          }
        }
      };
      if constexpr (PIPE) {
        // the consumers' side of the pipelined form (see PIPE above): per step gW of the NEXT tile into the image the
        // producers are not reading, the Gaussians of the tile after that into the fragment buffer they are not
        // reading, then the row gathers of that tile
        char* img0 = reinterpret_cast<char*>(hid_hi);
        char* img1 = reinterpret_cast<char*>(gw_img2f);
        fetch_d(T0);
        fetch_meta(T0);
        fill();
        share_fragments_to(frag_l);    // first tile
        fetch_d(T0 + S);
        share_fragments_to(frag_l2);   // second tile
        form_gw();
        gw_store(img0);                // gW of the first tile
        fetch_meta(T0 + S);
        fill();                        // X <- second tile
        fetch_meta(T0 + 2 * S);
        fetch_d(T0 + 2 * S);
        tile_barrier();  // P0
        tile_barrier();  // P1
        FMD_MARK(11)
        bool odd = false;
        for (int t = T0; t < t_end; t += S, odd = !odd) {
          form_gw();                                   // tile t+S
          gw_store(odd ? img0 : img1);
          share_fragments_to(odd ? frag_l2 : frag_l);  // tile t+2S
          FMD_MARK(7)
          fill();                                      // X <- tile t+2S
          fetch_meta(t + 3 * S);
          fetch_d(t + 3 * S);
          FMD_MARK(5)
          tile_barrier();  // the step's only barrier
          FMD_MARK(8)
        }
      } else {
      fetch_d(T0);
      fetch_meta(T0);
      fill();
      if (SHARE) share_fragments();
      fetch_d(T0 + S);
      fetch_meta(T0 + S);
      if (SHARE) {
        tile_barrier();  // first tile's fragments visible
        tile_barrier();  // ... and consumed
        share_fragments();
      }
      form_gw();  // first tile
      if (slice_owner) fetch_w(T0);
      tile_barrier();  // A(first)
      FMD_MARK(11)
      for (int t = T0; t < t_end; t += S) {
        fetch_d(t + 2 * S);  // for the fragments written after C(t)
        if (!slice_owner) {  // FS, waves 2 and 3: distances for the shared Gaussians and the barrier protocol
          if (!TWO) tile_barrier();  // B(t)
          tile_barrier();  // C(t)
          if (SHARE) share_fragments();
          if (!TWO) tile_barrier();  // D(t)
          tile_barrier();  // A(t+S)
          continue;
        }
        f32x16 fl = zero16();
        if (!FMD_DBG_SKIP(a, 0)) {
        FMD_MARK(0)
        // L1 recomputed: filt slice (lane = feature, registers = pairs)
        f32x16 acc1 = zero16();
        if constexpr (STASHED) {  // ... or read back from the forward kernel's stash, one tile ahead
          acc1 = fl_next;
        } else if constexpr (NEED_GX1) {  // (block 0 has no gx1: the filter values and the hidden tile are not needed at all)
          // fragments are read two k-steps ahead of their MFMAs; the fences keep the scheduler from
          // hoisting all sixteen 16-byte reads (64 registers) to the top of the loop
          const _Float16* ph = hid_hi + c32 * LDH + 8 * h;
          const _Float16* pl = hid_lo + c32 * LDH + 8 * h;
          // (W16A16: no low plane -- in the side-by-side form hid_lo is the gW image, which other waves write)
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
#pragma unroll
        for (int r = 0; r < 16; ++r) fl[r] = acc1[r];
        asm volatile("" : "+v"(fl));  // L1 (and its LDS reads) complete
        FMD_MARK(1)
        // gx1[i] += sum_j g[j] W, gx1[j] += g[i] W  (segment: one centre, distinct j), interleaved with the refill of
        // X for tile t+S: a segment's aggregate rows are read, the previous segment's gathers are issued while those
        // reads are in flight (a gather costs the CU's address unit ~16 cycles, an LDS round trip ~100), then the
        // segment is finished and its registers -- free from here on -- take the next tile's offsets.
        float vj[4], cval = 0.f;
        auto seg_reads = [&](auto qc) {
          constexpr int q = decltype(qc)::value;
          if (NEED_GX1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) vj[e] = agg_rd(X.oj[4 * q + e]);
            cval = agg_rd(X.oi[q]);
          }
        };
        auto seg_finish = [&](auto qc) {
          constexpr int q = decltype(qc)::value;
          if (NEED_GX1) {
            float si = 0.f;
            if constexpr (HALF) {  // packed pairs, as in the forward kernel
              const f32x2 w01 = f32x2{fl[4 * q], fl[4 * q + 1]}, w23 = f32x2{fl[4 * q + 2], fl[4 * q + 3]};
              const f32x2 n01 = f32x2{vj[0], vj[1]} + splat2(X.gi[q]) * w01;
              const f32x2 n23 = f32x2{vj[2], vj[3]} + splat2(X.gi[q]) * w23;
              const f32x2 s2 = f32x2{X.gj[4 * q], X.gj[4 * q + 1]} * w01 + f32x2{X.gj[4 * q + 2], X.gj[4 * q + 3]} * w23;
              agg_wr(X.oj[4 * q], n01[0]);
              agg_wr(X.oj[4 * q + 1], n01[1]);
              agg_wr(X.oj[4 * q + 2], n23[0]);
              agg_wr(X.oj[4 * q + 3], n23[1]);
              si = s2[0] + s2[1];
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float Wqe = fl[4 * q + e];  // already filt * C
                si += X.gj[4 * q + e] * Wqe;
                agg_wr(X.oj[4 * q + e], vj[e] + X.gi[q] * Wqe);
              }
            }
            si = fmd_segment_sum(si, X.oi[q]);
            agg_wr(X.oi[q], cval + si);
          }
        };
        auto seg_refill = [&](auto qc) {  // segment q of X <- tile t+S (offsets fetched one tile ago)
          constexpr int q = decltype(qc)::value;
          X.oi[q] = oi_m[q] + col4;
#pragma unroll
          for (int e = 0; e < 4; ++e) X.oj[4 * q + e] = oj_m[4 * q + e] + col4;
          X.xi[q] = *reinterpret_cast<const float*>(x1b + X.oi[q]);
          X.gi[q] = *reinterpret_cast<const float*>(gab + X.oi[q]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            X.xj[4 * q + e] = *reinterpret_cast<const float*>(x1b + X.oj[4 * q + e]);
            X.gj[4 * q + e] = *reinterpret_cast<const float*>(gab + X.oj[4 * q + e]);
          }
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
        FMD_MARK(5)
        fetch_meta(t + 2 * S);
        // The next tile's filter values are requested LAST: the memory counter retires in order, so everything issued
        // behind a stash load waits for its HBM round trip too -- here the next wait for a younger load is a whole
        // tile away (the row gathers above are consumed after barrier C, ~0.6 us from now).
        fetch_w(t + S);
        FMD_MARK(6)
        }
        if (!TWO) tile_barrier();  // B(t): hidden tiles consumed
        FMD_MARK(2)
        gw_store(gw_base);
        FMD_MARK(3)
        tile_barrier();  // C(t): gW visible
        FMD_MARK(4)
        // the producers read fragments only while staging (between A and B): the tile they stage in the
        // next iteration, t + 2S, is written here
        if (SHARE) share_fragments();
        form_gw();  // tile t+S
        FMD_MARK(7)
        if (!TWO) tile_barrier();  // D(t)
        FMD_MARK(8)
        tile_barrier();  // A(t+S)
        FMD_MARK(9)
      }
      }  // !PIPE
    }
  }
  if (NEED_GX1) {
    __syncthreads();
    float* out = a.slabs + (int64_t)dst_row * F;
    if (FS) {  // this half's 64 features, zeros in the other half
      for (int i = tid * 4; i < n * 64; i += 2048) {
        const int row = i >> 6, c = i & 63;
        *reinterpret_cast<f32x4*>(out + row * F + 64 * split + c) = *reinterpret_cast<const f32x4*>(gx1 + i);
        *reinterpret_cast<f32x4*>(out + row * F + 64 * (1 - split) + c) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    } else {
      for (int i = tid * 4; i < n * F; i += 2048)
        *reinterpret_cast<f32x4*>(out + i) = *reinterpret_cast<const f32x4*>(gx1 + i);
    }
  }
  (void)dst_row;
  if (queued && tid0 == 0) *next_item_p = (int)atomicAdd(a.queue, 1) + n_wg;   // read behind the next item's first barrier
  }  // work items
  };  // run
  if (producer) run(std::true_type{});
  else run(std::false_type{});
  FMD_STAMP_FLUSH(a, wave)
}

}  // namespace

namespace fmd_eng {

// (NEED_GX1, FIRST, SHARE, HALF, FS, STASH); the stash form only exists with a gx1 aggregate and in fp32-grade
// precision (carve_ws allocates no stash for W16A16: ADVICE r4 -- those instantiations could never be launched)
#define FMD_BWD_F32(M_, ST_)                                                                    \
  M_(true, true, false, false, false, ST_) M_(true, false, false, false, false, ST_)           \
  M_(true, true, true, false, false, ST_) M_(true, false, true, false, false, ST_)             \
  M_(true, true, true, false, true, ST_) M_(true, false, true, false, true, ST_)
#define FMD_BWD_HALF(M_)                                                                        \
  M_(true, true, false, true, false, false) M_(true, false, false, true, false, false)         \
  M_(true, true, true, true, false, false) M_(true, false, true, true, false, false)           \
  M_(true, true, true, true, true, false) M_(true, false, true, true, true, false)
// block 0 (no gx1 aggregate)
#define FMD_BWD_B0_F32(M_)                                                                      \
  M_(false, true, false, false, false, false) M_(false, false, false, false, false, false)     \
  M_(false, true, true, false, false, false) M_(false, false, true, false, false, false)       \
  M_(false, true, true, false, true, false) M_(false, false, true, false, true, false)
#define FMD_BWD_B0_HALF(M_)                                                                     \
  M_(false, true, false, true, false, false) M_(false, false, false, true, false, false)       \
  M_(false, true, true, true, false, false) M_(false, false, true, true, false, false)         \
  M_(false, true, true, true, true, false) M_(false, false, true, true, true, false)
// NS0 = 8 (65..128 Gaussians): the per-wave Gaussian form; round 6: also in feature-split mode (284..535 beads)
#define FMD_BWD_WIDE_F32(M_)                                                                    \
  M_(true, true, false, false, false, false) M_(true, false, false, false, false, false)       \
  M_(true, true, false, false, false, true) M_(true, false, false, false, false, true)         \
  M_(false, true, false, false, false, false) M_(false, false, false, false, false, false)     \
  M_(true, true, false, false, true, false) M_(true, false, false, false, true, false)         \
  M_(true, true, false, false, true, true) M_(true, false, false, false, true, true)           \
  M_(false, true, false, false, true, false) M_(false, false, false, false, true, false)
#define FMD_BWD_WIDE_HALF(M_)                                                                   \
  M_(true, true, false, true, false, false) M_(true, false, false, true, false, false)         \
  M_(false, true, false, true, false, false) M_(false, false, false, true, false, false)       \
  M_(true, true, false, true, true, false) M_(true, false, false, true, true, false)           \
  M_(false, true, false, true, true, false) M_(false, false, false, true, true, false)
#ifdef FMD_HALF_TU   // this translation unit: the W16A16 kernels, reached through the *_half entry points
#define FMD_TU(x) x##_half
#define FMD_BWD_MINE(M_) if constexpr (NS0 > 4) { FMD_BWD_WIDE_HALF(M_) } else { FMD_BWD_HALF(M_) FMD_BWD_B0_HALF(M_) }
#else
#define FMD_TU(x) x
#define FMD_BWD_MINE(M_)                                 \
  if constexpr (NS0 > 4) { FMD_BWD_WIDE_F32(M_) }        \
  else { FMD_BWD_F32(M_, false) FMD_BWD_F32(M_, true) FMD_BWD_B0_F32(M_) }
#endif
int setup_edge_bwd_attrs_half();
int launch_edge_bwd_half(const EdgeArgs& a, int num_rbf, bool half, bool need_gx1, bool first, int64_t n_mol,
                         hipStream_t s);

template <int NS0>
static int set_attrs() {
#define FMD_ONE(N_, F_, S_, H_, FS_, ST_)                                                                     \
  if (hipFuncSetAttribute((const void*)edge_bwd_split_kernel<NS0, N_, F_, S_, H_, FS_, ST_>,                  \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)MAX_LDS_BYTES) != hipSuccess)      \
    return FMD_ELAUNCH;
  FMD_BWD_MINE(FMD_ONE)
#undef FMD_ONE
  return FMD_OK;
}
int FMD_TU(setup_edge_bwd_attrs)() {
  int rc = set_attrs<2>();
  if (rc == FMD_OK) rc = set_attrs<4>();
  if (rc == FMD_OK) rc = set_attrs<8>();
#ifndef FMD_HALF_TU
  if (rc == FMD_OK) rc = setup_edge_bwd_attrs_half();
#endif
  return rc;
}

template <int NS0>
static int launch_bwd(const EdgeArgs& a, bool half, bool need, bool first, dim3 grid, hipStream_t s) {
  // [tile region][red 128][centres 64][bias 128]([fragments])[gx1 rows]; block 0 (no gx1) keeps no aggregate
  const size_t lds = (size_t)(TILE_FLOATS + 128 + 128 + MAX_RBF + 128 + a.max_beads * (a.fs ? 64 : F)) * sizeof(float);
  const size_t frag = (size_t)a.nb * 1024;
  // shared-Gaussian form when its fragment region fits next to the aggregate (block 0 keeps none)
  // (block 0, whole-aggregate mode: the pipelined form -- second fragment buffer, second gW image, second partial sums)
  const size_t lds_sh = need ? lds + frag
                             : (size_t)(TILE_FLOATS + 128 + 128 + MAX_RBF + 128) * sizeof(float) + frag +
                                   (a.fs ? 0 : frag + (size_t)(4096 + 128) * sizeof(float));
  const bool share = NS0 <= 4 && lds_sh <= MAX_LDS_BYTES;
  const bool stash = need && !half && a.wst != nullptr;
  if (lds > MAX_LDS_BYTES || (a.fs && !share && NS0 <= 4)) return FMD_EUNSUPPORTED;  // up to 64 Gaussians feature-split mode always shares
#define FMD_ONE(N_, F_, S_, H_, FS_, ST_)                                                                  \
  if (need == N_ && first == F_ && (share || (a.fs && NS0 <= 4)) == S_ && half == H_ && a.fs == FS_ && stash == ST_) {  \
    edge_bwd_split_kernel<NS0, N_, F_, S_, H_, FS_, ST_><<<grid, 512, (S_) ? lds_sh : lds, s>>>(a);      \
    return hipGetLastError() == hipSuccess ? FMD_OK : FMD_ELAUNCH;                                         \
  }
  FMD_BWD_MINE(FMD_ONE)
#undef FMD_ONE
  return FMD_EUNSUPPORTED;
}

int FMD_TU(launch_edge_bwd)(const EdgeArgs& a, int num_rbf, bool half, bool need_gx1, bool first, int64_t n_mol,
                            hipStream_t s) {
#ifndef FMD_HALF_TU
  if (half) return launch_edge_bwd_half(a, num_rbf, half, need_gx1, first, n_mol, s);
#endif
  const dim3 grid((unsigned)a.n_split, (unsigned)n_mol);
  return num_rbf <= 32   ? launch_bwd<2>(a, half, need_gx1, first, grid, s)
         : num_rbf <= 64 ? launch_bwd<4>(a, half, need_gx1, first, grid, s)
                         : launch_bwd<8>(a, half, need_gx1, first, grid, s);
}

}  // namespace fmd_eng
