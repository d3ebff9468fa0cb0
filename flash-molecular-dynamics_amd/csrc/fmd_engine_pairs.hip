// Fused engine, first and last launch of a force evaluation: the per-molecule pair list (replaces
// torch_cluster.radius_graph as called at neighbor_list/torch_impl.py:216-224, for undirected pairs) and the
// final reduction dE/dd -> forces, per-node energies -> energy[B] (models/gradients.py:265-280, schnet.py:352-357).
#include "fmd_engine.h"

namespace {
// ---------------------------------------------------------------------------------------------
// pair list: undirected pairs, strict d2 < rc2 with unfused fp32, each listed once under ONE of its two beads (the
// "centre": the smaller index unless the orientation step below moved the pair), sorted by centre then neighbour.
// Each centre's entries are padded to a multiple of 4 (pad = (i,i) with d = rc, whose cutoff is 0), so a
// 4-entry half of an 8-entry SEGMENT of a 32-entry tile always belongs to one centre, and both halves do
// unless the rows are disjoint (see the conflict flags below): inside a segment the neighbours j are
// distinct, which lets the edge kernels accumulate with plain LDS read-modify-write (no float atomics) and
// one in-register reduction per segment.
// ---------------------------------------------------------------------------------------------
// GROUPED: the "molecule" is a group of mol.stride > 1 consecutive molecules (fmd_engine.h MolView): the hit test also
// compares the beads' molecules.  The plain instantiation carries none of it.
template <bool GROUPED>
__global__ __launch_bounds__(1024) void pairs_kernel(const float* __restrict__ pos,
                                                    const MolView mol, float rc2,
                                                    float rc, int cap, uint32_t* __restrict__ poi,
                                                    uint32_t* __restrict__ poj,
                                                    float* __restrict__ pd, float* __restrict__ pc,
                                                    float* __restrict__ pdc, int32_t* __restrict__ npairs,
                                                    int32_t* __restrict__ report, int max_beads, float* __restrict__ forces, int ent_cap,
                                                    int* __restrict__ extra_row, int* __restrict__ queue) {
  // Two workgroups (16 waves each) per molecule.  Both build the hit masks, counts and offsets of the whole
  // molecule (one wave per centre row, rows dealt round-robin: the row's candidates j > i are tested 64 at a time,
  // the ballots are kept as hit masks in LDS); each then fills and finishes the rows of ITS half of the entries.
  // squared distances as separately rounded fp32 products and sums in the oracle's order: plain operators under
  // contract(off) (the __f*_rn intrinsics are header functions that keep contract = fast and were fused into v_fma_f32)
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int m = blockIdx.y, part = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (m: a GROUP of mol.stride consecutive molecules, see MolView; max_beads: the bound on a group's beads)
  const int64_t a0 = grp_begin(mol, m);
  const int n_all = (int)(grp_begin(mol, m + 1) - a0);
  int n = n_all;
  if (n > max_beads) {  // the caller's bound is wrong: stay inside the LDS that was sized for it, and say so
    n = max_beads;
  }
  const bool bad_bound = n_all > max_beads;
  if (m == 0 && part == 0 && tid < 16) queue[tid] = 0;   // the pair kernels' item counters (queue mode) of this evaluation
  float* sx = smem;
  float* sy = sx + n;
  float* sz = sy + n;
  int* cnt = reinterpret_cast<int*>(sz + n);   // [n]     pairs of row i (j > i)
  int* rowstart = cnt + n;                      // [n + 1] offset of the row's run
  int* rowend = rowstart + n + 1;               // [n]     end of the run including its pad entries
  int* conf = rowend + n;                       // [n]     row i must not share an 8-entry segment with the row before
  int* mid = conf + n;                          // [n]     molecule of the bead inside its group (0 when stride = 1)
  const int NC = (n + 63) >> 6;                 // 64-candidate chunks of a row (at most 9: 535 beads)
  // [n][NC] hit masks: bit b of masks[i NC + c] <-> (i, 64 c + b) is a pair; chunks below (i + 1) / 64 are not written
  unsigned long long* masks = reinterpret_cast<unsigned long long*>(smem + ((8 * n + 2) & ~1));
  __shared__ int tot_s[3];
  for (int i = tid; i < n; i += 1024) {
    sx[i] = pos[3 * (a0 + i)];
    sy[i] = pos[3 * (a0 + i) + 1];
    sz[i] = pos[3 * (a0 + i) + 2];
    if constexpr (GROUPED) {
      int k = 0;   // the bead's molecule inside the group: how many of the group's later molecules start at or before it
      for (int j = 1; j < mol.stride; ++j) k += grp_begin(MolView{mol.ptr, mol.n_real, 1}, (int64_t)m * mol.stride + j) - a0 <= i ? 1 : 0;
      mid[i] = k;
    }
  }
  // finish_kernel's two workgroups per molecule ADD their halves to the forces: start them from zero
  if (part == 0) {
    for (int i = tid; i < 3 * n_all; i += 1024) forces[3 * a0 + i] = 0.f;
    for (int i = tid; i < n_all; i += 1024) extra_row[a0 + i] = -1;  // sched_body names the exceptions
  }
  __syncthreads();
  auto dist2 = [&](int i, int j) {
    const float dx = sx[j] - sx[i], dy = sy[j] - sy[i], dz = sz[j] - sz[i];
    return (dx * dx + dy * dy) + dz * dz;
  };
  // ---- pass 1: the hit masks and row counts.  A lane keeps its candidates j = 64 c + lane in registers (beads
  //      beyond the molecule sit at 1e30: their d2 is +inf and never a hit), a wave takes one centre row at a time.
  {
    float px[9], py[9], pz[9];
    int pm[9];   // the candidates' molecules inside the group: a pair needs both beads in one molecule
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const int j = 64 * c + lane;
      const bool ok = j < n;
      px[c] = ok ? sx[j] : 1e30f;
      py[c] = ok ? sy[j] : 1e30f;
      pz[c] = ok ? sz[j] : 1e30f;
      pm[c] = (GROUPED && ok) ? mid[j] : 0;
    }
    for (int i = wv; i < n; i += 16) {
      const float xi = sx[i], yi = sy[i], zi = sz[i];
      const int mi = GROUPED ? mid[i] : 0;
      const int c_first = (i + 1) >> 6;
      // Branch-free chunk body (round 6): a lane of the row's FIRST chunk is a candidate only above the centre -- a scalar
      // mask on the ballot instead of a per-lane test (beyond the first chunk 64 c + lane > i always holds) --, and the
      // ballot goes into lane c of the mask registers with v_writelane.  The compiler had turned the two-sided
      // test into nested exec-mask regions, ~30 instructions and two branches per chunk (profiles/r06_pairs_phases.txt).
      const unsigned long long first_mask = ~0ull << ((i + 1) & 63);
      int c_tot = 0;
      unsigned mlo = 0u, mhi = 0u;  // lane c collects the mask of chunk c: one LDS store per row
      static_for<0, 9>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if (c >= c_first && c < NC) {   // (wave-uniform)
          const float dx = px[c] - xi, dy = py[c] - yi, dz = pz[c] - zi;
          const float d2 = (dx * dx + dy * dy) + dz * dz;
          unsigned long long mk = __ballot(GROUPED ? (d2 < rc2) & (pm[c] == mi) : d2 < rc2);
          if (c == c_first) mk &= first_mask;
          // (no clang builtin for v_writelane_b32; the lane select is an inline constant: gfx9's one-SGPR limit)
          asm("v_writelane_b32 %0, %1, %2" : "+v"(mlo) : "s"((unsigned)mk), "n"(c));
          asm("v_writelane_b32 %0, %1, %2" : "+v"(mhi) : "s"((unsigned)(mk >> 32)), "n"(c));
          c_tot += __popcll(mk);
        }
      });
      if (lane < NC) masks[i * NC + lane] = ((unsigned long long)mhi << 32) | mlo;  // (zero below c_first)
      if (lane == 0) cnt[i] = c_tot;
    }
  }
  __syncthreads();
  // ---- orientation: an undirected pair may sit in either endpoint's run (the edge kernels update both rows of an
  //      entry, finish_kernel takes the direction from the positions), so a row whose run is r = cnt mod 4 entries past a
  //      multiple of four GIVES r of its pairs (i, j) to the rows j: they become (j, i), and the run needs no pad
  //      entries.  A decision needs the final count of its row, so the rows are taken in a few blocks (each half of
  //      what remains) and a row gives only to rows of LATER blocks -- its largest neighbours -- which makes the rows
  //      of one block independent: 5.7 % pad entries -> 0.4 % at the bench size (7 barriers).
  for (int lo_b = 0; n - lo_b > 8;) {  // uniform over the workgroup
    const int hi_b = lo_b + max(4, (n - lo_b) >> 1);
    const int i = lo_b + tid;
    if (i < hi_b) {
      int c_i = cnt[i], give = c_i & 3;
      for (int c = NC - 1; c >= (hi_b >> 6) && give > 0; --c) {
        const unsigned long long orig = masks[i * NC + c];
        unsigned long long m2 = orig;
        if (c == (hi_b >> 6)) m2 &= ~0ull << (hi_b & 63);  // neighbours in later blocks only (all of them are > i)
        while (m2 && give > 0) {
          const int bt = 63 - __clzll((long long)m2);
          m2 ^= 1ull << bt;
          const int j = 64 * c + bt;
          atomicOr(&masks[j * NC + (i >> 6)], 1ull << (i & 63));
          atomicAdd(&cnt[j], 1);
          --give;
          --c_i;
        }
        const unsigned long long keep = (c == (hi_b >> 6)) ? (m2 | (orig & ~(~0ull << (hi_b & 63)))) : m2;
        masks[i * NC + c] = keep;
      }
      cnt[i] = c_i;
    }
    __syncthreads();
    lo_b = hi_b;
  }
  // ---- which rows may share an 8-entry segment with the row before them: the two halves of a segment are
  //      updated side by side, so the centre and the (at most four) first neighbours of row i must not appear among
  //      the neighbours in the last group of four of the row before, nor the centre of the row before among those
  //      first neighbours (a pair it gave away)
  for (int i = tid; i < n; i += 1024) {
    bool cf = false;
    if (cnt[i] > 0) {
      int p = i - 1;
      while (p >= 0 && cnt[p] == 0) --p;
      if (p >= 0) {
        // the row's first (at most four) neighbours and the last rp of the row before, as 16-bit fields
        unsigned long long hd = ~0ull, tl = ~0ull;
        int hn = 0, tn = 0;
        for (int c = 0; c < NC && hn < 4; ++c) {
          unsigned long long m2 = masks[i * NC + c];
          while (m2 && hn < 4) {
            hd = (hd << 16) | (unsigned long long)(64 * c + __ffsll((long long)m2) - 1);
            m2 &= m2 - 1;
            ++hn;
          }
        }
        const int rp = ((cnt[p] - 1) & 3) + 1;
        for (int c = NC - 1; c >= 0 && tn < rp; --c) {
          unsigned long long m2 = masks[p * NC + c];
          while (m2 && tn < rp) {
            const int bt = 63 - __clzll((long long)m2);
            m2 ^= 1ull << bt;
            tl = (tl << 16) | (unsigned long long)(64 * c + bt);
            ++tn;
          }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const unsigned tv = (unsigned)(tl >> (16 * t)) & 0xffffu;  // 0xffff: no such neighbour
          bool hit = tv == (unsigned)i;
#pragma unroll
          for (int y = 0; y < 4; ++y) hit |= tv == ((unsigned)(hd >> (16 * y)) & 0xffffu);
          cf |= tv != 0xffffu && hit;
          cf |= ((unsigned)(hd >> (16 * t)) & 0xffffu) == (unsigned)p;  // a pair the row before gave to this one
        }
      }
    }
    conf[i] = cf ? 1 : 0;
  }
  __syncthreads();
  // ---- offsets: every run is padded to a multiple of FOUR entries; a run that would start in the second half of
  //      an 8-entry segment whose first half belongs to a conflicting row is pushed to the next segment (the gap
  //      becomes pad entries of the row before); the total is padded to a multiple of 8.
  //      Done by one wave in three short steps instead of one 269-step loop: the only state a row inherits is
  //      whether it starts in the second half of a segment, so each lane runs its chunk of rows for both start
  //      states, the chunk functions are composed across lanes by a shuffle scan, and each lane replays its chunk
  //      from its true start offset.
  if (wv == 0) {
    const int chunk = (n + 63) >> 6, lo_r = min(n, lane * chunk), hi_r = min(n, lo_r + chunk);
    int dl[2], es[2], last_ne = -1, real = 0;
#pragma unroll
    for (int s0 = 0; s0 < 2; ++s0) {
      int st = s0, delta = 0;
      for (int i = lo_r; i < hi_r; ++i) {
        const int c = cnt[i];
        if (c == 0) continue;
        if (st && conf[i]) { delta += 4; st = 0; }
        const int pad = (c + 3) & ~3;
        delta += pad;
        st ^= (pad >> 2) & 1;
        if (s0 == 0) { last_ne = i; real += c; }
      }
      dl[s0] = delta;
      es[s0] = st;
    }
    // inclusive scan over lanes of the chunk functions  state -> (added entries, end state): earlier lanes first
    int d0 = dl[0], d1 = dl[1], nx = es[0] | (es[1] << 1), ln = last_ne;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int pd0 = __shfl_up(d0, o, 64), pd1 = __shfl_up(d1, o, 64), pnx = __shfl_up(nx, o, 64);
      const int pln = __shfl_up(ln, o, 64);
      if (lane >= o) {
        const int n0 = pnx & 1, n1 = (pnx >> 1) & 1;  // the earlier part's end states
        const int c0 = pd0 + (n0 ? d1 : d0), c1 = pd1 + (n1 ? d1 : d0);
        nx = ((nx >> n0) & 1) | (((nx >> n1) & 1) << 1);
        d0 = c0;
        d1 = c1;
        ln = max(ln, pln);
      }
    }
    int base = __shfl_up(d0, 1, 64), prev = __shfl_up(ln, 1, 64);  // exclusive: what the lanes before this one add
    if (lane == 0) { base = 0; prev = -1; }
    int off = base;
    for (int i = lo_r; i < hi_r; ++i) {
      const int c = cnt[i];
      rowstart[i] = off;
      if (c == 0) continue;
      if ((off & 4) && conf[i]) off += 4;
      rowstart[i] = off;
      if (prev >= 0) rowend[prev] = off;
      off += (c + 3) & ~3;
      prev = i;
    }
    // totals: the last lane's end state; real pairs summed over the wave
    int real_w = real;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) real_w += __shfl_xor(real_w, o, 64);
    if (lane == 63) {
      const int tot = (off + 7) & ~7;
      if (prev >= 0) rowend[prev] = tot;
      rowstart[n] = tot;
      tot_s[0] = tot;
      tot_s[1] = real_w;
    }
  }
  __syncthreads();
  const int total = tot_s[0];
  // this workgroup's rows [r_lo, r_hi) and entries [k_lo, k_hi): the split row is the first one that starts in the
  // second half of the list (row offsets ascend)
  if (tid < 64) {
    int below = 0;
    for (int i = tid; i < n; i += 64) below += rowstart[i] < (total >> 1) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o, 64);
    if (tid == 0) tot_s[2] = below;
  }
  __syncthreads();
  const int r_split = tot_s[2];
  int r_next = r_split;  // the first non-empty row of the second part: its (final) offset ends the first part's pads
  while (r_next < n && cnt[r_next] == 0) ++r_next;
  const int k_split = r_next < n ? rowstart[r_next] : total;
  // (small molecules are launched with ONE workgroup per molecule, which then fills every row: launch_pairs)
  const bool whole = gridDim.x == 1;
  const int r_lo = part ? r_split : 0, r_hi = (part || whole) ? n : r_split;
  const int k_lo = part ? k_split : 0, k_hi = (part || whole) ? total : k_split;
  // ---- pass 2: fill from the masks (sorted by j)
  uint32_t* oi_out = poi + (int64_t)m * cap;  // centre row byte offset  i * 512
  uint32_t* oj_out = poj + (int64_t)m * cap;  // neighbour row byte offset j * 512
  float* d_out = pd + (int64_t)m * cap;
  float* c_out = pc + (int64_t)m * cap;
  float* dc_out = pdc + (int64_t)m * cap;
  // The (centre, neighbour) entries are staged in LDS as i | j << 16 when the list fits (ent_cap entries): the sweep's
  // scattered writes (a dozen active lanes per store) then stay on chip, and the entry arrays are written once,
  // densely, by the pass below.  A longer list is written straight to the arrays and read back.
  uint32_t* ent = reinterpret_cast<uint32_t*>(masks + n * NC);
  const bool staged = total <= ent_cap;
  for (int i = r_lo + wv; i < r_hi; i += 16) {
    if (cnt[i] == 0) continue;
    int k = rowstart[i];
    k = __builtin_amdgcn_readfirstlane(k);
    const unsigned long long mrow = lane < NC ? masks[i * NC + lane] : 0ull;  // lane c: the mask of chunk c
    const unsigned rlo = (unsigned)mrow, rhi = (unsigned)(mrow >> 32);
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      if (c < NC) {
        const unsigned lo = __builtin_amdgcn_readlane(rlo, c), hi = __builtin_amdgcn_readlane(rhi, c);
        const unsigned long long mk = ((unsigned long long)hi << 32) | lo;  // wave-uniform: used as the execution mask
        if (__builtin_amdgcn_inverse_ballot_w64(mk)) {
          const int kk = k + (int)__builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
          if (staged) {
            ent[kk] = (uint32_t)i | ((uint32_t)(64 * c + lane) << 16);
          } else if (kk < cap) {
            oi_out[kk] = (uint32_t)i * (F * 4);
            oj_out[kk] = (uint32_t)(64 * c + lane) * (F * 4);
          }
        }
        k += __popcll(mk);
      }
    }
    const int kend = rowend[i];  // pad entries (i, i), C = 0: to the next run (at most 3 + 4 of them)
    if (k + lane < kend) {
      if (staged) {
        ent[k + lane] = (uint32_t)i | ((uint32_t)i << 16);
      } else if (k + lane < cap) {
        oi_out[k + lane] = (uint32_t)i * (F * 4);
        oj_out[k + lane] = (uint32_t)i * (F * 4);
      }
    }
  }
  __syncthreads();  // (workgroup-scope release/acquire of the entries written above)
  // ---- distances and cutoff values, one entry per lane (in a candidate sweep only one lane in five holds a pair,
  //      so sqrt / sin / cos there would run at a fifth of the lanes)
  for (int k = k_lo + tid; k < min(k_hi, cap); k += 1024) {
    int i, j;
    if (staged) {
      const uint32_t e = ent[k];
      i = (int)(e & 0xffffu);
      j = (int)(e >> 16);
      oi_out[k] = (uint32_t)i * (F * 4);
      oj_out[k] = (uint32_t)j * (F * 4);
    } else {
      i = (int)(oi_out[k] / (F * 4));
      j = (int)(oj_out[k] / (F * 4));
    }
    float d = rc, C = 0.f, dC = 0.f;
    if (i != j) {
      d = sqrtf(dist2(i, j));
      fmd_cutoff2(d, rc, C, dC);
    }
    d_out[k] = d;
    c_out[k] = C;
    dc_out[k] = dC;
  }
  // fill the last 32-entry tile with null entries (row 0, zero cutoff) so the edge kernels never
  // need a validity test
  if (part == 1 || whole) {
    const int tot_c = min(total, cap) & ~7;
    const int tot32 = min((tot_c + 31) & ~31, cap);
    for (int k = tot_c + tid; k < tot32; k += 1024) {
      oi_out[k] = 0u;
      oj_out[k] = 0u;
      d_out[k] = rc;
      c_out[k] = 0.f;
      dc_out[k] = 0.f;
    }
  }
  // per-molecule report (error code, entries needed, real pairs): finish_kernel reduces it into the status words,
  // so nothing has to be zeroed before the call
  if (part == 0 && tid == 0) {
    npairs[m] = min(total, cap) & ~7;
    report[3 * m] = bad_bound ? FMD_EINVAL : (total > cap ? FMD_ECAPACITY : 0);
    report[3 * m + 1] = total;
    report[3 * m + 2] = tot_s[1];
  }
}

// ---------------------------------------------------------------------------------------------
// Work items of the pair kernels (fmd_engine.h ITEM_INTS): one workgroup of 256 threads, once per force evaluation
// (the first workgroup of node_embed_kernel, beside the embedding lookup: no launch of its own).
// A pair-kernel workgroup keeps ONE molecule's aggregate in LDS, so a molecule's tiles can only be shared among
// workgroups that each pay for an aggregate (and a slab in HBM).  Static form (balanced = 0): workgroup (m, s) walks
// the tiles s, s + S, ... of molecule m.  When all workgroups fit the chip at once (one per CU), the launch lasts as
// long as its LARGEST molecule needs -- 11 % above the mean at the bench size -- so the balanced form caps every run
// at q tiles: workgroup (m, s) takes tiles [s q, (s + 1) q) of m, and what a large molecule has beyond S q goes, as a
// second item with its own (extra) slab, to the last workgroup of a small molecule: the k-th largest is paired with
// the k-th smallest.  q is the smallest cap for which every such pair fits into q tiles.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void sched_body(const SchedArgs& a) {
  __shared__ int key[256];    // tiles * 256 + molecule, sorted descending
  __shared__ int xoff[256];   // first row of a large molecule's extra slab
  __shared__ int tot_s;
  const int tid = threadIdx.x, S = a.n_split, B = a.n_mol, N = (int)a.n_nodes;
  auto put = [&](int64_t w, int slot, int mol, int tb, int te, int stride, int dst) {
    int4* o = reinterpret_cast<int4*>(a.items + (w * 2 + slot) * ITEM_INTS);
    o[0] = int4{mol, tb, te, stride};
    o[1] = int4{dst, 0, 0, 0};
  };
  if (!a.balanced) {
    for (int64_t w = tid; w < (int64_t)B * S; w += 256) {
      const int m = (int)(w / S), s = (int)(w % S);
      const int nt = (a.npairs[m] + 31) >> 5, a0 = (int)grp_begin(a.mol, m);
      if (a.fs) put(w, 0, m, 0, nt, 1, s * N + a0);  // feature split: both workgroups walk all tiles
      else put(w, 0, m, s, nt, S, s * N + a0);
      put(w, 1, -1, 0, 0, 1, 0);
    }
    return;
  }
  // (all global reads up front, one round trip: the lookup's traffic runs beside this workgroup)
  if (a.balanced == 2) {
    // Chained form (129..255 molecules: one workgroup per molecule would leave CUs idle, two would need a second
    // round).  Workgroup w of n_wg takes tiles [w q, (w + 1) q) of the molecules' tiles laid end to end, q = ceil(T /
    // n_wg): with q below the smallest and at least half the largest molecule a chunk touches at most two molecules
    // and a molecule at most three chunks -- piece 0 and 1 go to slabs 0 and 1 (every molecule has both), piece 2 to an
    // extra slab.  Batches that do not satisfy the two conditions fall back to halves: the first half of molecule m
    // on workgroup m, the second on workgroup B + m while those last, else on workgroup m as well.
    __shared__ int tt[256], pp[257], a0c[256], nbc[256], x3[257];
    __shared__ int tmin_s, tmax_s;
    const int G = a.n_wg;
    const int t_c = tid < B ? (a.npairs[tid] + 31) >> 5 : 0;
    tt[tid] = t_c;
    a0c[tid] = tid < B ? (int)grp_begin(a.mol, tid) : 0;
    nbc[tid] = tid < B ? (int)(grp_begin(a.mol, tid + 1) - grp_begin(a.mol, tid)) : 0;
    if (tid == 0) { tmin_s = 0x7fffffff; tmax_s = 0; }
    __syncthreads();
    if (tid < B) { atomicMin(&tmin_s, t_c); atomicMax(&tmax_s, t_c); }
    if (tid == 0) {  // exclusive prefix of the tile counts (B <= 256: a serial pass over LDS)
      int acc = 0;
      for (int m = 0; m < B; ++m) { pp[m] = acc; acc += tt[m]; }
      pp[B] = acc;
    }
    __syncthreads();
    const int T = pp[B], q = max(1, (T + G - 1) / G);
    const bool chain = T > 0 && q < tmin_s && tmax_s <= 2 * q;
    if (chain) {
      // molecules with a third piece and the rows of their extra slabs
      const bool three = tid < B && (pp[tid] + t_c - 1) / q - pp[tid] / q == 2;
      x3[tid] = three ? nbc[tid] : 0;
      __syncthreads();
      if (tid == 0) {
        int acc = 0;
        for (int m = 0; m < B; ++m) { const int v = x3[m]; x3[m] = acc; acc += v; }
      }
      __syncthreads();
      auto dst = [&](int m, int k) { return k < 2 ? k * N + a0c[m] : 2 * N + x3[m]; };
      if (tid < G) {
        const int lo = tid * q, hi = min(lo + q, T);
        if (lo >= T) {
          put(tid, 0, -1, 0, 0, 1, 0);
          put(tid, 1, -1, 0, 0, 1, 0);
        } else {
          int m0 = 0;  // the molecule that holds global tile lo: the last m with pp[m] <= lo
          for (int step = 128; step > 0; step >>= 1)
            if (m0 + step < B && pp[m0 + step] <= lo) m0 += step;
          put(tid, 0, m0, lo - pp[m0], min(hi, pp[m0 + 1]) - pp[m0], 1, dst(m0, tid - pp[m0] / q));
          if (hi > pp[m0 + 1]) put(tid, 1, m0 + 1, 0, hi - pp[m0 + 1], 1, dst(m0 + 1, 0));
          else put(tid, 1, -1, 0, 0, 1, 0);
        }
      }
      if (three)
        for (int r = 0; r < nbc[tid]; ++r) a.extra_row[a0c[tid] + r] = 2 * N + x3[tid] + r;
    } else if (tid < G) {
      if (tid < B) {
        const int hf = (t_c + 1) >> 1;
        put(tid, 0, tid, 0, hf, 1, a0c[tid]);
        if (tid + B < G) put(tid, 1, -1, 0, 0, 1, 0);
        else put(tid, 1, tid, hf, t_c, 1, N + a0c[tid]);
      } else {
        const int m = tid - B, hf = (tt[m] + 1) >> 1;
        put(tid, 0, m, hf, tt[m], 1, N + a0c[m]);
        put(tid, 1, -1, 0, 0, 1, 0);
      }
    }
    return;
  }
  const int t_me = tid < B ? (a.npairs[tid] + 31) >> 5 : 0;
  const int a0_me = tid < B ? (int)grp_begin(a.mol, tid) : 0, n_me = tid < B ? (int)(grp_begin(a.mol, tid + 1) - grp_begin(a.mol, tid)) : 0;
  __shared__ int a0_m[256], n_m[256];
  a0_m[tid] = a0_me;
  n_m[tid] = n_me;
  __shared__ __attribute__((aligned(16))) int raw[256];
  const int my = tid < B ? t_me * 256 + tid : -1;
  raw[tid] = my;
  if (tid == 0) tot_s = 0;
  __syncthreads();
  // rank sort, descending (the keys are distinct: every thread counts the keys above its own; broadcast LDS reads)
  int rank = 0;
#pragma unroll 8
  for (int k = 0; k < ((B + 3) & ~3); k += 4) {  // (unused slots hold -1: never above a key)
    const int4 v = *reinterpret_cast<const int4*>(raw + k);
    rank += (v.x > my ? 1 : 0) + (v.y > my ? 1 : 0) + (v.z > my ? 1 : 0) + (v.w > my ? 1 : 0);
  }
  key[tid < B ? rank : tid] = my;  // (my = -1 beyond the batch)
  {
    int tw = t_me;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tw += __shfl_xor(tw, o, 64);
    if ((tid & 63) == 0) atomicAdd(&tot_s, tw);
  }
  __syncthreads();
  const int tk = tid < B ? key[tid] >> 8 : 0;  // tiles of the tid-th largest molecule
  const int mk = tid < B ? key[tid] & 255 : 0;  // ... which is molecule mk: rows [a0k, a0k + nk)
  const int a0k = tid < B ? a0_m[mk] : 0, nk = tid < B ? n_m[mk] : 0;
  // the cap q.  A molecule that is large at a cap (tk > S q) objects to it while its partner, the tid-th smallest, is
  // large as well or the two do not fit into the partner's last workgroup; both conditions relax as q grows, so every
  // thread finds the first cap it accepts by bisection (at q = ceil(tk / S) it is not large any more) and the
  // largest of those wins.
  const int q0 = max(1, (tot_s + S * B - 1) / (S * B));
  const int tp = tid < B ? key[B - 1 - tid] >> 8 : 0;
  auto objects = [&](int qc) {
    return tk > S * qc && (tp > S * qc || (tk - S * qc) + max(0, tp - (S - 1) * qc) > qc);
  };
  int qk = q0;
  if (tid < B && objects(q0)) {
    int lo = q0, hi = max(q0, (tk + S - 1) / S);  // objects at lo, accepts hi
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (objects(mid)) lo = mid; else hi = mid;
    }
    qk = hi;
  }
  __shared__ int qmax_s[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) qk = max(qk, __shfl_xor(qk, o, 64));
  if ((tid & 63) == 0) qmax_s[tid >> 6] = qk;
  __syncthreads();
  const int q = max(max(qmax_s[0], qmax_s[1]), max(qmax_s[2], qmax_s[3]));
  const int n_large = __syncthreads_count(tid < B && tk > S * q);
  // rows of the extra slabs: exclusive prefix sum of the large molecules' bead counts (one wave-free LDS pass)
  __shared__ int nb_s[256], a0_s[256];
  nb_s[tid] = nk;
  a0_s[tid] = a0k;
  __syncthreads();
  if (tid == 0) {
    int off = 0;
    for (int k = 0; k < n_large; ++k) {
      xoff[k] = off;
      off += nb_s[k];
    }
  }
  if (tid < B) {
    for (int s = 0; s < S; ++s) {  // an empty run still zeroes and writes its slab
      put((int64_t)mk * S + s, 0, mk, min(s * q, tk), min((s + 1) * q, tk), 1, s * N + a0k);
      put((int64_t)mk * S + s, 1, -1, 0, 0, 1, 0);
    }
  }
  __syncthreads();
  if (tid < n_large) {
    const int mp = key[B - 1 - tid] & 255;
    put((int64_t)mp * S + (S - 1), 1, mk, S * q, tk, 1, S * N + xoff[tid]);
  }
  for (int k = tid >> 6; k < n_large; k += 4) {  // a wave per large molecule
    const int nb = nb_s[k], a0 = a0_s[k], x0 = S * N + xoff[k];
    for (int r = tid & 63; r < nb; r += 64) a.extra_row[a0 + r] = x0 + r;
  }
}

// second launch of a force evaluation: x = embedding[type], x1 = (embedding lin1^T)[type] from the packed table
// (32 lanes x 16 bytes per bead and array); its first workgroup writes the pair kernels' work items instead
__global__ __launch_bounds__(256) void node_embed_kernel(const int64_t* __restrict__ types,
                                                         const float* __restrict__ emb,
                                                         const float* __restrict__ emb_x1,
                                                         float* __restrict__ x, float* __restrict__ x1,
                                                         int64_t n_nodes, SchedArgs sa) {
  if (blockIdx.x == 0) {  // dispatched first: runs beside the lookup
    sched_body(sa);
    return;
  }
  const int64_t total = n_nodes * 32, stride = (int64_t)(gridDim.x - 1) * 256;
  for (int64_t idx = (int64_t)(blockIdx.x - 1) * 256 + threadIdx.x; idx < total; idx += stride) {
    const int64_t n = idx >> 5;
    const int c = (int)(idx & 31) * 4;
    const int64_t ty = types[n];
    *reinterpret_cast<f32x4*>(x + n * F + c) = *reinterpret_cast<const f32x4*>(emb + ty * F + c);
    *reinterpret_cast<f32x4*>(x1 + n * F + c) = *reinterpret_cast<const f32x4*>(emb_x1 + ty * F + c);
  }
}


// two workgroups (4 waves each) per molecule, each with half of its pair list: forces from dE/dd of the pairs,
// energy from the beads.  Each wave accumulates a private copy of dE/dpos in LDS as 64-bit FIXED POINT (2^-32): a
// scattered ds_add_u64 costs 229 cycles per wave instruction against 768 for ds_add_f32
// (tools/ubench/lds_atomic_int.hip), integer addition commutes -- any order of the atomics gives the same bits.
// The conversion is sign-symmetric, fixed(-x) = -fixed(x): |x| is cut into its integer part and a 32-bit fraction
// (both exact in fp32: x * 2^32 only moves the exponent), every addend is therefore exact whenever |x| >= 2^-32 * 2^23
// and truncated towards zero by less than 2^-32 otherwise.  The 64-bit sum carries |integer part| < 2^31 and collects
// at most 2 (n - 1) addends per bead and wave copy, four copies, n <= 535: fewer than 2^12 addends in all.  An addend
// is therefore accepted only below FMD_FIXED_LIMIT = 2^19 (2^12 x 2^19 = 2^31: the sum cannot wrap); anything larger,
// Inf or NaN raises `bad`, and the caller of this helper then poisons the molecule's forces with NaN and sets the
// sticky status word, so a blow-up stays as visible as it was with float atomics and never turns into a finite wrong
// force (a per-pair force of 5e5 in the model's units is a diverged system by any measure).  The four copies are added as
// integers, converted once; the two workgroups then add their halves to the force array that pairs_kernel zeroed: two
// addends commute.
constexpr float FMD_FIXED_LIMIT = 524288.0f;  // 2^19
__device__ __forceinline__ unsigned long long fmd_to_fixed(float x, bool& bad) {
  const float ax = fabsf(x);
  bad = bad || !(ax < FMD_FIXED_LIMIT);        // also true for NaN
  const float ip = truncf(ax);
  const float fr = (ax - ip) * 4294967296.0f;  // exact, in [0, 2^32)
  const unsigned long long q = ((unsigned long long)(unsigned)ip << 32) | (unsigned long long)(unsigned)fr;
  return x < 0.f ? 0ull - q : q;
}
__device__ __forceinline__ float fmd_from_fixed(unsigned long long q) {
  return (float)((double)(long long)q * (1.0 / 4294967296.0));
}
// threads of a finish_kernel workgroup: the kernel is a stream of scattered 64-bit LDS atomics (~229 cycles each for a lone
// wave), and sixteen waves keep four times as many in flight as four (same-box A/B, profiles/r06_ab_finish.txt: 16.3 us with
// 256 threads, 12.7 with 512, 11.3 with 1024; the same bits -- the groups of eight entries do not depend on who takes them)
constexpr int FIN_NT = 1024;
__global__ __launch_bounds__(FIN_NT) void finish_kernel(const float* __restrict__ pos,
                                                     const MolView mol,
                                                     const uint32_t* __restrict__ poi,
                                                     const uint32_t* __restrict__ poj,
                                                     const float* __restrict__ pd,
                                                     const float* __restrict__ pgd,
                                                     const float* __restrict__ pgd2,  // FS: second slab or null
                                                     const int32_t* __restrict__ npairs, int cap,
                                                     const float* __restrict__ e_node,
                                                     float* __restrict__ energy,
                                                     float* __restrict__ forces, int max_beads,
                                                     const int32_t* __restrict__ report,
                                                     int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int m = blockIdx.y, half = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t a0 = grp_begin(mol, m);
  const int n = min((int)(grp_begin(mol, m + 1) - a0), max_beads);
  unsigned long long* gall = reinterpret_cast<unsigned long long*>(smem);
  unsigned long long* g = gall + (wv & 3) * 3 * n;  // this wave's [3n] dE/dpos, fixed point (four copies: waves w, w + 4, ... share one)
  __shared__ float esum[4];
  __shared__ int bad_s;
  bool bad = false;  // a contribution that the fixed-point sum cannot carry (Inf, NaN, |x| >= FMD_FIXED_LIMIT)
  if (tid == 0) bad_s = 0;
  for (int i = tid; i < 12 * n; i += FIN_NT) gall[i] = 0ull;
  float* const spos = reinterpret_cast<float*>(gall + 12 * n);   // [3 n] positions of the molecule
  for (int i = tid; i < 3 * n; i += FIN_NT) spos[i] = pos[3 * a0 + i];
  __syncthreads();
  const int np = npairs[m];
  const int np_half = ((np + 15) >> 4) << 3;  // the first workgroup's share (whole 8-entry segments)
  const int q_lo = half ? min(np, np_half) : 0, q_hi = half ? np : min(np, np_half);
  // Every thread takes GROUPS of eight consecutive entries of the (centre-sorted) list -- 32 contiguous bytes of each
  // list array, fetched as two 16-byte loads per array, all of a group's loads in flight together, a wave's loads 2 KB
  // of contiguous memory (round 6: the former walk, one entry at a time with a per-thread stride of ~15 entries, was a
  // chain of ~15 uncoalesced round trips; 20.3 -> 15.4 us, profiles/r06_ab_finish.txt).  Inside a group the centre's
  // contribution is summed in registers and flushed when the centre changes, so lanes rarely collide in LDS.  The
  // positions come from LDS.  (q_lo, q_hi and the list base are multiples of 8 entries: the loads are aligned.)
  const uint32_t* const poi_m = poi + (int64_t)m * cap;
  const uint32_t* const poj_m = poj + (int64_t)m * cap;
  const float* const pd_m = pd + (int64_t)m * cap;
  const float* const pgd_m = pgd + (int64_t)m * cap;
  const float* const pgd2_m = pgd2 ? pgd2 + (int64_t)m * cap : nullptr;
  for (int p0 = q_lo + 8 * tid; p0 < q_hi; p0 += 8 * FIN_NT) {
    u32x4 vi[2], vj[2];
    f32x4 vd[2], vg[2], vg2[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      vi[u] = *reinterpret_cast<const u32x4*>(poi_m + p0 + 4 * u);
      vj[u] = *reinterpret_cast<const u32x4*>(poj_m + p0 + 4 * u);
      vd[u] = *reinterpret_cast<const f32x4*>(pd_m + p0 + 4 * u);
      vg[u] = *reinterpret_cast<const f32x4*>(pgd_m + p0 + 4 * u);
      vg2[u] = pgd2_m ? *reinterpret_cast<const f32x4*>(pgd2_m + p0 + 4 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    int ci = -1;
    float cx = 0.f, cy = 0.f, cz = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = (int)(vi[e >> 2][e & 3] / (F * 4));
      const int j = (int)(vj[e >> 2][e & 3] / (F * 4));
      if (i != ci) {
        if (ci >= 0) {
          atomicAdd(&g[3 * ci], fmd_to_fixed(cx, bad));
          atomicAdd(&g[3 * ci + 1], fmd_to_fixed(cy, bad));
          atomicAdd(&g[3 * ci + 2], fmd_to_fixed(cz, bad));
        }
        ci = i;
        cx = cy = cz = 0.f;
      }
      if (i != j) {  // pad entries carry nothing
        const float d = fmaxf(vd[e >> 2][e & 3], 1e-8f);
        const float gdd = vg[e >> 2][e & 3] + (pgd2_m ? vg2[e >> 2][e & 3] : 0.f);
        const float s = gdd / d;
        const float dx = s * (spos[3 * j] - spos[3 * i]);
        const float dy = s * (spos[3 * j + 1] - spos[3 * i + 1]);
        const float dz = s * (spos[3 * j + 2] - spos[3 * i + 2]);
        atomicAdd(&g[3 * j], fmd_to_fixed(dx, bad));
        atomicAdd(&g[3 * j + 1], fmd_to_fixed(dy, bad));
        atomicAdd(&g[3 * j + 2], fmd_to_fixed(dz, bad));
        cx -= dx;
        cy -= dy;
        cz -= dz;
      }
    }
    if (ci >= 0) {
      atomicAdd(&g[3 * ci], fmd_to_fixed(cx, bad));
      atomicAdd(&g[3 * ci + 1], fmd_to_fixed(cy, bad));
      atomicAdd(&g[3 * ci + 2], fmd_to_fixed(cz, bad));
    }
  }
  if (half == 0 && mol.stride == 1) {
    if (wv < 4) {   // (the first four waves, whatever the workgroup size: the energy's summation order is part of the result)
      float e = 0.f;
      for (int i = tid; i < n; i += 256) e += e_node[a0 + i];
      e = fmd_wave_sum(e);
      if (lane == 0) esum[wv] = e;
    }
  } else if (half == 0) {   // a group: one wave per molecule in turn (small molecules: at most 64 beads each)
    const MolView one{mol.ptr, mol.n_real, 1};
    for (int j = wv; j < mol.stride; j += FIN_NT / 64) {
      const int64_t mj = (int64_t)m * mol.stride + j;
      if (mj >= mol.n_real) break;
      const int64_t b0 = grp_begin(one, mj), b1 = grp_begin(one, mj + 1);
      float e = 0.f;
      for (int64_t i = b0 + lane; i < b1; i += 64) e += e_node[i];
      e = fmd_wave_sum(e);
      if (lane == 0) energy[mj] = e;
    }
  }
  if (bad) bad_s = 1;  // (benign race: every writer stores 1)
  __syncthreads();
  const bool poisoned = bad_s != 0;
  for (int i = tid; i < 3 * n; i += FIN_NT)
    unsafeAtomicAdd(&forces[3 * a0 + i],
                    poisoned ? __builtin_nanf("")
                             : -fmd_from_fixed(((gall[i] + gall[3 * n + i]) + gall[6 * n + i]) + gall[9 * n + i]));
  if (poisoned && tid == 0) atomicMin(&status[3], FMD_ENONFINITE);
  if (half == 0 && tid == 0 && mol.stride == 1) energy[m] = ((esum[0] + esum[1]) + esum[2]) + esum[3];
  // the call's status words from pairs_kernel's per-molecule reports (one wave of the first workgroup; a bead-bound
  // violation outranks a capacity overflow)
  if (blockIdx.x == 0 && blockIdx.y == 0 && wv == 0) {
    int err = 0, need = 0, real = 0;
    for (int q = lane; q < (int)gridDim.y; q += 64) {
      const int e = report[3 * q];
      err = (e == FMD_EINVAL || err == FMD_EINVAL) ? FMD_EINVAL : (e != 0 ? e : err);
      need = max(need, report[3 * q + 1]);
      real += report[3 * q + 2];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int e = __shfl_xor(err, o, 64);
      err = (e == FMD_EINVAL || err == FMD_EINVAL) ? FMD_EINVAL : (e != 0 ? e : err);
      need = max(need, __shfl_xor(need, o, 64));
      real += __shfl_xor(real, o, 64);
    }
    if (lane == 0) {
      status[0] = err;
      status[1] = need;
      status[2] = real;
      if (err != 0) atomicMin(&status[3], err);  // sticky: only the caller clears it
    }
  }
}

}  // namespace

namespace fmd_eng {

constexpr size_t LDS_PAIRS_MAX = 128 * 1024;

int setup_pairs_attrs() {
  if (hipFuncSetAttribute((const void*)pairs_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)LDS_PAIRS_MAX) != hipSuccess)
    return FMD_ELAUNCH;
  return hipFuncSetAttribute((const void*)pairs_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)LDS_PAIRS_MAX) == hipSuccess ? FMD_OK : FMD_ELAUNCH;
}

int launch_pairs(const PairsArgs& a, int64_t n_mol, hipStream_t s) {
  // positions, counts / offsets / flags, [n][n / 64] 64-bit hit masks (53.5 KB at 535 beads), then the entry staging
  // area: as much of the worst-case list as fits in 128 KB (the list of a 1ENH-sized molecule takes 32 KB)
  const int64_t nb = a.max_beads;
  const size_t base = (size_t)(8 * nb + 2 + 2 * ((nb + 63) / 64) * nb) * sizeof(float);
  const size_t lds = base + (size_t)a.cap * 4 < LDS_PAIRS_MAX ? base + (size_t)a.cap * 4 : LDS_PAIRS_MAX;
  const int ent_cap = (int)((lds - base) / 4);
  // Two workgroups per molecule build the masks and offsets redundantly and fill half of the entries each; below ~100
  // beads the fill is a small part of the kernel and the second workgroup only doubles the launch (1024 molecules of
  // 35 beads: 70 us with two, see profiles/r05_*small_molecules*)
  const unsigned parts = a.max_beads <= 96 ? 1u : 2u;
  if (a.mol.stride > 1)
    pairs_kernel<true><<<dim3(parts, (unsigned)n_mol), 1024, lds, s>>>(a.pos, a.mol, a.rc * a.rc, a.rc, a.cap, a.poi, a.poj,
                                                                 a.pd, a.pc, a.pdc, a.npairs, a.report, a.max_beads,
                                                                 a.forces, ent_cap, a.extra_row, a.queue);
  else
    pairs_kernel<false><<<dim3(parts, (unsigned)n_mol), 1024, lds, s>>>(a.pos, a.mol, a.rc * a.rc, a.rc, a.cap, a.poi, a.poj,
                                                                  a.pd, a.pc, a.pdc, a.npairs, a.report, a.max_beads,
                                                                  a.forces, ent_cap, a.extra_row, a.queue);
  return hipGetLastError() == hipSuccess ? FMD_OK : FMD_ELAUNCH;
}

int launch_node_embed(const int64_t* types, const float* emb, const float* emb_x1, float* x, float* x1,
                      int64_t n_nodes, const SchedArgs& sa, hipStream_t s) {
  const int64_t want = (n_nodes * 32 + 255) / 256;
  node_embed_kernel<<<(unsigned)(want < 4096 ? want : 4096) + 1, 256, 0, s>>>(types, emb, emb_x1, x, x1, n_nodes, sa);
  return hipGetLastError() == hipSuccess ? FMD_OK : FMD_ELAUNCH;
}

int launch_finish(const FinishArgs& a, int64_t n_mol, hipStream_t s) {
  // four per-wave copies of dE/dpos in 64-bit fixed point, then the molecule's positions
  finish_kernel<<<dim3(2, (unsigned)n_mol), FIN_NT, 12 * (size_t)a.max_beads * sizeof(unsigned long long) + 3 * (size_t)a.max_beads * sizeof(float), s>>>(
      a.pos, a.mol, a.poi, a.poj, a.pd, a.pgd, a.pgd2, a.npairs, a.cap, a.e_node, a.energy, a.forces, a.max_beads,
      a.report, a.status);
  return hipGetLastError() == hipSuccess ? FMD_OK : FMD_ELAUNCH;
}

}  // namespace fmd_eng
