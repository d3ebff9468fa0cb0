// Fused engine, node kernels: lin2 -> tanh -> lin -> residual -> next block's lin1 (models/schnet.py:497-548,644,719),
// the output network with its own backward (schnet.py:346, mlp.py:41-57; W16A16: gptq.py:266-306) and the node side
// of the backward pass.  32-row tiles, chained 128x128 products on the split-operand f16 MFMA.
#include "fmd_engine.h"

namespace {
// ---- register-resident, transposed-product forms of node_fwd / node_bwd (split-operand engines).
// The streamed 32-row chains above are bound by latency and by VALU work around the GEMMs, not by the matrix
// pipe: every stage waits for its operand fragments from L2 and every tile for its rows from HBM; every output
// element costs a scalar tanh / split / 4-byte LDS write / 4-byte global store, every operand element a byte
// permute.  Here
//  * one workgroup per CU keeps the three operands of its chain in registers for all of its tiles (4 waves x 192
//    registers; the kernel runs one wave per SIMD, so the 512-register budget is there) and requests the NEXT
//    tile's rows before it runs the current one;
//  * a stage computes the TRANSPOSED product  D[m = output column][n = row] = sum_k W[m][k] act[n][k]: the
//    packed operand fragment is the MFMA A operand, the activation rows come from LDS as the B operand.  A lane
//    then owns one row and, per group of four registers, four CONSECUTIVE output columns
//    (register r = 4 q + e  <->  column 32 w + 8 q + 4 h + e): global loads / stores are 16 bytes wide, the bias /
//    tanh / split arithmetic runs on packed pairs, and the next stage's operand goes to LDS as two fp16 planes
//    (hi, lo) with 8-byte writes that the next GEMM reads back as ready 16-byte fragments (no permutes).
constexpr int NLD = 144;              // halves per row of an activation plane (288 B: the b128 fragment reads are conflict free)
constexpr int NPLANE = 32 * NLD;      // halves of one plane; a tile is the hi plane followed by the lo plane
struct RowsPre {
  f32x4 s0[4], s1[4];
  int er;  // row of the slab buffer with one more partial of this thread's row (sched_body), or -1
};
// Loads are UNCONDITIONAL (rows past the end are clamped to the last row and zeroed at commit; a missing second slab
// re-reads the first) and stores go to arrays padded to whole tiles: with a fixed number of memory operations per tile
// the compiler can wait for "all but the N youngest" at the top of the next tile instead of vmcnt(0), which -- the
// counter is in order -- would also wait for the tile's last stores to reach memory (~1.5 us per tile).
template <bool XT>
__device__ __forceinline__ void rows_fetch(RowsPre& p, const float* __restrict__ src, int64_t slab_stride,
                                           int n_slabs, const int* __restrict__ extra_row, int64_t row0,
                                           int64_t n_rows, int tid) {
  const int r = tid >> 3, c0 = (tid & 7) * 4;
  const int64_t row = row0 + r < n_rows ? row0 + r : n_rows - 1;
  p.er = -1;
  if (XT) p.er = extra_row[row];
  const float* q = src + row * F + c0;
  const int64_t s1_off = n_slabs > 1 ? slab_stride : 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    p.s0[i] = *reinterpret_cast<const f32x4*>(q + 32 * i);
    p.s1[i] = *reinterpret_cast<const f32x4*>(q + 32 * i + s1_off);
  }
}
__device__ __forceinline__ void planes_store4(_Float16* tile, int r, int c, const f32x4& v) {
  uint32_t h0, l0, h1, l1;
  split_pair(f32x2{v[0], v[1]}, h0, l0);
  split_pair(f32x2{v[2], v[3]}, h1, l1);
  *reinterpret_cast<uint2*>(tile + r * NLD + c) = uint2{h0, h1};
  *reinterpret_cast<uint2*>(tile + NPLANE + r * NLD + c) = uint2{l0, l1};
}
// same summation order as rows_to_lds: ((s0 + s1) + s2) + ...
__device__ __forceinline__ void rows_commit(_Float16* tile, const RowsPre& p, const float* __restrict__ src,
                                            int64_t slab_stride, int n_slabs, const int* __restrict__ extra_row,
                                            int64_t row0, int64_t n_rows, int tid) {
  const int r = tid >> 3, c0 = (tid & 7) * 4;
  // a molecule whose tiles were spread over one more workgroup (sched_body) has one more partial per row
  const bool in = row0 + r < n_rows;
  const int er = (extra_row && in) ? p.er : -1;
  const float k0 = in ? 1.0f : 0.0f, k1 = (in && n_slabs > 1) ? 1.0f : 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + 32 * i;
    f32x4 v = p.s0[i] * k0;
    v += p.s1[i] * k1;
    if (n_slabs > 2 && row0 + r < n_rows) {
      const float* q = src + (row0 + r) * F + c;
      for (int s = 2; s < n_slabs; ++s) v += *reinterpret_cast<const f32x4*>(q + s * slab_stride);
    }
    if (er >= 0) v += *reinterpret_cast<const f32x4*>(src + (int64_t)er * F + c);
    planes_store4(tile, r, c, v);
  }
}
// this lane's 16 columns of one row of a [N,128] array (zeros when the row does not exist)
__device__ __forceinline__ f32x16 load_cols(const float* __restrict__ src, int64_t row, int64_t n_rows, int w, int h) {
  f32x16 v;
  const int64_t rc = row < n_rows ? row : n_rows - 1;  // clamped: the load is unconditional (see rows_fetch)
  const float k = row < n_rows ? 1.0f : 0.0f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(src + rc * F + 32 * w + 8 * q + 4 * h);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * q + e] = t[e] * k;
  }
  return v;
}
// (the arrays the node kernels write are padded to a multiple of 32 rows: carve_ws)
__device__ __forceinline__ void store_cols(float* __restrict__ dst, int64_t row, int w, int h, const f32x16& v) {
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<f32x4*>(dst + row * F + 32 * w + 8 * q + 4 * h) =
        f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}
__device__ __forceinline__ void cols_to_planes(_Float16* tile, const f32x16& v, int w, int lane) {
  const int r = lane & 31, c = 32 * w + 4 * (lane >> 5);
#pragma unroll
  for (int q = 0; q < 4; ++q)
    planes_store4(tile, r, c + 8 * q, f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]});
}
// three accumulators: with one wave per SIMD two dependent MFMAs in a row would stall on each other
__device__ __forceinline__ f32x16 gemm_cols_frag(const _Float16* tile, const BFrag& wf, int lane) {
  f32x16 a1 = zero16(), a2 = zero16(), a3 = zero16();
  const _Float16* ph = tile + (lane & 31) * NLD + 8 * (lane >> 5);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const f16x8 bh = *reinterpret_cast<const f16x8*>(ph + 16 * s);
    const f16x8 bl = *reinterpret_cast<const f16x8*>(ph + NPLANE + 16 * s);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf.h[s], bh, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf.h[s], bl, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf.l[s], bh, a3, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) a1[r] += (a2[r] + a3[r]) * SPLIT_INV;
  return a1;
}

__global__ __launch_bounds__(256) void node_fwd_res_kernel(NodeFwdArgs a) {
  __shared__ __attribute__((aligned(16))) _Float16 bufX[2 * NPLANE];
  __shared__ __attribute__((aligned(16))) _Float16 bufY[2 * NPLANE];
  __shared__ __attribute__((aligned(16))) _Float16 bufZ[2 * NPLANE];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, nl = lane & 31;
  const int64_t n_tiles = (a.n_nodes + 31) / 32;
  int64_t tile = blockIdx.x;
  if (tile >= n_tiles) return;
  BFrag w1, w2, w3;
  RowsPre pre;
  rows_fetch<true>(pre, a.slabs, a.slab_stride, a.n_slabs, a.extra_row, tile * 32, a.n_nodes, tid);
  f32x16 xo = load_cols(a.x, tile * 32 + nl, a.n_nodes, w, h);
  load_bfrag(w2, a.l2, w, lane);
  load_bfrag(w3, a.l3, w, lane);
  load_bfrag(w1, a.l1, w, lane);
  const f32x16 b2 = load_cols(a.b2, 0, 1, w, h), b3 = load_cols(a.b3, 0, 1, w, h);
  // The operands must have landed BEFORE the loop: left to the compiler, the wait for them sits behind the first
  // barrier of the loop body, i.e. right after the next tile's rows are requested, in every iteration -- and the
  // prefetch overlaps nothing (s_waitcnt vmcnt(0), expcnt / lgkmcnt untouched).
  __builtin_amdgcn_s_waitcnt(0x0F70);
  for (; tile < n_tiles; tile += gridDim.x) {
    const int64_t row = tile * 32 + nl, next = tile + gridDim.x;
    rows_commit(bufX, pre, a.slabs, a.slab_stride, a.n_slabs, a.extra_row, tile * 32, a.n_nodes, tid);
    const f32x16 x_old = xo;
    {  // (past the last tile: the last tile again, unused)
      const int64_t nx = next < n_tiles ? next : n_tiles - 1;
      rows_fetch<true>(pre, a.slabs, a.slab_stride, a.n_slabs, a.extra_row, nx * 32, a.n_nodes, tid);
      xo = load_cols(a.x, nx * 32 + nl, a.n_nodes, w, h);
    }
    lds_barrier();
    f32x16 acc = gemm_cols_frag(bufX, w2, lane);  // conv = agg lin2^T + b2
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 t = tanh_fast2(f32x2{acc[r], acc[r + 1]} + f32x2{b2[r], b2[r + 1]});
      acc[r] = t[0];
      acc[r + 1] = t[1];
    }
    store_cols(a.t_out, row, w, h, acc);
    cols_to_planes(bufY, acc, w, lane);
    lds_barrier();
    acc = gemm_cols_frag(bufY, w3, lane);  // y = tanh(conv) lin^T + b3 ; x += y
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = x_old[r] + (acc[r] + b3[r]);
    store_cols(a.x, row, w, h, acc);
    cols_to_planes(bufZ, acc, w, lane);
    lds_barrier();
    const f32x16 x1 = gemm_cols_frag(bufZ, w1, lane);  // x1 = x lin1^T
    store_cols(a.x1_out, row, w, h, x1);
    // no barrier here: bufX / bufY / bufZ are next written after the first / second / third barrier of the next
    // tile, each of which every wave reaches only after its last read of that buffer
  }
}

template <bool HAS_SLABS>
__global__ __launch_bounds__(256) void node_bwd_res_kernel(NodeBwdArgs a) {
  __shared__ __attribute__((aligned(16))) _Float16 bufX[2 * NPLANE];
  __shared__ __attribute__((aligned(16))) _Float16 bufY[2 * NPLANE];
  __shared__ __attribute__((aligned(16))) _Float16 bufZ[2 * NPLANE];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, nl = lane & 31;
  const int64_t n_tiles = (a.n_nodes + 31) / 32;
  int64_t tile = blockIdx.x;
  if (tile >= n_tiles) return;
  BFrag w1, w3, w2;
  RowsPre pre;
  f32x16 t_nx, gx_nx;
  auto fetch = [&](int64_t t) {
    const int64_t row = t * 32 + nl;
    t_nx = load_cols(a.t, row, a.n_nodes, w, h);
    if (HAS_SLABS) {
      gx_nx = load_cols(a.gx, row, a.n_nodes, w, h);
      rows_fetch<true>(pre, a.slabs, a.slab_stride, a.n_slabs, a.extra_row, t * 32, a.n_nodes, tid);
    } else {
      rows_fetch<false>(pre, a.gx, 0, 1, nullptr, t * 32, a.n_nodes, tid);
    }
  };
  fetch(tile);
  if (HAS_SLABS) load_bfrag(w1, a.l1t, w, lane);
  load_bfrag(w3, a.l3t, w, lane);
  load_bfrag(w2, a.l2t, w, lane);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // operands landed before the loop (see node_fwd_res_kernel)
  for (; tile < n_tiles; tile += gridDim.x) {
    const int64_t row = tile * 32 + nl, next = tile + gridDim.x;
    const f32x16 t_pre = t_nx;
    f32x16 gx_pre;
    if (HAS_SLABS) {
      gx_pre = gx_nx;
      rows_commit(bufX, pre, a.slabs, a.slab_stride, a.n_slabs, a.extra_row, tile * 32, a.n_nodes, tid);
    } else {
      rows_commit(bufY, pre, a.gx, 0, 1, nullptr, tile * 32, a.n_nodes, tid);
    }
    fetch(next < n_tiles ? next : n_tiles - 1);  // (past the last tile: the last tile again, unused)
    f32x16 acc;
    if (HAS_SLABS) {  // gx += (sum of gx1 partials) lin1
      lds_barrier();
      acc = gemm_cols_frag(bufX, w1, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = gx_pre[r] + acc[r];
      store_cols(a.gx, row, w, h, acc);
      cols_to_planes(bufY, acc, w, lane);
    }
    lds_barrier();
    acc = gemm_cols_frag(bufY, w3, lane);  // d/d tanh(conv) = gx lin
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] *= 1.0f - t_pre[r] * t_pre[r];
    cols_to_planes(bufZ, acc, w, lane);
    lds_barrier();
    acc = gemm_cols_frag(bufZ, w2, lane);  // d/d agg = gconv lin2
    store_cols(a.gagg, row, w, h, acc);
    if (!HAS_SLABS) lds_barrier();  // bufY is refilled at the top of the next tile
  }
}

// W16A16 output network in the transposed form: one fp16 plane (round to nearest even, the value later layers and the
// backward see is written back to v); its GEMMs are one MFMA per k-step against the hi record (= Linear.weight.half())
__device__ __forceinline__ void cols_to_plane_half(_Float16* tile, f32x16& v, int w, int lane) {
  _Float16* p = tile + (lane & 31) * NLD + 32 * w + 4 * (lane >> 5);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f16x2 h0 = __builtin_convertvector(f32x2{v[4 * q], v[4 * q + 1]}, f16x2);
    const f16x2 h1 = __builtin_convertvector(f32x2{v[4 * q + 2], v[4 * q + 3]}, f16x2);
    v[4 * q] = (float)h0[0]; v[4 * q + 1] = (float)h0[1];
    v[4 * q + 2] = (float)h1[0]; v[4 * q + 3] = (float)h1[1];
    uint32_t u0, u1;
    __builtin_memcpy(&u0, &h0, 4);
    __builtin_memcpy(&u1, &h1, 4);
    *reinterpret_cast<uint2*>(p + 8 * q) = uint2{u0, u1};
  }
}

__device__ __forceinline__ f32x16 gemm_cols_half(const _Float16* tile, const BFrag& wf, int lane) {
  f32x16 acc = zero16();
  const _Float16* ph = tile + (lane & 31) * NLD + 8 * (lane >> 5);
#pragma unroll
  for (int s = 0; s < 8; ++s)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf.h[s], *reinterpret_cast<const f16x8*>(ph + 16 * s), acc, 0, 0, 0);
  return acc;
}

// node_out in the transposed form (see node_fwd_res_kernel); its operands do not fit the register file, so they stay
// double-buffered: the next GEMM's operand is requested (L2) before the current GEMM runs.
// The output network ends in Linear(w, 1) without bias and the backward pass starts from dE/de = 1, so its last layer
// is a dot product of each row with ONE weight vector and the first backward stage is that vector times (1 - h^2):
// both run on the registers that hold the last hidden layer (no GEMM, no operand, no LDS tile; they were two padded
// 128 x 128 stages of eight, each with its 64 KB operand streamed from L2).
template <int NUM_OUT, bool HOUT>
__global__ __launch_bounds__(256) void node_out_res_kernel(NodeOutArgs a) {
  __shared__ __attribute__((aligned(16))) _Float16 bufX[2 * NPLANE];
  __shared__ __attribute__((aligned(16))) _Float16 bufY[2 * NPLANE];
  __shared__ float e_part[2][4][32];  // per-wave partial energies of a tile's rows (double-buffered over tiles)
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, nl = lane & 31;
  const int64_t n_tiles = (a.n_nodes + 31) / 32;
  int64_t tile = blockIdx.x;
  if (tile >= n_tiles) return;
  constexpr int NH = NUM_OUT - 1;      // hidden layers of the output network (tanh, with bias)
  constexpr int NOPS = 2 + 2 * NH;     // lin2, lin, po[0..NH-1], pot[NH-1..0]
  const float* ops[NOPS];
  ops[0] = a.l2;
  ops[1] = a.l3;
#pragma unroll
  for (int k = 0; k < NH; ++k) {
    ops[2 + k] = a.po[k];
    ops[2 + NH + k] = a.pot[NH - 1 - k];
  }
  // Operand idx lives in buffer idx & 1.  The request for operand idx + 2 goes out right BEHIND GEMM idx, into the
  // buffer that GEMM has just read: it then has the whole next stage plus this stage's epilogue to arrive (requested
  // at the start of GEMM idx + 1, as before, it had one GEMM: every stage waited ~1 us for its operand).  All requests
  // are unconditional (past the last tile they re-read operands 0 and 1) so that the compiler can count them.
  BFrag fb0, fb1;
  auto gemm = [&](auto idx_c, const _Float16* t) -> f32x16 {
    constexpr int idx = decltype(idx_c)::value;
    BFrag& cur = (idx & 1) ? fb1 : fb0;
    f32x16 r;
    if constexpr (HOUT && idx >= 2) r = gemm_cols_half(t, cur, lane);
    else r = gemm_cols_frag(t, cur, lane);
    load_bfrag(cur, ops[(idx + 2) % NOPS], w, lane);  // NOPS is even: the next tile's operands 0, 1 land in fb0, fb1
    return r;
  };
  auto put = [&](_Float16* t, f32x16& v) {  // the output network's activations
    if (HOUT) cols_to_plane_half(t, v, w, lane);
    else cols_to_planes(t, v, w, lane);
  };
  RowsPre pre;
  rows_fetch<true>(pre, a.slabs, a.slab_stride, a.n_slabs, a.extra_row, tile * 32, a.n_nodes, tid);
  f32x16 xo = load_cols(a.x, tile * 32 + nl, a.n_nodes, w, h);
  load_bfrag(fb0, ops[0], w, lane);
  load_bfrag(fb1, ops[1], w, lane);
  const f32x16 b2 = load_cols(a.b2, 0, 1, w, h), b3 = load_cols(a.b3, 0, 1, w, h);
  // the hidden layers' biases stay in registers: loaded inside the tile loop, each stage's wait for its 64 bytes of
  // bias was a wait for every operand request in flight as well (the counter is in order)
  f32x16 bo_r[NH];
#pragma unroll
  for (int k = 0; k < NH; ++k) bo_r[k] = load_cols(a.bo[k], 0, 1, w, h);
  // the last layer's weight vector, this lane's 16 columns (zero beyond the layer's width; W16A16: its fp16 values)
  const f32x16 wl = load_cols(a.w_last, 0, 1, w, h);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // everything requested so far has landed (see node_fwd_res_kernel)
  int par = 0;
  for (; tile < n_tiles; tile += gridDim.x, par ^= 1) {
    const int64_t row = tile * 32 + nl, next = tile + gridDim.x;
    const bool ok = row < a.n_nodes;
    rows_commit(bufX, pre, a.slabs, a.slab_stride, a.n_slabs, a.extra_row, tile * 32, a.n_nodes, tid);
    const f32x16 x_old = xo;
    lds_barrier();
    f32x16 acc = gemm(std::integral_constant<int, 0>{}, bufX);
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 t = tanh_fast2(f32x2{acc[r], acc[r + 1]} + f32x2{b2[r], b2[r + 1]});
      acc[r] = t[0];
      acc[r + 1] = t[1];
    }
    store_cols(a.t_out, row, w, h, acc);
    cols_to_planes(bufY, acc, w, lane);
    lds_barrier();
    acc = gemm(std::integral_constant<int, 1>{}, bufY);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = ok ? x_old[r] + (acc[r] + b3[r]) : 0.f;
    put(bufX, acc);
    lds_barrier();
    // ---- output network forward: tanh hidden layers ...
    f32x16 hk[NH];
    _Float16* cur = bufX;
    _Float16* nxt = bufY;
    static_for<0, NH>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      hk[k] = gemm(std::integral_constant<int, 2 + k>{}, cur);
      const f32x16& bk = bo_r[k];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 t = tanh_fast2(f32x2{hk[k][r], hk[k][r + 1]} + f32x2{bk[r], bk[r + 1]});
        hk[k][r] = t[0];
        hk[k][r + 1] = t[1];
      }
      if constexpr (k + 1 < NH) {
        put(nxt, hk[k]);  // W16A16: hk <- f16(tanh), what the next layer and the backward see
        lds_barrier();
        _Float16* tmp = cur; cur = nxt; nxt = tmp;
      } else if (HOUT) {  // the last hidden layer feeds no GEMM: only its fp16 rounding point
#pragma unroll
        for (int r = 0; r < 16; ++r) hk[k][r] = (float)(_Float16)hk[k][r];
      }
    });
    // ---- ... last layer: e = <w_last, h>, this lane's 16 columns, then the other half of the wave; the four waves'
    //      parts meet in LDS behind the next barrier
    {
      float e0 = 0.f, e1 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        e0 = __builtin_fmaf(wl[r], hk[NH - 1][r], e0);
        e1 = __builtin_fmaf(wl[r + 1], hk[NH - 1][r + 1], e1);
      }
      const float e = fmd_xor32_sum(e0 + e1);
      if (h == 0) e_part[par][w][nl] = e;
    }
    // ---- output network backward: dE/de = 1, so dE/d(pre-activation of the last hidden layer) = w_last (1 - h^2)
    f32x16 g;
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = wl[r] * (1.0f - hk[NH - 1][r] * hk[NH - 1][r]);
    put(nxt, g);  // W16A16: f16(g (1 - y^2)) before the dot (cfconv_kernels.py:1019-1023)
    lds_barrier();
    { _Float16* tmp = cur; cur = nxt; nxt = tmp; }
    if (w == 0 && h == 0 && ok)
      a.e_node[row] = ((e_part[par][0][nl] + e_part[par][1][nl]) + e_part[par][2][nl]) + e_part[par][3][nl];
    {  // The next tile's rows are requested HERE: behind the only conditional store of the tile (what is older than a
       // request does not enter the count of operations the wait for it may leave outstanding) and the remaining
       // backward stages ahead of their use.  Past the last tile: the last tile again, unused.
      const int64_t nx = next < n_tiles ? next : n_tiles - 1;
      rows_fetch<true>(pre, a.slabs, a.slab_stride, a.n_slabs, a.extra_row, nx * 32, a.n_nodes, tid);
      xo = load_cols(a.x, nx * 32 + nl, a.n_nodes, w, h);
    }
    static_for<0, NH - 1>([&](auto jc) {
      constexpr int j = decltype(jc)::value;   // operand pot[NH-1-j], then times 1 - hk[NH-2-j]^2
      g = gemm(std::integral_constant<int, 2 + NH + j>{}, cur);
#pragma unroll
      for (int r = 0; r < 16; ++r) g[r] *= 1.0f - hk[NH - 2 - j][r] * hk[NH - 2 - j][r];
      put(nxt, g);
      lds_barrier();
      _Float16* tmp = cur; cur = nxt; nxt = tmp;
    });
    g = gemm(std::integral_constant<int, NOPS - 1>{}, cur);
    store_cols(a.gx, row, w, h, g);
    lds_barrier();
  }
}
}  // namespace

namespace fmd_eng {

int launch_node_fwd(const NodeFwdArgs& a, unsigned grid, hipStream_t s) {
  node_fwd_res_kernel<<<grid, 256, 0, s>>>(a);
  return hipGetLastError() == hipSuccess ? FMD_OK : FMD_ELAUNCH;
}
int launch_node_out(const NodeOutArgs& a, int num_out, bool hout, unsigned grid, hipStream_t s) {
  if (num_out == 3 && hout) node_out_res_kernel<3, true><<<grid, 256, 0, s>>>(a);
  else if (num_out == 3) node_out_res_kernel<3, false><<<grid, 256, 0, s>>>(a);
  else if (hout) node_out_res_kernel<2, true><<<grid, 256, 0, s>>>(a);
  else node_out_res_kernel<2, false><<<grid, 256, 0, s>>>(a);
  return hipGetLastError() == hipSuccess ? FMD_OK : FMD_ELAUNCH;
}
int launch_node_bwd(const NodeBwdArgs& a, bool has_slabs, unsigned grid, hipStream_t s) {
  if (has_slabs) node_bwd_res_kernel<true><<<grid, 256, 0, s>>>(a);
  else node_bwd_res_kernel<false><<<grid, 256, 0, s>>>(a);
  return hipGetLastError() == hipSuccess ? FMD_OK : FMD_ELAUNCH;
}

}  // namespace fmd_eng
