// Internal header of the fused CGSchNet engine (fmd_engine_*.hip): sizes, kernel argument records, the launchers each
// translation unit exports to the host side (fmd_engine_host.hip), and the device helpers the kernels share.
// Nothing here is part of the C ABI (include/flashmd_hip.h).
//
// Design (MI355X first):
//  * the radius graph is symmetric and the filter W_ij = filt(d_ij) * C(d_ij) depends on the pair
//    only, so work is organised over UNDIRECTED pairs i<j: the filter network runs once per pair and
//    feeds both directions (half the MFMA work of a per-edge formulation, same arithmetic);
//  * one workgroup owns a share of one molecule's pair tiles (32 pairs) and keeps the molecule's
//    aggregate (<= 283 beads x 128 features, fp32) in LDS until one final flush; 8 waves, specialised:
//    producers (rbf, filter layer 0, tanh -> hidden tile in LDS) and consumers (filter layer 1, row
//    gathers, aggregate update), each owning a 32-feature slice whose weight fragments live in VGPRs.
//    Every aggregate element is touched by exactly one wave in program order => bit-reproducible sums;
//  * contractions: precision FMD_PREC_F32 = split-operand f16 MFMA (x = hi + lo, three products, fp32
//    accumulation, 2e-7 = the error of the fp32 MFMA at 3/16 of its cycles; bounded operands carry an unscaled low
//    word and share ONE accumulator, only gW keeps lo = f16((x - hi) * 2048) and a second one, see split_pair_u /
//    split_pair); FMD_PREC_W16A16 = one product of fp16-rounded operands;
//  * backward gets d/d(distance) from the forward-mode product z' = W0 rbf'(d) and u = gW W1 (two extra slice
//    GEMMs), so the only per-pair OUTPUT that reaches HBM is one float (dE/dd) per pair.  The filter values W of
//    blocks 1.. are kept by the forward kernel (the filter stash, EdgeArgs::wst: one float per pair and feature)
//    and read back once instead of recomputing layer 1 (FMD_PREC_F32; block 0 and W16A16 recompute / need none).
#pragma once
#include <string.h>
#include <type_traits>

#include "fmd_common.h"

// Diagnostic build only (`make stamps` -> build_stamps/libflashmd_hip_stamps.so, never loaded by the package): every
// wave accumulates the cycles it spends in each phase of the pair kernels' tile loop; tools/phase_stamps.py reads them.
#ifdef FMD_PHASE_STAMPS
#define FMD_NSTAMP 12
#define FMD_STAMP_DECL unsigned long long ts_[FMD_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev_ = 0;
#define FMD_STAMP_START tprev_ = __builtin_readcyclecounter();
#define FMD_MARK(k)                                                  \
  {                                                                  \
    const unsigned long long now_ = __builtin_readcyclecounter();    \
    ts_[k] += now_ - tprev_;                                         \
    tprev_ = now_;                                                   \
  }
#define FMD_STAMP_FLUSH(args_, wave_)                                                                        \
  if ((args_).stamps && (threadIdx.x & 63) == 0) {                                                           \
    unsigned long long* o_ = (args_).stamps + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (wave_)) * FMD_NSTAMP; \
    for (int k_ = 0; k_ < FMD_NSTAMP; ++k_) o_[k_] = ts_[k_];                                                \
  }
#define FMD_DBG_SKIP(args_, bit_) (((args_).dbg >> (bit_)) & 1)
#else
#define FMD_DBG_SKIP(args_, bit_) false
#define FMD_STAMP_DECL
#define FMD_STAMP_START
#define FMD_MARK(k)
#define FMD_STAMP_FLUSH(args_, wave_)
#endif

namespace fmd_eng {

constexpr int F = 128;            // hidden = filters
constexpr int LDA = 132;          // floats per row of the tile region (its size is kept: 32 x 132 floats)
constexpr int TILE_FLOATS = 32 * LDA;
constexpr size_t MAX_LDS_BYTES = 160 * 1024;
constexpr int MAX_LDS_ROWS = 283; // beads per molecule whose aggregate fits the 160 KB LDS next to the tile buffer
constexpr int MAX_LDS_ROWS_FS = 535;  // ... in feature-split mode (64 features per workgroup, any num_rbf)
constexpr int MAX_RBF = 128;                // Gaussians of the basis the pair kernels take (k-steps of 16: NS0 = 2, 4 or 8)
constexpr int Q0_FLOATS = 4 * 8 * 64 * 4;   // split operand, K padded to 64 or 128: [wave][k-step of 16][lane] x 8 halves (16 B)
// record of (wave w, k-step s) of a packed layer-0 operand: 4 k-steps per wave up to 64 Gaussians, 8 beyond
template <int NS0>
constexpr int q0_rec(int w, int s) { return w * (NS0 > 4 ? 8 : 4) + s; }
constexpr int Q1_FLOATS = 4 * 8 * 64 * 4;   // same for K = 128
constexpr int SLICE = 2 * Q1_FLOATS;        // floats of one packed 128x128 node operand: hi record, lo record

// A pair-kernel workgroup walks up to two WORK ITEMS (sched_body writes them once per force evaluation): item =
// ITEM_INTS ints {molecule (-1: none), first tile, end tile, tile stride, row of the slab buffer that takes the item's
// aggregate, 0, 0, 0}; workgroup (molecule m, split s) reads items[(m * n_split + s) * 2 + 0 / 1].
constexpr int ITEM_INTS = 8;
constexpr int WST_TILE_FLOATS = 32 * F;     // one tile of the filter stash (EdgeArgs::wst): 32 entries x 128 features

// Molecule GROUPS (round 6).  Every engine kernel that walks "a molecule" -- its pair list, its aggregate in LDS, its work
// items -- really walks a group of `mol_stride` consecutive molecules of the batch: group g = molecules [g stride,
// (g + 1) stride), beads [mol_ptr[g stride], mol_ptr[(g + 1) stride]).  stride = 1 is the plain case.  Batches of many
// small molecules run with stride > 1 (run_engine): a 10-bead molecule is 45 pairs = two tiles, and one workgroup item,
// one aggregate staging and one pairs_kernel workgroup of 1024 threads per such molecule was all fixed cost.  Rows of
// different molecules of a group never meet in a pair (pairs_kernel tests the molecule index), so the group's pair list is
// block diagonal and everything downstream is unchanged.
struct MolView {
  const int64_t* ptr;      // the batch's mol_ptr [n_mol_real + 1]
  int64_t n_real;          // molecules of the batch
  int stride;              // molecules per group
};
__device__ __forceinline__ int64_t grp_begin(const MolView& v, int64_t g) {
  const int64_t m = g * v.stride;
  return v.ptr[m < v.n_real ? m : v.n_real];
}

struct EdgeArgs {
  const int* items;        // [n_mol * n_split][2][ITEM_INTS]
  const uint32_t* poi;     // centre row byte offsets (i * 512)
  const uint32_t* poj;     // neighbour row byte offsets (j * 512)
  const float* pd;
  const float* pc;         // C(d) per pair
  const float* pdc;        // C'(d) per pair
  float* pgd;              // dE/dd per pair (backward)
  int64_t pgd_stride;      // FS mode: second dE/dd slab at pgd + pgd_stride
  bool fs;                 // feature-split mode (host dispatch only)
  const int32_t* npairs;
  MolView mol;
  const float* x1;         // [N,128] of this block
  const float* gagg;       // [N,128] dE/d(agg) (backward)
  const float* b0;
  const float *q0h, *q0l, *q1h, *q1l, *q1th, *q1tl;  // fp16 hi/lo split operands (16-byte records)
  const float *q0ch, *q0cl;                          // centre-weighted W0 (backward, shared-Gaussian form)
  float c0;                                          // cutoff / 2
  int nb;                                            // ceil(num_rbf / 8): half k-steps that hold Gaussians
  const float* centers;    // [MAX_RBF]
  float* slabs;            // [n_split + 1][N,128] partial aggregates (fwd: agg, bwd: gx1); the last: extra pieces
  int64_t slab_stride;
  float gamma, rc;
  int cap, n_split;
  int max_beads;           // the caller's bound on beads per molecule: what the dynamic LDS was sized for
  // Filter stash of this block, or null: W = filt(d) C(d) of every pair and feature as the forward kernel's consumers
  // hold it (lane = feature, registers = the tile's entries), written by edge_fwd and read back by edge_bwd, which then
  // does not recompute layer 1.  Tile t of molecule m: WST_TILE_FLOATS floats at (m * cap / 32 + t) * WST_TILE_FLOATS,
  // record [feature slice 0..3][register group q 0..3][lane 0..63] x 16 bytes = registers 4q .. 4q+3 of that lane.
  float* wst;
  // Queue mode (more work items than CUs, e.g. a thousand small molecules): one workgroup per CU, each starts with item
  // <its linear id> and draws the next from this device counter (zeroed by pairs_kernel) until n_items are taken; the item
  // records are items[2 q].  Null: slot mode, the workgroup's own two records.
  int* queue;
  int n_items;
#ifdef FMD_PHASE_STAMPS
  unsigned long long* stamps;  // diagnostic build (make stamps) only: [workgroup][8 waves][FMD_NSTAMP] cycle totals
  int dbg;                     // diagnostic build only: bit 0 = consumers skip their A-B work, bit 1 = producers skip staging
#endif
};


struct NodeFwdArgs {
  float* x;            // [N,128] in/out
  const float* slabs;  // agg partials of the previous block
  int64_t slab_stride;
  int n_slabs;
  const int* extra_row;  // [N] row of the slab buffer that holds one more partial of this row, or -1
  const float *l2, *b2, *l3, *b3;  // previous block's lin2 / lin
  float* t_out;        // tanh(conv) of the previous block (kept for backward)
  const float* l1;     // this block's lin1
  float* x1_out;
  int64_t n_nodes;
};

struct NodeOutArgs {
  float* x;
  const float* slabs;
  int64_t slab_stride;
  int n_slabs;
  const int* extra_row;
  const float *l2, *b2, *l3, *b3;
  float* t_out;
  const float* po[3];   // packed output linears
  const float* bo[3];   // biases (zero-filled when absent)
  const float* pot[3];  // packed transposes
  const float* w_last;  // [128] the last layer's weight vector (Linear(w, 1), no bias), zero beyond its width
  float* e_node;        // [N]
  float* gx;            // [N,128] dE/dx_L
  const float *l3t, *l2t;  // the last block's lin / lin2, transposed operands: its node backward is fused into node_out
  float* gagg;          // [N,128] dE/d(agg) of the last block
  int64_t n_nodes;
};

struct NodeBwdArgs {
  float* gx;             // [N,128] dE/dx_{l+1} in, dE/dx_l... (updated when slabs are given)
  const float* slabs;    // gx1 partials of block l+1
  int64_t slab_stride;
  int n_slabs;
  const int* extra_row;
  const float* l1t;      // lin1 of block l+1, transposed operand
  const float* t;        // tanh(conv) of block l
  const float *l3t, *l2t;
  float* gagg;           // [N,128] out
  int64_t n_nodes;
};


struct PairsArgs {
  const float* pos;
  MolView mol;
  float rc;
  int cap;                 // pair-list capacity per molecule (multiple of 32)
  uint32_t *poi, *poj;
  float *pd, *pc, *pdc;
  int32_t* npairs;
  int32_t* report;         // [n_mol][3] pairs_kernel -> finish_kernel
  int max_beads;
  float* forces;           // zeroed here, accumulated by finish
  int* extra_row;          // [N] set to -1 here (sched_body names the rows that have an extra partial)
  int* queue;              // [16] item counters of the pair kernels' queue mode, zeroed here
};
struct SchedArgs {
  const int32_t* npairs;
  MolView mol;
  int n_mol, n_split;
  int balanced;            // 1: even out the tiles over the workgroups (one wave of workgroups, n_mol <= 256);
                           // 2: chained -- n_wg workgroups (<= 256) take consecutive chunks of the global tile sequence
  int n_wg;                // chained form: workgroups of the pair-kernel launch (grid (1, n_wg))
  int fs;
  int64_t n_nodes;
  int* items;
  int* extra_row;
};
struct FinishArgs {
  const float* pos;
  MolView mol;
  const uint32_t *poi, *poj;
  const float* pd;
  const float* pgd;
  const float* pgd2;       // feature-split mode: second dE/dd slab, else null
  const int32_t* npairs;
  int cap;
  const float* e_node;
  float* energy;
  float* forces;
  int max_beads;
  const int32_t* report;
  int32_t* status;
};

// launchers (each returns FMD_OK / FMD_ELAUNCH / FMD_EUNSUPPORTED); setup_* raise the kernels' dynamic-LDS ceiling
// once per device
int setup_pairs_attrs();
int setup_edge_fwd_attrs();
int setup_edge_bwd_attrs();
int launch_pairs(const PairsArgs& a, int64_t n_mol, hipStream_t s);
int launch_finish(const FinishArgs& a, int64_t n_mol, hipStream_t s);
int launch_edge_fwd(const EdgeArgs& a, int num_rbf, bool half, int64_t n_mol, hipStream_t s);
int launch_edge_bwd(const EdgeArgs& a, int num_rbf, bool half, bool need_gx1, bool first, int64_t n_mol, hipStream_t s);  // a.wst != null with need_gx1: the stash form
int launch_node_embed(const int64_t* types, const float* emb, const float* emb_x1, float* x, float* x1,
                      int64_t n_nodes, const SchedArgs& sa, hipStream_t s);
int launch_node_fwd(const NodeFwdArgs& a, unsigned grid, hipStream_t s);
int launch_node_out(const NodeOutArgs& a, int num_out, bool hout, unsigned grid, hipStream_t s);
int launch_node_bwd(const NodeBwdArgs& a, unsigned grid, hipStream_t s);

}  // namespace fmd_eng

namespace {
using namespace fmd_eng;

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int tile_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// compile-time unrolled loop: the body receives std::integral_constant<int, I> (true constants keep
// every register-array index static, so nothing is demoted to scratch)
template <int I, int N, class Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

constexpr int LDH = 136;                       // halves per row of an fp16 LDS tile (272 B)
constexpr int TILE_H_BYTES = 32 * LDH * 2;     // one fp16 tile
constexpr float SPLIT_SCALE = 2048.0f, SPLIT_INV = 1.0f / 2048.0f;
typedef __fp16 hv2 __attribute__((ext_vector_type(2)));

// ---- arithmetic on pairs of values.  Packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) halves the
// instruction count, but such an instruction cannot issue while the OTHER wave of the SIMD has an MFMA in flight
// (tools/ubench/coissue.hip: a v_pk_mul_f32 stream beside an MFMA stream takes the SUM of both times, older or
// younger; plain v_fma_f32 / v_cvt_pkrtz / v_fma_mix / transcendentals overlap within ~15 %).  In the pair kernels
// the two waves of a SIMD are one producer and one consumer whose VALU phases are meant to run under each other's
// MFMA bursts, so they use the scalar pair type (FMD_SCALAR_PAIRS, set by their translation units, which are also
// compiled with -fno-slp-vectorize); the node kernels run one wave per SIMD and keep the packed form.
#ifdef FMD_SCALAR_PAIRS
struct f32x2 {
  float x, y;
  __device__ __forceinline__ float operator[](int i) const { return i ? y : x; }
};
__device__ __forceinline__ f32x2 operator*(f32x2 a, f32x2 b) { return f32x2{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ f32x2 operator+(f32x2 a, f32x2 b) { return f32x2{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ f32x2 operator-(f32x2 a, f32x2 b) { return f32x2{a.x - b.x, a.y - b.y}; }
#else
typedef float f32x2 __attribute__((ext_vector_type(2)));
#endif
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also carries a workgroup-scope release of GLOBAL
// stores: with stores outstanding it waits for vmcnt(0), and since the counter is in order that drains every prefetch
// issued before it as well (node kernels: ~2.5 us per tile).  Kernels whose workgroups exchange data through LDS only
// use this form between their stages.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr float LOG2E = 1.4426950408889634f;
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 exp2_2(f32x2 x) {
  return f32x2{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
}
// tanh = 1 - 2 rcp(e^{2x} + 1) on a pair (see fmd_tanh_fast)
__device__ __forceinline__ f32x2 tanh_fast2(f32x2 x) {
  const f32x2 e = exp2_2(x * splat2(2.0f * LOG2E)) + splat2(1.0f);
  const f32x2 rc = f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
  return splat2(1.0f) - splat2(2.0f) * rc;
}
// the same for an argument already multiplied by 2 log2(e) (the pair kernels fold that factor into C and the bias)
__device__ __forceinline__ f32x2 tanh_from_scaled2(f32x2 y) {
  const f32x2 e = exp2_2(y) + splat2(1.0f);
  const f32x2 rc = f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
  return splat2(1.0f) - splat2(2.0f) * rc;
}
// W16A16 forms (template flag HALF): operands are rounded to fp16 once (round to nearest even, as the
// reference's .half()) and a product is the single hi.hi MFMA; the lo words/tiles/fragments are unused.
template <bool HALF>
__device__ __forceinline__ void mfma_prod(const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl,
                                          f32x16& acc1, f32x16& acc2);
template <bool HALF>
__device__ __forceinline__ f32x2 comb2(const f32x16& a1, const f32x16& a2, int r) {
  if (HALF) return f32x2{a1[r], a1[r + 1]};
  return f32x2{a1[r], a1[r + 1]} + f32x2{a2[r], a2[r + 1]} * f32x2{1.0f / 2048.0f, 1.0f / 2048.0f};
}
__device__ __forceinline__ uint32_t half_pair(f32x2 x) {  // v_cvt_pk_f16_f32 (RTE)
#ifdef FMD_SCALAR_PAIRS
  const f16x2 h = {(_Float16)x.x, (_Float16)x.y};
#else
  const f16x2 h = __builtin_convertvector(x, f16x2);
#endif
  uint32_t u;
  __builtin_memcpy(&u, &h, 4);
  return u;
}
// one pair -> packed hi (2 halves) and lo (2 halves) words
__device__ __forceinline__ void split_pair(f32x2 x, uint32_t& hi, uint32_t& lo) {
  const hv2 h = __builtin_amdgcn_cvt_pkrtz(x[0], x[1]);
  const f32x2 xs = x * splat2(SPLIT_SCALE);  // (x - h) 2048 as x 2048 - h 2048: v_fma_mix reads the half directly
  const hv2 l = __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)h[0], -SPLIT_SCALE, xs[0]),
                                           __builtin_fmaf((float)h[1], -SPLIT_SCALE, xs[1]));
  __builtin_memcpy(&hi, &h, 4);
  __builtin_memcpy(&lo, &l, 4);
}
// The same with an UNSCALED low word, lo = f16(x - hi): hi + lo then carries x to ~2^-24 absolute for |x| <~ 2 (fp16
// denormals are honoured by the conversions and by the MFMA: tools/ubench/denorm_mfma.hip), which is fp32 grade for
// operands of bounded magnitude -- weights, Gaussians, C tanh -- and lets all three products of a split contraction
// share ONE accumulator (no 1/2048 recombination, 16 registers less per GEMM).  Unbounded operands (gW) keep the
// scaled form above.
__device__ __forceinline__ void split_pair_u(f32x2 x, uint32_t& hi, uint32_t& lo) {
  const hv2 h = __builtin_amdgcn_cvt_pkrtz(x[0], x[1]);
  const hv2 l = __builtin_amdgcn_cvt_pkrtz(x[0] - (float)h[0], x[1] - (float)h[1]);
  __builtin_memcpy(&hi, &h, 4);
  __builtin_memcpy(&lo, &l, 4);
}
// raw Gaussians of one k-step (8 centres of this lane half) as split fragments; with WITH_DIFF also
// raw * (d - centre), the operand of z' = W0 . d(rbf)/dd.  gl = gamma * log2(e).
template <bool WITH_DIFF, bool HALF = false>
__device__ __forceinline__ void rbf_fragments(float d, float gl, const f32x4& c0, const f32x4& c1, f16x8& bh,
                                              f16x8& bl, f16x8& ph, f16x8& pl) {
  uint32_t xh[4], xl[4], qh[4], ql[4];
  const f32x2 cc[4] = {f32x2{c0[0], c0[1]}, f32x2{c0[2], c0[3]}, f32x2{c1[0], c1[1]}, f32x2{c1[2], c1[3]}};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2 diff = splat2(d) - cc[j];
    const f32x2 raw = exp2_2((diff * diff) * splat2(gl));
    if (HALF) {
      xh[j] = half_pair(raw);
      xl[j] = 0;
      if (WITH_DIFF) { qh[j] = half_pair(raw * diff); ql[j] = 0; }
    } else {
      split_pair_u(raw, xh[j], xl[j]);
      if (WITH_DIFF) split_pair_u(raw * diff, qh[j], ql[j]);
    }
  }
  __builtin_memcpy(&bh, xh, 16);
  __builtin_memcpy(&bl, xl, 16);
  if (WITH_DIFF) {
    __builtin_memcpy(&ph, qh, 16);
    __builtin_memcpy(&pl, ql, 16);
  }
}

template <bool HALF>
__device__ __forceinline__ void mfma_prod(const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl,
                                          f32x16& acc1, f32x16& acc2) {
  acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1, 0, 0, 0);
  if (!HALF) {
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2, 0, 0, 0);
  }
}
// one accumulator, operands with unscaled low words (split_pair_u / pack scale 1): acc += a_hi b_hi + a_hi b_lo + a_lo b_hi
template <bool HALF>
__device__ __forceinline__ void mfma_merged(const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl,
                                            f32x16& acc) {
  if (!HALF) {  // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
  }
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
}
// Diagnostic build only: the same with one of the small products left out (drop = 1: a_hi b_lo, i.e. the B operand
// reduced to fp16; drop = 2: a_lo b_hi, the A operand reduced to fp16) -- the "two-product" variants whose force error
// and time tools/two_product.py reports.  The product path never instantiates this.
#ifdef FMD_PHASE_STAMPS
template <bool HALF>
__device__ __forceinline__ void mfma_merged_drop(const f16x8& ah, const f16x8& al, const f16x8& bh, const f16x8& bl,
                                                 f32x16& acc, int drop) {
  if (!HALF) {
    if (drop != 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    if (drop != 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
  }
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
}
#define FMD_MERGED(HALF_, ah_, al_, bh_, bl_, acc_, drop_) mfma_merged_drop<HALF_>(ah_, al_, bh_, bl_, acc_, drop_)
#define FMD_DROP(args_, shift_) (((args_).dbg >> (shift_)) & 3)
#else
#define FMD_MERGED(HALF_, ah_, al_, bh_, bl_, acc_, drop_) mfma_merged<HALF_>(ah_, al_, bh_, bl_, acc_)
#define FMD_DROP(args_, shift_) 0
#endif
// acc1 += a_hi b_hi ; acc2 += a_hi b_lo + a_lo b_hi
__device__ __forceinline__ void mfma_split(const f16x8& ah, const f16x8& al, const f16x8& bh,
                                           const f16x8& bl, f32x16& acc1, f32x16& acc2) {
  acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2, 0, 0, 0);
}


// Centre-row sum of a segment.  A 32-entry tile is four 8-entry segments; the lower and the upper lane half hold
// four entries each (register r = 4q + e <-> entry 8q + 4h + e).  Usually both halves belong to one centre and the
// centre row gets the sum over all eight entries (both halves then store the same value to the same address).  Where
// a centre's run ends on a 4-entry boundary the other half may belong to the next centre: pairs_kernel only pairs
// them when (centre, neighbours) of the two halves are disjoint, and each half then keeps its own sum.
__device__ __forceinline__ float fmd_segment_sum(float si, uint32_t centre_off) {
  const auto c = __builtin_amdgcn_permlane32_swap(centre_off, centre_off, false, false);
  const unsigned u = __builtin_bit_cast(unsigned, si);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const float both = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
  return (unsigned)c[0] == (unsigned)c[1] ? both : si;
}

// Workgroup -> (molecule, split), XCD aware.  Workgroups are dealt to the 8 XCDs round-robin in linear-id order
// (id = blockIdx.y * gridDim.x + blockIdx.x), and every XCD has its own L2.  The S workgroups of one molecule
// gather the same x1 / d(agg) rows, so they are given ids that are equal mod 8: ids are taken in groups of
// 8 S = 8 molecules x S splits, id = 8 S g + 8 split + r  <->  molecule 8 g + r.  The last n_mol % 8 molecules
// keep the plain (molecule-major) order.
__device__ __forceinline__ void wg_to_mol_split(int& m, int& split) {
  const int S = gridDim.x, n_mol = gridDim.y;
  const int lid = blockIdx.y * S + blockIdx.x;
  const int grouped = (n_mol & ~7) * S;
  if (lid < grouped) {
    const int g = lid / (8 * S), rem = lid - g * 8 * S;
    split = rem >> 3;
    m = 8 * g + (rem & 7);
  } else {
    m = blockIdx.y;
    split = blockIdx.x;
  }
}

}  // namespace
