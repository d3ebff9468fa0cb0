// Whole-path engine: energy[B] and forces[N,3] of a batch with no [E,F] tensor in HBM.
// Replaces SchNet.forward + GradientsOut.forward of the reference (models/schnet.py:178-369,
// models/gradients.py:226-290) for StandardSchNet models with hidden = filters = 128.
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
//  * contractions: precision FMD_PREC_F32 = split-operand f16 MFMA (x = hi + lo/2048, three products,
//    fp32 accumulation, 2e-7 = the error of the fp32 MFMA at 3/16 of its cycles);
//    FMD_PREC_F32_MFMA = exact fp32 on v_mfma_f32_32x32x2_f32 (the 4-wave / wave-specialised fp32 kernels);
//  * backward recomputes the filter instead of storing [E,F] tensors, and gets d/d(distance) from
//    the forward-mode product z' = W0 rbf'(d) and u = gW W1 (two extra slice GEMMs), so the only
//    per-pair quantity that reaches HBM is one float (dE/dd) per pair.
#include <string.h>
#include <atomic>
#include <type_traits>
#include <vector>
#include <stdlib.h>

#include "fmd_common.h"

namespace {

constexpr int F = 128;            // hidden = filters
constexpr int LDA = 132;          // LDS row stride (floats) of a 32 x 128 operand tile
constexpr int TILE_FLOATS = 32 * LDA;
constexpr size_t MAX_LDS_BYTES = 160 * 1024;
constexpr int MAX_LDS_ROWS = 283; // beads per molecule whose aggregate fits the 160 KB LDS next to the tile buffer
constexpr int MAX_LDS_ROWS_FS = 535;  // ... in feature-split mode (64 features per workgroup, any num_rbf)
constexpr int SLICE = 4 * 64 * 64;  // floats of one packed 128x128 operand ([wave][k-step][lane])
constexpr int P0_FLOATS = 4 * 32 * 64;

// ---------------------------------------------------------------------------------------------
// packed weight layout (floats)
// ---------------------------------------------------------------------------------------------
struct BlockOff {
  int64_t p0, b0, p1, p1t, l1, l1t, l2, b2, l2t, l3, b3, l3t;
  int64_t q0h, q0l, q1h, q1l, q1th, q1tl;  // fp16 hi/lo split operands of the filter network
  int64_t q0ch, q0cl;                      // W0 with column k scaled by (centre_k - cutoff/2): backward only
};
struct PackLayout {
  int64_t emb, centers, blocks, block_stride, outs, out_stride, emb_x1, total;
};
constexpr int Q0_FLOATS = 4 * 4 * 64 * 4;   // [wave][k-step of 16][lane] x 8 halves (16 B), K padded to 64
constexpr int Q1_FLOATS = 4 * 8 * 64 * 4;   // same for K = 128
constexpr int64_t BLOCK_FLOATS = P0_FLOATS + F + 8 * (int64_t)SLICE + 2 * F + 4 * Q0_FLOATS + 4 * Q1_FLOATS;
constexpr int64_t OUT_FLOATS = 2 * (int64_t)SLICE + F;

PackLayout pack_layout(const fmd_schnet_desc* d) {
  PackLayout L;
  L.emb = 0;
  L.centers = (int64_t)d->embedding_size * F;
  L.blocks = L.centers + 64;
  L.block_stride = BLOCK_FLOATS;
  L.outs = L.blocks + (int64_t)d->num_blocks * BLOCK_FLOATS;
  L.out_stride = OUT_FLOATS;
  L.emb_x1 = L.outs + (int64_t)d->num_out * OUT_FLOATS;  // [embedding_size][128]: embedding lin1(block 0)^T
  L.total = L.emb_x1 + (int64_t)d->embedding_size * F;
  return L;
}
BlockOff block_off(const PackLayout& L, int l) {
  BlockOff o;
  int64_t p = L.blocks + l * L.block_stride;
  o.p0 = p; p += P0_FLOATS;
  o.b0 = p; p += F;
  o.p1 = p; p += SLICE;
  o.p1t = p; p += SLICE;
  o.l1 = p; p += SLICE;
  o.l1t = p; p += SLICE;
  o.l2 = p; p += SLICE;
  o.b2 = p; p += F;
  o.l2t = p; p += SLICE;
  o.l3 = p; p += SLICE;
  o.b3 = p; p += F;
  o.l3t = p; p += SLICE;
  o.q0h = p; p += Q0_FLOATS;
  o.q0l = p; p += Q0_FLOATS;
  o.q1h = p; p += Q1_FLOATS;
  o.q1l = p; p += Q1_FLOATS;
  o.q1th = p; p += Q1_FLOATS;
  o.q1tl = p; p += Q1_FLOATS;
  o.q0ch = p; p += Q0_FLOATS;
  o.q0cl = p; p += Q0_FLOATS;
  return o;
}

// M is [n_out][n_k] row-major (or its transpose when `transposed`): the operand that multiplies a
// row tile as  out[row][o] = sum_k a[row][k] * M[o][k].  Lane l of wave w keeps, for k-step s,
// M[32w + (l&31)][s + 64 (l>>5)]  (zero outside the matrix).
void pack_slice(float* dst, const float* M, int n_out, int n_k, bool transposed) {
  for (int w = 0; w < 4; ++w)
    for (int s = 0; s < 64; ++s)
      for (int l = 0; l < 64; ++l) {
        const int o = 32 * w + (l & 31), k = s + 64 * (l >> 5);
        float v = 0.f;
        if (o < n_out && k < n_k) v = transposed ? M[(int64_t)k * n_out + o] : M[(int64_t)o * n_k + k];
        dst[(w * 64 + s) * 64 + l] = v;
      }
}
// fp16 hi/lo split of an operand for v_mfma_f32_32x32x16_f16: x ~ hi + lo / 2048 with hi = f16(x),
// lo = f16((x - hi) * 2048).  Record (w, s, lane) holds the 8 halves M[32w + (lane&31)][16s + 8(lane>>5) + j]
// (the A-operand and the B-operand fragment of that instruction index their matrix the same way).
void pack_split(float* dst_hi, float* dst_lo, const float* M, int n_out, int n_k, int n_steps,
                bool transposed) {
  _Float16* hi = reinterpret_cast<_Float16*>(dst_hi);
  _Float16* lo = reinterpret_cast<_Float16*>(dst_lo);
  for (int w = 0; w < 4; ++w)
    for (int s = 0; s < n_steps; ++s)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const int o = 32 * w + (l & 31), k = 16 * s + 8 * (l >> 5) + j;
          float v = 0.f;
          if (o < n_out && k < n_k) v = transposed ? M[(int64_t)k * n_out + o] : M[(int64_t)o * n_k + k];
          const _Float16 h = (_Float16)v;
          const int64_t idx = (((int64_t)w * n_steps + s) * 64 + l) * 8 + j;
          hi[idx] = h;
          lo[idx] = (_Float16)((v - (float)h) * 2048.0f);
        }
}
// node-kernel operand: fp32 slice, or (same footprint: SLICE == 2 * Q1_FLOATS) its hi/lo fp16 split
void pack_node(float* dst, const float* M, int n_out, int n_k, bool transposed, bool split) {
  static_assert(SLICE == 2 * Q1_FLOATS, "split records must fit the fp32 slice");
  if (split) pack_split(dst, dst + Q1_FLOATS, M, n_out, n_k, 8, transposed);
  else pack_slice(dst, M, n_out, n_k, transposed);
}
// filter layer 0 as the A operand of hidden^T = W0 rbf^T: lane l, step s: f0_w[32w+(l&31)][2s+(l>>5)]
void pack_p0(float* dst, const float* f0_w, int n_rbf) {
  for (int w = 0; w < 4; ++w)
    for (int s = 0; s < 32; ++s)
      for (int l = 0; l < 64; ++l) {
        const int o = 32 * w + (l & 31), k = 2 * s + (l >> 5);
        dst[(w * 32 + s) * 64 + l] = k < n_rbf ? f0_w[(int64_t)o * n_rbf + k] : 0.f;
      }
}

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

// 8 fp32 values -> hi / lo fp16 fragments
__device__ __forceinline__ void split8(const float (&x)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const hv2 h = __builtin_amdgcn_cvt_pkrtz(x[2 * j], x[2 * j + 1]);
    const float r0 = (x[2 * j] - (float)h[0]) * SPLIT_SCALE;
    const float r1 = (x[2 * j + 1] - (float)h[1]) * SPLIT_SCALE;
    const hv2 l = __builtin_amdgcn_cvt_pkrtz(r0, r1);
    hi[2 * j] = (_Float16)h[0]; hi[2 * j + 1] = (_Float16)h[1];
    lo[2 * j] = (_Float16)l[0]; lo[2 * j + 1] = (_Float16)l[1];
  }
}
__device__ __forceinline__ void split4(const float (&x)[4], f16x4& hi, f16x4& lo) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const hv2 h = __builtin_amdgcn_cvt_pkrtz(x[2 * j], x[2 * j + 1]);
    const float r0 = (x[2 * j] - (float)h[0]) * SPLIT_SCALE;
    const float r1 = (x[2 * j + 1] - (float)h[1]) * SPLIT_SCALE;
    const hv2 l = __builtin_amdgcn_cvt_pkrtz(r0, r1);
    hi[2 * j] = (_Float16)h[0]; hi[2 * j + 1] = (_Float16)h[1];
    lo[2 * j] = (_Float16)l[0]; lo[2 * j + 1] = (_Float16)l[1];
  }
}
// ---- packed-fp32 forms (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 process two values per VALU slot):
// the producers of the split pair kernels are VALU-issue bound, so their arithmetic is written on pairs.
typedef float f32x2 __attribute__((ext_vector_type(2)));
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
  const f16x2 h = __builtin_convertvector(x, f16x2);
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
      split_pair(raw, xh[j], xl[j]);
      if (WITH_DIFF) split_pair(raw * diff, qh[j], ql[j]);
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
// acc1 += a_hi b_hi ; acc2 += a_hi b_lo + a_lo b_hi
__device__ __forceinline__ void mfma_split(const f16x8& ah, const f16x8& al, const f16x8& bh,
                                           const f16x8& bl, f32x16& acc1, f32x16& acc2) {
  acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2, 0, 0, 0);
  acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2, 0, 0, 0);
}


// rows x 128 tile in LDS (stride LDA) times a packed operand streamed from global/L2:
// result lane = output column 32w + (lane&31), register r = tile row tile_row(r, lane>>5)
__device__ __forceinline__ f32x16 gemm_rows_packed(const float* lds_a, const float* __restrict__ pk,
                                                   int w, int lane) {
  f32x16 acc = zero16();
  const float* ap = lds_a + (lane & 31) * LDA + 64 * (lane >> 5);
  const float* bp = pk + (int64_t)w * 64 * 64 + lane;
#pragma unroll
  for (int s4 = 0; s4 < 16; ++s4) {
    const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + 4 * s4);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q], bp[(4 * s4 + q) * 64], acc, 0, 0, 0);
  }
  return acc;
}

// same, but the operand pointer is made opaque so the loads stay inside the caller's tile loop
// (hoisting eight 64-register operands out of the output-network loop would spill)
__device__ __forceinline__ f32x16 gemm_rows_streamed(const float* lds_a, const float* pk, int w,
                                                     int lane) {
  asm volatile("" : "+s"(pk));
  return gemm_rows_packed(lds_a, pk, w, lane);
}

__device__ __forceinline__ void tile_to_lds(float* buf, const f32x16& v, int w, int lane) {
  const int col = 32 * w + (lane & 31), h = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) buf[tile_row(r, h) * LDA + col] = v[r];
}

// ---- split-operand forms of the node helpers (precision FMD_PREC_F32): the LDS tile holds
// u32 = f16 hi | f16 lo << 16 per element, the operand record is [hi 8 steps][lo 8 steps] of 16-byte
// fragments (pack_split), a 128-deep slice costs 24 fp16 MFMAs instead of 64 fp32 ones.
__device__ __forceinline__ uint32_t pack_hl(float v) {
  const hv2 h1 = __builtin_amdgcn_cvt_pkrtz(v, 0.0f);
  const hv2 pk = __builtin_amdgcn_cvt_pkrtz(v, (v - (float)h1[0]) * SPLIT_SCALE);
  uint32_t u;
  __builtin_memcpy(&u, &pk, 4);
  return u;
}
// 8 consecutive packed elements -> hi / lo fragments
__device__ __forceinline__ void unpack_frag(const u32x4& v0, const u32x4& v1, f16x8& hi, f16x8& lo) {
  u32x4 hi4, lo4;
  hi4[0] = __builtin_amdgcn_perm(v0[1], v0[0], 0x05040100u);
  hi4[1] = __builtin_amdgcn_perm(v0[3], v0[2], 0x05040100u);
  hi4[2] = __builtin_amdgcn_perm(v1[1], v1[0], 0x05040100u);
  hi4[3] = __builtin_amdgcn_perm(v1[3], v1[2], 0x05040100u);
  lo4[0] = __builtin_amdgcn_perm(v0[1], v0[0], 0x07060302u);
  lo4[1] = __builtin_amdgcn_perm(v0[3], v0[2], 0x07060302u);
  lo4[2] = __builtin_amdgcn_perm(v1[1], v1[0], 0x07060302u);
  lo4[3] = __builtin_amdgcn_perm(v1[3], v1[2], 0x07060302u);
  __builtin_memcpy(&hi, &hi4, 16);
  __builtin_memcpy(&lo, &lo4, 16);
}
__device__ __forceinline__ f32x16 gemm_rows_split(const float* lds_a, const float* __restrict__ pk,
                                                  int w, int lane) {
  f32x16 acc1 = zero16(), acc2 = zero16();
  const uint32_t* ap = reinterpret_cast<const uint32_t*>(lds_a) + (lane & 31) * LDA + 8 * (lane >> 5);
  const f16x8* bh = reinterpret_cast<const f16x8*>(pk) + (int64_t)w * 8 * 64 + lane;
  const f16x8* bl = reinterpret_cast<const f16x8*>(pk + Q1_FLOATS) + (int64_t)w * 8 * 64 + lane;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const u32x4 v0 = *reinterpret_cast<const u32x4*>(ap + 16 * s);
    const u32x4 v1 = *reinterpret_cast<const u32x4*>(ap + 16 * s + 4);
    f16x8 ah, al;
    unpack_frag(v0, v1, ah, al);
    mfma_split(ah, al, bh[s * 64], bl[s * 64], acc1, acc2);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc1[r] += acc2[r] * SPLIT_INV;
  return acc1;
}
// The same GEMM with the operand's fragments already in registers: lets a kernel that chains several GEMMs per
// tile issue the NEXT operand's loads (L2, ~1-2 us) before it runs the current one.
struct BFrag {
  f16x8 h[8], l[8];
};
__device__ __forceinline__ void load_bfrag(BFrag& b, const float* pk, int w, int lane) {
  asm volatile("" : "+s"(pk));  // the loads stay where they are written (not hoisted out of the tile loop)
  const f16x8* bh = reinterpret_cast<const f16x8*>(pk) + (int64_t)w * 8 * 64 + lane;
  const f16x8* bl = reinterpret_cast<const f16x8*>(pk + Q1_FLOATS) + (int64_t)w * 8 * 64 + lane;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    b.h[s] = bh[s * 64];
    b.l[s] = bl[s * 64];
  }
}
__device__ __forceinline__ f32x16 gemm_rows_frag(const float* lds_a, const BFrag& b, int lane) {
  f32x16 acc1 = zero16(), acc2 = zero16();
  const uint32_t* ap = reinterpret_cast<const uint32_t*>(lds_a) + (lane & 31) * LDA + 8 * (lane >> 5);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const u32x4 v0 = *reinterpret_cast<const u32x4*>(ap + 16 * s);
    const u32x4 v1 = *reinterpret_cast<const u32x4*>(ap + 16 * s + 4);
    f16x8 ah, al;
    unpack_frag(v0, v1, ah, al);
    mfma_split(ah, al, b.h[s], b.l[s], acc1, acc2);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc1[r] += acc2[r] * SPLIT_INV;
  return acc1;
}
template <bool SPLIT>
__device__ __forceinline__ f32x16 gemm_rows(const float* lds_a, const float* __restrict__ pk, int w,
                                            int lane) {
  if (SPLIT) return gemm_rows_split(lds_a, pk, w, lane);
  return gemm_rows_packed(lds_a, pk, w, lane);
}
// W16A16 output network: tile elements are u32 = f16(x) (round to nearest even), the operand is the hi
// record of the split form (= Linear.weight.half()), one MFMA per k-step
__device__ __forceinline__ f32x16 gemm_rows_half(const float* lds_a, const float* pk, int w, int lane) {
  asm volatile("" : "+s"(pk));
  f32x16 acc = zero16();
  const uint32_t* ap = reinterpret_cast<const uint32_t*>(lds_a) + (lane & 31) * LDA + 8 * (lane >> 5);
  const f16x8* bh = reinterpret_cast<const f16x8*>(pk) + (int64_t)w * 8 * 64 + lane;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const u32x4 v0 = *reinterpret_cast<const u32x4*>(ap + 16 * s);
    const u32x4 v1 = *reinterpret_cast<const u32x4*>(ap + 16 * s + 4);
    f16x8 ah, al;
    unpack_frag(v0, v1, ah, al);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[s * 64], acc, 0, 0, 0);
  }
  return acc;
}
// rounds v to fp16 in place (the value later layers and the backward see) and stores it
__device__ __forceinline__ void tile_store_half(float* buf, f32x16& v, int w, int lane) {
  const int col = 32 * w + (lane & 31), h = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const _Float16 hv = (_Float16)v[r];
    v[r] = (float)hv;
    uint16_t bits;
    __builtin_memcpy(&bits, &hv, 2);
    reinterpret_cast<uint32_t*>(buf)[tile_row(r, h) * LDA + col] = bits;
  }
}
template <bool SPLIT>
__device__ __forceinline__ f32x16 gemm_rows_nohoist(const float* lds_a, const float* pk, int w, int lane) {
  asm volatile("" : "+s"(pk));
  return gemm_rows<SPLIT>(lds_a, pk, w, lane);
}
template <bool SPLIT>
__device__ __forceinline__ void tile_store(float* buf, const f32x16& v, int w, int lane) {
  const int col = 32 * w + (lane & 31), h = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (SPLIT) reinterpret_cast<uint32_t*>(buf)[tile_row(r, h) * LDA + col] = pack_hl(v[r]);
    else buf[tile_row(r, h) * LDA + col] = v[r];
  }
}
template <bool SPLIT>
__device__ __forceinline__ void tile_store4(float* buf, int r, int c, const f32x4& v) {
  if (SPLIT) {
    const u32x4 u = {pack_hl(v[0]), pack_hl(v[1]), pack_hl(v[2]), pack_hl(v[3])};
    *reinterpret_cast<u32x4*>(reinterpret_cast<uint32_t*>(buf) + r * LDA + c) = u;
  } else {
    *reinterpret_cast<f32x4*>(buf + r * LDA + c) = v;
  }
}

// 32 rows x 128 floats global -> LDS tile (rows beyond n_rows zero-filled), optional sum of slabs
template <bool SPLIT>
__device__ __forceinline__ void rows_to_lds(float* buf, const float* __restrict__ src,
                                            int64_t slab_stride, int n_slabs, int64_t row0,
                                            int64_t n_rows, int tid) {
  const int r = tid >> 3, c0 = (tid & 7) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + 32 * i;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row0 + r < n_rows) {
      const float* p = src + (row0 + r) * F + c;
      v = *reinterpret_cast<const f32x4*>(p);
      for (int s = 1; s < n_slabs; ++s) v += *reinterpret_cast<const f32x4*>(p + s * slab_stride);
    }
    tile_store4<SPLIT>(buf, r, c, v);
  }
}

// ---------------------------------------------------------------------------------------------
// pair list: undirected i<j, strict d2 < rc2 with unfused fp32, sorted by centre i then j.
// Each centre's entries are padded to a multiple of 4 (pad = (i,i) with d = rc, whose cutoff is 0), so a
// 4-entry half of an 8-entry SEGMENT of a 32-entry tile always belongs to one centre, and both halves do
// unless the rows are disjoint (see the conflict flags below): inside a segment the neighbours j are
// distinct, which lets the edge kernels accumulate with plain LDS read-modify-write (no float atomics) and
// one in-register reduction per segment.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void pairs_kernel(const float* __restrict__ pos,
                                                    const int64_t* __restrict__ mol_ptr, float rc2,
                                                    float rc, int cap, uint32_t* __restrict__ poi,
                                                    uint32_t* __restrict__ poj,
                                                    float* __restrict__ pd, float* __restrict__ pc,
                                                    float* __restrict__ pdc, int32_t* __restrict__ npairs,
                                                    int32_t* __restrict__ report, int max_beads, float* __restrict__ forces, int ent_cap) {
  // Two workgroups (16 waves each) per molecule.  Both build the hit masks, counts and offsets of the whole
  // molecule (one wave per centre row, rows dealt round-robin: the row's candidates j > i are tested 64 at a time,
  // the ballots are kept as hit masks in LDS); each then fills and finishes the rows of ITS half of the entries.
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int m = blockIdx.y, part = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t a0 = mol_ptr[m];
  int n = (int)(mol_ptr[m + 1] - a0);
  if (n > max_beads) {  // the caller's bound is wrong: stay inside the LDS that was sized for it, and say so
    n = max_beads;
  }
  const bool bad_bound = (int)(mol_ptr[m + 1] - a0) > max_beads;
  float* sx = smem;
  float* sy = sx + n;
  float* sz = sy + n;
  int* cnt = reinterpret_cast<int*>(sz + n);   // [n]     pairs of row i (j > i)
  int* rowstart = cnt + n;                      // [n + 1] offset of the row's run
  int* rowend = rowstart + n + 1;               // [n]     end of the run including its pad entries
  int* conf = rowend + n;                       // [n]     row i must not share an 8-entry segment with the row before
  const int NC = (n + 63) >> 6;                 // 64-candidate chunks of a row (at most 9: 535 beads)
  // [n][NC] hit masks: bit b of masks[i NC + c] <-> (i, 64 c + b) is a pair; chunks below (i + 1) / 64 are not written
  unsigned long long* masks = reinterpret_cast<unsigned long long*>(smem + ((7 * n + 2) & ~1));
  __shared__ int tot_s[3];
  for (int i = tid; i < n; i += 1024) {
    sx[i] = pos[3 * (a0 + i)];
    sy[i] = pos[3 * (a0 + i) + 1];
    sz[i] = pos[3 * (a0 + i) + 2];
  }
  // finish_kernel's two workgroups per molecule ADD their halves to the forces: start them from zero
  if (part == 0)
    for (int i = tid; i < 3 * (int)(mol_ptr[m + 1] - a0); i += 1024) forces[3 * a0 + i] = 0.f;
  __syncthreads();
  auto dist2 = [&](int i, int j) {
    const float dx = __fsub_rn(sx[j], sx[i]), dy = __fsub_rn(sy[j], sy[i]), dz = __fsub_rn(sz[j], sz[i]);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
  };
  // ---- pass 1: the hit masks and row counts.  A lane keeps its candidates j = 64 c + lane in registers (beads
  //      beyond the molecule sit at 1e30: their d2 is +inf and never a hit), a wave takes one centre row at a time.
  {
    float px[9], py[9], pz[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const int j = 64 * c + lane;
      const bool ok = j < n;
      px[c] = ok ? sx[j] : 1e30f;
      py[c] = ok ? sy[j] : 1e30f;
      pz[c] = ok ? sz[j] : 1e30f;
    }
    for (int i = wv; i < n; i += 16) {
      const float xi = sx[i], yi = sy[i], zi = sz[i];
      const int c_first = (i + 1) >> 6;
      int c_tot = 0;
      unsigned mlo = 0u, mhi = 0u;  // lane c collects the mask of chunk c: one LDS store per row
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        if (c >= c_first && c < NC) {
          const float dx = __fsub_rn(px[c], xi), dy = __fsub_rn(py[c], yi), dz = __fsub_rn(pz[c], zi);
          const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
          const unsigned long long mk = __ballot(64 * c + lane > i && d2 < rc2);
          mlo = lane == c ? (unsigned)mk : mlo;
          mhi = lane == c ? (unsigned)(mk >> 32) : mhi;
          c_tot += __popcll(mk);
        }
      }
      if (lane < NC) masks[i * NC + lane] = ((unsigned long long)mhi << 32) | mlo;  // (zero below c_first)
      if (lane == 0) cnt[i] = c_tot;
    }
  }
  __syncthreads();
  // ---- which rows may share an 8-entry segment with the row before them: the two halves of a segment are
  //      updated side by side, so the centre and the (at most four) first neighbours of row i must not appear among
  //      the neighbours in the last group of four of the row before (centres ascend, so they cannot meet otherwise)
  for (int i = tid; i < n; i += 1024) {
    bool cf = false;
    if (cnt[i] > 0) {
      int p = i - 1;
      while (p >= 0 && cnt[p] == 0) --p;
      if (p >= 0) {
        // the row's first (at most four) neighbours and the last rp of the row before, as 16-bit fields
        unsigned long long hd = ~0ull, tl = ~0ull;
        int hn = 0, tn = 0;
        for (int c = (i + 1) >> 6; c < NC && hn < 4; ++c) {
          unsigned long long m2 = masks[i * NC + c];
          while (m2 && hn < 4) {
            hd = (hd << 16) | (unsigned long long)(64 * c + __ffsll((long long)m2) - 1);
            m2 &= m2 - 1;
            ++hn;
          }
        }
        const int rp = ((cnt[p] - 1) & 3) + 1;
        for (int c = NC - 1; c >= ((p + 1) >> 6) && tn < rp; --c) {
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
  const int r_lo = part ? r_split : 0, r_hi = part ? n : r_split;
  const int k_lo = part ? k_split : 0, k_hi = part ? total : k_split;
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
      if (c >= ((i + 1) >> 6) && c < NC) {
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
  if (part == 1) {
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
// edge kernels
// ---------------------------------------------------------------------------------------------
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
struct EdgeArgs {
  const uint32_t* poi;     // centre row byte offsets (i * 512)
  const uint32_t* poj;     // neighbour row byte offsets (j * 512)
  const float* pd;
  const float* pc;         // C(d) per pair
  const float* pdc;        // C'(d) per pair
  float* pgd;              // dE/dd per pair (backward)
  int64_t pgd_stride;      // FS mode: second dE/dd slab at pgd + pgd_stride
  bool fs;                 // feature-split mode (host dispatch only)
  const int32_t* npairs;
  const int64_t* mol_ptr;
  const float* x1;         // [N,128] of this block
  const float* gagg;       // [N,128] dE/d(agg) (backward)
  const float* p0;         // packed filter layer 0 (A operand)
  const float* b0;
  const float* p1;         // packed filter layer 1 (B operand)
  const float* p1t;        // packed filter layer 1 transposed (A operand of u)
  const float *q0h, *q0l, *q1h, *q1l, *q1th, *q1tl;  // fp16 hi/lo split operands (16-byte records)
  const float *q0ch, *q0cl;                          // centre-weighted W0 (backward, shared-Gaussian form)
  float c0;                                          // cutoff / 2
  int nb;                                            // ceil(num_rbf / 8): half k-steps that hold Gaussians
  const float* centers;    // [64]
  float* slabs;            // [n_split][N,128] partial aggregates (fwd: agg, bwd: gx1)
  int64_t slab_stride;
  float gamma, rc;
  int cap, n_split;
  int max_beads;           // the caller's bound on beads per molecule: what the dynamic LDS was sized for
  unsigned long long* stamps;  // diagnostic builds only (forward kernels)
  unsigned long long* stamps2; // diagnostic builds only (backward split kernel, middle block)
};

template <int KS0, bool STAMP>
__global__ __launch_bounds__(256, 1) void edge_fwd_kernel(EdgeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* hid = smem;                 // [32][LDA]
  float* agg = smem + TILE_FLOATS;   // [n][128]
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, c32 = lane & 31;
  int m, split;
  wg_to_mol_split(m, split);
  const int64_t a0 = a.mol_ptr[m];
  const int n = min((int)(a.mol_ptr[m + 1] - a0), a.max_beads);  // LDS is sized for max_beads (pairs_kernel flags a larger molecule)
  const int np = a.npairs[m];
  for (int i = tid * 4; i < n * F; i += 1024) *reinterpret_cast<f32x4*>(agg + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  float w0[KS0], w1[64], b0r[16], cen[KS0];
#pragma unroll
  for (int s = 0; s < KS0; ++s) w0[s] = a.p0[(w * 32 + s) * 64 + lane];
#pragma unroll
  for (int s = 0; s < 64; ++s) w1[s] = a.p1[(w * 64 + s) * 64 + lane];
#pragma unroll
  for (int r = 0; r < 16; ++r) b0r[r] = a.b0[32 * w + tile_row(r, h)];
#pragma unroll
  for (int s = 0; s < KS0; ++s) cen[s] = a.centers[2 * s + h];
  const uint32_t col4 = (uint32_t)(32 * w + c32) * 4u;         // byte offset of this lane's feature
  const char* x1b = reinterpret_cast<const char*>(a.x1 + a0 * F);
  char* aggb = reinterpret_cast<char*>(agg);
  const uint32_t* poi = a.poi + (int64_t)m * a.cap;
  const uint32_t* poj = a.poj + (int64_t)m * a.cap;
  const float* pdist = a.pd + (int64_t)m * a.cap;
  const float* pcut = a.pc + (int64_t)m * a.cap;
  const int ntiles = (np + 31) >> 5;
  const int S = a.n_split;

  // Tile meta, fetched one tile ahead with unconditional loads (index clamped into the list):
  //   lane = pair view    : d, C of entry 32 t + c32                        (feeds the rbf / L0)
  //   lane = feature view : row offsets and C of the 16 entries this lane's accumulator registers
  //                         stand for, entries 8q + 4h + {0..3}, q = 0..3    (16-byte broadcast loads)
  float d_n, c_n;
  u32x4 oi_n[4], oj_n[4];
  f32x4 cr_n[4];
  auto fetch = [&](int t) {
    const int base = min(t * 32, a.cap - 32);
    d_n = pdist[base + c32];
    c_n = pcut[base + c32];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      oi_n[q] = *reinterpret_cast<const u32x4*>(poi + base + 8 * q + 4 * h);
      oj_n[q] = *reinterpret_cast<const u32x4*>(poj + base + 8 * q + 4 * h);
      cr_n[q] = *reinterpret_cast<const f32x4*>(pcut + base + 8 * q + 4 * h);
    }
  };
  // state of the tile in flight: register r = 4q + e  <->  entry 8q + 4h + e
  uint32_t oi_c[16], oj_c[16];
  float cr_c[16], xi_c[16], xj_c[16];

  // diagnostic build only (STAMP): per-phase cycle totals of this wave -> a.stamps[wg][wave][phase]
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = 0;
  auto stamp = [&](int ph) {
    if (STAMP) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long now = __builtin_readcyclecounter();
      __builtin_amdgcn_sched_barrier(0);
      tacc[ph] += now - tprev;
      tprev = now;
    }
  };
  auto stage = [&](int t) {
    const float d = d_n, C = c_n;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        oi_c[4 * q + e] = oi_n[q][e] + col4;
        oj_c[4 * q + e] = oj_n[q][e] + col4;
        cr_c[4 * q + e] = cr_n[q][e];
      }
    fetch(t + S);
    // x1 gathers: issued now, first used in the message phase (after L1)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      xi_c[r] = *reinterpret_cast<const float*>(x1b + oi_c[r]);
      xj_c[r] = *reinterpret_cast<const float*>(x1b + oj_c[r]);
    }
    stamp(5);  // meta select, prefetch + gather issue
    // L0: hidden^T slice = W0 slice . rbf^T  (lane = pair, registers = features)
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < KS0; ++s) {
      const float diff = d - cen[s];
      const float b = __expf(a.gamma * diff * diff) * C;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[s], b, acc, 0, 0, 0);
    }
    if (STAMP) asm volatile("" : "+v"(acc));
    stamp(6);  // L0 (+rbf)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmd_tanh_fast(acc[4 * q + e] + b0r[4 * q + e]);
      *reinterpret_cast<f32x4*>(hid + c32 * LDA + 32 * w + 8 * q + 4 * h) = v;
    }
  };
  // Segment q (registers 4q..4q+3 of both lane halves) belongs to one centre with distinct
  // neighbours: neighbour rows are updated by plain read-modify-write, the centre row once.
  auto message = [&](const f32x16& fl) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float vj[4], Wq[4];
      float si = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        Wq[e] = fl[4 * q + e] * cr_c[4 * q + e];
        vj[e] = *reinterpret_cast<float*>(aggb + oj_c[4 * q + e]);
        si += xj_c[4 * q + e] * Wq[e];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        *reinterpret_cast<float*>(aggb + oj_c[4 * q + e]) = vj[e] + xi_c[4 * q + e] * Wq[e];
      si = fmd_segment_sum(si, oi_c[4 * q]);
      float* ci = reinterpret_cast<float*>(aggb + oi_c[4 * q]);
      *ci += si;  // same centre in both halves: both store the same value to the same address
    }
  };

  fetch(split);
  __syncthreads();
  if (STAMP) tprev = __builtin_readcyclecounter();
  for (int t = split; t < ntiles; t += S) {
    stage(t);
    stamp(0);  // tanh, hidden write
    __syncthreads();
    stamp(1);  // barrier A
    f32x16 fl = zero16();
    {
      const float* ap = hid + c32 * LDA + 64 * h;
#pragma unroll
      for (int s4 = 0; s4 < 16; ++s4) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + 4 * s4);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          fl = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q], w1[4 * s4 + q], fl, 0, 0, 0);
      }
    }
    if (STAMP) asm volatile("" : "+v"(fl));
    stamp(2);  // L1
    message(fl);
    stamp(3);  // message
    __syncthreads();
    stamp(4);  // barrier B
  }
  if (STAMP && lane == 0) {
    unsigned long long* o = a.stamps + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) * 8;
    for (int k = 0; k < 8; ++k) o[k] = tacc[k];
  }
  __syncthreads();
  float* out = a.slabs + split * a.slab_stride + a0 * F;
  for (int i = tid * 4; i < n * F; i += 1024)
    *reinterpret_cast<f32x4*>(out + i) = *reinterpret_cast<const f32x4*>(agg + i);
}

// =================================================================================================
// Wave-specialised forward pair kernel: 8 waves per workgroup, two per SIMD.
//   waves 0-3 (consumers): own the 32-feature slices -- L1 MFMAs, x1 gathers, aggregate update in LDS
//   waves 4-7 (producers): rbf values on the VALU, L0 MFMAs, tanh, hidden tile -> LDS
// A single wave per SIMD is instruction-issue bound here (about 800 non-MFMA instructions per 89 MFMAs
// per tile); two waves per SIMD give the MFMA pipe a second instruction stream to overlap with, and the
// split keeps the LDS footprint of the 4-wave design (one hidden tile + one aggregate).
// Per tile two workgroup barriers:  A(t) "hidden(t) visible", B(t) "hidden(t) consumed".
// =================================================================================================
struct SegRegs {              // consumer state of one 32-entry tile, register r = 4q + e <-> entry 8q+4h+e
  uint32_t oi[4];             // centre row byte offset of segment q (+ this lane's column)
  uint32_t oj[16];            // neighbour row byte offsets
  float cr[16];               // C(d) per entry
  float xi[4], xj[16];        // gathered x1[centre], x1[neighbour] (this lane's feature)
};

template <int KS0, bool STAMP>
__global__ __launch_bounds__(512, 2) void edge_fwd_ws_kernel(EdgeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* hid = smem;                 // [32][LDA]
  float* agg = smem + TILE_FLOATS;   // [n][128]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, c32 = lane & 31;
  const bool producer = wave >= 4;   // wave-uniform
  const int w = wave & 3;            // feature slice
  int m, split;
  wg_to_mol_split(m, split);
  const int64_t a0 = a.mol_ptr[m];
  const int n = min((int)(a.mol_ptr[m + 1] - a0), a.max_beads);  // LDS is sized for max_beads (pairs_kernel flags a larger molecule)
  const int np = a.npairs[m];
  for (int i = tid * 4; i < n * F; i += 2048) *reinterpret_cast<f32x4*>(agg + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t* poi = a.poi + (int64_t)m * a.cap;
  const uint32_t* poj = a.poj + (int64_t)m * a.cap;
  const float* pdist = a.pd + (int64_t)m * a.cap;
  const float* pcut = a.pc + (int64_t)m * a.cap;
  const int ntiles = (np + 31) >> 5;
  const int S = a.n_split;
  const int last_base = (ntiles - 1) * 32;
  __syncthreads();
  if (split < ntiles) {  // uniform over the workgroup
    if (producer) {
      // ------------------------------------------------------------------ producer waves
      float w0[KS0], cen[KS0], b0r[16];
#pragma unroll
      for (int s = 0; s < KS0; ++s) w0[s] = a.p0[(w * 32 + s) * 64 + lane];
#pragma unroll
      for (int s = 0; s < KS0; ++s) cen[s] = a.centers[2 * s + h];
#pragma unroll
      for (int r = 0; r < 16; ++r) b0r[r] = a.b0[32 * w + tile_row(r, h)];
      float d_n = pdist[min(split * 32, last_base) + c32], c_n = pcut[min(split * 32, last_base) + c32];
      auto compute = [&](int t_next) {  // hidden^T slice of the tile whose meta is in (d_n, c_n)
        const float d = d_n, C = c_n;
        const int nb = min(t_next * 32, last_base);
        d_n = pdist[nb + c32];
        c_n = pcut[nb + c32];
        // rbf values first (independent VALU chains pipeline; interleaving them one-to-one with the
        // dependent MFMA chain costs ~116 cycles per step instead of 64), then the MFMA chain
        float bval[KS0];
#pragma unroll
        for (int s = 0; s < KS0; ++s) {
          const float diff = d - cen[s];
          bval[s] = __expf(a.gamma * diff * diff) * C;
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc = zero16();
#pragma unroll
        for (int s = 0; s < KS0; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[s], bval[s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = fmd_tanh_fast(acc[r] + b0r[r]);
        // pin the whole computation in front of the barrier that follows (it has no memory side
        // effect, so the scheduler would otherwise sink it behind the barrier, onto the critical path)
        asm volatile("" : "+v"(acc));
        return acc;
      };
      auto put = [&](const f32x16& v) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(hid + c32 * LDA + 32 * w + 8 * q + 4 * h) =
              f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
      };
      __builtin_amdgcn_s_setprio(1);  // the short MFMA chain of the producers goes first on the shared pipe
      put(compute(split + S));
      __syncthreads();  // A(first)
      for (int t = split; t < ntiles; t += S) {
        const f32x16 hv = compute(t + 2 * S);  // tile t+S, while the consumers contract tile t
        __syncthreads();                       // B(t)
        put(hv);
        __syncthreads();                       // A(t+S)
      }
    } else {
      // ------------------------------------------------------------------ consumer waves
      float w1[64];
#pragma unroll
      for (int s = 0; s < 64; ++s) w1[s] = a.p1[(w * 64 + s) * 64 + lane];
      const uint32_t col4 = (uint32_t)(32 * w + c32) * 4u;
      const char* x1b = reinterpret_cast<const char*>(a.x1 + a0 * F);
      char* aggb = reinterpret_cast<char*>(agg);
      SegRegs X;
      // meta of the tile being loaded into X (flat scalars, static indices: stays in registers)
      uint32_t oj_m[16], oi_m[4];
      float cr_m[16];
      float Wq[4], vj[4], cval = 0.f, si = 0.f, so = 0.f;  // aggregate-update state of one segment
      // side work of region gap `i` for the tile whose meta/gathers go into X (entry base `base`)
      // and the aggregate update of the previous tile (filter slice fl_p)
      auto side_fetch = [&](auto ic, int base) {
        constexpr int i = decltype(ic)::value;
        // ---- meta loads of the new tile: gaps 0..11
        if constexpr (i < 4) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(poj + base + 8 * i + 4 * h);
          oj_m[4 * i] = v[0]; oj_m[4 * i + 1] = v[1]; oj_m[4 * i + 2] = v[2]; oj_m[4 * i + 3] = v[3];
        } else if constexpr (i < 8) {
          constexpr int k = i - 4;
          const f32x4 v = *reinterpret_cast<const f32x4*>(pcut + base + 8 * k + 4 * h);
          cr_m[4 * k] = v[0]; cr_m[4 * k + 1] = v[1]; cr_m[4 * k + 2] = v[2]; cr_m[4 * k + 3] = v[3];
        } else if constexpr (i < 12) {
          oi_m[i - 8] = poi[base + 8 * (i - 8) + 4 * h];
        }
      };
      auto side_msg = [&](auto ic, const f32x16& fl_p) {
        constexpr int i = decltype(ic)::value;
        // ---- aggregate update of the previous tile (X still holds it): gaps 2..25, segment q in
        //      four steps (one centre, distinct neighbours: plain read-modify-write)
        if constexpr (i >= 2 && i < 26) {
          constexpr int q = (i - 2) / 6, ph = (i - 2) % 6;
          if constexpr (ph == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              Wq[e] = fl_p[4 * q + e] * X.cr[4 * q + e];
              vj[e] = *reinterpret_cast<const float*>(aggb + X.oj[4 * q + e]);
            }
            cval = *reinterpret_cast<const float*>(aggb + X.oi[q]);
          } else if constexpr (ph == 1) {
            si = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) si += X.xj[4 * q + e] * Wq[e];
            so = fmd_segment_sum(si, X.oi[q]) - si;  // the other half's share (0 when it belongs to another centre)
          } else if constexpr (ph == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              *reinterpret_cast<float*>(aggb + X.oj[4 * q + e]) = vj[e] + X.xi[q] * Wq[e];
          } else if constexpr (ph == 3) {
            // both halves store the same value to the same address
            *reinterpret_cast<float*>(aggb + X.oi[q]) = cval + (si + so);
          }
        }
      };
      auto side_fill = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        // ---- X <- new tile: offsets (gaps 30..33), x1 gathers (gaps 34..53)
        if constexpr (i >= 30 && i < 34) {
          constexpr int q = i - 30;
          X.oi[q] = oi_m[q] + col4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            X.oj[4 * q + e] = oj_m[4 * q + e] + col4;
            X.cr[4 * q + e] = cr_m[4 * q + e];
          }
        }
        if constexpr (i >= 34 && i < 38) X.xi[i - 34] = *reinterpret_cast<const float*>(x1b + X.oi[i - 34]);
        if constexpr (i >= 38 && i < 54) X.xj[i - 38] = *reinterpret_cast<const float*>(x1b + X.oj[i - 38]);
      };
      // prologue: X <- tile `split` (its aggregate update runs, with a zero filter, inside region(split))
      f32x16 fl_prev = zero16();
      {
        const int base = min(split * 32, last_base);
        static_for<0, 12>([&](auto ic) { side_fetch(ic, base); });
        static_for<30, 54>([&](auto ic) { side_fill(ic); });
      }
      unsigned long long t_region = 0, t_bb = 0, t_ba = 0, tprev = 0;
      __syncthreads();  // A(first)
      if (STAMP) tprev = __builtin_readcyclecounter();
      for (int t = split; t < ntiles; t += S) {
        // region(t): 64 L1 MFMAs of tile t; in their shadows the aggregate update of tile t-S (held in
        // X, fl_prev), then X is refilled with tile t: meta loads, offsets, x1 gathers.
        const int base = min(t * 32, last_base);
        f32x16 fl = zero16();
        const float* ap = hid + c32 * LDA + 64 * h;
        f32x4 a4 = *reinterpret_cast<const f32x4*>(ap);
        f32x4 a4n = a4;
        static_for<0, 64>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          if constexpr ((i & 3) == 0 && i < 60) a4n = *reinterpret_cast<const f32x4*>(ap + i + 4);
          fl = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i & 3], w1[i], fl, 0, 0, 0);
          side_fetch(ic, base);
          side_msg(ic, fl_prev);
          side_fill(ic);
          if constexpr ((i & 3) == 3) a4 = a4n;
          __builtin_amdgcn_sched_barrier(0);
        });
        fl_prev = fl;
        if (STAMP) {
          asm volatile("" : "+v"(fl_prev));
          const unsigned long long now = __builtin_readcyclecounter();
          t_region += now - tprev;
          tprev = now;
        }
        __syncthreads();  // B(t): hidden tile consumed
        if (STAMP) {
          const unsigned long long now = __builtin_readcyclecounter();
          t_bb += now - tprev;
          tprev = now;
        }
        __syncthreads();  // A(t+S)
        if (STAMP) {
          const unsigned long long now = __builtin_readcyclecounter();
          t_ba += now - tprev;
          tprev = now;
        }
      }
      if (STAMP && lane == 0) {
        unsigned long long* o = a.stamps + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) * 8;
        o[0] = t_region; o[1] = t_bb; o[2] = t_ba;
        for (int k = 3; k < 8; ++k) o[k] = 0;
      }
      // the last tile's aggregate update
      static_for<2, 26>([&](auto ic) { side_msg(ic, fl_prev); });
    }
  }
  __syncthreads();
  float* out = a.slabs + split * a.slab_stride + a0 * F;
  for (int i = tid * 4; i < n * F; i += 2048)
    *reinterpret_cast<f32x4*>(out + i) = *reinterpret_cast<const f32x4*>(agg + i);
}

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
// SH: the tile's Gaussians are computed once, by the consumers (consumer w: k-step w) between barriers B and A,
// and reach the producers as split fragments in LDS (see the backward kernel); used when the region fits.
template <int NS0, bool STAMP, bool HALF, bool FS, bool SH>  // NS0: k-steps of 16 covering the rbf index; HALF: W16A16
__global__ __launch_bounds__(512, 2) void edge_fwd_split_kernel(EdgeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  _Float16* hid_hi = reinterpret_cast<_Float16*>(smem);          // [32][LDH]
  _Float16* hid_lo = hid_hi + 32 * LDH;                          // [32][LDH]
  float* frag_l = smem + TILE_FLOATS + 128;                      // SH: [nb][hi, lo][32] x 16 B
  float* agg = smem + TILE_FLOATS + 128 + (SH ? a.nb * 256 : 0); // [n][128]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, c32 = lane & 31;
  const bool producer = wave >= 4;   // wave-uniform
  const int w = wave & 3;            // feature slice
  int m, split;
  wg_to_mol_split(m, split);
  const int64_t a0 = a.mol_ptr[m];
  const int n = min((int)(a.mol_ptr[m + 1] - a0), a.max_beads);  // LDS is sized for max_beads (pairs_kernel flags a larger molecule)
  const int np = a.npairs[m];
  constexpr int AW = FS ? 64 : F;  // aggregate row width kept in LDS
  for (int i = tid * 4; i < n * AW; i += 2048) *reinterpret_cast<f32x4*>(agg + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t* poi = a.poi + (int64_t)m * a.cap;
  const uint32_t* poj = a.poj + (int64_t)m * a.cap;
  const float* pdist = a.pd + (int64_t)m * a.cap;
  const float* pcut = a.pc + (int64_t)m * a.cap;
  const int ntiles = (np + 31) >> 5;
  const int S = FS ? 1 : a.n_split;  // tile stride of this workgroup
  const int T0 = FS ? 0 : split;     // its first tile
  const int last_base = (ntiles - 1) * 32;
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, tprev = 0;  // diagnostic build only
#define FMD_STAMP(acc)                                         \
  if (STAMP) {                                                 \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    acc += now_ - tprev;                                       \
    tprev = now_;                                              \
  }
  __syncthreads();
  if (STAMP) tprev = __builtin_readcyclecounter();
  if (T0 < ntiles) {  // uniform over the workgroup
    if (producer) {
      // ------------------------------------------------------------------ producer waves
      f16x8 a0h[NS0], a0l[NS0];
      f32x4 cen[SH ? 1 : 2 * NS0], b0r[4];
      const float gl = a.gamma * LOG2E;
#pragma unroll
      for (int s = 0; s < NS0; ++s) {
        a0h[s] = reinterpret_cast<const f16x8*>(a.q0h)[(w * 4 + s) * 64 + lane];
        if (!HALF) a0l[s] = reinterpret_cast<const f16x8*>(a.q0l)[(w * 4 + s) * 64 + lane];
        if (!SH) {
          cen[2 * s] = *reinterpret_cast<const f32x4*>(a.centers + 16 * s + 8 * h);
          cen[2 * s + 1] = *reinterpret_cast<const f32x4*>(a.centers + 16 * s + 8 * h + 4);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) b0r[q] = *reinterpret_cast<const f32x4*>(a.b0 + 32 * w + 8 * q + 4 * h);
      float d_n = pdist[min(T0 * 32, last_base) + c32], c_n = pcut[min(T0 * 32, last_base) + c32];
      f16x4 hh[4], hl[4];  // hidden slice of the staged tile, split, 4 features per register group
      auto compute = [&](int t_next) {
        const float d = d_n, C = c_n;
        const int nb = min(t_next * 32, last_base);
        d_n = pdist[nb + c32];
        c_n = pcut[nb + c32];
        f32x16 acc1 = zero16(), acc2 = zero16();
#pragma unroll
        for (int s = 0; s < NS0; ++s) {
          f16x8 bh, bl, ph, pl;  // C is a per-pair (per-lane) scalar: applied to the sums
          if (SH) {
            const int blk = min(2 * s + h, a.nb - 1);  // past the last Gaussian: any finite data (zero weights)
            bh = *reinterpret_cast<const f16x8*>(frag_l + ((blk * 2 + 0) * 32 + c32) * 4);
            if (!HALF) bl = *reinterpret_cast<const f16x8*>(frag_l + ((blk * 2 + 1) * 32 + c32) * 4);
          } else {
            rbf_fragments<false, HALF>(d, gl, cen[SH ? 0 : 2 * s], cen[SH ? 0 : 2 * s + 1], bh, bl, ph, pl);
          }
          mfma_prod<HALF>(a0h[s], a0l[s], bh, bl, acc1, acc2);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          uint32_t th[2], tl[2];
#pragma unroll
          for (int ep = 0; ep < 2; ++ep) {
            const int r = 4 * q + 2 * ep;
            // z = C (W0 raw) + b0 ; the tile carries C tanh(z): the consumers' L1 then yields W = filt C
            const f32x2 g0 = comb2<HALF>(acc1, acc2, r);
            const f32x2 hv = tanh_fast2(splat2(C) * g0 + f32x2{b0r[q][2 * ep], b0r[q][2 * ep + 1]});
            if (HALF) { th[ep] = half_pair(splat2(C) * hv); tl[ep] = 0; }
            else split_pair(splat2(C) * hv, th[ep], tl[ep]);
          }
          __builtin_memcpy(&hh[q], th, 8);
          __builtin_memcpy(&hl[q], tl, 8);
        }
        // pin the computation in front of the barrier that follows
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(hh[q]), "+v"(hl[q]));
      };
      auto put = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          *reinterpret_cast<f16x4*>(hid_hi + c32 * LDH + 32 * w + 8 * q + 4 * h) = hh[q];
          if (!HALF) *reinterpret_cast<f16x4*>(hid_lo + c32 * LDH + 32 * w + 8 * q + 4 * h) = hl[q];
        }
      };
      if (SH) __syncthreads();  // fragments of the first tile visible
      compute(T0 + S);
      if (SH) __syncthreads();  // ... and consumed: the consumers write the second tile's
      put();
      __syncthreads();  // A(first)
      for (int t = T0; t < ntiles; t += S) {
        compute(t + 2 * S);  // tile t+S, while the consumers work on tile t
        FMD_STAMP(ts0)
        __syncthreads();     // B(t)
        FMD_STAMP(ts1)
        put();
        FMD_STAMP(ts2)
        __syncthreads();     // A(t+S)
        FMD_STAMP(ts3)
      }
    } else {
      // ------------------------------------------------------------------ consumer waves
      if (FS && w >= 2) {  // no feature slice for this wave in FS mode: keep the barrier protocol
        __syncthreads();  // A(first)
        for (int t = T0; t < ntiles; t += S) {
          __syncthreads();  // B(t)
          __syncthreads();  // A(t+S)
        }
      } else {
      const int cw = FS ? 2 * split + w : w;  // feature slice of this wave
      f16x8 b1h[8], b1l[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        b1h[s] = reinterpret_cast<const f16x8*>(a.q1h)[(cw * 8 + s) * 64 + lane];
        if (!HALF) b1l[s] = reinterpret_cast<const f16x8*>(a.q1l)[(cw * 8 + s) * 64 + lane];
      }
      const uint32_t col4 = (uint32_t)(32 * cw + c32) * 4u;
      // byte offset inside the LDS aggregate of the row/column a global offset o = row * 512 + col4 names
      const int lds_adj = (32 * w + c32) * 4 - (int)(col4 >> 1);
      auto lo = [&](uint32_t o) -> uint32_t { return FS ? (uint32_t)((int)(o >> 1) + lds_adj) : o; };
      const char* x1b = reinterpret_cast<const char*>(a.x1 + a0 * F);
      char* aggb = reinterpret_cast<char*>(agg);
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
            vj[e] = *reinterpret_cast<const float*>(aggb + lo(X.oj[4 * q + e]));
            si += X.xj[4 * q + e] * Wq[e];
          }
          const float cval = *reinterpret_cast<const float*>(aggb + lo(X.oi[q]));
#pragma unroll
          for (int e = 0; e < 4; ++e)
            *reinterpret_cast<float*>(aggb + lo(X.oj[4 * q + e])) = vj[e] + X.xi[q] * Wq[e];
          si = fmd_segment_sum(si, X.oi[q]);
          // both halves store the same value to the same address
          *reinterpret_cast<float*>(aggb + lo(X.oi[q])) = cval + si;
        });
      };
      // The aggregate update of a tile is deferred by one tile: it does not depend on the hidden
      // tile, so it runs (with the refill of X) while the producers stage the next tile, instead of
      // between the two barriers where the producers only write LDS.
      // SH: Gaussians of one tile (k-step w of this wave) as split fragments in LDS
      float d_m = 0.f;
      f32x4 cs0, cs1;
      if (SH && w < NS0) {
        cs0 = *reinterpret_cast<const f32x4*>(a.centers + 16 * w + 8 * h);
        cs1 = *reinterpret_cast<const f32x4*>(a.centers + 16 * w + 8 * h + 4);
      }
      auto fetch_d = [&](int t) {
        if (SH) d_m = pdist[min(t * 32, last_base) + c32];
      };
      auto share_fragments = [&]() {  // of the tile whose distances were fetched last
        if (SH && w < NS0 && 2 * w + h < a.nb) {
          f16x8 bh, bl, ph, pl;
          rbf_fragments<false, HALF>(d_m, a.gamma * LOG2E, cs0, cs1, bh, bl, ph, pl);
          const int blk = 2 * w + h;
          *reinterpret_cast<f16x8*>(frag_l + ((blk * 2 + 0) * 32 + c32) * 4) = bh;
          if (!HALF) *reinterpret_cast<f16x8*>(frag_l + ((blk * 2 + 1) * 32 + c32) * 4) = bl;
        }
      };
      if (SH) {
        fetch_d(T0);
        share_fragments();
        fetch_d(T0 + S);
        __syncthreads();  // first tile's fragments visible
        __syncthreads();  // ... and consumed
        share_fragments();
        fetch_d(T0 + 2 * S);
      }
      fetch_meta(T0);
      fill();            // X <- first tile (valid rows for the first, zero-weight, update)
      f32x16 fl_prev = zero16();
      __syncthreads();  // A(first)
      for (int t = T0; t < ntiles; t += S) {
        // L1: filt slice = hidden . W1^T slice (lane = feature, registers = pairs)
        f32x16 acc1 = zero16(), acc2 = zero16();
        {
          // fragments are read two k-steps ahead of their MFMAs; the fences keep the scheduler from
          // hoisting all sixteen 16-byte reads (64 registers) to the top of the loop
          const _Float16* ph = hid_hi + c32 * LDH + 8 * h;
          const _Float16* pl = hid_lo + c32 * LDH + 8 * h;
          f16x8 ah0 = *reinterpret_cast<const f16x8*>(ph), al0 = *reinterpret_cast<const f16x8*>(pl);
          f16x8 ah1 = *reinterpret_cast<const f16x8*>(ph + 16), al1 = *reinterpret_cast<const f16x8*>(pl + 16);
          static_for<0, 8>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            f16x8 ahn = ah1, aln = al1;
            if constexpr (s + 2 < 8) {
              ahn = *reinterpret_cast<const f16x8*>(ph + 16 * (s + 2));
              aln = *reinterpret_cast<const f16x8*>(pl + 16 * (s + 2));
            }
            mfma_prod<HALF>(ah0, al0, b1h[s], b1l[s], acc1, acc2);
            ah0 = ah1; al0 = al1; ah1 = ahn; al1 = aln;
            __builtin_amdgcn_sched_barrier(0);
          });
        }
        FMD_STAMP(ts0)
        message(fl_prev);  // tile t-S (X holds its gathers)
        fill();            // X <- tile t (meta fetched one tile ago; lands before the next update)
        fetch_meta(t + S);
#pragma unroll
        for (int r = 0; r < 16; ++r) fl_prev[r] = HALF ? acc1[r] : acc1[r] + acc2[r] * SPLIT_INV;
        asm volatile("" : "+v"(fl_prev));  // L1 (and its LDS reads) complete before the barrier
        FMD_STAMP(ts2)
        __syncthreads();  // B(t): hidden tiles consumed
        FMD_STAMP(ts1)
        if (SH) {  // the producers only write the hidden tile now: the fragments they stage from next are replaced
          share_fragments();
          fetch_d(t + 3 * S);
        }
        __syncthreads();  // A(t+S)
        FMD_STAMP(ts3)
      }
      message(fl_prev);  // last tile
      }
    }
  }
  if (STAMP && lane == 0) {
    unsigned long long* o = a.stamps + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) * 8 + (producer ? 4 : 0);
    o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = ts3;
  }
#undef FMD_STAMP
  __syncthreads();
  float* out = a.slabs + split * a.slab_stride + a0 * F;
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
}

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

// FS: feature-split mode for molecules of 284..566 beads, see edge_fwd_split_kernel (each of the molecule's two
// workgroups: all tiles, 64 features of gx1, its own dE/dd slab -- the u GEMM only sees its 64 features of gW).
template <int NS0, bool NEED_GX1, bool FIRST, bool STAMP, bool SHARE, bool HALF, bool FS>
__global__ __launch_bounds__(512, 2) void edge_bwd_split_kernel(EdgeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  _Float16* hid_hi = reinterpret_cast<_Float16*>(smem);          // [32][LDH]
  _Float16* hid_lo = hid_hi + 32 * LDH;                          // [32][LDH]
  // the gW tile reuses the two fp16 tiles (hi, lo), [pair][feature]
  float* red = smem + TILE_FLOATS + 128;                         // [4][32]
  float* cen_l = red + 128;                                      // [64] rbf centres
  float* b0_l = cen_l + 64;                                      // [128] filter layer-0 bias
  // SHARE: the Gaussians of a tile are computed once, by the consumers (consumer w: k-step w), and cross
  // LDS as split fragments [half k-step][hi, lo][pair] x 16 B; the producers then run W0 and
  // Wc = W0 diag(centre - c0) on the same fragments (W0 (raw (d - centre)) = (d - c0) W0 raw - Wc raw).
  float* frag_l = b0_l + 128;                                    // [nb][2][32] x 16 B   (SHARE only)
  float* gx1 = frag_l + (SHARE ? a.nb * 256 : 0);                // [n][128]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, c32 = lane & 31;
  const bool producer = wave >= 4;   // wave-uniform
  const int w = wave & 3;            // feature slice
  int m, split;
  wg_to_mol_split(m, split);
  const int64_t a0 = a.mol_ptr[m];
  const int n = min((int)(a.mol_ptr[m + 1] - a0), a.max_beads);  // LDS is sized for max_beads (pairs_kernel flags a larger molecule)
  const int np = a.npairs[m];
  constexpr int AW = FS ? 64 : F;  // gx1 row width kept in LDS
  if (NEED_GX1)
    for (int i = tid * 4; i < n * AW; i += 2048) *reinterpret_cast<f32x4*>(gx1 + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid < 64) cen_l[tid] = a.centers[tid];
  if (tid < 128) b0_l[tid] = a.b0[tid];
  const uint32_t* poi = a.poi + (int64_t)m * a.cap;
  const uint32_t* poj = a.poj + (int64_t)m * a.cap;
  const float* pdist = a.pd + (int64_t)m * a.cap;
  const float* pcut = a.pc + (int64_t)m * a.cap;
  const float* pdcut = a.pdc + (int64_t)m * a.cap;
  float* pgd = a.pgd + (int64_t)m * a.cap + (FS ? split * a.pgd_stride : 0);  // FS: one dE/dd slab per half
  const int ntiles = (np + 31) >> 5;
  const int S = FS ? 1 : a.n_split;  // tile stride of this workgroup
  const int T0 = FS ? 0 : split;     // its first tile
  const int last_base = (ntiles - 1) * 32;
  unsigned long long ts[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = 0;  // diagnostic build only
#define FMD_STAMP(k)                                           \
  if (STAMP) {                                                 \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    ts[k] += now_ - tprev;                                     \
    tprev = now_;                                              \
  }
  __syncthreads();
  if (STAMP) tprev = __builtin_readcyclecounter();
  if (T0 < ntiles) {  // uniform over the workgroup
    if (producer) {
      // ------------------------------------------------------------------ producer waves
      f16x8 a0h[NS0], a0l[NS0], ach[SHARE ? NS0 : 1], acl[SHARE ? NS0 : 1], wth[8], wtl[8];
#pragma unroll
      for (int s = 0; s < NS0; ++s) {
        a0h[s] = reinterpret_cast<const f16x8*>(a.q0h)[(w * 4 + s) * 64 + lane];
        if (!HALF) a0l[s] = reinterpret_cast<const f16x8*>(a.q0l)[(w * 4 + s) * 64 + lane];
        if (SHARE && !HALF) {
          ach[s] = reinterpret_cast<const f16x8*>(a.q0ch)[(w * 4 + s) * 64 + lane];
          acl[s] = reinterpret_cast<const f16x8*>(a.q0cl)[(w * 4 + s) * 64 + lane];
        }
      }
      constexpr int NKU = FS ? 4 : 8;          // k-steps of the u GEMM: the features of gW this workgroup holds
      const int ku0 = FS ? 4 * split : 0;
#pragma unroll
      for (int s = 0; s < NKU; ++s) {
        wth[s] = reinterpret_cast<const f16x8*>(a.q1th)[(w * 8 + ku0 + s) * 64 + lane];
        if (!HALF) wtl[s] = reinterpret_cast<const f16x8*>(a.q1tl)[(w * 8 + ku0 + s) * 64 + lane];
      }
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
      float k_nx[16], gd_nx = 0.f;
      float k_cu[16], gd_cu = 0.f;
      const float gl = a.gamma * LOG2E;
      auto compute = [&](int t_fetch) {
        const float d = d_n, C = c_n, dC = dc_n;
        gd_nx = gd_n;
        fetch(t_fetch);
        f32x16 z1 = zero16(), z2 = zero16(), p1 = zero16(), p2 = zero16();
#pragma unroll
        for (int s = 0; s < NS0; ++s) {
          f32x4 c0, c1;
          if (!SHARE) {
            c0 = *reinterpret_cast<const f32x4*>(cen_l + 16 * s + 8 * h);
            c1 = *reinterpret_cast<const f32x4*>(cen_l + 16 * s + 8 * h + 4);
          }
          // z  = C G0 + b0,            G0 = W0 raw
          // z' = 2 gamma C G1 + C' G0, G1 = W0 (raw diff)
          f16x8 bh, bl, ph, pl;
          if (SHARE) {  // p accumulates Gc = Wc raw
            const int blk = min(2 * s + h, a.nb - 1);  // past the last Gaussian: any finite data (zero weights)
            bh = *reinterpret_cast<const f16x8*>(frag_l + ((blk * 2 + 0) * 32 + c32) * 4);
            bl = *reinterpret_cast<const f16x8*>(frag_l + ((blk * 2 + 1) * 32 + c32) * 4);
            if (HALF) {  // W16A16: the second record holds f16(raw (d - centre)); no cancellation-prone Wc form
              z1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h[s], bh, z1, 0, 0, 0);
              p1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h[s], bl, p1, 0, 0, 0);
            } else {
              mfma_split(a0h[s], a0l[s], bh, bl, z1, z2);
              mfma_split(ach[s], acl[s], bh, bl, p1, p2);
            }
          } else {
            rbf_fragments<true, HALF>(d, gl, c0, c1, bh, bl, ph, pl);
            mfma_prod<HALF>(a0h[s], a0l[s], bh, bl, z1, z2);
            mfma_prod<HALF>(a0h[s], a0l[s], ph, pl, p1, p2);
          }
          __builtin_amdgcn_sched_barrier(0);  // one k-step of temporaries live at a time
        }
        const float g2c = 2.0f * a.gamma * C, dm = d - a.c0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(b0_l + 32 * w + 8 * q + 4 * h);
          uint32_t th[2], tl[2];
#pragma unroll
          for (int ep = 0; ep < 2; ++ep) {
            const int r = 4 * q + 2 * ep;
            const f32x2 g0 = comb2<HALF>(z1, z2, r);
            f32x2 g1 = comb2<HALF>(p1, p2, r);
            if (SHARE && !HALF) g1 = splat2(dm) * g0 - g1;
            const f32x2 hv = tanh_fast2(splat2(C) * g0 + f32x2{b4[2 * ep], b4[2 * ep + 1]});
            // the tile carries C h: L1 then yields W = filt C
            if (HALF) { th[ep] = half_pair(splat2(C) * hv); tl[ep] = 0; }
            else split_pair(splat2(C) * hv, th[ep], tl[ep]);
            const f32x2 kk = splat2(C) * ((splat2(1.0f) - hv * hv) * (splat2(g2c) * g1 + splat2(dC) * g0)) +
                             splat2(dC) * hv;
            k_nx[r] = kk[0];
            k_nx[r + 1] = kk[1];
          }
          __builtin_memcpy(&hh[q], th, 8);
          __builtin_memcpy(&hl[q], tl, 8);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(hh[q]), "+v"(hl[q]));
        // ... and the k array too: left alone, the compiler sinks its arithmetic behind the reduction
        // and carries both accumulators of G1 and the hidden values (48 registers) across it instead
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(k_nx[r]));
      };
      auto put = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          *reinterpret_cast<f16x4*>(hid_hi + c32 * LDH + 32 * w + 8 * q + 4 * h) = hh[q];
          if (!HALF) *reinterpret_cast<f16x4*>(hid_lo + c32 * LDH + 32 * w + 8 * q + 4 * h) = hl[q];
        }
      };
      auto rotate = [&]() {
#pragma unroll
        for (int r = 0; r < 16; ++r) k_cu[r] = k_nx[r];
        gd_cu = gd_nx;
      };
      fetch(T0);
      if (SHARE) __syncthreads();  // fragments of the first tile visible
      compute(T0 + S);
      if (SHARE) __syncthreads();  // ... and consumed: the consumers write the second tile's
      rotate();
      put();
      __syncthreads();  // A(first)
      for (int t = T0; t < ntiles; t += S) {
        compute(t + 2 * S);  // stages tile t+S while the consumers contract tile t
        FMD_STAMP(0)
        __syncthreads();     // B(t)
        FMD_STAMP(1)
        __syncthreads();     // C(t): gW(t) visible
        FMD_STAMP(2)
        // u^T slice = W1^T slice . gW^T  (lane = pair, registers = hidden features)
        f32x16 u1 = zero16(), u2 = zero16();
        {
          // gW fragments read one k-step ahead (fenced: see the consumers' L1 loop)
          const _Float16* gph = hid_hi + c32 * LDH + 8 * h + 16 * ku0;
          const _Float16* gpl = hid_lo + c32 * LDH + 8 * h + 16 * ku0;
          f16x8 bh = *reinterpret_cast<const f16x8*>(gph), bl = *reinterpret_cast<const f16x8*>(gpl);
          static_for<0, NKU>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            f16x8 nh = bh, nl = bl;
            if constexpr (s + 1 < NKU) {
              nh = *reinterpret_cast<const f16x8*>(gph + 16 * (s + 1));
              nl = *reinterpret_cast<const f16x8*>(gpl + 16 * (s + 1));
            }
            mfma_prod<HALF>(wth[s], wtl[s], bh, bl, u1, u2);
            bh = nh; bl = nl;
            __builtin_amdgcn_sched_barrier(0);
          });
        }
        // dE/dd = C sum_h u (1-h^2) z'  +  C' sum_h u h   (this wave's 32 hidden features)
        float tot = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 pr = comb2<HALF>(u1, u2, r) *
                           f32x2{k_cu[r], k_cu[r + 1]};
          tot += pr[0] + pr[1];
        }
        tot = fmd_xor32_sum(tot);
        if (h == 0) red[w * 32 + c32] = tot;
        FMD_STAMP(3)
        __syncthreads();     // D(t): gW consumed, partial sums visible
        FMD_STAMP(4)
        if (w == 0 && h == 0) {
          const float sum = ((red[c32] + red[32 + c32]) + red[64 + c32]) + red[96 + c32];
          pgd[t * 32 + c32] = FIRST ? sum : gd_cu + sum;
        }
        put();               // hidden tiles of t+S
        rotate();
        FMD_STAMP(5)
        __syncthreads();     // A(t+S)
        FMD_STAMP(6)
      }
    } else {
      // ------------------------------------------------------------------ consumer waves
      const bool slice_owner = !FS || w < 2;  // FS: waves 2, 3 only share Gaussians and keep the barriers
      const int cw = FS ? 2 * split + (w & 1) : w;
      f16x8 b1h[8], b1l[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        b1h[s] = reinterpret_cast<const f16x8*>(a.q1h)[(cw * 8 + s) * 64 + lane];
        if (!HALF) b1l[s] = reinterpret_cast<const f16x8*>(a.q1l)[(cw * 8 + s) * 64 + lane];
      }
      const int col = 32 * cw + c32;
      const uint32_t col4 = (uint32_t)col * 4u;
      const int lds_adj = (32 * (w & 1) + c32) * 4 - (int)(col4 >> 1);
      auto lo = [&](uint32_t o) -> uint32_t { return FS ? (uint32_t)((int)(o >> 1) + lds_adj) : o; };
      const char* x1b = reinterpret_cast<const char*>(a.x1 + a0 * F);
      const char* gab = reinterpret_cast<const char*>(a.gagg + a0 * F);
      char* gx1b = reinterpret_cast<char*>(gx1);
      SegRegsB X;
      uint32_t oj_m[16], oi_m[4];
      float d_m = 0.f;  // SHARE: distance of pair c32 of the fetched tile
      auto fetch_meta = [&](int t) {
        const int base = min(t * 32, last_base);
        if (SHARE) d_m = pdist[base + c32];
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
      auto share_fragments = [&]() {  // of the tile whose meta were fetched last
        if (w < NS0 && 2 * w + h < a.nb) {
          const float d = d_m;
          const f32x4 c0 = *reinterpret_cast<const f32x4*>(cen_l + 16 * w + 8 * h);
          const f32x4 c1 = *reinterpret_cast<const f32x4*>(cen_l + 16 * w + 8 * h + 4);
          f16x8 bh, bl, ph, pl;
          rbf_fragments<HALF, HALF>(d, gl, c0, c1, bh, bl, ph, pl);
          const int blk = 2 * w + h;
          *reinterpret_cast<f16x8*>(frag_l + ((blk * 2 + 0) * 32 + c32) * 4) = bh;
          *reinterpret_cast<f16x8*>(frag_l + ((blk * 2 + 1) * 32 + c32) * 4) = HALF ? ph : bl;
        }
      };
      fetch_meta(T0);
      fill();
      if (SHARE) share_fragments();
      fetch_meta(T0 + S);
      if (SHARE) {
        __syncthreads();  // first tile's fragments visible
        __syncthreads();  // ... and consumed
        share_fragments();
      }
      __syncthreads();  // A(first)
      for (int t = T0; t < ntiles; t += S) {
        if (!slice_owner) {  // FS, waves 2 and 3: distances for the shared Gaussians and the barrier protocol
          fetch_meta(t + 2 * S);
          __syncthreads();  // B(t)
          __syncthreads();  // C(t)
          if (SHARE) share_fragments();
          __syncthreads();  // D(t)
          __syncthreads();  // A(t+S)
          continue;
        }
        // L1 recomputed: filt slice (lane = feature, registers = pairs)
        f32x16 acc1 = zero16(), acc2 = zero16();
        {
          // fragments are read two k-steps ahead of their MFMAs; the fences keep the scheduler from
          // hoisting all sixteen 16-byte reads (64 registers) to the top of the loop
          const _Float16* ph = hid_hi + c32 * LDH + 8 * h;
          const _Float16* pl = hid_lo + c32 * LDH + 8 * h;
          f16x8 ah0 = *reinterpret_cast<const f16x8*>(ph), al0 = *reinterpret_cast<const f16x8*>(pl);
          f16x8 ah1 = *reinterpret_cast<const f16x8*>(ph + 16), al1 = *reinterpret_cast<const f16x8*>(pl + 16);
          static_for<0, 8>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            f16x8 ahn = ah1, aln = al1;
            if constexpr (s + 2 < 8) {
              ahn = *reinterpret_cast<const f16x8*>(ph + 16 * (s + 2));
              aln = *reinterpret_cast<const f16x8*>(pl + 16 * (s + 2));
            }
            mfma_prod<HALF>(ah0, al0, b1h[s], b1l[s], acc1, acc2);
            ah0 = ah1; al0 = al1; ah1 = ahn; al1 = aln;
            __builtin_amdgcn_sched_barrier(0);
          });
        }
        f32x16 fl;
#pragma unroll
        for (int r = 0; r < 16; ++r) fl[r] = HALF ? acc1[r] : acc1[r] + acc2[r] * SPLIT_INV;
        // gW = x1[i] g[j] + x1[j] g[i], split two values at a time: gwh[p] = hi halves of entries 2p, 2p+1,
        // gwl[p] = their lo halves
        f16x2 gwh[8], gwl[8];
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
        asm volatile("" : "+v"(fl));  // L1 (and its LDS reads) complete before the barrier
        // The aggregate update and the refill of X need nothing from the producers: they run here,
        // while the producers are still staging the next tile, not between the later barriers.
        if (NEED_GX1) {   // gx1[i] += sum_j g[j] W, gx1[j] += g[i] W  (segment: one centre, distinct j)
          static_for<0, 4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            float Wq[4], vj[4];
            float si = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              Wq[e] = fl[4 * q + e];  // already filt * C
              vj[e] = *reinterpret_cast<const float*>(gx1b + lo(X.oj[4 * q + e]));
              si += X.gj[4 * q + e] * Wq[e];
            }
            const float cval = *reinterpret_cast<const float*>(gx1b + lo(X.oi[q]));
#pragma unroll
            for (int e = 0; e < 4; ++e)
              *reinterpret_cast<float*>(gx1b + lo(X.oj[4 * q + e])) = vj[e] + X.gi[q] * Wq[e];
            si = fmd_segment_sum(si, X.oi[q]);
            *reinterpret_cast<float*>(gx1b + lo(X.oi[q])) = cval + si;
          });
        }
        FMD_STAMP(0)
        fill();           // X <- tile t+S
        fetch_meta(t + 2 * S);
        FMD_STAMP(4)
        __syncthreads();  // B(t): hidden tiles consumed
        FMD_STAMP(1)
#pragma unroll
        for (int r = 0; r < 16; ++r) {  // ds_write_b16 of each half
          hid_hi[tile_row(r, h) * LDH + col] = gwh[r >> 1][r & 1];
          if (!HALF) hid_lo[tile_row(r, h) * LDH + col] = gwl[r >> 1][r & 1];
        }
        FMD_STAMP(2)
        __syncthreads();  // C(t): gW visible
        FMD_STAMP(3)
        // the producers read fragments only while staging (between A and B): the tile they stage in the
        // next iteration, t + 2S, is written here, in the consumers' idle window
        if (SHARE) share_fragments();
        __syncthreads();  // D(t)
        FMD_STAMP(5)
        __syncthreads();  // A(t+S)
        FMD_STAMP(6)
      }
    }
  }
  if (STAMP && lane == 0) {
    unsigned long long* o = a.stamps2 + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) * 16 + (producer ? 8 : 0);
    for (int k = 0; k < 7; ++k) o[k] = ts[k];
  }
#undef FMD_STAMP
  if (NEED_GX1) {
    __syncthreads();
    float* out = a.slabs + split * a.slab_stride + a0 * F;
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
}

template <int KS0, bool NEED_GX1, bool FIRST>
__global__ __launch_bounds__(256, 1) void edge_bwd_kernel(EdgeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* buf = smem;                       // [32][LDA]: hidden tile, then gW tile
  float* red = smem + TILE_FLOATS;         // [4][32]
  float* gx1 = smem + TILE_FLOATS + 128;   // [n][128]
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, c32 = lane & 31;
  int m, split;
  wg_to_mol_split(m, split);
  const int64_t a0 = a.mol_ptr[m];
  const int n = min((int)(a.mol_ptr[m + 1] - a0), a.max_beads);  // LDS is sized for max_beads (pairs_kernel flags a larger molecule)
  const int np = a.npairs[m];
  if (NEED_GX1)
    for (int i = tid * 4; i < n * F; i += 1024) *reinterpret_cast<f32x4*>(gx1 + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  float w0[KS0], w1[64], w1t[64], b0r[16], cen[KS0];
#pragma unroll
  for (int s = 0; s < KS0; ++s) w0[s] = a.p0[(w * 32 + s) * 64 + lane];
#pragma unroll
  for (int s = 0; s < 64; ++s) w1[s] = a.p1[(w * 64 + s) * 64 + lane];
#pragma unroll
  for (int s = 0; s < 64; ++s) w1t[s] = a.p1t[(w * 64 + s) * 64 + lane];
#pragma unroll
  for (int r = 0; r < 16; ++r) b0r[r] = a.b0[32 * w + tile_row(r, h)];
#pragma unroll
  for (int s = 0; s < KS0; ++s) cen[s] = a.centers[2 * s + h];
  const int col = 32 * w + c32;
  const uint32_t col4 = (uint32_t)col * 4u;
  const char* x1b = reinterpret_cast<const char*>(a.x1 + a0 * F);
  const char* gab = reinterpret_cast<const char*>(a.gagg + a0 * F);
  char* gx1b = reinterpret_cast<char*>(gx1);
  const uint32_t* poi = a.poi + (int64_t)m * a.cap;
  const uint32_t* poj = a.poj + (int64_t)m * a.cap;
  const float* pdist = a.pd + (int64_t)m * a.cap;
  const float* pcut = a.pc + (int64_t)m * a.cap;
  const float* pdcut = a.pdc + (int64_t)m * a.cap;
  float* pgd = a.pgd + (int64_t)m * a.cap;
  const int ntiles = (np + 31) >> 5;
  const int S = a.n_split;

  // tile meta one tile ahead (see edge_fwd_kernel); wave 0 also prefetches the running dE/dd
  float d_n, c_n, dc_n, gd_n = 0.f;
  u32x4 oi_n[4], oj_n[4];
  f32x4 cr_n[4];
  auto fetch = [&](int t) {
    const int base = min(t * 32, a.cap - 32);
    d_n = pdist[base + c32];
    c_n = pcut[base + c32];
    dc_n = pdcut[base + c32];
    if (!FIRST) gd_n = pgd[base + c32];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      oi_n[q] = *reinterpret_cast<const u32x4*>(poi + base + 8 * q + 4 * h);
      oj_n[q] = *reinterpret_cast<const u32x4*>(poj + base + 8 * q + 4 * h);
      cr_n[q] = *reinterpret_cast<const f32x4*>(pcut + base + 8 * q + 4 * h);
    }
  };
  fetch(split);
  __syncthreads();
  for (int t = split; t < ntiles; t += S) {
    const float d = d_n, C = c_n, dC = dc_n, gd_old = gd_n;
    uint32_t oi_c[16], oj_c[16];
    float cr_c[16], xi[16], xj[16], gi[16], gj[16];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        oi_c[4 * q + e] = oi_n[q][e] + col4;
        oj_c[4 * q + e] = oj_n[q][e] + col4;
        cr_c[4 * q + e] = cr_n[q][e];
      }
    fetch(t + S);
    // gathers of x1 and d/d(agg) rows: in flight under the L0 / L1 MFMAs
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      xi[r] = *reinterpret_cast<const float*>(x1b + oi_c[r]);
      xj[r] = *reinterpret_cast<const float*>(x1b + oj_c[r]);
      gi[r] = *reinterpret_cast<const float*>(gab + oi_c[r]);
      gj[r] = *reinterpret_cast<const float*>(gab + oj_c[r]);
    }
    // ---- L0 and its forward-mode derivative: z = W0 rbf(d), z' = W0 d(rbf)/dd
    f32x16 az = zero16(), azp = zero16();
#pragma unroll
    for (int s = 0; s < KS0; ++s) {
      const float diff = d - cen[s];
      const float raw = __expf(a.gamma * diff * diff);
      const float b = raw * C;
      const float bp = raw * (2.0f * a.gamma * diff * C + dC);
      az = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[s], b, az, 0, 0, 0);
      azp = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[s], bp, azp, 0, 0, 0);
    }
    float hr[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hr[4 * q + e] = fmd_tanh_fast(az[4 * q + e] + b0r[4 * q + e]);
        v[e] = hr[4 * q + e];
      }
      *reinterpret_cast<f32x4*>(buf + c32 * LDA + 32 * w + 8 * q + 4 * h) = v;
    }
    __syncthreads();
    // ---- L1 recomputed: filt slice (lane = feature, registers = pairs)
    f32x16 fl = zero16();
    {
      const float* ap = buf + c32 * LDA + 64 * h;
#pragma unroll
      for (int s4 = 0; s4 < 16; ++s4) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + 4 * s4);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          fl = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q], w1[4 * s4 + q], fl, 0, 0, 0);
      }
    }
    __syncthreads();  // every wave is done with the hidden tile; buf becomes the gW tile
    // ---- gW = x1[i] gagg[j] + x1[j] gagg[i]
#pragma unroll
    for (int r = 0; r < 16; ++r) buf[tile_row(r, h) * LDA + col] = xi[r] * gj[r] + xj[r] * gi[r];
    __syncthreads();
    // ---- u^T slice = W1^T slice . gW^T   (lane = pair, registers = hidden features), with the
    //      aggregate update gx1[i] += gagg[j] W, gx1[j] += gagg[i] W in the same scheduling region
    f32x16 u = zero16();
    {
      const float* bp = buf + c32 * LDA + 64 * h;
#pragma unroll
      for (int s4 = 0; s4 < 16; ++s4) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + 4 * s4);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          u = __builtin_amdgcn_mfma_f32_32x32x2f32(w1t[4 * s4 + q], b4[q], u, 0, 0, 0);
      }
    }
    if (NEED_GX1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float Wq[4], vj[4];
        float si = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          Wq[e] = fl[4 * q + e] * cr_c[4 * q + e];
          vj[e] = *reinterpret_cast<float*>(gx1b + oj_c[4 * q + e]);
          si += gj[4 * q + e] * Wq[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
          *reinterpret_cast<float*>(gx1b + oj_c[4 * q + e]) = vj[e] + gi[4 * q + e] * Wq[e];
        si = fmd_segment_sum(si, oi_c[4 * q]);
        float* ci = reinterpret_cast<float*>(gx1b + oi_c[4 * q]);
        *ci += si;
      }
    }
    // dE/dd = sum_h [C u (1-h^2)] z'  +  C' sum_h u h
    float g_rbf = 0.f, g_c = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float hh = hr[r];
      g_c += u[r] * hh;
      g_rbf += u[r] * (1.0f - hh * hh) * azp[r];
    }
    float tot = C * g_rbf + dC * g_c;
    tot = fmd_xor32_sum(tot);
    if (h == 0) red[w * 32 + c32] = tot;
    __syncthreads();
    if (w == 0 && h == 0) {
      const float sum = ((red[c32] + red[32 + c32]) + red[64 + c32]) + red[96 + c32];
      pgd[t * 32 + c32] = FIRST ? sum : gd_old + sum;
    }
  }
  if (NEED_GX1) {
    __syncthreads();
    float* out = a.slabs + split * a.slab_stride + a0 * F;
    for (int i = tid * 4; i < n * F; i += 1024)
      *reinterpret_cast<f32x4*>(out + i) = *reinterpret_cast<const f32x4*>(gx1 + i);
  }
}

// ---------------------------------------------------------------------------------------------
// node kernels (32-row tiles, chained 128x128 slices through two LDS tile buffers)
// ---------------------------------------------------------------------------------------------
// this lane's 16 elements (its column, the tile's rows) of a [N,128] array: issued at the top of a tile so
// the loads fly under the GEMM chain instead of adding a memory latency to every stage
__device__ __forceinline__ f32x16 prefetch_col(const float* __restrict__ src, int64_t row0, int64_t n_rows,
                                               int col, int h) {
  f32x16 v;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t row = row0 + tile_row(r, h);
    v[r] = row < n_rows ? src[row * F + col] : 0.f;
  }
  return v;
}
struct NodeFwdArgs {
  float* x;            // [N,128] in/out
  const float* slabs;  // agg partials of the previous block
  int64_t slab_stride;
  int n_slabs;
  const float *l2, *b2, *l3, *b3;  // previous block's lin2 / lin
  float* t_out;        // tanh(conv) of the previous block (kept for backward)
  const float* l1;     // this block's lin1
  float* x1_out;
  int64_t n_nodes;
};

template <bool SPLIT>
__global__ __launch_bounds__(256) void node_fwd_kernel(NodeFwdArgs a) {
  __shared__ __attribute__((aligned(16))) float bufA[TILE_FLOATS];
  __shared__ __attribute__((aligned(16))) float bufB[TILE_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
  const int col = 32 * w + (lane & 31);
  const int64_t n_tiles = (a.n_nodes + 31) / 32;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * 32;
    const f32x16 x_old = prefetch_col(a.x, row0, a.n_nodes, col, h);
    rows_to_lds<SPLIT>(bufA, a.slabs, a.slab_stride, a.n_slabs, row0, a.n_nodes, tid);
    __syncthreads();
    f32x16 acc = gemm_rows<SPLIT>(bufA, a.l2, w, lane);  // conv = agg lin2^T + b2
    const float b2 = a.b2[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[r] = fmd_tanh_fast(acc[r] + b2);
      const int64_t row = row0 + tile_row(r, h);
      if (row < a.n_nodes) a.t_out[row * F + col] = acc[r];
    }
    tile_store<SPLIT>(bufB, acc, w, lane);
    __syncthreads();
    acc = gemm_rows<SPLIT>(bufB, a.l3, w, lane);  // y = tanh(conv) lin^T + b3 ; x += y
    const float b3 = a.b3[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = row0 + tile_row(r, h);
      float xn = 0.f;
      if (row < a.n_nodes) {
        xn = x_old[r] + (acc[r] + b3);
        a.x[row * F + col] = xn;
      }
      acc[r] = xn;
    }
    tile_store<SPLIT>(bufA, acc, w, lane);
    __syncthreads();
    const f32x16 x1 = gemm_rows<SPLIT>(bufA, a.l1, w, lane);  // x1 = x lin1^T
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = row0 + tile_row(r, h);
      if (row < a.n_nodes) a.x1_out[row * F + col] = x1[r];
    }
    __syncthreads();
  }
}

struct NodeOutArgs {
  float* x;
  const float* slabs;
  int64_t slab_stride;
  int n_slabs;
  const float *l2, *b2, *l3, *b3;
  float* t_out;
  const float* po[3];   // packed output linears
  const float* bo[3];   // biases (zero-filled when absent)
  const float* pot[3];  // packed transposes
  float* e_node;        // [N]
  float* gx;            // [N,128] dE/dx_L
  int64_t n_nodes;
};

// HOUT: W16A16 output network (fp16 operands and stored activations, fp32 accumulation; reference
// models/gptq.py:266-306 with the rounding points of SURVEY row A9)
template <int NUM_OUT, bool SPLIT, bool HOUT>
__global__ __launch_bounds__(256) void node_out_kernel(NodeOutArgs a) {
  __shared__ __attribute__((aligned(16))) float bufA[TILE_FLOATS];
  __shared__ __attribute__((aligned(16))) float bufB[TILE_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
  const int col = 32 * w + (lane & 31);
  const int64_t n_tiles = (a.n_nodes + 31) / 32;
  // the 2 + 2 NUM_OUT GEMMs of a tile, in order: lin2, lin, output net forward, output net backward
  constexpr int NOPS = 2 + 2 * NUM_OUT;
  const float* ops[NOPS];
  ops[0] = a.l2;
  ops[1] = a.l3;
#pragma unroll
  for (int k = 0; k < NUM_OUT; ++k) {
    ops[2 + k] = a.po[k];
    ops[2 + NUM_OUT + k] = a.pot[NUM_OUT - 1 - k];
  }
  // fp32-grade form: operand fragments double-buffered in registers, the next GEMM's operand is requested (L2)
  // before the current GEMM runs (this kernel has one wave per SIMD: the registers are there)
  constexpr bool PIPE = SPLIT && !HOUT;
  BFrag fb0, fb1;
  auto gemm = [&](auto idx_c, const float* tile, bool more_tiles) -> f32x16 {
    constexpr int idx = decltype(idx_c)::value;
    if constexpr (PIPE) {
      BFrag& cur = (idx & 1) ? fb1 : fb0;
      BFrag& nxt = (idx & 1) ? fb0 : fb1;
      if constexpr (idx + 1 < NOPS) load_bfrag(nxt, ops[idx + 1], w, lane);
      else if (more_tiles) load_bfrag(nxt, ops[0], w, lane);  // NOPS is even: nxt is fb0 again
      return gemm_rows_frag(tile, cur, lane);
    } else if constexpr (HOUT && idx >= 2) {
      return gemm_rows_half(tile, ops[idx], w, lane);
    } else {
      return gemm_rows_nohoist<SPLIT>(tile, ops[idx], w, lane);
    }
  };
  if (PIPE && (int64_t)blockIdx.x < n_tiles) load_bfrag(fb0, ops[0], w, lane);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * 32;
    const bool more = tile + gridDim.x < n_tiles;
    const f32x16 x_old = prefetch_col(a.x, row0, a.n_nodes, col, h);
    rows_to_lds<SPLIT>(bufA, a.slabs, a.slab_stride, a.n_slabs, row0, a.n_nodes, tid);
    __syncthreads();
    f32x16 acc = gemm(std::integral_constant<int, 0>{}, bufA, more);
    const float b2 = a.b2[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[r] = fmd_tanh_fast(acc[r] + b2);
      const int64_t row = row0 + tile_row(r, h);
      if (row < a.n_nodes) a.t_out[row * F + col] = acc[r];
    }
    tile_store<SPLIT>(bufB, acc, w, lane);
    __syncthreads();
    acc = gemm(std::integral_constant<int, 1>{}, bufB, more);
    const float b3 = a.b3[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = row0 + tile_row(r, h);
      acc[r] = row < a.n_nodes ? x_old[r] + (acc[r] + b3) : 0.f;
    }
    if (HOUT) tile_store_half(bufA, acc, w, lane);
    else tile_store<SPLIT>(bufA, acc, w, lane);
    __syncthreads();
    // ---- output network forward (tanh hidden layers, last layer linear without bias)
    f32x16 hk[2];
    float* cur = bufA;
    float* nxt = bufB;
    static_for<0, NUM_OUT - 1>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      hk[k] = gemm(std::integral_constant<int, 2 + k>{}, cur, more);
      const float bk = a.bo[k][col];
#pragma unroll
      for (int r = 0; r < 16; ++r) hk[k][r] = fmd_tanh_fast(hk[k][r] + bk);
      if (HOUT) tile_store_half(nxt, hk[k], w, lane);  // hk <- f16(tanh): what the backward multiplies by
      else tile_store<SPLIT>(nxt, hk[k], w, lane);
      __syncthreads();
      float* tmp = cur; cur = nxt; nxt = tmp;
    });
    const f32x16 e = gemm(std::integral_constant<int, 2 + NUM_OUT - 1>{}, cur, more);
    if (col == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + tile_row(r, h);
        if (row < a.n_nodes) a.e_node[row] = e[r];
      }
    }
    // ---- output network backward: dE/de = 1 in column 0
    f32x16 g;
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = col == 0 ? 1.0f : 0.0f;
    if (HOUT) tile_store_half(nxt, g, w, lane);
    else tile_store<SPLIT>(nxt, g, w, lane);
    __syncthreads();
    { float* tmp = cur; cur = nxt; nxt = tmp; }
    static_for<0, NUM_OUT - 1>([&](auto jc) {
      constexpr int j = decltype(jc)::value;   // operand pot[NUM_OUT-1-j], multiplies by 1 - hk[NUM_OUT-2-j]^2
      constexpr int k = NUM_OUT - 1 - j;
      g = gemm(std::integral_constant<int, 2 + NUM_OUT + j>{}, cur, more);
#pragma unroll
      for (int r = 0; r < 16; ++r) g[r] *= 1.0f - hk[k - 1][r] * hk[k - 1][r];
      if (HOUT) tile_store_half(nxt, g, w, lane);  // f16(g (1 - y^2)) before the dot (cfconv_kernels.py:1019-1023)
      else tile_store<SPLIT>(nxt, g, w, lane);
      __syncthreads();
      float* tmp = cur; cur = nxt; nxt = tmp;
    });
    g = gemm(std::integral_constant<int, NOPS - 1>{}, cur, more);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = row0 + tile_row(r, h);
      if (row < a.n_nodes) a.gx[row * F + col] = g[r];
    }
    __syncthreads();
  }
}

struct NodeBwdArgs {
  float* gx;             // [N,128] dE/dx_{l+1} in, dE/dx_l... (updated when slabs are given)
  const float* slabs;    // gx1 partials of block l+1
  int64_t slab_stride;
  int n_slabs;
  const float* l1t;      // lin1 of block l+1, transposed operand
  const float* t;        // tanh(conv) of block l
  const float *l3t, *l2t;
  float* gagg;           // [N,128] out
  int64_t n_nodes;
};

template <bool HAS_SLABS, bool SPLIT>
__global__ __launch_bounds__(256) void node_bwd_kernel(NodeBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float bufA[TILE_FLOATS];
  __shared__ __attribute__((aligned(16))) float bufB[TILE_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
  const int col = 32 * w + (lane & 31);
  const int64_t n_tiles = (a.n_nodes + 31) / 32;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = tile * 32;
    const f32x16 t_pre = prefetch_col(a.t, row0, a.n_nodes, col, h);
    f32x16 gx_pre;
    if (HAS_SLABS) gx_pre = prefetch_col(a.gx, row0, a.n_nodes, col, h);
    f32x16 acc;
    if (HAS_SLABS) {  // gx += (sum of gx1 partials) lin1
      rows_to_lds<SPLIT>(bufA, a.slabs, a.slab_stride, a.n_slabs, row0, a.n_nodes, tid);
      __syncthreads();
      acc = gemm_rows<SPLIT>(bufA, a.l1t, w, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + tile_row(r, h);
        float v = 0.f;
        if (row < a.n_nodes) {
          v = gx_pre[r] + acc[r];
          a.gx[row * F + col] = v;
        }
        acc[r] = v;
      }
      tile_store<SPLIT>(bufB, acc, w, lane);
    } else {
      rows_to_lds<SPLIT>(bufB, a.gx, 0, 1, row0, a.n_nodes, tid);
    }
    __syncthreads();
    acc = gemm_rows<SPLIT>(bufB, a.l3t, w, lane);  // d/d tanh(conv) = gx lin
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = row0 + tile_row(r, h);
      const float tv = t_pre[r];
      acc[r] *= 1.0f - tv * tv;
    }
    tile_store<SPLIT>(bufA, acc, w, lane);
    __syncthreads();
    acc = gemm_rows<SPLIT>(bufA, a.l2t, w, lane);  // d/d agg = gconv lin2
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = row0 + tile_row(r, h);
      if (row < a.n_nodes) a.gagg[row * F + col] = acc[r];
    }
    __syncthreads();
  }
}

// first node launch of a timestep: x = embedding[type], x1 = (embedding lin1^T)[type] from the packed table
// (32 lanes x 16 bytes per bead and array)
__global__ __launch_bounds__(256) void node_embed_kernel(const int64_t* __restrict__ types,
                                                         const float* __restrict__ emb,
                                                         const float* __restrict__ emb_x1,
                                                         float* __restrict__ x, float* __restrict__ x1,
                                                         int64_t n_nodes) {
  const int64_t total = n_nodes * 32, stride = (int64_t)gridDim.x * 256;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
    const int64_t n = idx >> 5;
    const int c = (int)(idx & 31) * 4;
    const int64_t ty = types[n];
    *reinterpret_cast<f32x4*>(x + n * F + c) = *reinterpret_cast<const f32x4*>(emb + ty * F + c);
    *reinterpret_cast<f32x4*>(x1 + n * F + c) = *reinterpret_cast<const f32x4*>(emb_x1 + ty * F + c);
  }
}

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
};
__device__ __forceinline__ void rows_fetch(RowsPre& p, const float* __restrict__ src, int64_t slab_stride,
                                           int n_slabs, int64_t row0, int64_t n_rows, int tid) {
  const int r = tid >> 3, c0 = (tid & 7) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    p.s0[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    p.s1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row0 + r < n_rows) {
      const float* q = src + (row0 + r) * F + c0 + 32 * i;
      p.s0[i] = *reinterpret_cast<const f32x4*>(q);
      if (n_slabs > 1) p.s1[i] = *reinterpret_cast<const f32x4*>(q + slab_stride);
    }
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
                                            int64_t slab_stride, int n_slabs, int64_t row0, int64_t n_rows,
                                            int tid) {
  const int r = tid >> 3, c0 = (tid & 7) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + 32 * i;
    f32x4 v = p.s0[i];
    if (n_slabs > 1) v += p.s1[i];
    if (n_slabs > 2 && row0 + r < n_rows) {
      const float* q = src + (row0 + r) * F + c;
      for (int s = 2; s < n_slabs; ++s) v += *reinterpret_cast<const f32x4*>(q + s * slab_stride);
    }
    planes_store4(tile, r, c, v);
  }
}
// this lane's 16 columns of one row of a [N,128] array (zeros when the row does not exist)
__device__ __forceinline__ f32x16 load_cols(const float* __restrict__ src, int64_t row, bool ok, int w, int h) {
  f32x16 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (ok) t = *reinterpret_cast<const f32x4*>(src + row * F + 32 * w + 8 * q + 4 * h);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * q + e] = t[e];
  }
  return v;
}
__device__ __forceinline__ void store_cols(float* __restrict__ dst, int64_t row, bool ok, int w, int h,
                                           const f32x16& v) {
  if (!ok) return;
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
  rows_fetch(pre, a.slabs, a.slab_stride, a.n_slabs, tile * 32, a.n_nodes, tid);
  f32x16 xo = load_cols(a.x, tile * 32 + nl, tile * 32 + nl < a.n_nodes, w, h);
  load_bfrag(w2, a.l2, w, lane);
  load_bfrag(w3, a.l3, w, lane);
  load_bfrag(w1, a.l1, w, lane);
  const f32x16 b2 = load_cols(a.b2, 0, true, w, h), b3 = load_cols(a.b3, 0, true, w, h);
  for (; tile < n_tiles; tile += gridDim.x) {
    const int64_t row = tile * 32 + nl, next = tile + gridDim.x;
    const bool ok = row < a.n_nodes;
    rows_commit(bufX, pre, a.slabs, a.slab_stride, a.n_slabs, tile * 32, a.n_nodes, tid);
    const f32x16 x_old = xo;
    if (next < n_tiles) {
      rows_fetch(pre, a.slabs, a.slab_stride, a.n_slabs, next * 32, a.n_nodes, tid);
      xo = load_cols(a.x, next * 32 + nl, next * 32 + nl < a.n_nodes, w, h);
    }
    __syncthreads();
    f32x16 acc = gemm_cols_frag(bufX, w2, lane);  // conv = agg lin2^T + b2
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 t = tanh_fast2(f32x2{acc[r], acc[r + 1]} + f32x2{b2[r], b2[r + 1]});
      acc[r] = t[0];
      acc[r + 1] = t[1];
    }
    store_cols(a.t_out, row, ok, w, h, acc);
    cols_to_planes(bufY, acc, w, lane);
    __syncthreads();
    acc = gemm_cols_frag(bufY, w3, lane);  // y = tanh(conv) lin^T + b3 ; x += y
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = x_old[r] + (acc[r] + b3[r]);
    store_cols(a.x, row, ok, w, h, acc);
    cols_to_planes(bufZ, acc, w, lane);
    __syncthreads();
    const f32x16 x1 = gemm_cols_frag(bufZ, w1, lane);  // x1 = x lin1^T
    store_cols(a.x1_out, row, ok, w, h, x1);
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
    t_nx = load_cols(a.t, row, row < a.n_nodes, w, h);
    if (HAS_SLABS) {
      gx_nx = load_cols(a.gx, row, row < a.n_nodes, w, h);
      rows_fetch(pre, a.slabs, a.slab_stride, a.n_slabs, t * 32, a.n_nodes, tid);
    } else {
      rows_fetch(pre, a.gx, 0, 1, t * 32, a.n_nodes, tid);
    }
  };
  fetch(tile);
  if (HAS_SLABS) load_bfrag(w1, a.l1t, w, lane);
  load_bfrag(w3, a.l3t, w, lane);
  load_bfrag(w2, a.l2t, w, lane);
  for (; tile < n_tiles; tile += gridDim.x) {
    const int64_t row = tile * 32 + nl, next = tile + gridDim.x;
    const bool ok = row < a.n_nodes;
    const f32x16 t_pre = t_nx;
    f32x16 gx_pre;
    if (HAS_SLABS) {
      gx_pre = gx_nx;
      rows_commit(bufX, pre, a.slabs, a.slab_stride, a.n_slabs, tile * 32, a.n_nodes, tid);
    } else {
      rows_commit(bufY, pre, a.gx, 0, 1, tile * 32, a.n_nodes, tid);
    }
    if (next < n_tiles) fetch(next);
    f32x16 acc;
    if (HAS_SLABS) {  // gx += (sum of gx1 partials) lin1
      __syncthreads();
      acc = gemm_cols_frag(bufX, w1, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = gx_pre[r] + acc[r];
      store_cols(a.gx, row, ok, w, h, acc);
      cols_to_planes(bufY, acc, w, lane);
    }
    __syncthreads();
    acc = gemm_cols_frag(bufY, w3, lane);  // d/d tanh(conv) = gx lin
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] *= 1.0f - t_pre[r] * t_pre[r];
    cols_to_planes(bufZ, acc, w, lane);
    __syncthreads();
    acc = gemm_cols_frag(bufZ, w2, lane);  // d/d agg = gconv lin2
    store_cols(a.gagg, row, ok, w, h, acc);
    if (!HAS_SLABS) __syncthreads();  // bufY is refilled at the top of the next tile
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

// node_out in the transposed form (see node_fwd_res_kernel); its 2 + 2 NUM_OUT operands do not fit the register
// file, so they stay double-buffered: the next GEMM's operand is requested (L2) before the current GEMM runs
template <int NUM_OUT, bool HOUT>
__global__ __launch_bounds__(256) void node_out_res_kernel(NodeOutArgs a) {
  __shared__ __attribute__((aligned(16))) _Float16 bufX[2 * NPLANE];
  __shared__ __attribute__((aligned(16))) _Float16 bufY[2 * NPLANE];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, nl = lane & 31;
  const int64_t n_tiles = (a.n_nodes + 31) / 32;
  int64_t tile = blockIdx.x;
  if (tile >= n_tiles) return;
  constexpr int NOPS = 2 + 2 * NUM_OUT;
  const float* ops[NOPS];
  ops[0] = a.l2;
  ops[1] = a.l3;
#pragma unroll
  for (int k = 0; k < NUM_OUT; ++k) {
    ops[2 + k] = a.po[k];
    ops[2 + NUM_OUT + k] = a.pot[NUM_OUT - 1 - k];
  }
  BFrag fb0, fb1;
  auto gemm = [&](auto idx_c, const _Float16* t, bool more_tiles) -> f32x16 {
    constexpr int idx = decltype(idx_c)::value;
    BFrag& cur = (idx & 1) ? fb1 : fb0;
    BFrag& nxt = (idx & 1) ? fb0 : fb1;
    if constexpr (idx + 1 < NOPS) load_bfrag(nxt, ops[idx + 1], w, lane);
    else if (more_tiles) load_bfrag(nxt, ops[0], w, lane);  // NOPS is even: nxt is fb0 again
    if constexpr (HOUT && idx >= 2) return gemm_cols_half(t, cur, lane);
    else return gemm_cols_frag(t, cur, lane);
  };
  auto put = [&](_Float16* t, f32x16& v) {  // the output network's activations
    if (HOUT) cols_to_plane_half(t, v, w, lane);
    else cols_to_planes(t, v, w, lane);
  };
  RowsPre pre;
  rows_fetch(pre, a.slabs, a.slab_stride, a.n_slabs, tile * 32, a.n_nodes, tid);
  f32x16 xo = load_cols(a.x, tile * 32 + nl, tile * 32 + nl < a.n_nodes, w, h);
  load_bfrag(fb0, ops[0], w, lane);
  const f32x16 b2 = load_cols(a.b2, 0, true, w, h), b3 = load_cols(a.b3, 0, true, w, h);
  const bool col0 = w == 0 && h == 0;  // this lane's register 0 is column 0
  for (; tile < n_tiles; tile += gridDim.x) {
    const int64_t row = tile * 32 + nl, next = tile + gridDim.x;
    const bool ok = row < a.n_nodes, more = next < n_tiles;
    rows_commit(bufX, pre, a.slabs, a.slab_stride, a.n_slabs, tile * 32, a.n_nodes, tid);
    const f32x16 x_old = xo;
    if (more) {
      rows_fetch(pre, a.slabs, a.slab_stride, a.n_slabs, next * 32, a.n_nodes, tid);
      xo = load_cols(a.x, next * 32 + nl, next * 32 + nl < a.n_nodes, w, h);
    }
    __syncthreads();
    f32x16 acc = gemm(std::integral_constant<int, 0>{}, bufX, more);
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 t = tanh_fast2(f32x2{acc[r], acc[r + 1]} + f32x2{b2[r], b2[r + 1]});
      acc[r] = t[0];
      acc[r + 1] = t[1];
    }
    store_cols(a.t_out, row, ok, w, h, acc);
    cols_to_planes(bufY, acc, w, lane);
    __syncthreads();
    acc = gemm(std::integral_constant<int, 1>{}, bufY, more);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = ok ? x_old[r] + (acc[r] + b3[r]) : 0.f;
    put(bufX, acc);
    __syncthreads();
    // ---- output network forward (tanh hidden layers, last layer linear without bias)
    f32x16 hk[2];
    _Float16* cur = bufX;
    _Float16* nxt = bufY;
    static_for<0, NUM_OUT - 1>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      hk[k] = gemm(std::integral_constant<int, 2 + k>{}, cur, more);
      const f32x16 bk = load_cols(a.bo[k], 0, true, w, h);
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 t = tanh_fast2(f32x2{hk[k][r], hk[k][r + 1]} + f32x2{bk[r], bk[r + 1]});
        hk[k][r] = t[0];
        hk[k][r + 1] = t[1];
      }
      put(nxt, hk[k]);  // W16A16: hk <- f16(tanh), what the backward multiplies by
      __syncthreads();
      _Float16* tmp = cur; cur = nxt; nxt = tmp;
    });
    const f32x16 e = gemm(std::integral_constant<int, 2 + NUM_OUT - 1>{}, cur, more);
    if (col0 && ok) a.e_node[row] = e[0];
    // ---- output network backward: dE/de = 1 in column 0
    f32x16 g = zero16();
    if (col0) g[0] = 1.0f;
    put(nxt, g);
    __syncthreads();
    { _Float16* tmp = cur; cur = nxt; nxt = tmp; }
    static_for<0, NUM_OUT - 1>([&](auto jc) {
      constexpr int j = decltype(jc)::value;   // operand pot[NUM_OUT-1-j], multiplies by 1 - hk[NUM_OUT-2-j]^2
      constexpr int k = NUM_OUT - 1 - j;
      g = gemm(std::integral_constant<int, 2 + NUM_OUT + j>{}, cur, more);
#pragma unroll
      for (int r = 0; r < 16; ++r) g[r] *= 1.0f - hk[k - 1][r] * hk[k - 1][r];
      put(nxt, g);  // W16A16: f16(g (1 - y^2)) before the dot (cfconv_kernels.py:1019-1023)
      __syncthreads();
      _Float16* tmp = cur; cur = nxt; nxt = tmp;
    });
    g = gemm(std::integral_constant<int, NOPS - 1>{}, cur, more);
    store_cols(a.gx, row, ok, w, h, g);
    __syncthreads();
  }
}

// two workgroups (4 waves each) per molecule, each with half of its pair list: forces from dE/dd of the pairs,
// energy from the beads.  Each wave accumulates a private copy of dE/dpos in LDS (float LDS atomics are ordered
// inside one wave) and the four copies are added in fixed order; the two workgroups then add their halves to
// the force array that pairs_kernel zeroed: two addends commute, so the result is reproducible.
__global__ __launch_bounds__(256) void finish_kernel(const float* __restrict__ pos,
                                                     const int64_t* __restrict__ mol_ptr,
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
  const int64_t a0 = mol_ptr[m];
  const int n = min((int)(mol_ptr[m + 1] - a0), max_beads);
  float* g = smem + wv * 3 * n;  // this wave's [3n] dE/dpos
  __shared__ float esum[4];
  for (int i = tid; i < 12 * n; i += 256) smem[i] = 0.f;
  __syncthreads();
  const int np = npairs[m];
  const int np_half = ((np + 15) >> 4) << 3;  // the first workgroup's share (whole 8-entry segments)
  const int q_lo = half ? min(np, np_half) : 0, q_hi = half ? np : min(np, np_half);
  // every thread walks a contiguous run of the (centre-sorted) list: the centre's contribution is
  // summed in registers and flushed when the centre changes, so lanes rarely collide in LDS
  const int chunk = (q_hi - q_lo + 255) / 256;
  const int p_lo = min(q_hi, q_lo + tid * chunk), p_hi = min(q_hi, p_lo + chunk);
  int ci = -1;
  float cx = 0.f, cy = 0.f, cz = 0.f;
  for (int p = p_lo; p < p_hi; ++p) {
    const int i = (int)(poi[(int64_t)m * cap + p] / (F * 4));
    const int j = (int)(poj[(int64_t)m * cap + p] / (F * 4));
    if (i != ci) {
      if (ci >= 0) {
        atomicAdd(&g[3 * ci], cx);
        atomicAdd(&g[3 * ci + 1], cy);
        atomicAdd(&g[3 * ci + 2], cz);
      }
      ci = i;
      cx = cy = cz = 0.f;
    }
    if (i != j) {  // pad entries carry nothing
      const float d = fmaxf(pd[(int64_t)m * cap + p], 1e-8f);
      const float gdd = pgd[(int64_t)m * cap + p] + (pgd2 ? pgd2[(int64_t)m * cap + p] : 0.f);
      const float s = gdd / d;
      const float dx = s * (pos[3 * (a0 + j)] - pos[3 * (a0 + i)]);
      const float dy = s * (pos[3 * (a0 + j) + 1] - pos[3 * (a0 + i) + 1]);
      const float dz = s * (pos[3 * (a0 + j) + 2] - pos[3 * (a0 + i) + 2]);
      atomicAdd(&g[3 * j], dx);
      atomicAdd(&g[3 * j + 1], dy);
      atomicAdd(&g[3 * j + 2], dz);
      cx -= dx;
      cy -= dy;
      cz -= dz;
    }
  }
  if (ci >= 0) {
    atomicAdd(&g[3 * ci], cx);
    atomicAdd(&g[3 * ci + 1], cy);
    atomicAdd(&g[3 * ci + 2], cz);
  }
  if (half == 0) {
    float e = 0.f;
    for (int i = tid; i < n; i += 256) e += e_node[a0 + i];
    e = fmd_wave_sum(e);
    if (lane == 0) esum[wv] = e;
  }
  __syncthreads();
  for (int i = tid; i < 3 * n; i += 256)
    unsafeAtomicAdd(&forces[3 * a0 + i], -(((smem[i] + smem[3 * n + i]) + smem[6 * n + i]) + smem[9 * n + i]));
  if (half == 0 && tid == 0) energy[m] = ((esum[0] + esum[1]) + esum[2]) + esum[3];
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
      status[3] = 0;
    }
  }
}

void* g_stamp_buffer = nullptr;
void* g_stamp_buffer2 = nullptr;  // set by fmd_debug_set_stamp_buffer (diagnostics, never in timed runs)

int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

struct Workspace {
  float *x, *gx, *gagg, *e_node, *pd, *pc, *pdc, *pgd;
  float* x1[8];
  float* t[8];
  float* slabs;
  uint32_t *poi, *poj;
  int32_t* npairs;
  int32_t* report;  // [n_mol][3] pairs_kernel -> finish_kernel
  int64_t total;
};

constexpr int MAX_SPLIT = 32;  // workgroups per molecule (small batches: 8 molecules x 32 = 256 workgroups)
int device_cus() {
  int cus = 256;
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  return cus;
}
int resolve_split(const fmd_schnet_desc* d, int64_t n_mol) {
  if (d->n_split > 0) return d->n_split > MAX_SPLIT ? MAX_SPLIT : d->n_split;
  const int cus = device_cus();
  int64_t s = n_mol > 0 ? (cus + n_mol - 1) / n_mol : 1;
  if (s < 1) s = 1;
  if (s > MAX_SPLIT) s = MAX_SPLIT;
  return (int)s;
}

Workspace carve_ws(const fmd_schnet_desc* d, void* base, int64_t n_mol, int64_t n_nodes, int64_t cap,
                   int n_split) {
  Workspace w;
  char* p = reinterpret_cast<char*>(base);
  auto take = [&](int64_t bytes) {
    char* q = p;
    p += align256(bytes);
    return q;
  };
  const int64_t nf = n_nodes * F * (int64_t)sizeof(float);
  w.x = (float*)take(nf);
  w.gx = (float*)take(nf);
  w.gagg = (float*)take(nf);
  for (int l = 0; l < d->num_blocks; ++l) {
    w.x1[l] = (float*)take(nf);
    w.t[l] = (float*)take(nf);
  }
  w.slabs = (float*)take(nf * n_split);
  w.e_node = (float*)take(n_nodes * 4);
  w.poi = (uint32_t*)take(n_mol * cap * 4);
  w.poj = (uint32_t*)take(n_mol * cap * 4);
  w.pd = (float*)take(n_mol * cap * 4);
  w.pc = (float*)take(n_mol * cap * 4);
  w.pdc = (float*)take(n_mol * cap * 4);
  w.pgd = (float*)take(2 * n_mol * cap * 4);  // second slab: feature-split mode
  w.npairs = (int32_t*)take(n_mol * 4);
  w.report = (int32_t*)take(n_mol * 12);
  w.total = p - reinterpret_cast<char*>(base);
  return w;
}

bool desc_ok(const fmd_schnet_desc* d) {
  if (!d) return false;
  if (d->hidden != F || d->num_rbf < 2 || d->num_rbf > 64) return false;
  if (d->num_blocks < 1 || d->num_blocks > 8) return false;
  if (d->num_out < 2 || d->num_out > 3) return false;
  for (int k = 0; k < d->num_out; ++k)
    if (d->out_width[k] < 1 || d->out_width[k] > F) return false;
  if (d->out_width[d->num_out - 1] != 1) return false;
  if (d->precision != FMD_PREC_F32 && d->precision != FMD_PREC_F32_MFMA && d->precision != FMD_PREC_W16A16)
    return false;
  if (d->embedding_size < 1) return false;
  return true;
}

template <int KS0>
int run_edges(const fmd_schnet_desc* d, bool fwd, int l, int L, EdgeArgs& ea, dim3 grid, size_t lds,
              hipStream_t s) {
  if (fwd) {
    const char* mode = getenv("FMD_EDGE_FWD");  // experiments: "4wave" | "ws" | default split
    if (mode && mode[0] == '4') {
      if (ea.stamps) edge_fwd_kernel<KS0, true><<<grid, 256, lds, s>>>(ea);
      else edge_fwd_kernel<KS0, false><<<grid, 256, lds, s>>>(ea);
    } else if ((mode && mode[0] == 'w') || d->precision == FMD_PREC_F32_MFMA) {
      if (ea.stamps) edge_fwd_ws_kernel<KS0, true><<<grid, 512, lds, s>>>(ea);
      else edge_fwd_ws_kernel<KS0, false><<<grid, 512, lds, s>>>(ea);
    } else {
      constexpr int NS0 = (2 * KS0 + 15) / 16;
      const bool half = d->precision == FMD_PREC_W16A16;
      // shared-Gaussian form when its fragment region fits next to the aggregate
      const size_t lds_fsh = (size_t)(TILE_FLOATS + 128 + ea.nb * 256 + ea.max_beads * F) * sizeof(float);
      const char* mode_fsh = getenv("FMD_EDGE_FWD_SHARE");  // experiments: "0" disables
      // (not in W16A16: without operand splits the Gaussians are cheap for the producers, and the consumers'
      //  extra work between B and A costs more than it saves: 0.156 vs 0.144 ms)
      const bool fsh = !ea.fs && !half && !ea.stamps && lds_fsh <= MAX_LDS_BYTES && !(mode_fsh && mode_fsh[0] == '0');
      if (ea.fs && half) edge_fwd_split_kernel<NS0, false, true, true, false><<<grid, 512, lds, s>>>(ea);
      else if (ea.fs) edge_fwd_split_kernel<NS0, false, false, true, false><<<grid, 512, lds, s>>>(ea);
      else if (fsh) edge_fwd_split_kernel<NS0, false, false, false, true><<<grid, 512, lds_fsh, s>>>(ea);
      else if (half) edge_fwd_split_kernel<NS0, false, true, false, false><<<grid, 512, lds, s>>>(ea);
      else if (ea.stamps) edge_fwd_split_kernel<NS0, true, false, false, false><<<grid, 512, lds, s>>>(ea);
      else edge_fwd_split_kernel<NS0, false, false, false, false><<<grid, 512, lds, s>>>(ea);
    }
  } else {
    const bool first = (l == L - 1), need = (l > 0);
    const char* mode = getenv("FMD_EDGE_BWD");  // experiments: "4wave" | default split
    if ((mode && mode[0] == '4') || d->precision == FMD_PREC_F32_MFMA) {
      if (need && first) edge_bwd_kernel<KS0, true, true><<<grid, 256, lds, s>>>(ea);
      else if (need) edge_bwd_kernel<KS0, true, false><<<grid, 256, lds, s>>>(ea);
      else if (first) edge_bwd_kernel<KS0, false, true><<<grid, 256, lds, s>>>(ea);
      else edge_bwd_kernel<KS0, false, false><<<grid, 256, lds, s>>>(ea);
    } else {
      constexpr int NS0 = (2 * KS0 + 15) / 16;
      // shared-Gaussian form when its fragment region fits next to the aggregate (block 0 keeps none)
      const size_t frag = (size_t)ea.nb * 1024;
      const size_t lds_sh = need ? lds + frag : (size_t)(TILE_FLOATS + 128 + 128 + 64 + 128) * sizeof(float) + frag;
      const char* mode_sh = getenv("FMD_EDGE_BWD_SHARE");  // experiments: "0" disables
      const bool share = lds_sh <= MAX_LDS_BYTES && !(mode_sh && mode_sh[0] == '0');
      const bool half = d->precision == FMD_PREC_W16A16;
#define FMD_BWD(N_, F_, T_, S_, H_) edge_bwd_split_kernel<NS0, N_, F_, T_, S_, H_, false><<<grid, 512, (S_) ? lds_sh : lds, s>>>(ea)
#define FMD_BWD_FS(N_, F_, H_) edge_bwd_split_kernel<NS0, N_, F_, false, true, H_, true><<<grid, 512, lds_sh, s>>>(ea)
      if (ea.fs) {  // feature-split mode: the fragment region always fits beside 64-wide rows
        if (lds_sh > MAX_LDS_BYTES) return FMD_EUNSUPPORTED;
        if (half) {
          if (need && first) FMD_BWD_FS(true, true, true);
          else if (need) FMD_BWD_FS(true, false, true);
          else if (first) FMD_BWD_FS(false, true, true);
          else FMD_BWD_FS(false, false, true);
        } else {
          if (need && first) FMD_BWD_FS(true, true, false);
          else if (need) FMD_BWD_FS(true, false, false);
          else if (first) FMD_BWD_FS(false, true, false);
          else FMD_BWD_FS(false, false, false);
        }
      } else if (share && half) {
        if (need && first) FMD_BWD(true, true, false, true, true);
        else if (need) FMD_BWD(true, false, false, true, true);
        else if (first) FMD_BWD(false, true, false, true, true);
        else FMD_BWD(false, false, false, true, true);
      } else if (half) {
        if (need && first) FMD_BWD(true, true, false, false, true);
        else if (need) FMD_BWD(true, false, false, false, true);
        else if (first) FMD_BWD(false, true, false, false, true);
        else FMD_BWD(false, false, false, false, true);
      } else if (share) {
        if (need && first) FMD_BWD(true, true, false, true, false);
        else if (need && ea.stamps2) FMD_BWD(true, false, true, true, false);
        else if (need) FMD_BWD(true, false, false, true, false);
        else if (first) FMD_BWD(false, true, false, true, false);
        else FMD_BWD(false, false, false, true, false);
      } else {
        if (need && first) FMD_BWD(true, true, false, false, false);
        else if (need) FMD_BWD(true, false, false, false, false);
        else if (first) FMD_BWD(false, true, false, false, false);
        else FMD_BWD(false, false, false, false, false);
      }
#undef FMD_BWD
#undef FMD_BWD_FS
    }
  }
  return hipGetLastError() == hipSuccess ? FMD_OK : FMD_ELAUNCH;
}

template <int KS0>
hipError_t set_edge_attrs(size_t lds) {
  hipError_t e = hipFuncSetAttribute((const void*)edge_fwd_kernel<KS0, false>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)edge_fwd_kernel<KS0, true>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)edge_fwd_ws_kernel<KS0, false>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)edge_fwd_ws_kernel<KS0, true>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  {
    constexpr int NSF = (2 * KS0 + 15) / 16;
    const void* fwd[] = {(const void*)edge_fwd_split_kernel<NSF, false, false, false, false>,
                         (const void*)edge_fwd_split_kernel<NSF, true, false, false, false>,
                         (const void*)edge_fwd_split_kernel<NSF, false, true, false, false>,
                         (const void*)edge_fwd_split_kernel<NSF, false, false, true, false>,
                         (const void*)edge_fwd_split_kernel<NSF, false, true, true, false>,
                         (const void*)edge_fwd_split_kernel<NSF, false, false, false, true>};
    for (const void* f : fwd) {
      e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MAX_LDS_BYTES);
      if (e != hipSuccess) return e;
    }
  }
  e = hipFuncSetAttribute((const void*)edge_bwd_kernel<KS0, true, true>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)edge_bwd_kernel<KS0, true, false>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)edge_bwd_kernel<KS0, false, true>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)edge_bwd_kernel<KS0, false, false>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  constexpr int NS0 = (2 * KS0 + 15) / 16;
  const void* plain[] = {(const void*)edge_bwd_split_kernel<NS0, true, true, false, false, false, false>,
                         (const void*)edge_bwd_split_kernel<NS0, true, false, false, false, false, false>,
                         (const void*)edge_bwd_split_kernel<NS0, false, true, false, false, false, false>,
                         (const void*)edge_bwd_split_kernel<NS0, false, false, false, false, false, false>,
                         (const void*)edge_bwd_split_kernel<NS0, true, true, false, false, true, false>,
                         (const void*)edge_bwd_split_kernel<NS0, true, false, false, false, true, false>,
                         (const void*)edge_bwd_split_kernel<NS0, false, true, false, false, true, false>,
                         (const void*)edge_bwd_split_kernel<NS0, false, false, false, false, true, false>};
  for (const void* f : plain) {
    e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  const void* shared[] = {(const void*)edge_bwd_split_kernel<NS0, true, true, false, true, false, false>,
                          (const void*)edge_bwd_split_kernel<NS0, true, false, true, true, false, false>,
                          (const void*)edge_bwd_split_kernel<NS0, true, false, false, true, false, false>,
                          (const void*)edge_bwd_split_kernel<NS0, false, true, false, true, false, false>,
                          (const void*)edge_bwd_split_kernel<NS0, false, false, false, true, false, false>,
                          (const void*)edge_bwd_split_kernel<NS0, true, true, false, true, true, false>,
                          (const void*)edge_bwd_split_kernel<NS0, true, false, false, true, true, false>,
                          (const void*)edge_bwd_split_kernel<NS0, false, true, false, true, true, false>,
                          (const void*)edge_bwd_split_kernel<NS0, false, false, false, true, true, false>,
                          (const void*)edge_bwd_split_kernel<NS0, true, true, false, true, false, true>,
                          (const void*)edge_bwd_split_kernel<NS0, true, false, false, true, false, true>,
                          (const void*)edge_bwd_split_kernel<NS0, false, true, false, true, false, true>,
                          (const void*)edge_bwd_split_kernel<NS0, false, false, false, true, false, true>,
                          (const void*)edge_bwd_split_kernel<NS0, true, true, false, true, true, true>,
                          (const void*)edge_bwd_split_kernel<NS0, true, false, false, true, true, true>,
                          (const void*)edge_bwd_split_kernel<NS0, false, true, false, true, true, true>,
                          (const void*)edge_bwd_split_kernel<NS0, false, false, false, true, true, true>};
  for (const void* f : shared) {
    e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MAX_LDS_BYTES);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace

extern "C" int64_t fmd_schnet_pack_bytes(const fmd_schnet_desc* desc) {
  if (!desc_ok(desc)) return FMD_EINVAL;
  return pack_layout(desc).total * (int64_t)sizeof(float);
}

extern "C" int fmd_schnet_pack(const fmd_schnet_desc* desc, const fmd_schnet_weights* wt,
                               const float* centers, void* packed_host) {
  if (!desc_ok(desc) || !wt || !centers || !packed_host || !wt->embedding || !wt->blocks) return FMD_EINVAL;
  const PackLayout L = pack_layout(desc);
  float* out = reinterpret_cast<float*>(packed_host);
  // node operands in split (hi/lo fp16) form; W16A16 keeps lin1/lin2/lin fp32-grade like the reference
  const bool split = desc->precision != FMD_PREC_F32_MFMA;
  memset(out, 0, L.total * sizeof(float));
  memcpy(out + L.emb, wt->embedding, (size_t)desc->embedding_size * F * sizeof(float));
  for (int k = 0; k < desc->num_rbf; ++k) out[L.centers + k] = centers[k];
  // The first block's x1 = lin1(embedding[type]) only depends on the bead type: one table row per type (fp64
  // accumulation, rounded once), so the first node launch of a timestep is a gather instead of a GEMM.
  if (desc->num_blocks > 0 && wt->blocks[0].lin1_w) {
    const float* l1 = reinterpret_cast<const float*>(wt->blocks[0].lin1_w);
    const float* em = reinterpret_cast<const float*>(wt->embedding);
    for (int t = 0; t < desc->embedding_size; ++t)
      for (int o = 0; o < F; ++o) {
        double acc = 0.0;
        for (int k = 0; k < F; ++k) acc += (double)em[(int64_t)t * F + k] * (double)l1[(int64_t)o * F + k];
        out[L.emb_x1 + (int64_t)t * F + o] = (float)acc;
      }
  }
  for (int l = 0; l < desc->num_blocks; ++l) {
    const fmd_schnet_block_weights& b = wt->blocks[l];
    if (!b.lin1_w || !b.f0_w || !b.f0_b || !b.f1_w || !b.lin2_w || !b.lin2_b || !b.lin_w || !b.lin_b)
      return FMD_EINVAL;
    const BlockOff o = block_off(L, l);
    pack_p0(out + o.p0, b.f0_w, desc->num_rbf);
    memcpy(out + o.b0, b.f0_b, F * sizeof(float));
    if (desc->precision == FMD_PREC_W16A16)  // the fp16 filter network carries its bias in fp16 (gptq.py:178-184)
      for (int f = 0; f < F; ++f) out[o.b0 + f] = (float)(_Float16)out[o.b0 + f];
    pack_slice(out + o.p1, b.f1_w, F, F, false);
    pack_slice(out + o.p1t, b.f1_w, F, F, true);
    pack_node(out + o.l1, b.lin1_w, F, F, false, split);
    pack_node(out + o.l1t, b.lin1_w, F, F, true, split);
    pack_node(out + o.l2, b.lin2_w, F, F, false, split);
    memcpy(out + o.b2, b.lin2_b, F * sizeof(float));
    pack_node(out + o.l2t, b.lin2_w, F, F, true, split);
    pack_node(out + o.l3, b.lin_w, F, F, false, split);
    memcpy(out + o.b3, b.lin_b, F * sizeof(float));
    pack_node(out + o.l3t, b.lin_w, F, F, true, split);
    pack_split(out + o.q0h, out + o.q0l, b.f0_w, F, desc->num_rbf, 4, false);
    {  // Wc[f][k] = (centre_k - cutoff/2) W0[f][k]:  W0 (raw (d - centre)) = (d - cutoff/2) W0 raw - Wc raw
      std::vector<float> wc((size_t)F * desc->num_rbf);
      const float c0 = 0.5f * desc->cutoff;
      for (int f = 0; f < F; ++f)
        for (int k = 0; k < desc->num_rbf; ++k)
          wc[(size_t)f * desc->num_rbf + k] = (centers[k] - c0) * b.f0_w[(size_t)f * desc->num_rbf + k];
      pack_split(out + o.q0ch, out + o.q0cl, wc.data(), F, desc->num_rbf, 4, false);
    }
    pack_split(out + o.q1h, out + o.q1l, b.f1_w, F, F, 8, false);
    pack_split(out + o.q1th, out + o.q1tl, b.f1_w, F, F, 8, true);
  }
  int in_w = F;
  for (int k = 0; k < desc->num_out; ++k) {
    if (!wt->out_w[k]) return FMD_EINVAL;
    float* base = out + L.outs + k * L.out_stride;
    const int ow = desc->out_width[k];
    pack_node(base, wt->out_w[k], ow, in_w, false, split);
    if (wt->out_b[k]) memcpy(base + SLICE, wt->out_b[k], ow * sizeof(float));
    if (desc->precision == FMD_PREC_W16A16)
      for (int f = 0; f < F; ++f) base[SLICE + f] = (float)(_Float16)base[SLICE + f];
    pack_node(base + SLICE + F, wt->out_w[k], in_w, ow, true, split);  // (W^T)[o=in][k=out] = W[out][in]
    in_w = ow;
  }
  return FMD_OK;
}

extern "C" int64_t fmd_schnet_workspace_bytes(const fmd_schnet_desc* desc, int64_t n_mol,
                                              int64_t n_nodes, int64_t max_pairs_per_mol) {
  if (!desc_ok(desc) || n_mol < 0 || n_nodes < 0 || max_pairs_per_mol < 0) return FMD_EINVAL;
  const int64_t cap = (max_pairs_per_mol + 31) / 32 * 32;
  const int split = resolve_split(desc, n_mol);  // the slabs scale with it; >= 2 for the feature-split mode
  return carve_ws(desc, nullptr, n_mol, n_nodes, cap, split < 2 ? 2 : split).total + 256;
}

static int run_engine(const fmd_schnet_desc* desc, const void* packed_dev, const float* pos,
                      const int64_t* atom_types, const int64_t* mol_ptr, int64_t n_mol,
                      int64_t n_nodes, int64_t max_beads_per_mol, int64_t max_pairs_per_mol,
                      float* energy, float* forces, void* workspace, int32_t* status, void* stream,
                      hipEvent_t* ev) {
  if (!desc_ok(desc) || n_mol < 0 || n_nodes < 0) return FMD_EINVAL;
  if (n_mol == 0 || n_nodes == 0) return FMD_OK;
  if (!packed_dev || !pos || !atom_types || !mol_ptr || !energy || !forces || !workspace || !status)
    return FMD_EINVAL;
  if (max_beads_per_mol < 1 || max_beads_per_mol > MAX_LDS_ROWS_FS) return FMD_EUNSUPPORTED;
  // feature-split mode: the split kernels only (fp32-grade default and W16A16)
  const bool fs = max_beads_per_mol > MAX_LDS_ROWS;
  if (fs && desc->precision == FMD_PREC_F32_MFMA) return FMD_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return FMD_EINVAL;
  hipStream_t s = fmd_stream(stream);
  const int L = desc->num_blocks;
  const int64_t cap = (max_pairs_per_mol + 31) / 32 * 32;
  if (cap <= 0 || cap > (int64_t)1 << 30) return FMD_EINVAL;
  const int split = fs ? 2 : resolve_split(desc, n_mol);  // FS: the two feature halves
  const Workspace ws = carve_ws(desc, workspace, n_mol, n_nodes, cap, split);
  const PackLayout PL = pack_layout(desc);
  const float* pk = reinterpret_cast<const float*>(packed_dev);

  int kidx = 0;  // kernel index in launch order (fmd_schnet_kernel_name)
  // a NULL slot skips that record: a caller may time a single kernel without paying for 2 records per launch
  auto ev_begin = [&]() { if (ev && ev[2 * kidx]) (void)hipEventRecord(ev[2 * kidx], s); };
  auto ev_end = [&]() { if (ev && ev[2 * kidx + 1]) (void)hipEventRecord(ev[2 * kidx + 1], s); ++kidx; };
  // positions, counts / offsets / flags, [n][n / 64] 64-bit hit masks (53.5 KB at 535 beads), then the entry staging
  // area: as much of the worst-case list as fits in 128 KB (the list of a 1ENH-sized molecule takes 32 KB)
  const size_t lds_pairs_base =
      (7 * max_beads_per_mol + 2 + 2 * ((max_beads_per_mol + 63) / 64) * max_beads_per_mol) * sizeof(float);
  const size_t lds_pairs_max = 128 * 1024;
  const size_t lds_pairs = lds_pairs_base + (size_t)cap * 4 < lds_pairs_max ? lds_pairs_base + (size_t)cap * 4 : lds_pairs_max;
  const int ent_cap = (int)((lds_pairs - lds_pairs_base) / 4);
  {
    static std::atomic<int> pairs_attr[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return FMD_ELAUNCH;
    if (!pairs_attr[dev].load(std::memory_order_relaxed)) {
      if (hipFuncSetAttribute((const void*)pairs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_pairs_max) != hipSuccess)
        return FMD_ELAUNCH;
      pairs_attr[dev].store(1, std::memory_order_relaxed);
    }
  }
  ev_begin();
  pairs_kernel<<<dim3(2, (unsigned)n_mol), 1024, lds_pairs, s>>>(pos, mol_ptr, desc->cutoff * desc->cutoff,
                                                     desc->cutoff, (int)cap, ws.poi, ws.poj, ws.pd,
                                                     ws.pc, ws.pdc, ws.npairs, ws.report, (int)max_beads_per_mol, forces, ent_cap);
  ev_end();
  FMD_CHECK_LAUNCH();

  const int ks0 = (desc->num_rbf + 1) / 2;
  // [tile region 4352 floats][red 128][centres 64][bias 128][aggregate rows]
  const size_t lds_edge =
      (size_t)(TILE_FLOATS + 128 + 128 + 64 + 128 + max_beads_per_mol * (fs ? 64 : F)) * sizeof(float);
  // the dynamic-LDS ceilings of the pair kernels only need raising when a larger molecule shows up
  // (per device: the attribute belongs to the device's copy of the kernel)
  {
    static std::atomic<size_t> lds_set[3][64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return FMD_ELAUNCH;
    const int cls = ks0 <= 16 ? 0 : (ks0 <= 25 ? 1 : 2);
    if (lds_set[cls][dev].load(std::memory_order_relaxed) < lds_edge) {
      hipError_t he = cls == 0 ? set_edge_attrs<16>(lds_edge)
                      : cls == 1 ? set_edge_attrs<25>(lds_edge) : set_edge_attrs<32>(lds_edge);
      if (he != hipSuccess) return FMD_ELAUNCH;
      lds_set[cls][dev].store(lds_edge, std::memory_order_relaxed);
    }
  }

  const bool sp = desc->precision != FMD_PREC_F32_MFMA;
  const int64_t n_tiles = (n_nodes + 31) / 32;
  // node kernels: workgroups loop over 32-row tiles; FMD_NODE_GRID (experiments) overrides the grid
  int64_t ng_max = 768;
  if (const char* e = getenv("FMD_NODE_GRID")) ng_max = atoi(e) > 0 ? atoi(e) : 768;
  const unsigned node_grid = (unsigned)(n_tiles < ng_max ? n_tiles : ng_max);
  // register-resident node_fwd / node_bwd (split-operand engines): one workgroup per CU; FMD_NODE_RES=0 selects
  // the streamed kernels (experiments)
  bool node_res = sp;
  if (const char* e = getenv("FMD_NODE_RES")) node_res = sp && atoi(e) != 0;
  const int64_t cus = device_cus();
  const unsigned res_grid = (unsigned)(n_tiles < cus ? n_tiles : cus);
  int64_t og = cus;  // (one workgroup per CU is resident: the kernel needs most of the register file)
  if (const char* e = getenv("FMD_OUT_GRID")) og = atoi(e) > 0 ? atoi(e) : og;
  const unsigned out_grid = (unsigned)(n_tiles < og ? n_tiles : og);
  const int64_t slab_stride = n_nodes * F;
  dim3 egrid((unsigned)split, (unsigned)n_mol);

  auto edge = [&](bool fwd, int l) -> int {
    const BlockOff o = block_off(PL, l);
    EdgeArgs ea;
    ea.poi = ws.poi; ea.poj = ws.poj; ea.pd = ws.pd; ea.pc = ws.pc; ea.pdc = ws.pdc; ea.pgd = ws.pgd; ea.npairs = ws.npairs;
    ea.mol_ptr = mol_ptr; ea.x1 = ws.x1[l]; ea.gagg = ws.gagg;
    ea.p0 = pk + o.p0; ea.b0 = pk + o.b0; ea.p1 = pk + o.p1; ea.p1t = pk + o.p1t;
    ea.q0h = pk + o.q0h; ea.q0l = pk + o.q0l; ea.q1h = pk + o.q1h; ea.q1l = pk + o.q1l;
    ea.q1th = pk + o.q1th; ea.q1tl = pk + o.q1tl;
    ea.q0ch = pk + o.q0ch; ea.q0cl = pk + o.q0cl; ea.c0 = 0.5f * desc->cutoff; ea.nb = (desc->num_rbf + 7) / 8;
    ea.centers = pk + PL.centers; ea.slabs = ws.slabs; ea.slab_stride = slab_stride;
    ea.gamma = desc->gamma; ea.rc = desc->cutoff; ea.cap = (int)cap; ea.n_split = split;
    ea.pgd_stride = n_mol * cap; ea.fs = fs; ea.max_beads = (int)max_beads_per_mol;
    ea.stamps = reinterpret_cast<unsigned long long*>(g_stamp_buffer);
    ea.stamps2 = reinterpret_cast<unsigned long long*>(g_stamp_buffer2);
    if (ks0 <= 16) return run_edges<16>(desc, fwd, l, L, ea, egrid, lds_edge, s);
    if (ks0 <= 25) return run_edges<25>(desc, fwd, l, L, ea, egrid, lds_edge, s);
    return run_edges<32>(desc, fwd, l, L, ea, egrid, lds_edge, s);
  };

  // ---------------- forward
  for (int l = 0; l < L; ++l) {
    const BlockOff o = block_off(PL, l);
    NodeFwdArgs na;
    na.x = ws.x; na.slabs = ws.slabs;
    na.slab_stride = slab_stride; na.n_slabs = split; na.l1 = pk + o.l1; na.x1_out = ws.x1[l];
    na.n_nodes = n_nodes;
    na.l2 = na.b2 = na.l3 = na.b3 = nullptr; na.t_out = nullptr;
    ev_begin();
    if (l == 0) {
      const int64_t want = (n_nodes * 32 + 255) / 256;
      node_embed_kernel<<<(unsigned)(want < 4096 ? want : 4096), 256, 0, s>>>(atom_types, pk + PL.emb, pk + PL.emb_x1,
                                                                          ws.x, ws.x1[0], n_nodes);
    } else {
      const BlockOff q = block_off(PL, l - 1);
      na.l2 = pk + q.l2; na.b2 = pk + q.b2; na.l3 = pk + q.l3; na.b3 = pk + q.b3; na.t_out = ws.t[l - 1];
      if (node_res) node_fwd_res_kernel<<<res_grid, 256, 0, s>>>(na);
      else if (sp) node_fwd_kernel<true><<<node_grid, 256, 0, s>>>(na);
      else node_fwd_kernel<false><<<node_grid, 256, 0, s>>>(na);
    }
    ev_end();
    FMD_CHECK_LAUNCH();
    ev_begin();
    int rc = edge(true, l);
    ev_end();
    if (rc != FMD_OK) return rc;
  }
  {
    const BlockOff q = block_off(PL, L - 1);
    NodeOutArgs oa;
    oa.x = ws.x; oa.slabs = ws.slabs; oa.slab_stride = slab_stride; oa.n_slabs = split;
    oa.l2 = pk + q.l2; oa.b2 = pk + q.b2; oa.l3 = pk + q.l3; oa.b3 = pk + q.b3; oa.t_out = ws.t[L - 1];
    for (int k = 0; k < 3; ++k) {
      const int kk = k < desc->num_out ? k : desc->num_out - 1;
      const float* base = pk + PL.outs + kk * PL.out_stride;
      oa.po[k] = base; oa.bo[k] = base + SLICE; oa.pot[k] = base + SLICE + F;
    }
    oa.e_node = ws.e_node; oa.gx = ws.gx; oa.n_nodes = n_nodes;
    ev_begin();
    if (desc->num_out == 3) {
      if (node_res && desc->precision == FMD_PREC_W16A16) node_out_res_kernel<3, true><<<out_grid, 256, 0, s>>>(oa);
      else if (node_res) node_out_res_kernel<3, false><<<out_grid, 256, 0, s>>>(oa);
      else if (desc->precision == FMD_PREC_W16A16) node_out_kernel<3, true, true><<<node_grid, 256, 0, s>>>(oa);
      else if (sp) node_out_kernel<3, true, false><<<node_grid, 256, 0, s>>>(oa);
      else node_out_kernel<3, false, false><<<node_grid, 256, 0, s>>>(oa);
    } else {
      if (node_res && desc->precision == FMD_PREC_W16A16) node_out_res_kernel<2, true><<<out_grid, 256, 0, s>>>(oa);
      else if (node_res) node_out_res_kernel<2, false><<<out_grid, 256, 0, s>>>(oa);
      else if (desc->precision == FMD_PREC_W16A16) node_out_kernel<2, true, true><<<node_grid, 256, 0, s>>>(oa);
      else if (sp) node_out_kernel<2, true, false><<<node_grid, 256, 0, s>>>(oa);
      else node_out_kernel<2, false, false><<<node_grid, 256, 0, s>>>(oa);
    }
    ev_end();
    FMD_CHECK_LAUNCH();
  }
  // ---------------- backward
  for (int l = L - 1; l >= 0; --l) {
    const BlockOff o = block_off(PL, l);
    NodeBwdArgs ba;
    ba.gx = ws.gx; ba.slabs = ws.slabs; ba.slab_stride = slab_stride; ba.n_slabs = split;
    ba.t = ws.t[l]; ba.l3t = pk + o.l3t; ba.l2t = pk + o.l2t; ba.gagg = ws.gagg; ba.n_nodes = n_nodes;
    ba.l1t = nullptr;
    ev_begin();
    if (l == L - 1) {
      if (node_res) node_bwd_res_kernel<false><<<res_grid, 256, 0, s>>>(ba);
      else if (sp) node_bwd_kernel<false, true><<<node_grid, 256, 0, s>>>(ba);
      else node_bwd_kernel<false, false><<<node_grid, 256, 0, s>>>(ba);
    } else {
      ba.l1t = pk + block_off(PL, l + 1).l1t;
      if (node_res) node_bwd_res_kernel<true><<<res_grid, 256, 0, s>>>(ba);
      else if (sp) node_bwd_kernel<true, true><<<node_grid, 256, 0, s>>>(ba);
      else node_bwd_kernel<true, false><<<node_grid, 256, 0, s>>>(ba);
    }
    ev_end();
    FMD_CHECK_LAUNCH();
    ev_begin();
    int rc = edge(false, l);
    ev_end();
    if (rc != FMD_OK) return rc;
  }
  ev_begin();
  finish_kernel<<<dim3(2, (unsigned)n_mol), 256, 12 * max_beads_per_mol * sizeof(float), s>>>(
      pos, mol_ptr, ws.poi, ws.poj, ws.pd, ws.pgd, fs ? ws.pgd + n_mol * cap : nullptr, ws.npairs, (int)cap,
      ws.e_node, energy, forces, (int)max_beads_per_mol, ws.report, status);
  ev_end();
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}

extern "C" int fmd_schnet_energy_forces(const fmd_schnet_desc* desc, const void* packed_dev,
                                        const float* pos, const int64_t* atom_types,
                                        const int64_t* mol_ptr, int64_t n_mol, int64_t n_nodes,
                                        int64_t max_beads_per_mol, int64_t max_pairs_per_mol,
                                        float* energy, float* forces, void* workspace,
                                        int32_t* status, void* stream) {
  return run_engine(desc, packed_dev, pos, atom_types, mol_ptr, n_mol, n_nodes, max_beads_per_mol,
                    max_pairs_per_mol, energy, forces, workspace, status, stream, nullptr);
}

// Diagnostic hook: device buffer of [workgroups][4 waves][6 phases] uint64 cycle totals written by
// the stamped build of the forward pair kernel; NULL (default) selects the normal kernels.
extern "C" int fmd_debug_set_stamp_buffer(void* device_buffer) {
  g_stamp_buffer = device_buffer;
  return FMD_OK;
}
// same for the backward split kernel (middle blocks): [workgroups][4 slices][16] uint64
extern "C" int fmd_debug_set_stamp_buffer_bwd(void* device_buffer) {
  g_stamp_buffer2 = device_buffer;
  return FMD_OK;
}

extern "C" int fmd_schnet_kernel_count(const fmd_schnet_desc* desc) {
  return desc_ok(desc) ? 4 * desc->num_blocks + 3 : FMD_EINVAL;
}

extern "C" const char* fmd_schnet_kernel_name(const fmd_schnet_desc* desc, int idx) {
  if (!desc_ok(desc) || idx < 0 || idx >= 4 * desc->num_blocks + 3) return "";
  const int L = desc->num_blocks;
  if (idx == 0) return "pairs_kernel";
  if (idx <= 2 * L) return (idx & 1) ? "node_fwd_kernel" : "edge_fwd_kernel";
  if (idx == 2 * L + 1) return "node_out_kernel";
  if (idx <= 4 * L + 1) return (idx & 1) ? "edge_bwd_kernel" : "node_bwd_kernel";
  return "finish_kernel";
}

extern "C" int fmd_schnet_energy_forces_timed(const fmd_schnet_desc* desc, const void* packed_dev,
                                              const float* pos, const int64_t* atom_types,
                                              const int64_t* mol_ptr, int64_t n_mol, int64_t n_nodes,
                                              int64_t max_beads_per_mol, int64_t max_pairs_per_mol,
                                              float* energy, float* forces, void* workspace,
                                              int32_t* status, void* stream, void** events,
                                              int64_t n_events) {
  if (!desc_ok(desc) || !events || n_events < 2 * (4 * desc->num_blocks + 3)) return FMD_EINVAL;
  return run_engine(desc, packed_dev, pos, atom_types, mol_ptr, n_mol, n_nodes, max_beads_per_mol,
                    max_pairs_per_mol, energy, forces, workspace, status, stream,
                    reinterpret_cast<hipEvent_t*>(events));
}
