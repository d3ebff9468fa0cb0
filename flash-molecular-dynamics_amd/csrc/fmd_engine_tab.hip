// Fused engine, TABULATED-FILTER pair kernels (opt-in: FMD_FLAG_TABULATED_FILTER; never the default path).
//
// At inference the weights are frozen (reference simulation/base.py:357-358) and the message weight of a block,
//   W_l(d) = filt_l(rbf(d)) * C(d)   (models/schnet.py:710-713 with the filter MLP of models/mlp.py:41-48),
// is a function of the scalar pair distance alone.  fmd_schnet_pack tabulates it per block on a uniform grid of
// spacing h = 2^e (fp64 evaluation of the network and of its analytic derivative, cubic Hermite data turned into one
// cubic per interval), and these kernels evaluate   W(d) = a0 + t (a1 + t (a2 + t a3)),  t = d / h - k,  from the
// 16-byte record of (interval k, feature) instead of running filter layers 0 / 1 on the MFMA: no rbf, no tanh, no
// operand splits, no GEMM.  Interpolation error ~ (h / sigma)^4 / 384 ~ 1e-8 relative for the 1ENH basis (sigma = Gaussian
// width 0.1, h = 2^-8), below fp32 rounding; the backward form uses the interpolant's own derivative, so forces stay the
// exact gradient of the tabulated energy.
//
// Layout of the work (no matrix core involved -- the bound is the vector-memory path that fetches the records, 16 bytes
// per pair and feature from L2, and the SIMDs' issue rate):
//   * one workgroup = one molecule x 64 features: x1[:, 64 s .. 64 s + 63] AND the aggregate slice live in LDS
//     (2 x 269 x 72 floats = 155 KB), so the only global traffic of the tile loop is the pair list and the records;
//   * 8 waves, wave w owns 8 feature columns; lane = 8 e + f walks the list one 8-entry SEGMENT per step (entry e,
//     feature f).  A segment's neighbour rows are distinct and differ from its centre rows (pairs_kernel guarantees it),
//     so the 64 read-modify-writes of a step are conflict free; waves never touch each other's columns: the tile loop
//     has NO barrier;
//   * the centre-row sum of a 4-entry half segment is a cross-lane reduction (lanes 8 apart).
#include "fmd_engine.h"

namespace {

constexpr int TAB_FW = 64;    // features per workgroup
constexpr int TAB_LDW = 72;   // floats per LDS row: 64 + 8, so that rows r, r+1, r+2, r+3 start 8 banks apart

struct TabSeg {   // one 8-entry segment as this lane sees it
  uint32_t oi, oj;  // row byte offsets (row * 512) of the entry's centre and neighbour
  float d;
};

__device__ __forceinline__ float tab_row_ror8_add(float v) {  // v + v(lane ^ 8): DPP row_ror:8 inside a 16-lane row
  const int u = __builtin_bit_cast(int, v);
  const int r = __builtin_amdgcn_update_dpp(0, u, 0x128, 0xf, 0xf, false);
  return v + __builtin_bit_cast(float, r);
}

__global__ __launch_bounds__(512) void tab_fwd_kernel(EdgeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  int m, s;
  wg_to_mol_split(m, s);   // grid (2, n_mol): s = feature half; both halves of a molecule on one XCD (same records' rows)
  const int64_t a0 = a.mol_ptr[m];
  const int n = min((int)(a.mol_ptr[m + 1] - a0), a.max_beads);
  const int np = a.npairs[m];
  const int ntiles = (np + 31) >> 5;
  float* x1s = smem;                                  // [max_beads][TAB_LDW]
  float* aggs = smem + (int64_t)a.max_beads * TAB_LDW;
  for (int i = tid; i < n * 16; i += 512) {
    const int row = i >> 4, c4 = (i & 15) * 4;
    *reinterpret_cast<f32x4*>(x1s + row * TAB_LDW + c4) =
        *reinterpret_cast<const f32x4*>(a.x1 + (a0 + row) * F + TAB_FW * s + c4);
    *reinterpret_cast<f32x4*>(aggs + row * TAB_LDW + c4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  const int e = lane >> 3, f = lane & 7, col = 8 * w + f;
  const float* tabc = a.tab + (TAB_FW * s + col) * 4;   // this lane's feature column: record k at tabc + k * 512
  const uint32_t* poi = a.poi + (int64_t)m * a.cap;
  const uint32_t* poj = a.poj + (int64_t)m * a.cap;
  const float* pd = a.pd + (int64_t)m * a.cap;
  const int last = max(ntiles - 1, 0);
  const float inv_h = a.tab_inv_h;
  const int kmax = a.tab_k - 1;
  // LDS byte offset of (row, col) from a list offset o = row * 512:  row * 288 + col * 4 = (o >> 4) * 9 + col * 4
  const uint32_t colb = (uint32_t)col * 4u;
  typedef __attribute__((address_space(3))) float* lds_fp;
  const uint32_t x1_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(x1s);
  const uint32_t agg_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(aggs);
  auto lds_rd = [&](uint32_t addr) -> float { return *(lds_fp)(uintptr_t)addr; };
  auto lds_wr = [&](uint32_t addr, float v) { *(lds_fp)(uintptr_t)addr = v; };

  // two-stage software pipeline: list entries two tiles ahead, table records one tile ahead
  TabSeg L0[4], L1[4], L2[4];
  f32x4 R0[4], R1[4];
  auto load_list = [&](TabSeg* L, int t) {
    const int base = min(t, last) * 32 + e;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      L[q].oi = poi[base + 8 * q];
      L[q].oj = poj[base + 8 * q];
      L[q].d = pd[base + 8 * q];
    }
  };
  auto load_recs = [&](f32x4* R, const TabSeg* L) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = min((int)(L[q].d * inv_h), kmax);
      R[q] = *reinterpret_cast<const f32x4*>(tabc + (int64_t)k * (F * 4));
    }
  };
  if (ntiles > 0) {
    load_list(L0, 0);
    load_list(L1, 1);
    load_recs(R0, L0);
    for (int t = 0; t < ntiles; ++t) {
      load_list(L2, t + 2);
      load_recs(R1, L1);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float dh = L0[q].d * inv_h;
        const float tt = dh - (float)min((int)dh, kmax);
        float W = R0[q][0] + tt * (R0[q][1] + tt * (R0[q][2] + tt * R0[q][3]));
        W = L0[q].oi == L0[q].oj ? 0.f : W;    // pad entries (i, i) carry nothing
        const uint32_t rj = (L0[q].oj >> 4) * 9u + colb, ri = (L0[q].oi >> 4) * 9u + colb;
        const float xj = lds_rd(x1_lds + rj), xi = lds_rd(x1_lds + ri), vj = lds_rd(agg_lds + rj);
        lds_wr(agg_lds + rj, vj + xi * W);
        // centre row: sum over the 4 entries of this half segment (lanes 8 and 16 apart), plus the other half when
        // both halves belong to one centre (fmd_segment_sum's rule)
        float si = xj * W;
        si = tab_row_ror8_add(si);
        si += __shfl_xor(si, 16, 64);
        si = fmd_segment_sum(si, L0[q].oi);
        if ((e & 3) == 0) {   // (both halves store the same value to the same address when they share the centre)
          const float cv = lds_rd(agg_lds + ri);
          lds_wr(agg_lds + ri, cv + si);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) { L0[q] = L1[q]; L1[q] = L2[q]; R0[q] = R1[q]; }
    }
  }
  __syncthreads();
  float* out = a.slabs + (int64_t)s * a.slab_stride + a0 * F;   // slab s holds this half's 64 features, zeros elsewhere
  for (int i = tid; i < n * 16; i += 512) {
    const int row = i >> 4, c4 = (i & 15) * 4;
    *reinterpret_cast<f32x4*>(out + row * F + TAB_FW * s + c4) = *reinterpret_cast<const f32x4*>(aggs + row * TAB_LDW + c4);
    *reinterpret_cast<f32x4*>(out + row * F + TAB_FW * (1 - s) + c4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

}  // namespace

namespace fmd_eng {

int setup_tab_attrs() {
  return hipFuncSetAttribute((const void*)tab_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)MAX_LDS_BYTES) == hipSuccess ? FMD_OK : FMD_ELAUNCH;
}

int launch_tab_fwd(const EdgeArgs& a, int64_t n_mol, hipStream_t s) {
  const size_t lds = (size_t)2 * a.max_beads * TAB_LDW * sizeof(float);
  if (lds > MAX_LDS_BYTES || !a.tab) return FMD_EUNSUPPORTED;
  tab_fwd_kernel<<<dim3(2, (unsigned)n_mol), 512, lds, s>>>(a);
  return hipGetLastError() == hipSuccess ? FMD_OK : FMD_ELAUNCH;
}

}  // namespace fmd_eng
