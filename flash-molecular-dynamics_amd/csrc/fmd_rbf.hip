// A2: fused distance + Gaussian RBF + cosine cutoff, forward and backward.
// Replaces the reference's fused_distance_gaussian_rbf_cutoff (kernels/cfconv_kernels.py:1578-1656)
// and the backward formula at :1696-1733.  HBM-bound: the [E,R] fp32 expansion is the traffic;
// a wave owns 64 consecutive edges and writes/reads their 64*R values as one contiguous stream.
#include "fmd_common.h"

namespace {

constexpr int RBF_WAVES = 4;
constexpr int RBF_KC = 61;  // grad_rbf columns staged per pass in the backward kernel (62.5 KB of LDS)

__global__ __launch_bounds__(RBF_WAVES * 64) void rbf_fwd_kernel(
    const float* __restrict__ pos, const int64_t* __restrict__ esrc, const int64_t* __restrict__ edst,
    const float* __restrict__ centers, float gamma, float rc, int64_t n_edges, int n_rbf,
    float* __restrict__ dist, float* __restrict__ rbf) {
  __shared__ float sd[RBF_WAVES][64], sc[RBF_WAVES][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t e0 = ((int64_t)blockIdx.x * RBF_WAVES + wave) * 64;
  if (e0 >= n_edges) return;
  const int64_t e = e0 + lane;
  float d = 0.f, c = 0.f;
  if (e < n_edges) {
    const int64_t s = esrc[e], t = edst[e];
    const float dx = pos[3 * t] - pos[3 * s], dy = pos[3 * t + 1] - pos[3 * s + 1],
                dz = pos[3 * t + 2] - pos[3 * s + 2];
    d = sqrtf(dx * dx + dy * dy + dz * dz);
    c = fmd_cutoff(d, rc);
    dist[e] = d;
  }
  sd[wave][lane] = d;
  sc[wave][lane] = c;
  // same-wave LDS write -> read: program order + waitcnt is enough, no barrier needed
  const int ne = (int)min((int64_t)64, n_edges - e0);
  const int total = ne * n_rbf;
  float* out = rbf + e0 * n_rbf;
  for (int t = lane; t < total; t += 64) {
    const int el = t / n_rbf, k = t - el * n_rbf;
    const float diff = sd[wave][el] - centers[k];
    out[t] = __expf(gamma * diff * diff) * sc[wave][el];
  }
}

// grad wrt distance, then scattered onto the two end points with float atomics (the reference
// uses index_add_, equally unordered).
// SCATTER = true: float atomics onto gpos (any edge list).  SCATTER = false: the per-edge vector
// gd (r_dst - r_src) / d is stored to edge_vec[E][4] and rbf_gather_kernel sums it per node over the caller's CSR
// indices -- no atomics, fixed summation order.
template <bool SCATTER>
__global__ __launch_bounds__(RBF_WAVES * 64) void rbf_bwd_kernel(
    const float* __restrict__ pos, const int64_t* __restrict__ esrc, const int64_t* __restrict__ edst,
    const float* __restrict__ centers, float gamma, float rc, int64_t n_edges, int n_rbf,
    const float* __restrict__ dist, const float* __restrict__ gdist, const float* __restrict__ grbf,
    float* __restrict__ gpos, f32x4* __restrict__ edge_vec) {
  // grad_rbf is [E, R] row-major: the wave's 64 x R block is read coalesced and staged in LDS ([64][R + 1]: odd
  // stride, conflict-free), then lane e sums its own row against d(rbf)/dd.  (The first version added every
  // element with an LDS float atomic: ~3 cycles per lane on this chip, 10x the cost of the staging.)
  extern __shared__ float stage[];  // [RBF_WAVES][64 * (R | 1 + ...)]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t e0 = ((int64_t)blockIdx.x * RBF_WAVES + wave) * 64;
  if (e0 >= n_edges) return;
  const int64_t e = e0 + lane;
  float d = 0.f, c = 0.f, dc = 0.f;
  if (e < n_edges) {
    d = dist[e];
    fmd_cutoff2(d, rc, c, dc);
  }
  float acc = 0.f;
  if (grbf) {
    const int kc = min(n_rbf, RBF_KC), ld = kc | 1;  // columns per pass (one pass for R <= 61)
    float* st = stage + (size_t)wave * 64 * ld;
    const int ne = (int)min((int64_t)64, n_edges - e0);
    const float* g = grbf + e0 * n_rbf;
    const float g2c = 2.0f * gamma * c;
    for (int k0 = 0; k0 < n_rbf; k0 += kc) {
      const int kn = min(kc, n_rbf - k0);
      for (int t = lane; t < ne * kn; t += 64) {
        const int el = t / kn, k = t - el * kn;
        st[el * ld + k] = g[(int64_t)el * n_rbf + k0 + k];
      }
      // written and read by the same wave: its LDS operations complete in order
      if (lane < ne) {
        for (int k = 0; k < kn; ++k) {
          const float diff = d - centers[k0 + k];
          const float ex = __expf(gamma * diff * diff);
          // d rbf / d d = 2 gamma diff ex C + ex C'            (cfconv_kernels.py:1716-1719)
          acc += st[lane * ld + k] * (ex * (g2c * diff + dc));
        }
      }
    }
  }
  if (e < n_edges) {
    float gd = acc;
    if (gdist) gd += gdist[e];
    const int64_t s = esrc[e], t = edst[e];
    const float dx = pos[3 * t] - pos[3 * s], dy = pos[3 * t + 1] - pos[3 * s + 1],
                dz = pos[3 * t + 2] - pos[3 * s + 2];
    const float inv = gd / fmaxf(d, 1e-8f);  // clamp as cfconv_kernels.py:1697
    const float gx = dx * inv, gy = dy * inv, gz = dz * inv;
    if (!SCATTER) {
      edge_vec[e] = f32x4{gx, gy, gz, 0.f};
      return;
    }
    atomicAdd(&gpos[3 * t], gx);
    atomicAdd(&gpos[3 * t + 1], gy);
    atomicAdd(&gpos[3 * t + 2], gz);
    atomicAdd(&gpos[3 * s], -gx);
    atomicAdd(&gpos[3 * s + 1], -gy);
    atomicAdd(&gpos[3 * s + 2], -gz);
  }
}

// gpos[n] = sum_{e: dst(e) = n} v_e - sum_{e: src(e) = n} v_e over the two CSR indices (perm[k] = edge id).  One wave
// per node: lanes stride over the segment, 16-byte gathers, one xor-tree reduction per component.
__global__ __launch_bounds__(256) void rbf_gather_kernel(const f32x4* __restrict__ edge_vec,
                                                         const int64_t* __restrict__ dst_ptr,
                                                         const int64_t* __restrict__ dst_perm,
                                                         const int64_t* __restrict__ src_ptr,
                                                         const int64_t* __restrict__ src_perm, int64_t n_nodes,
                                                         float* __restrict__ gpos) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= n_nodes) return;
  float ax = 0.f, ay = 0.f, az = 0.f;
  for (int64_t k = dst_ptr[n] + lane; k < dst_ptr[n + 1]; k += 64) {
    const f32x4 v = edge_vec[dst_perm[k]];
    ax += v[0]; ay += v[1]; az += v[2];
  }
  for (int64_t k = src_ptr[n] + lane; k < src_ptr[n + 1]; k += 64) {
    const f32x4 v = edge_vec[src_perm[k]];
    ax -= v[0]; ay -= v[1]; az -= v[2];
  }
  ax = fmd_wave_sum(ax); ay = fmd_wave_sum(ay); az = fmd_wave_sum(az);
  if (lane == 0) { gpos[3 * n] = ax; gpos[3 * n + 1] = ay; gpos[3 * n + 2] = az; }
}

}  // namespace

extern "C" int fmd_distance_rbf_cutoff_fwd(const float* pos, const int64_t* edge_src,
                                           const int64_t* edge_dst, const float* centers, float gamma,
                                           float cutoff, int64_t n_edges, int64_t n_rbf, float* dist,
                                           float* rbf, void* stream) {
  if (n_edges < 0 || n_rbf <= 0 || n_rbf > 65536) return FMD_EINVAL;
  if (n_edges == 0) return FMD_OK;
  if (!pos || !edge_src || !edge_dst || !centers || !dist || !rbf) return FMD_EINVAL;
  const int64_t nblk = (n_edges + RBF_WAVES * 64 - 1) / (RBF_WAVES * 64);
  rbf_fwd_kernel<<<(unsigned)nblk, RBF_WAVES * 64, 0, fmd_stream(stream)>>>(
      pos, edge_src, edge_dst, centers, gamma, cutoff, n_edges, (int)n_rbf, dist, rbf);
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}

extern "C" int fmd_distance_rbf_cutoff_bwd(const float* pos, const int64_t* edge_src,
                                           const int64_t* edge_dst, const float* centers, float gamma,
                                           float cutoff, int64_t n_edges, int64_t n_rbf,
                                           int64_t n_nodes, const float* dist, const float* grad_dist,
                                           const float* grad_rbf, float* grad_pos, void* stream) {
  if (n_edges < 0 || n_rbf <= 0 || n_nodes < 0 || !grad_pos) return FMD_EINVAL;
  hipStream_t s = fmd_stream(stream);
  if (hipMemsetAsync(grad_pos, 0, sizeof(float) * 3 * n_nodes, s) != hipSuccess) return FMD_ELAUNCH;
  if (n_edges == 0) return FMD_OK;
  if (!pos || !edge_src || !edge_dst || !centers || !dist) return FMD_EINVAL;
  const int64_t nblk = (n_edges + RBF_WAVES * 64 - 1) / (RBF_WAVES * 64);
  const int kc = (int)(n_rbf < RBF_KC ? n_rbf : RBF_KC);
  const size_t lds = grad_rbf ? (size_t)RBF_WAVES * 64 * (kc | 1) * sizeof(float) : 0;
  rbf_bwd_kernel<true><<<(unsigned)nblk, RBF_WAVES * 64, lds, s>>>(pos, edge_src, edge_dst, centers, gamma,
                                                               cutoff, n_edges, (int)n_rbf, dist, grad_dist,
                                                               grad_rbf, grad_pos, nullptr);
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}

extern "C" int fmd_distance_rbf_cutoff_bwd_csr(const float* pos, const int64_t* edge_src, const int64_t* edge_dst,
                                               const float* centers, float gamma, float cutoff, int64_t n_edges,
                                               int64_t n_rbf, int64_t n_nodes, const float* dist,
                                               const float* grad_dist, const float* grad_rbf,
                                               const int64_t* dst_ptr, const int64_t* dst_perm,
                                               const int64_t* src_ptr, const int64_t* src_perm, void* edge_vec,
                                               float* grad_pos, void* stream) {
  if (n_edges < 0 || n_rbf <= 0 || n_nodes < 0 || !grad_pos) return FMD_EINVAL;
  hipStream_t s = fmd_stream(stream);
  if (n_edges == 0) return hipMemsetAsync(grad_pos, 0, sizeof(float) * 3 * n_nodes, s) == hipSuccess ? FMD_OK : FMD_ELAUNCH;
  if (!pos || !edge_src || !edge_dst || !centers || !dist || !dst_ptr || !dst_perm || !src_ptr || !src_perm ||
      !edge_vec || (reinterpret_cast<uintptr_t>(edge_vec) & 15))
    return FMD_EINVAL;
  const int64_t nblk = (n_edges + RBF_WAVES * 64 - 1) / (RBF_WAVES * 64);
  const int kc = (int)(n_rbf < RBF_KC ? n_rbf : RBF_KC);
  const size_t lds = grad_rbf ? (size_t)RBF_WAVES * 64 * (kc | 1) * sizeof(float) : 0;
  rbf_bwd_kernel<false><<<(unsigned)nblk, RBF_WAVES * 64, lds, s>>>(
      pos, edge_src, edge_dst, centers, gamma, cutoff, n_edges, (int)n_rbf, dist, grad_dist, grad_rbf, nullptr,
      reinterpret_cast<f32x4*>(edge_vec));
  FMD_CHECK_LAUNCH();
  if (n_nodes > 0)
    rbf_gather_kernel<<<(unsigned)((n_nodes + 3) / 4), 256, 0, s>>>(reinterpret_cast<const f32x4*>(edge_vec), dst_ptr,
                                                                dst_perm, src_ptr, src_perm, n_nodes, grad_pos);
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}
