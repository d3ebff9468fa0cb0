// A5/A6: CFConv gather-multiply-reduce kernels at operator granularity.
//   csr_cfconv_fwd   <- fused_csr_cfconv / fused_src_csr_grad_x (reference kernels/csr_kernels.py:727-810, 399-482)
//   scatter_atomic   <- fused_cutoff_gather_multiply_scatter     (reference kernels/cfconv_kernels.py:93-170)
//   grad_edge        <- fused_grad_filter_out (:261-337) + the d(cutoff)/d(distance) term the
//                       reference's Triton path drops
//   segment_reduce   <- csr_segment_reduce (kernels/csr_kernels.py:559-617)
// All are HBM-bound on the [E,F] filter tensor.  wave64 mapping: a half-wave (32 lanes x float4)
// covers 128 features of one edge row with 16-byte loads, so a wave streams two edges per step and
// the two halves are combined with one DPP/permute add at the end of a segment.
#include "fmd_common.h"

namespace {

template <bool F16>
__device__ __forceinline__ f32x4 load_filt4(const void* filt, int64_t e, int n_feat, int f) {
  if (F16) {
    const f16x4 h = *reinterpret_cast<const f16x4*>(reinterpret_cast<const _Float16*>(filt) + e * n_feat + f);
    return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  } else {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(filt) + e * n_feat + f);
  }
}

// ---- segment reduce over a CSR: one wave per node, features in chunks of 128 -----------------
// MODE 0: out[n] = sum x[gather[perm[k]]] * filt[perm[k]] * C(d[perm[k]])
// MODE 1: out[n] = sum msg[perm[k]]                                  (csr_segment_reduce)
template <bool F16, int MODE>
__global__ __launch_bounds__(256) void csr_reduce_kernel(
    const float* __restrict__ x, const void* __restrict__ filt, const float* __restrict__ dist,
    const int64_t* __restrict__ gather, const int64_t* __restrict__ ptr,
    const int64_t* __restrict__ perm, int64_t n_nodes, int n_feat, float rc, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
  const int64_t node = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (node >= n_nodes) return;
  const int64_t a = ptr[node], b = ptr[node + 1];
  for (int f0 = 0; f0 < n_feat; f0 += 128) {
    const int f = f0 + 4 * l32;
    const bool fok = f < n_feat;  // n_feat % 4 == 0 is checked on the host
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // two edges per step (one per half-wave), 2-deep unroll to keep 4 row loads in flight
    int64_t k = a + half;
    for (; k + 2 < b; k += 4) {
      const int64_t e0 = perm[k], e1 = perm[k + 2];
      if (MODE == 0) {
        const int64_t s0 = gather[e0], s1 = gather[e1];
        const float c0 = fmd_cutoff(dist[e0], rc), c1 = fmd_cutoff(dist[e1], rc);
        if (fok) {
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + s0 * n_feat + f);
          const f32x4 x1 = *reinterpret_cast<const f32x4*>(x + s1 * n_feat + f);
          const f32x4 w0 = load_filt4<F16>(filt, e0, n_feat, f);
          const f32x4 w1 = load_filt4<F16>(filt, e1, n_feat, f);
          acc += x0 * (w0 * c0);
          acc += x1 * (w1 * c1);
        }
      } else if (fok) {
        acc += *reinterpret_cast<const f32x4*>(x + e0 * n_feat + f);
        acc += *reinterpret_cast<const f32x4*>(x + e1 * n_feat + f);
      }
    }
    for (; k < b; k += 2) {
      const int64_t e0 = perm[k];
      if (MODE == 0) {
        const int64_t s0 = gather[e0];
        const float c0 = fmd_cutoff(dist[e0], rc);
        if (fok) {
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(x + s0 * n_feat + f);
          acc += x0 * (load_filt4<F16>(filt, e0, n_feat, f) * c0);
        }
      } else if (fok) {
        acc += *reinterpret_cast<const f32x4*>(x + e0 * n_feat + f);
      }
    }
    // combine the two half-waves (lane l and l+32 hold the same features)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += __shfl_xor(acc[i], 32, 64);
    if (half == 0 && fok) *reinterpret_cast<f32x4*>(out + node * n_feat + f) = acc;
  }
}

// ---- generic-width fallback (n_feat not a multiple of 4): one wave per node, scalar lanes ------
template <bool F16, int MODE>
__global__ __launch_bounds__(256) void csr_reduce_scalar_kernel(
    const float* __restrict__ x, const void* __restrict__ filt, const float* __restrict__ dist,
    const int64_t* __restrict__ gather, const int64_t* __restrict__ ptr,
    const int64_t* __restrict__ perm, int64_t n_nodes, int n_feat, float rc, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t node = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (node >= n_nodes) return;
  const int64_t a = ptr[node], b = ptr[node + 1];
  for (int f = lane; f < n_feat; f += 64) {
    float acc = 0.f;
    for (int64_t k = a; k < b; ++k) {
      const int64_t e = perm[k];
      if (MODE == 0) {
        const float w = F16 ? (float)reinterpret_cast<const _Float16*>(filt)[e * n_feat + f]
                            : reinterpret_cast<const float*>(filt)[e * n_feat + f];
        acc += x[gather[e] * n_feat + f] * (w * fmd_cutoff(dist[e], rc));
      } else {
        acc += x[e * n_feat + f];
      }
    }
    out[node * n_feat + f] = acc;
  }
}

// ---- edge-parallel kernels: half-wave per edge, feature chunks of 128 ---------------------------
template <bool F16>
__global__ __launch_bounds__(256) void scatter_atomic_kernel(
    const float* __restrict__ x, const void* __restrict__ filt, const float* __restrict__ dist,
    const int64_t* __restrict__ esrc, const int64_t* __restrict__ edst, int64_t n_edges, int n_feat,
    float rc, float* __restrict__ out) {
  const int l32 = threadIdx.x & 31;
  const int64_t e = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (e >= n_edges) return;
  const int64_t s = esrc[e], t = edst[e];
  const float c = fmd_cutoff(dist[e], rc);
  for (int f = l32; f < n_feat; f += 32) {
    const float w = F16 ? (float)reinterpret_cast<const _Float16*>(filt)[e * n_feat + f]
                        : reinterpret_cast<const float*>(filt)[e * n_feat + f];
    atomicAdd(&out[t * n_feat + f], x[s * n_feat + f] * (w * c));
  }
}

// gfilt[e,:] = x[src] * g[dst] * C ; gd[e] = C'(d) * sum_f x[src] filt[e] g[dst]
template <bool F16>
__global__ __launch_bounds__(256) void grad_edge_kernel(
    const float* __restrict__ x, const float* __restrict__ g, const void* __restrict__ filt,
    const float* __restrict__ dist, const int64_t* __restrict__ esrc,
    const int64_t* __restrict__ edst, int64_t n_edges, int n_feat, float rc,
    void* __restrict__ gfilt, float* __restrict__ gdist) {
  const int l32 = threadIdx.x & 31;
  const int64_t e = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  const bool ok = e < n_edges;
  float dot = 0.f, c = 0.f, dc = 0.f;
  if (ok) {
    const int64_t s = esrc[e], t = edst[e];
    fmd_cutoff2(dist[e], rc, c, dc);
    if ((n_feat & 3) == 0) {
      for (int f = 4 * l32; f < n_feat; f += 128) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + s * n_feat + f);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + t * n_feat + f);
        const f32x4 xg = xv * gv;
        if (gfilt) {
          const f32x4 o = xg * c;
          if (F16) {
            const f16x4 h = {(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
            *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(gfilt) + e * n_feat + f) = h;
          } else {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(gfilt) + e * n_feat + f) = o;
          }
        }
        if (gdist) {
          const f32x4 w = load_filt4<F16>(filt, e, n_feat, f);
          dot += xg[0] * w[0] + xg[1] * w[1] + xg[2] * w[2] + xg[3] * w[3];
        }
      }
    } else {
      for (int f = l32; f < n_feat; f += 32) {
        const float xg = x[s * n_feat + f] * g[t * n_feat + f];
        if (gfilt) {
          if (F16) reinterpret_cast<_Float16*>(gfilt)[e * n_feat + f] = (_Float16)(xg * c);
          else reinterpret_cast<float*>(gfilt)[e * n_feat + f] = xg * c;
        }
        if (gdist) {
          const float w = F16 ? (float)reinterpret_cast<const _Float16*>(filt)[e * n_feat + f]
                              : reinterpret_cast<const float*>(filt)[e * n_feat + f];
          dot += xg * w;
        }
      }
    }
  }
  if (gdist) {  // reduce inside the half-wave (all 64 lanes take part in the shuffles)
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    if (ok && l32 == 0) gdist[e] = dot * dc;
  }
}

}  // namespace

#define FMD_DISPATCH_F16(flag, ...)            \
  do {                                         \
    if (flag) { constexpr bool F16 = true; __VA_ARGS__; } \
    else { constexpr bool F16 = false; __VA_ARGS__; }      \
  } while (0)

extern "C" int fmd_csr_cfconv_fwd(const float* x, const void* filt, int filt_is_f16, const float* dist,
                                  const int64_t* gather, const int64_t* ptr, const int64_t* perm,
                                  int64_t n_nodes, int64_t n_feat, float cutoff, float* out,
                                  void* stream) {
  if (n_nodes < 0 || n_feat <= 0 || n_feat > (1 << 20)) return FMD_EINVAL;
  if (n_nodes == 0) return FMD_OK;
  if (!x || !filt || !dist || !gather || !ptr || !perm || !out) return FMD_EINVAL;
  const unsigned nb = (unsigned)((n_nodes + 3) / 4);
  hipStream_t s = fmd_stream(stream);
  const bool vec = (n_feat % 4) == 0;
  FMD_DISPATCH_F16(filt_is_f16, {
    if (vec)
      csr_reduce_kernel<F16, 0><<<nb, 256, 0, s>>>(x, filt, dist, gather, ptr, perm, n_nodes,
                                                 (int)n_feat, cutoff, out);
    else
      csr_reduce_scalar_kernel<F16, 0><<<nb, 256, 0, s>>>(x, filt, dist, gather, ptr, perm, n_nodes,
                                                        (int)n_feat, cutoff, out);
  });
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}

extern "C" int fmd_csr_segment_reduce(const float* msg, const int64_t* ptr, const int64_t* perm,
                                      int64_t n_nodes, int64_t n_feat, float* out, void* stream) {
  if (n_nodes < 0 || n_feat <= 0 || n_feat > (1 << 20)) return FMD_EINVAL;
  if (n_nodes == 0) return FMD_OK;
  if (!msg || !ptr || !perm || !out) return FMD_EINVAL;
  const unsigned nb = (unsigned)((n_nodes + 3) / 4);
  hipStream_t s = fmd_stream(stream);
  if (n_feat % 4 == 0)
    csr_reduce_kernel<false, 1><<<nb, 256, 0, s>>>(msg, nullptr, nullptr, nullptr, ptr, perm, n_nodes,
                                                 (int)n_feat, 0.f, out);
  else
    csr_reduce_scalar_kernel<false, 1><<<nb, 256, 0, s>>>(msg, nullptr, nullptr, nullptr, ptr, perm,
                                                        n_nodes, (int)n_feat, 0.f, out);
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}

extern "C" int fmd_cfconv_scatter_atomic(const float* x, const void* filt, int filt_is_f16,
                                         const float* dist, const int64_t* edge_src,
                                         const int64_t* edge_dst, int64_t n_edges, int64_t n_nodes,
                                         int64_t n_feat, float cutoff, float* out, void* stream) {
  if (n_edges < 0 || n_nodes < 0 || n_feat <= 0 || !out) return FMD_EINVAL;
  hipStream_t s = fmd_stream(stream);
  if (hipMemsetAsync(out, 0, sizeof(float) * n_nodes * n_feat, s) != hipSuccess) return FMD_ELAUNCH;
  if (n_edges == 0) return FMD_OK;
  if (!x || !filt || !dist || !edge_src || !edge_dst) return FMD_EINVAL;
  const unsigned nb = (unsigned)((n_edges + 7) / 8);
  FMD_DISPATCH_F16(filt_is_f16, {
    scatter_atomic_kernel<F16><<<nb, 256, 0, s>>>(x, filt, dist, edge_src, edge_dst, n_edges,
                                                (int)n_feat, cutoff, out);
  });
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}

extern "C" int fmd_cfconv_grad_edge(const float* x, const float* grad_out, const void* filt,
                                    int filt_is_f16, const float* dist, const int64_t* edge_src,
                                    const int64_t* edge_dst, int64_t n_edges, int64_t n_feat,
                                    float cutoff, void* grad_filt, float* grad_dist, void* stream) {
  if (n_edges < 0 || n_feat <= 0) return FMD_EINVAL;
  if (n_edges == 0) return FMD_OK;
  if (!x || !grad_out || !dist || !edge_src || !edge_dst) return FMD_EINVAL;
  if (grad_dist && !filt) return FMD_EINVAL;
  if (!grad_filt && !grad_dist) return FMD_OK;
  const unsigned nb = (unsigned)((n_edges + 7) / 8);
  hipStream_t s = fmd_stream(stream);
  FMD_DISPATCH_F16(filt_is_f16, {
    grad_edge_kernel<F16><<<nb, 256, 0, s>>>(x, grad_out, filt, dist, edge_src, edge_dst, n_edges,
                                           (int)n_feat, cutoff, grad_filt, grad_dist);
  });
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}
