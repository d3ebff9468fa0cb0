// Dense contractions of the path at operator granularity.
//   fmd_gemm_f32              exact fp32 on v_mfma_f32_32x32x2_f32   (fused_tanh_linear, its backward,
//                             fused_linear_tanh fp32, lin1/lin2: reference kernels/cfconv_kernels.py:1846-1967,
//                             543-635; models/schnet.py:644,719)
//   fmd_linear_tanh_f16 ...   fp16 storage / fp32 accumulate on v_mfma_f32_32x32x16_f16
//                             (reference kernels/cfconv_kernels.py:657-758, 780-952, 973-1064, 1174-1226)
// wave64 tiling: a wave owns 32 output rows; the MFMA C layout (col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) makes every accumulator register a 128-byte row segment.
#include "fmd_common.h"

namespace {

// =================================================================================================
// fp32 GEMM: tile 64(M) x 128(N), BK = 32, 4 waves as 2(M) x 2(N), each wave 32 x 64.
// =================================================================================================
constexpr int G32_BM = 64, G32_BN = 128, G32_BK = 32;

template <int PRO, int EPI, bool B_NK>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A,
                                                       const float* __restrict__ B,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ aux,
                                                       float* __restrict__ C, int64_t M, int N, int K) {
  __shared__ float As[G32_BM][G32_BK + 1];
  __shared__ float Bs[G32_BK][G32_BN + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.x * G32_BM;
  const int n0 = blockIdx.y * G32_BN;
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  for (int k0 = 0; k0 < K; k0 += G32_BK) {
    __syncthreads();
    // A tile: 64 x 32, coalesced along k
    for (int t = tid; t < G32_BM * G32_BK; t += 256) {
      const int r = t / G32_BK, c = t - r * G32_BK;
      float v = 0.f;
      if (m0 + r < M && k0 + c < K) {
        v = A[(m0 + r) * K + k0 + c];
        if (PRO == FMD_PRO_TANH) v = fmd_tanh(v);
      }
      As[r][c] = v;
    }
    // B tile: 32 x 128
    if (B_NK) {  // B[n][k] in memory: read along k, store transposed
      for (int t = tid; t < G32_BK * G32_BN; t += 256) {
        const int n = t / G32_BK, c = t - n * G32_BK;
        float v = 0.f;
        if (n0 + n < N && k0 + c < K) v = B[(int64_t)(n0 + n) * K + k0 + c];
        Bs[c][n] = v;
      }
    } else {
      for (int t = tid; t < G32_BK * G32_BN; t += 256) {
        const int c = t / G32_BN, n = t - c * G32_BN;
        float v = 0.f;
        if (n0 + n < N && k0 + c < K) v = B[(int64_t)(k0 + c) * N + n0 + n];
        Bs[c][n] = v;
      }
    }
    __syncthreads();
    const int arow = wm * 32 + (lane & 31);
    const int kh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < G32_BK; kk += 2) {
      const float a = As[arow][kk + kh];
      const float b0 = Bs[kk + kh][wn * 64 + (lane & 31)];
      const float b1 = Bs[kk + kh][wn * 64 + 32 + (lane & 31)];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
    }
  }
  // epilogue
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int col = n0 + wn * 64 + t * 32 + (lane & 31);
    if (col >= N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row >= M) continue;
      float v = acc[t][r] + bv;
      if (EPI == FMD_EPI_TANH) v = fmd_tanh(v);
      if (EPI == FMD_EPI_MUL_DTANH) {
        const float y = aux[row * N + col];
        v *= (1.0f - y * y);
      }
      C[row * N + col] = v;
    }
  }
}

// =================================================================================================
// fp16 GEMM: tile 128(M) x 128(N), BK = 64 halves, 4 waves each 32 x 128 (4 accumulators).
//   A source variants:  AK_F32 (round to fp16 on load), AK_F16, AK_DTANH (f16(g * (1 - y^2)))
//   epilogue variants:  bias(fp16) + clamped tanh, or none; store fp16 or fp32
// LDS images are [row][k] with a 16-byte pad per row (conflict-free ds_read_b128 fragments).
// =================================================================================================
constexpr int G16_BM = 128, G16_BN = 128, G16_BK = 64, G16_LD = G16_BK + 8;
enum { AK_F32 = 0, AK_F16 = 1, AK_DTANH_F16G = 2, AK_DTANH_F32G = 3 };

template <int AK, bool TANH_EPI, bool OUT_F16>
__global__ __launch_bounds__(256) void gemm_f16_kernel(const void* __restrict__ Ap,
                                                       const void* __restrict__ Yp,
                                                       const _Float16* __restrict__ B,
                                                       const _Float16* __restrict__ bias,
                                                       void* __restrict__ Cp, int64_t M, int N, int K) {
  __shared__ __attribute__((aligned(16))) _Float16 As[G16_BM * G16_LD];
  __shared__ __attribute__((aligned(16))) _Float16 Bs[G16_BN * G16_LD];  // Bs[n][k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * G16_BM;
  const int n0 = blockIdx.y * G16_BN;
  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int64_t rows_here = min((int64_t)G16_BM, M - m0);
  for (int k0 = 0; k0 < K; k0 += G16_BK) {
    const int kc = min(G16_BK, K - k0);
    __syncthreads();
    // ---- stage A: rows_here x kc (zero-filled to 128 x 64)
    for (int t = tid; t < G16_BM * G16_BK; t += 256) {
      const int r = t / G16_BK, c = t - r * G16_BK;
      float v = 0.f;
      if (r < rows_here && c < kc) {
        const int64_t idx = (m0 + r) * K + k0 + c;
        if (AK == AK_F32) v = reinterpret_cast<const float*>(Ap)[idx];
        else if (AK == AK_F16) v = (float)reinterpret_cast<const _Float16*>(Ap)[idx];
        else {
          const float g = (AK == AK_DTANH_F16G) ? (float)reinterpret_cast<const _Float16*>(Ap)[idx]
                                                 : reinterpret_cast<const float*>(Ap)[idx];
          const float y = (float)reinterpret_cast<const _Float16*>(Yp)[idx];
          // the reference forms g * (1 - y*y) with separately rounded fp32 operations before the fp16
          // rounding (cfconv_kernels.py:1019-1023)
          {
#pragma clang fp contract(off)
            const float yy = y * y;
            const float om = 1.0f - yy;
            v = g * om;
            // keep the fp32 rounding of the product: folded into the f16 conversion below (v_fma_mixlo_f16) it would
            // be rounded once from the exact product, one fp16 ulp away from the reference in ~1 of 30 000 elements
            asm volatile("" : "+v"(v));
          }
        }
      }
      As[r * G16_LD + c] = (_Float16)v;
    }
    // ---- stage B transposed: B[k][n] -> Bs[n][k]
    for (int t = tid; t < G16_BK * G16_BN; t += 256) {
      const int c = t / G16_BN, n = t - c * G16_BN;
      _Float16 v = (_Float16)0.f;
      if (c < kc && n0 + n < N) v = B[(int64_t)(k0 + c) * N + n0 + n];
      Bs[n * G16_LD + c] = v;
    }
    __syncthreads();
    const int r32 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < G16_BK; ks += 16) {
      const f16x8 a = *reinterpret_cast<const f16x8*>(&As[(wave * 32 + r32) * G16_LD + ks + 8 * h]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f16x8 b = *reinterpret_cast<const f16x8*>(&Bs[(t * 32 + r32) * G16_LD + ks + 8 * h]);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int col = n0 + t * 32 + (lane & 31);
    if (col >= N) continue;
    const float bv = (TANH_EPI && bias) ? (float)bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row >= M) continue;
      float v = acc[t][r];
      if (TANH_EPI) v = fmd_tanh_clamped(v + bv);
      if (OUT_F16) reinterpret_cast<_Float16*>(Cp)[row * N + col] = (_Float16)v;
      else reinterpret_cast<float*>(Cp)[row * N + col] = v;
    }
  }
}

template <int AK, bool TANH_EPI, bool OUT_F16>
int launch_f16(const void* a, const void* y, const void* b, const void* bias, void* c, int64_t m,
               int64_t n, int64_t k, void* stream) {
  if (m < 0 || n <= 0 || k <= 0 || n > (1 << 24) || k > (1 << 24)) return FMD_EINVAL;
  if (m == 0) return FMD_OK;
  if (!a || !b || !c) return FMD_EINVAL;
  dim3 grid((unsigned)((m + G16_BM - 1) / G16_BM), (unsigned)((n + G16_BN - 1) / G16_BN));
  gemm_f16_kernel<AK, TANH_EPI, OUT_F16><<<grid, 256, 0, fmd_stream(stream)>>>(
      a, y, reinterpret_cast<const _Float16*>(b), reinterpret_cast<const _Float16*>(bias), c, m,
      (int)n, (int)k);
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}

}  // namespace

extern "C" int fmd_gemm_f32(const float* a, const float* b, const float* bias, const float* aux,
                            float* c, int64_t m, int64_t n, int64_t k, int b_is_nk, int prologue,
                            int epilogue, void* stream) {
  if (m < 0 || n <= 0 || k <= 0 || n > (1 << 24) || k > (1 << 24)) return FMD_EINVAL;
  if (prologue < 0 || prologue > 1 || epilogue < 0 || epilogue > 2) return FMD_EINVAL;
  if (epilogue == FMD_EPI_MUL_DTANH && !aux) return FMD_EINVAL;
  if (m == 0) return FMD_OK;
  if (!a || !b || !c) return FMD_EINVAL;
  dim3 grid((unsigned)((m + G32_BM - 1) / G32_BM), (unsigned)((n + G32_BN - 1) / G32_BN));
  hipStream_t s = fmd_stream(stream);
#define FMD_G32(P, E, T) \
  gemm_f32_kernel<P, E, T><<<grid, 256, 0, s>>>(a, b, bias, aux, c, m, (int)n, (int)k)
  const int key = prologue * 6 + epilogue * 2 + (b_is_nk ? 1 : 0);
  switch (key) {
    case 0: FMD_G32(0, 0, false); break;
    case 1: FMD_G32(0, 0, true); break;
    case 2: FMD_G32(0, 1, false); break;
    case 3: FMD_G32(0, 1, true); break;
    case 4: FMD_G32(0, 2, false); break;
    case 5: FMD_G32(0, 2, true); break;
    case 6: FMD_G32(1, 0, false); break;
    case 7: FMD_G32(1, 0, true); break;
    case 8: FMD_G32(1, 1, false); break;
    case 9: FMD_G32(1, 1, true); break;
    case 10: FMD_G32(1, 2, false); break;
    case 11: FMD_G32(1, 2, true); break;
    default: return FMD_EINVAL;
  }
#undef FMD_G32
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}

extern "C" int fmd_linear_tanh_f16(const void* x, int x_is_f16, const void* w, const void* bias,
                                   void* y, int64_t m, int64_t n, int64_t k, void* stream) {
  return x_is_f16 ? launch_f16<AK_F16, true, true>(x, nullptr, w, bias, y, m, n, k, stream)
                  : launch_f16<AK_F32, true, true>(x, nullptr, w, bias, y, m, n, k, stream);
}

extern "C" int fmd_linear_f16(const void* x, const void* w, void* y, int out_is_f16, int64_t m,
                              int64_t n, int64_t k, void* stream) {
  return out_is_f16 ? launch_f16<AK_F16, false, true>(x, nullptr, w, nullptr, y, m, n, k, stream)
                    : launch_f16<AK_F16, false, false>(x, nullptr, w, nullptr, y, m, n, k, stream);
}

// gx[M,N] = f16(g * (1 - y^2))[M,K] @ w_t[K,N]   (g, y are [M,K])
extern "C" int fmd_tanh_bwd_grad_x_f16(const void* g, int g_is_f16, const void* y, const void* w_t,
                                       float* gx, int64_t m, int64_t n, int64_t k, void* stream) {
  if (!y) return FMD_EINVAL;
  return g_is_f16 ? launch_f16<AK_DTANH_F16G, false, false>(g, y, w_t, nullptr, gx, m, n, k, stream)
                  : launch_f16<AK_DTANH_F32G, false, false>(g, y, w_t, nullptr, gx, m, n, k, stream);
}

extern "C" int fmd_matmul_f32_f16_to_f16(const float* a, const void* b, void* c, int64_t m, int64_t n,
                                         int64_t k, void* stream) {
  return launch_f16<AK_F32, false, true>(a, nullptr, b, nullptr, c, m, n, k, stream);
}
