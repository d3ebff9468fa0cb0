// Shared device/host helpers for the gfx950 CGSchNet kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "flashmd_hip.h"

#define FMD_PI_F 3.14159265358979323846f

#define FMD_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return FMD_ELAUNCH;               \
  } while (0)

static inline hipStream_t fmd_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// C(d) = 0.5 (cos(pi d / rc) + 1) [d < rc]      (reference models/cutoff.py:137-145)
__device__ __forceinline__ float fmd_cutoff(float d, float rc) {
  float c = 0.5f * (cosf(d * FMD_PI_F / rc) + 1.0f);
  return d < rc ? c : 0.0f;
}
// C'(d) = -(pi / 2 rc) sin(pi d / rc) [d < rc]
__device__ __forceinline__ float fmd_dcutoff(float d, float rc) {
  float s = -0.5f * (FMD_PI_F / rc) * sinf(d * FMD_PI_F / rc);
  return d < rc ? s : 0.0f;
}
// cutoff and derivative together
__device__ __forceinline__ void fmd_cutoff2(float d, float rc, float& c, float& dc) {
  float s, co;
  sincosf(d * FMD_PI_F / rc, &s, &co);
  bool in = d < rc;
  c = in ? 0.5f * (co + 1.0f) : 0.0f;
  dc = in ? -0.5f * (FMD_PI_F / rc) * s : 0.0f;
}

// fp32-accurate tanh: 1 - 2/(e^{2x}+1), |err| ~ 1e-7 absolute; exact saturation for large |x|.
__device__ __forceinline__ float fmd_tanh(float x) {
  float ax = fabsf(x);
  float r;
  if (ax < 0.25f) {  // odd polynomial (Taylor to x^9) keeps relative accuracy near 0
    float x2 = x * x;
    r = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * (-0.053968254f + x2 * 0.021869488f))));
  } else {
    float e = __expf(2.0f * ax);
    float t = 1.0f - 2.0f / (e + 1.0f);
    r = copysignf(t, x);
  }
  return r;
}

// branch-free tanh for the fused kernels: 1 - 2 rcp(e^{2x} + 1); |abs err| ~ 2e-7, exact at +-inf.
// (absolute accuracy is what the consumers need: the value feeds a contraction and 1 - t^2)
__device__ __forceinline__ float fmd_tanh_fast(float x) {
  const float e = __expf(2.0f * x);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// the reference's Triton tanh for the W16A16 kernels (kernels/cfconv_kernels.py:449-454)
__device__ __forceinline__ float fmd_tanh_clamped(float x) {
  float xc = fminf(fmaxf(x, -10.0f), 10.0f);
  float e = __expf(2.0f * xc);
  return (e - 1.0f) / (e + 1.0f);
}

// v + (v of the lane 32 away), in every lane: v_permlane32_swap (one VALU op on gfx950) instead of the LDS
// round trip of ds_bpermute.  swap(x, y): lanes 32-63 of x <-> lanes 0-31 of y, so with x = y = v the pair
// becomes (lower half everywhere, upper half everywhere).
__device__ __forceinline__ float fmd_xor32_sum(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
}

__device__ __forceinline__ float fmd_wave_sum(float v) {
  v = fmd_xor32_sum(v);
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
