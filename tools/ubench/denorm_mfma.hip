// Does v_mfma_f32_32x32x16_f16 honour fp16 denormal inputs?  A[m][k] = a (all), B[k][n] = 1: D = 16 a.
//   hipcc --offload-arch=gfx950 -O3 -o denorm_mfma denorm_mfma.hip && ./denorm_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out, float a) {
  f16x8 fa, fb;
  for (int j = 0; j < 8; ++j) { fa[j] = (_Float16)a; fb[j] = (_Float16)1.0f; }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)fa[0]; }
}
int main() {
  float* d; hipMalloc(&d, 8);
  const float vals[] = {1.0f, 6.2e-5f, 3.0e-5f, 1.0e-6f, 6.0e-8f, 2.4e-7f};
  for (float a : vals) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, a);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("a = %.4e  f16(a) = %.6e  mfma sum / 16 = %.6e\n", a, h[1], h[0] / 16.0f);
  }
  return 0;
}
