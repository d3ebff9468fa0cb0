// Check of the transposed LDS read used for the gW hand-over: image [k = feature][n = pair] (64-byte rows of 32
// halves); the B fragment of v_mfma_f32_32x32x16_f16 for lane l = (n = l & 31, half = l >> 5) is
// { B[8 half + j][n], j = 0..7 } = two ds_read_b64_tr_b16 (rows 8 half .. +3 and +4 .. +7).
//   hipcc --offload-arch=gfx950 -O3 -o tr_read tr_read.hip && ./tr_read
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(float* out) {
  __shared__ __attribute__((aligned(16))) __fp16 img[16 * 32];
  const int l = threadIdx.x;
  for (int i = l; i < 16 * 32; i += 64) img[i] = (__fp16)(float)((i / 32) * 64 + (i % 32));  // k * 64 + n
  __syncthreads();
  const int n = l & 31, half = l >> 5, g = (l >> 4) & 1, i = l & 15, q = i >> 2, p = i & 3;
  // lane 4q+p of a 16-lane group supplies the address of row (k0 + q), columns 16 g + 4p .. +3
  const int k0 = 8 * half;
  const __fp16* a0 = img + (k0 + q) * 32 + 16 * g + 4 * p;
  const __fp16* a1 = img + (k0 + 4 + q) * 32 + 16 * g + 4 * p;
  h4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)a0);
  h4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)a1);
  int bad = 0;
  for (int j = 0; j < 4; ++j) {
    bad += (float)v0[j] != (float)((k0 + j) * 64 + n);
    bad += (float)v1[j] != (float)((k0 + 4 + j) * 64 + n);
  }
  out[l] = bad;
  out[64 + l] = (float)v0[1];
}
int main() {
  float* d; hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0; for (int l = 0; l < 64; ++l) bad += (int)h[l];
  printf("transposed-read fragment check: %d mismatches (lane 5 element 1 = %.0f, expected %d; lane 37: %.0f, expected %d)\n",
         bad, h[64 + 5], 1 * 64 + 5, h[64 + 37], 9 * 64 + 5);
  return 0;
}
