// LDS float atomic add (ds_add_f32, no return) against read-modify-write (ds_read_b32 + v_fma + ds_write_b32):
// cycles per wave-instruction for the aggregate-update pattern of the pair kernels (lanes 0-31 one row, lanes 32-63
// another row of a [n][128] fp32 array; 8 waves per workgroup as in the kernels, only waves 0-3 active).
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic lds_atomic.hip && ./lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, const int* rows, float seed) {
  extern __shared__ float agg[];  // [256][128]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5, c = lane & 31;
  for (int i = threadIdx.x; i < 256 * 128; i += 512) agg[i] = 0.f;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  if (wave < 4) {
    float v = seed + lane;
    for (int it = 0; it < 64; ++it) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = rows[(it * 16 + e) * 2 + (MODE == 2 ? 0 : h)];  // MODE 2: both halves hit the same row (same address per column)
        float* p = agg + row * 128 + 32 * wave + c;
        if (MODE == 0) {
          *p = *p + v * 1.0001f;  // read-modify-write
        } else {
          asm volatile("ds_add_f32 %0, %1" : : "v"((unsigned)(uintptr_t)p), "v"(v * 1.0001f) : "memory");
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  __syncthreads();
  if (agg[threadIdx.x] == 12345.f) out[0] = 0;
}
int main() {
  unsigned long long* d; int* rows;
  hipMalloc(&d, 256 * 8 * 8); hipMalloc(&rows, 64 * 16 * 2 * 4);
  int hr[64 * 16 * 2];
  unsigned s = 12345;
  for (int i = 0; i < 64 * 16 * 2; ++i) { s = s * 1664525u + 1013904223u; hr[i] = (s >> 8) % 256; }
  hipMemcpy(rows, hr, sizeof(hr), hipMemcpyHostToDevice);
  const char* names[] = {"read + fma + write", "ds_add_f32 (halves on different rows)", "ds_add_f32 (both halves on the same address)"};
  for (int mode = 0; mode < 3; ++mode) {
    unsigned long long h[256 * 8];
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) { hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 131072, 0, d, rows, 1.f); }
      if (mode == 1) { hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 131072, 0, d, rows, 1.f); }
      if (mode == 2) { hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 131072, 0, d, rows, 1.f); }
      hipDeviceSynchronize();
    }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < 4; ++w) sum += h[b * 8 + w];
    printf("%-48s %.1f cycles per update (wave view, 4 waves per CU updating)\n", names[mode], sum / (256.0 * 4 * 64 * 16));
  }
  return 0;
}
