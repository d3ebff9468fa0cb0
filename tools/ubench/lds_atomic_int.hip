// Scattered LDS atomics as in finish_kernel (every lane adds to a random element of a small per-wave array):
// ds_add_f32 against ds_add_u32 and ds_add_u64 (fixed-point accumulation).  Cycles per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_int lds_atomic_int.hip && ./lds_atomic_int
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, const int* idx, float seed) {
  __shared__ unsigned long long arr[4 * 1024];  // per wave 1024 slots (8 KB)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 4 * 1024; i += 256) arr[i] = 0;
  __syncthreads();
  unsigned long long* mine = arr + wave * 1024;
  const unsigned long long t0 = __builtin_readcyclecounter();
  float v = seed + lane;
  for (int it = 0; it < 256; ++it) {
    const int j = idx[it * 64 + lane] & 1023;
    if (MODE == 0) asm volatile("ds_add_f32 %0, %1" : : "v"((unsigned)(uintptr_t)(mine + j)), "v"(v) : "memory");
    if (MODE == 1) asm volatile("ds_add_u32 %0, %1" : : "v"((unsigned)(uintptr_t)(mine + j)), "v"((unsigned)(v * 1024.f)) : "memory");
    if (MODE == 2) {
      const unsigned long long q = (unsigned long long)(long long)(v * 1048576.f);
      asm volatile("ds_add_u64 %0, %1" : : "v"((unsigned)(uintptr_t)(mine + j)), "v"(q) : "memory");
    }
    v += 0.5f;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
  __syncthreads();
  if (arr[threadIdx.x] == 12345ull) out[0] = 0;
}
int main() {
  unsigned long long* d; int* idx;
  hipMalloc(&d, 256 * 4 * 8); hipMalloc(&idx, 256 * 64 * 4);
  static int hi[256 * 64];
  unsigned s = 777;
  for (int i = 0; i < 256 * 64; ++i) { s = s * 1664525u + 1013904223u; hi[i] = (s >> 8); }
  hipMemcpy(idx, hi, sizeof(hi), hipMemcpyHostToDevice);
  const char* names[] = {"ds_add_f32", "ds_add_u32", "ds_add_u64"};
  for (int mode = 0; mode < 3; ++mode) {
    static unsigned long long h[256 * 4];
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, d, idx, 1.f);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, d, idx, 1.f);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, d, idx, 1.f);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0; for (int i = 0; i < 256 * 4; ++i) sum += h[i];
    printf("%-12s %.1f cycles per scattered wave-instruction (4 waves per CU)\n", names[mode], sum / (256.0 * 4 * 256));
  }
  return 0;
}
