// Which SIMD does each wave of a workgroup run on?  (HW_REG_HW_ID: wave_id[3:0], simd_id[5:4], cu_id[11:8])
//   hipcc --offload-arch=gfx950 -O3 -o simd_map simd_map.hip && ./simd_map
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int T>
__global__ __launch_bounds__(T) void probe(unsigned* out) {
  extern __shared__ float smem[];
  const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);  // 16 bits of HW_ID
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (T / 64) + (threadIdx.x >> 6)] = hw;
  if (smem[threadIdx.x] == 1234.5f) out[0] = 0;
}
template <int T>
void run(size_t lds, int nblk) {
  unsigned* d;
  hipMalloc(&d, nblk * (T / 64) * 4);
  hipFuncSetAttribute((const void*)probe<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(probe<T>, dim3(nblk), dim3(T), lds, 0, d);
  hipDeviceSynchronize();
  unsigned h[4096];
  hipMemcpy(h, d, nblk * (T / 64) * 4, hipMemcpyDeviceToHost);
  printf("threads %d, dynamic LDS %zu: simd id per wave, first 6 workgroups\n", T, lds);
  for (int b = 0; b < 6; ++b) {
    printf("  wg %d (cu %2u):", b, (h[b * (T / 64)] >> 8) & 15);
    for (int w = 0; w < T / 64; ++w) printf(" %u", (h[b * (T / 64) + w] >> 4) & 3);
    printf("\n");
  }
  int hist[8][4] = {};
  for (int b = 0; b < nblk; ++b)
    for (int w = 0; w < T / 64; ++w) hist[w][(h[b * (T / 64) + w] >> 4) & 3]++;
  for (int w = 0; w < T / 64; ++w) printf("  wave %d: simd histogram %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  int same04 = 0, same01 = 0;
  if (T == 512)
    for (int b = 0; b < nblk; ++b) {
      same04 += ((h[b * 8] >> 4) & 3) == ((h[b * 8 + 4] >> 4) & 3);
      same01 += ((h[b * 8] >> 4) & 3) == ((h[b * 8 + 1] >> 4) & 3);
    }
  if (T == 512) printf("  waves 0 and 4 share a SIMD in %d of %d workgroups; waves 0 and 1 in %d\n", same04, nblk, same01);
  hipFree(d);
}
int main() {
  run<512>(0, 256);
  run<512>(160 * 1024, 256);
  run<256>(160 * 1024, 256);
  run<1024>(64 * 1024, 256);
  return 0;
}
