// Row gathers as the pair kernels' consumers issue them: cycles per wave instruction when 4 waves of a CU (one per
// SIMD) gather from a 17.6 MB [N][128] fp32 table.
//   form 0: global_load_dword,   lanes 0-31 one row, lanes 32-63 another (2 x 128 B per instruction)
//   form 1: global_load_dwordx2, 16 lanes per row, 4 rows (4 x 128 B per instruction)
//   form 2: global_load_dwordx4, 8 lanes per row, 8 rows (8 x 128 B per instruction)
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip && ./gather_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int FORM>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, const float* table, const int* rows, int n_iter, float* sink) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= 4) return;
  float acc = 0.f;
  const int* r = rows + blockIdx.x * 4096;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < n_iter; ++it) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      if (FORM == 0) {
        const int row = r[(it * 16 + e) * 8 + (lane >> 5)];
        acc += table[(size_t)row * 128 + 32 * wave + (lane & 31)];
      } else if (FORM == 1) {
        const int row = r[(it * 16 + e) * 8 + (lane >> 4)];
        const f2 v = *reinterpret_cast<const f2*>(table + (size_t)row * 128 + 32 * wave + 2 * (lane & 15));
        acc += v[0] + v[1];
      } else {
        const int row = r[(it * 16 + e) * 8 + (lane >> 3)];
        const f4 v = *reinterpret_cast<const f4*>(table + (size_t)row * 128 + 32 * wave + 4 * (lane & 7));
        acc += v[0] + v[1] + v[2] + v[3];
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
  if (acc == 1234.5f) sink[0] = acc;
}
int main() {
  const int N = 34432, nblk = 256, n_iter = 16;
  float* table; int* rows; unsigned long long* d; float* sink;
  hipMalloc(&table, (size_t)N * 128 * 4); hipMemset(table, 0, (size_t)N * 128 * 4);
  hipMalloc(&rows, nblk * 4096 * 4); hipMalloc(&d, nblk * 4 * 8); hipMalloc(&sink, 4);
  int* hr = new int[nblk * 4096];
  unsigned s = 1;
  for (int b = 0; b < nblk; ++b) for (int i = 0; i < 4096; ++i) { s = s * 1664525u + 1013904223u; hr[b * 4096 + i] = (b / 2) * 269 + (s >> 8) % 269; }
  hipMemcpy(rows, hr, nblk * 4096 * 4, hipMemcpyHostToDevice);
  const char* names[] = {"dword   (2 rows x 128 B)", "dwordx2 (4 rows x 128 B)", "dwordx4 (8 rows x 128 B)"};
  for (int form = 0; form < 3; ++form) {
    unsigned long long h[256 * 4];
    for (int rep = 0; rep < 2; ++rep) {
      if (form == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(512), 0, 0, d, table, rows, n_iter, sink);
      if (form == 1) hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(512), 0, 0, d, table, rows, n_iter, sink);
      if (form == 2) hipLaunchKernelGGL(k<2>, dim3(nblk), dim3(512), 0, 0, d, table, rows, n_iter, sink);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0; for (int i = 0; i < 256 * 4; ++i) sum += h[i];
    const double per = sum / (256.0 * 4 * n_iter * 16);
    printf("%s: %.1f cycles per gather instruction (4 waves per CU issuing), %.2f cycles per 128-B row segment\n", names[form], per, per / (2 << form));
  }
  return 0;
}
