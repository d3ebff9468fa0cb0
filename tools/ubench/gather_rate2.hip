// What bounds the consumers' row gathers?  Cycles per 128-byte row segment per CU as a function of the number of
// gathering waves (4 or 8 per CU), of the loads a wave keeps in flight (16 / 32 / 48 dword gathers = 2 segments each),
// of the table size (one molecule: 137 KB -- L1-resident? -- or the whole batch) and of the cache policy (plain /
// nontemporal).   hipcc --offload-arch=gfx950 -O3 -o gather_rate2 gather_rate2.hip && ./gather_rate2
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int NW, int DEPTH, bool NT>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, const float* table, const int* rows, int n_iter, float* sink) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= NW) return;
  float acc = 0.f;
  const int* r = rows + blockIdx.x * 8192 + wave * 1024;   // (uniform per half wave)
  const int col = 32 * (wave & 3) + (lane & 31);
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < n_iter; ++it) {
    float v[DEPTH];
#pragma unroll
    for (int e = 0; e < DEPTH; ++e) {
      const int row = r[((it * DEPTH + e) * 2 + (lane >> 5)) & 1023];
      const float* p = table + (size_t)row * 128 + col;
      v[e] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int e = 0; e < DEPTH; ++e) acc += v[e];
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (acc == 1234.5f) sink[0] = acc;
}
template <int NW, int DEPTH, bool NT>
void run(const char* what, unsigned long long* d, const float* table, const int* rows, float* sink) {
  const int nblk = 256, n_iter = 4096 / DEPTH;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<NW, DEPTH, NT>), dim3(nblk), dim3(512), 0, 0, d, table, rows, n_iter, sink);
    hipDeviceSynchronize();
  }
  unsigned long long h[256 * 8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double mx = 0;
  for (int b = 0; b < nblk; ++b) { double m = 0; for (int w = 0; w < NW; ++w) m = h[b * 8 + w] > m ? h[b * 8 + w] : m; mx += m; }
  const double per_cu = mx / nblk / ((double)n_iter * DEPTH * 2 * NW);   // cycles per segment at the CU
  printf("%-28s waves %d in flight %2d %s: %6.2f cycles per 128-B segment per CU  (%5.1f B/clk/CU)\n", what, NW, DEPTH, NT ? "nt   " : "plain", per_cu, 128.0 / per_cu);
}
int main() {
  const int N = 34432, nblk = 256;
  float* table; int* rows; unsigned long long* d; float* sink;
  hipMalloc(&table, (size_t)N * 128 * 4); hipMemset(table, 0, (size_t)N * 128 * 4);
  hipMalloc(&rows, nblk * 8192 * 4); hipMalloc(&d, nblk * 8 * 8); hipMalloc(&sink, 4);
  int* hr = new int[nblk * 8192];
  for (int mode = 0; mode < 3; ++mode) {
    unsigned s = 1;
    // mode 0: rows of the workgroup's own molecule (269 rows, 137 KB); 1: a 16-row window (8 KB: L1 hits); 2: any row of the batch
    for (int b = 0; b < nblk; ++b) for (int i = 0; i < 8192; ++i) {
      s = s * 1664525u + 1013904223u;
      hr[b * 8192 + i] = mode == 0 ? (b / 2) * 269 + (s >> 8) % 269 : mode == 1 ? (b / 2) * 269 + (s >> 8) % 16 : (s >> 8) % N;
    }
    hipMemcpy(rows, hr, nblk * 8192 * 4, hipMemcpyHostToDevice);
    const char* nm = mode == 0 ? "molecule (137 KB)" : mode == 1 ? "16-row window (8 KB)" : "whole batch (17.6 MB)";
    run<4, 16, false>(nm, d, table, rows, sink);
    run<4, 32, false>(nm, d, table, rows, sink);
    run<4, 48, false>(nm, d, table, rows, sink);
    run<8, 16, false>(nm, d, table, rows, sink);
    run<8, 32, false>(nm, d, table, rows, sink);
    run<4, 32, true>(nm, d, table, rows, sink);
    run<1, 32, false>(nm, d, table, rows, sink);
    run<2, 32, false>(nm, d, table, rows, sink);
  }
  return 0;
}
