// Microbenchmark: do MFMA and VALU work of the TWO waves that share a SIMD overlap?  512-thread workgroup, one per
// CU; waves w and w+4 share SIMD.  Each mode gives the two halves (waves 0-3 / 4-7) a role and prints the cycles of
// the slowest wave.  Diagnostics only.
//   hipcc --offload-arch=gfx950 -O3 -o coissue coissue.hip && ./coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define REP4(x) x x x x
#define REP8(x) x x x x x x x x

enum Role { IDLE = 0, MFMA = 1, VALU = 2, MIX = 3, LDSRW = 4, TRANS = 5, PKV = 6, MIX3 = 7, CVT = 8, MIXF = 9, CVTS = 10, PKMUL = 11, PERM = 12, SWAP = 13, GATH = 14, LDSB = 15 };

__device__ const float* g_table;   // [N][128] fp32, the gather role's rows
// one "unit" of each role ~ 8 MFMAs (256 pipe cycles) or 64 VALU (256 issue cycles)
template <int role>
__device__ __forceinline__ void run_role(int iters, float& a0, float& a1, float& a2, float& a3,
                                         f32x16& acc0, f32x16& acc1, f16x8 fa, f16x8 fb, float* lds, int lane) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 p0 = {a0, a1}, p1 = {a2, a3};
  for (int it = 0; it < iters; ++it) {
    if (role == MFMA) {
      REP4(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);
           acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc1, 0, 0, 0);)
    } else if (role == VALU) {
      REP8(asm volatile("v_fma_f32 %0, %0, %0, 1.0\n v_fma_f32 %1, %1, %1, 1.0\n v_fma_f32 %2, %2, %2, 1.0\n v_fma_f32 %3, %3, %3, 1.0\n"
                        "v_fma_f32 %0, %0, %0, 1.0\n v_fma_f32 %1, %1, %1, 1.0\n v_fma_f32 %2, %2, %2, 1.0\n v_fma_f32 %3, %3, %3, 1.0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (role == TRANS) {  // 32 v_exp + 32 v_fma
      REP8(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, 1.0\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %3, 1.0\n"
                        "v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, 1.0\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %3, 1.0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (role == PKV) {  // 64 v_pk_fma_f32
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n"
                        "v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n"
                        : "+v"(p0), "+v"(p1));)
    } else if (role == CVT) {  // 64 v_cvt_pkrtz_f16_f32
      REP8(asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2\n v_cvt_pkrtz_f16_f32 %3, %1, %2\n v_cvt_pkrtz_f16_f32 %0, %1, %2\n v_cvt_pkrtz_f16_f32 %3, %1, %2\n"
                        "v_cvt_pkrtz_f16_f32 %0, %1, %2\n v_cvt_pkrtz_f16_f32 %3, %1, %2\n v_cvt_pkrtz_f16_f32 %0, %1, %2\n v_cvt_pkrtz_f16_f32 %3, %1, %2\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (role == MIXF) {  // 64 v_fma_mix_f32
      REP8(asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n"
                        "v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (role == CVTS) {  // 32 v_cvt_f16_f32 + 32 v_cvt_f32_f16
      REP8(asm volatile("v_cvt_f16_f32 %0, %1\n v_cvt_f32_f16 %2, %3\n v_cvt_f16_f32 %0, %1\n v_cvt_f32_f16 %2, %3\n"
                        "v_cvt_f16_f32 %0, %1\n v_cvt_f32_f16 %2, %3\n v_cvt_f16_f32 %0, %1\n v_cvt_f32_f16 %2, %3\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (role == PKMUL) {  // 64 v_pk_mul_f32
      REP8(asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0\n"
                        "v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %0\n"
                        : "+v"(p0), "+v"(p1));)
    } else if (role == PERM) {  // 64 v_perm_b32
      REP8(asm volatile("v_perm_b32 %0, %1, %2, %3\n v_perm_b32 %0, %1, %2, %3\n v_perm_b32 %0, %1, %2, %3\n v_perm_b32 %0, %1, %2, %3\n"
                        "v_perm_b32 %0, %1, %2, %3\n v_perm_b32 %0, %1, %2, %3\n v_perm_b32 %0, %1, %2, %3\n v_perm_b32 %0, %1, %2, %3\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (role == SWAP) {  // 64 v_permlane32_swap
      REP8(asm volatile("v_permlane32_swap_b32 %0, %1\n s_nop 0\n v_permlane32_swap_b32 %2, %3\n s_nop 0\n v_permlane32_swap_b32 %0, %1\n s_nop 0\n v_permlane32_swap_b32 %2, %3\n s_nop 0\n"
                        "v_permlane32_swap_b32 %0, %1\n s_nop 0\n v_permlane32_swap_b32 %2, %3\n s_nop 0\n v_permlane32_swap_b32 %0, %1\n s_nop 0\n v_permlane32_swap_b32 %2, %3\n s_nop 0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (role == MIX) {  // 8 MFMA, each followed by 5 independent VALU (program order interleave)
      REP4(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);
           asm volatile("v_fma_f32 %0, %0, %0, 1.0\n v_fma_f32 %1, %1, %1, 1.0\n v_fma_f32 %2, %2, %2, 1.0\n v_fma_f32 %3, %3, %3, 1.0\n v_fma_f32 %0, %0, %0, 1.0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
           acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc1, 0, 0, 0);
           asm volatile("v_fma_f32 %0, %0, %0, 1.0\n v_fma_f32 %1, %1, %1, 1.0\n v_fma_f32 %2, %2, %2, 1.0\n v_fma_f32 %3, %3, %3, 1.0\n v_fma_f32 %0, %0, %0, 1.0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (role == MIX3) {  // 8 MFMA, each followed by 3 independent VALU
      REP4(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);
           asm volatile("v_fma_f32 %0, %0, %0, 1.0\n v_fma_f32 %1, %1, %1, 1.0\n v_fma_f32 %2, %2, %2, 1.0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
           acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc1, 0, 0, 0);
           asm volatile("v_fma_f32 %0, %0, %0, 1.0\n v_fma_f32 %1, %1, %1, 1.0\n v_fma_f32 %2, %2, %2, 1.0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (role == GATH) {  // 32 row gathers (global_load_dword, lanes 0-31 one 128-byte row segment, lanes 32-63 another)
      float v[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const unsigned row = ((unsigned)(it * 32 + k) * 2654435761u + (lane >> 5) * 40503u) % 269u;
        v[k] = g_table[(size_t)(blockIdx.x / 2 * 269 + row) * 128 + (lane & 31) + 32 * (threadIdx.x >> 6 & 3)];
      }
#pragma unroll
      for (int k = 0; k < 32; ++k) a0 += v[k];
    } else if (role == LDSB) {  // 32 ds_read_b128 (fragment reads)
      typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const f4 v = *reinterpret_cast<const f4*>(&lds[((k * 64 + lane) * 4) & 4095]);
        a0 += v[0] + v[3];
      }
    } else if (role == LDSRW) {  // 16 ds_read_b32 + 16 fma + 16 ds_write_b32 (the aggregate update pattern)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        float v = lds[(k * 67 + lane) & 4095];
        lds[(k * 67 + lane) & 4095] = v * a0 + a1;
      }
    }
  }
  a0 += p0[0] + p1[1];
}

template <int role_lo, int role_hi>
__global__ __launch_bounds__(512, 2) void bench(unsigned long long* out, float seed, int iters_lo, int iters_hi) {
  __shared__ float lds[4096];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = seed;
  float a0 = seed + threadIdx.x, a1 = a0 * 0.5f, a2 = a0 * 0.25f, a3 = a0 * 0.125f;
  f32x16 acc0 = {0}, acc1 = {0};
  f16x8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 1, 1, 1, 1, 1, 1};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (wave < 4) run_role<role_lo>(iters_lo, a0, a1, a2, a3, acc0, acc1, fa, fb, lds, lane);
  else run_role<role_hi>(iters_hi, a0, a1, a2, a3, acc0, acc1, fa, fb, lds, lane);
  // wait for the MFMA results before the end stamp
  float sink = acc0[0] + acc1[3] + a0 + a1 + a2 + a3;
  asm volatile("" : "+v"(sink));
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (sink == 12345.678f) out[0] = 1;
}


typedef float f32x4v __attribute__((ext_vector_type(4)));
enum Role16 { I16 = 0, M16 = 1, V16 = 2, X16 = 3 };  // idle, 16 MFMA 16x16x32, 64 v_fma, 16 MFMA each followed by 3 VALU
template <int role>
__device__ __forceinline__ void run16(int iters, float& a0, float& a1, float& a2, float& a3, f32x4v& c0, f32x4v& c1,
                                      f16x8 fa, f16x8 fb) {
  for (int it = 0; it < iters; ++it) {
    if (role == M16) {
      REP8(c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c0, 0, 0, 0);
           c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c1, 0, 0, 0);)
    } else if (role == V16) {
      REP8(asm volatile("v_fma_f32 %0, %0, %0, 1.0\n v_fma_f32 %1, %1, %1, 1.0\n v_fma_f32 %2, %2, %2, 1.0\n v_fma_f32 %3, %3, %3, 1.0\n"
                        "v_fma_f32 %0, %0, %0, 1.0\n v_fma_f32 %1, %1, %1, 1.0\n v_fma_f32 %2, %2, %2, 1.0\n v_fma_f32 %3, %3, %3, 1.0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if (role == X16) {
      REP8(c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c0, 0, 0, 0);
           asm volatile("v_fma_f32 %0, %0, %0, 1.0\n v_fma_f32 %1, %1, %1, 1.0\n v_fma_f32 %2, %2, %2, 1.0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
           c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c1, 0, 0, 0);
           asm volatile("v_fma_f32 %0, %0, %0, 1.0\n v_fma_f32 %1, %1, %1, 1.0\n v_fma_f32 %2, %2, %2, 1.0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    }
  }
}
template <int R0, int R1, int R2, int R3>
__global__ __launch_bounds__(1024) void bench16(unsigned long long* out, float seed) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float a0 = seed + threadIdx.x, a1 = a0 * 0.5f, a2 = a0 * 0.25f, a3 = a0 * 0.125f;
  f32x4v c0 = {0}, c1 = {0};
  f16x8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 1, 1, 1, 1, 1, 1};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  const int g = wave >> 2;
  if (g == 0) run16<R0>(R0 ? 64 : 0, a0, a1, a2, a3, c0, c1, fa, fb);
  else if (g == 1) run16<R1>(R1 ? 64 : 0, a0, a1, a2, a3, c0, c1, fa, fb);
  else if (g == 2) run16<R2>(R2 ? 64 : 0, a0, a1, a2, a3, c0, c1, fa, fb);
  else run16<R3>(R3 ? 64 : 0, a0, a1, a2, a3, c0, c1, fa, fb);
  float sink = c0[0] + c1[3] + a0 + a1 + a2 + a3;
  asm volatile("" : "+v"(sink));
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
  if (sink == 12345.678f) out[0] = 1;
}
unsigned long long* d;
const int nblk = 256;
template <int R0, int R1, int R2, int R3>
void run16h(const char* name) {
  static unsigned long long h[nblk * 16];
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((bench16<R0, R1, R2, R3>), dim3(nblk), dim3(1024), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
  }
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double g[4] = {0, 0, 0, 0};
  for (int b = 0; b < nblk; ++b)
    for (int w = 0; w < 16; ++w) g[w >> 2] += h[b * 16 + w];
  printf("%-44s groups: %7.0f %7.0f %7.0f %7.0f cyc\n", name, g[0] / (4.0 * nblk), g[1] / (4.0 * nblk),
         g[2] / (4.0 * nblk), g[3] / (4.0 * nblk));
}
template <int LO, int HI>
void run(const char* name) {
  static unsigned long long h[nblk * 8];
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((bench<LO, HI>), dim3(nblk), dim3(512), 0, 0, d, 1.0f, LO == IDLE ? 0 : 64, HI == IDLE ? 0 : 64);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
  }
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long long lo = 0, hi = 0;
  for (int b = 0; b < nblk; ++b)
    for (int w = 0; w < 8; ++w) {
      unsigned long long v = h[b * 8 + w];
      if (w < 4) lo += v; else hi += v;
    }
  printf("%-40s waves0-3 %8.0f cyc   waves4-7 %8.0f cyc   kernel %.1f us\n", name, lo / (4.0 * nblk),
         hi / (4.0 * nblk), ms * 1000.f);
}

int main() {
  hipMalloc(&d, nblk * 16 * sizeof(unsigned long long));
  {
    float* t;
    hipMalloc(&t, (size_t)34432 * 128 * 4);
    hipMemset(t, 0, (size_t)34432 * 128 * 4);
    hipMemcpyToSymbol(HIP_SYMBOL(g_table), &t, sizeof(t));
  }
  // per 64 iterations: MFMA = 512 MFMA; VALU = 4096 v_fma; MIX = 512 MFMA + 2560 v_fma; MIX3 = 512 + 1536;
  // TRANS = 2048 v_exp + 2048 v_fma; PKV = 4096 v_pk_fma; LDSRW = 1024 (ds_read, fma, ds_write)
  run<MFMA, IDLE>("MFMA alone (waves 0-3)");
  run<IDLE, VALU>("VALU alone (waves 4-7)");
  run<VALU, IDLE>("VALU alone (waves 0-3)");
  run<MFMA, VALU>("older MFMA | younger VALU");
  run<VALU, MFMA>("older VALU | younger MFMA");
  run<MFMA, MFMA>("MFMA | MFMA");
  run<VALU, VALU>("VALU | VALU");
  run<MIX, IDLE>("MIX (8 MFMA + 40 VALU) alone");
  run<MIX, MIX>("MIX | MIX");
  run<MIX3, IDLE>("MIX3 (8 MFMA + 24 VALU) alone");
  run<MIX3, MIX3>("MIX3 | MIX3");
  run<MIX, VALU>("older MIX | younger VALU");
  run<TRANS, IDLE>("TRANS alone");
  run<MFMA, TRANS>("older MFMA | younger TRANS");
  run<PKV, IDLE>("PKV alone");
  run<MFMA, PKV>("older MFMA | younger PKV");
  run<CVT, IDLE>("CVT (cvt_pkrtz) alone");
  run<MFMA, CVT>("older MFMA | younger CVT");
  run<MIXF, IDLE>("MIXF (v_fma_mix_f32) alone");
  run<MFMA, MIXF>("older MFMA | younger MIXF");
  run<CVTS, IDLE>("CVTS (cvt_f16_f32 / cvt_f32_f16) alone");
  run<MFMA, CVTS>("older MFMA | younger CVTS");
  run<PKMUL, IDLE>("PKMUL alone");
  run<MFMA, PKMUL>("older MFMA | younger PKMUL");
  run<PKMUL, MFMA>("older PKMUL | younger MFMA");
  run<PERM, IDLE>("PERM alone");
  run<MFMA, PERM>("older MFMA | younger PERM");
  run<SWAP, IDLE>("SWAP (permlane32_swap) alone");
  run<MFMA, SWAP>("older MFMA | younger SWAP");
  run<LDSRW, IDLE>("LDSRW alone");
  run<LDSRW, MFMA>("older LDSRW | younger MFMA");
  run<MFMA, LDSRW>("older MFMA | younger LDSRW");
  run<LDSRW, VALU>("older LDSRW | younger VALU");
  run<LDSRW, LDSRW>("LDSRW | LDSRW");
  // row gathers (32 per iteration) and 16-byte LDS fragment reads (32 per iteration) beside the other wave's work
  run<GATH, IDLE>("GATH alone (waves 0-3)");
  run<GATH, MFMA>("older GATH | younger MFMA");
  run<MFMA, GATH>("older MFMA | younger GATH");
  run<GATH, VALU>("older GATH | younger VALU");
  run<VALU, GATH>("older VALU | younger GATH");
  run<GATH, GATH>("GATH | GATH");
  run<LDSB, IDLE>("LDSB alone");
  run<LDSB, MFMA>("older LDSB | younger MFMA");
  run<MFMA, LDSB>("older MFMA | younger LDSB");
  run<LDSB, VALU>("older LDSB | younger VALU");
  // 16 waves (4 per SIMD): per 64 iterations M16 = 1024 MFMA 16x16x32, V16 = 4096 v_fma, X16 = 1024 MFMA + 3072 v_fma
  run16h<M16, I16, I16, I16>("16w: M16 alone");
  run16h<M16, M16, I16, I16>("16w: M16 M16");
  run16h<V16, I16, I16, I16>("16w: V16 alone");
  run16h<V16, V16, V16, V16>("16w: V16 x4");
  run16h<X16, I16, I16, I16>("16w: X16 alone");
  run16h<X16, X16, I16, I16>("16w: X16 X16");
  run16h<X16, X16, X16, X16>("16w: X16 x4");
  run16h<M16, M16, V16, V16>("16w: M16 M16 V16 V16");
  run16h<M16, V16, M16, V16>("16w: M16 V16 M16 V16");
  run16h<X16, X16, V16, V16>("16w: X16 X16 V16 V16");
  return 0;
}
