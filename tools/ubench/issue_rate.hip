// Microbenchmark: cycles per wave-instruction of the VALU forms the pair-kernel producers use,
// for independent and dependent chains, at 1 and 2 waves per SIMD.  Diagnostics only.
//   hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip && ./issue_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void bench(unsigned long long* out, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  f32x16 acc0 = {0}, acc1 = {0};
  f16x8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 1, 1, 1, 1, 1, 1};
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 16; ++it) {
    if (KIND == 0) {  // independent v_fma_f32 (8 chains)
      REP8(asm volatile("v_fma_f32 %0, %0, %0, 1.0\n v_fma_f32 %1, %1, %1, 1.0\n v_fma_f32 %2, %2, %2, 1.0\n v_fma_f32 %3, %3, %3, 1.0\n"
                        "v_fma_f32 %4, %4, %4, 1.0\n v_fma_f32 %5, %5, %5, 1.0\n v_fma_f32 %6, %6, %6, 1.0\n v_fma_f32 %7, %7, %7, 1.0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 1) {  // dependent v_fma_f32 (1 chain)
      REP64(asm volatile("v_fma_f32 %0, %0, %0, 1.0\n" : "+v"(a0));)
    } else if (KIND == 2) {  // independent v_pk_fma_f32 (4 chains)
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                        "v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
    } else if (KIND == 3) {  // dependent v_pk_fma_f32
      REP64(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n" : "+v"(p0));)
    } else if (KIND == 4) {  // independent v_exp_f32 (8 chains)
      REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                        "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 5) {  // dependent v_exp_f32
      REP64(asm volatile("v_exp_f32 %0, %0\n s_nop 0\n" : "+v"(a0));)
    } else if (KIND == 6) {  // v_cvt_pkrtz independent
      REP8(asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2\n v_cvt_pkrtz_f16_f32 %3, %1, %2\n v_cvt_pkrtz_f16_f32 %4, %1, %2\n v_cvt_pkrtz_f16_f32 %5, %1, %2\n"
                        "v_cvt_pkrtz_f16_f32 %0, %1, %2\n v_cvt_pkrtz_f16_f32 %3, %1, %2\n v_cvt_pkrtz_f16_f32 %4, %1, %2\n v_cvt_pkrtz_f16_f32 %5, %1, %2\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5));)
    } else if (KIND == 7) {  // v_fma_mix_f32 independent
      REP8(asm volatile("v_fma_mix_f32 %0, %4, %5, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %4, %5, %6 op_sel_hi:[1,0,0]\n"
                        "v_fma_mix_f32 %2, %4, %5, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %4, %5, %6 op_sel_hi:[1,0,0]\n"
                        "v_fma_mix_f32 %0, %4, %5, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %4, %5, %6 op_sel_hi:[1,0,0]\n"
                        "v_fma_mix_f32 %2, %4, %5, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %4, %5, %6 op_sel_hi:[1,0,0]\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6));)
    } else if (KIND == 8) {  // MFMA 32x32x16 f16, two accumulators alternating
      REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);
           acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc1, 0, 0, 0);
           acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);
           acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc1, 0, 0, 0);
           acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);
           acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc1, 0, 0, 0);
           acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);
           acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc1, 0, 0, 0);)
    } else if (KIND == 9) {  // MFMA + 7 independent VALU between (does VALU hide under the MFMA?)
      REP8(acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);
           asm volatile("v_fma_f32 %0, %0, %0, 1.0\n v_fma_f32 %1, %1, %1, 1.0\n v_fma_f32 %2, %2, %2, 1.0\n v_fma_f32 %3, %3, %3, 1.0\n"
                        "v_fma_f32 %4, %4, %4, 1.0\n v_fma_f32 %5, %5, %5, 1.0\n v_fma_f32 %6, %6, %6, 1.0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6));)
    } else if (KIND == 10) {  // s_nop 0 only (pure issue)
      REP64(asm volatile("s_nop 0\n");)
    } else if (KIND == 11) {  // v_mov independent
      REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %2, %1\n v_mov_b32 %3, %1\n v_mov_b32 %4, %1\n"
                        "v_mov_b32 %0, %1\n v_mov_b32 %2, %1\n v_mov_b32 %3, %1\n v_mov_b32 %4, %1\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4));)
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1] + acc0[0] + acc1[3];
  if (sink == 12345.678f) out[1000] = 1;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, int n_inst, unsigned long long* d_out) {
  for (int threads : {256, 512, 1024}) {
    unsigned long long h[16];
    hipLaunchKernelGGL(bench<KIND>, dim3(1), dim3(threads), 0, 0, d_out, 1.0f);
    hipLaunchKernelGGL(bench<KIND>, dim3(1), dim3(threads), 0, 0, d_out, 1.0f);
    hipDeviceSynchronize();
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int w = 0; w < threads / 64; ++w) s += h[w];
    s /= threads / 64;
    printf("%-28s waves/SIMD %d : %7.2f cycles per instruction (wave view)\n", name, threads / 256, s / (16.0 * n_inst));
  }
}

int main() {
  unsigned long long* d_out;
  hipMalloc(&d_out, 2048 * sizeof(unsigned long long));
  run<0>("v_fma_f32 independent", 64, d_out);
  run<1>("v_fma_f32 dependent", 64, d_out);
  run<2>("v_pk_fma_f32 independent", 64, d_out);
  run<3>("v_pk_fma_f32 dependent", 64, d_out);
  run<4>("v_exp_f32 independent", 64, d_out);
  run<5>("v_exp_f32 dependent+nop", 128, d_out);
  run<6>("v_cvt_pkrtz independent", 64, d_out);
  run<7>("v_fma_mix_f32 independent", 64, d_out);
  run<8>("mfma 32x32x16 f16", 64, d_out);
  run<9>("mfma + 7 valu", 64, d_out);
  run<10>("s_nop 0", 64, d_out);
  run<11>("v_mov_b32 independent", 64, d_out);
  return 0;
}
