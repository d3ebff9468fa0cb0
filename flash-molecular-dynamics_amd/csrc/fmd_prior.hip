// Row N2: classical prior terms of the coarse-grained force field, one fused kernel per term kind in the
// "Static" form (per-term parameter arrays; reference simulation/specialize_prior.py:76-207):
//   bonds      prior/harmonic.py:91-123,200-225       U = k (d - x0)^2
//   angles     prior/harmonic.py:228-262              U = k (cos(theta) - x0)^2
//   repulsion  prior/repulsion.py:80-122              U = (sigma / d)^6
//   dihedrals  prior/fourier_series.py:97-192         U = v0 + sum_n k1_n sin(n phi) + k2_n cos(n phi)
// with the features of geometry/internal_coordinates.py:72-223.  Energy and the analytic -dU/dpos are added to
// energy[B] / forces[N,3] in one pass: one workgroup per molecule keeps the molecule's positions in LDS, each
// of its 4 waves accumulates a private copy of the gradient (LDS float atomics are ordered inside one wave),
// the copies and the energy partials are combined in fixed order => reproducible sums.  Terms are molecule-major
// (collated batches are), term_ptr[m]..term_ptr[m+1] is molecule m's range.  HBM-bound on the term list:
// 8*order + 4*P bytes per term.
#include "fmd_common.h"

namespace {

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

template <int KIND>  // 0 bonds, 1 angles (cos), 2 repulsion, 3 dihedrals
__global__ __launch_bounds__(256) void prior_kernel(const float* __restrict__ pos,
                                                    const int64_t* __restrict__ mol_ptr,
                                                    const int64_t* __restrict__ mapping,
                                                    const float* __restrict__ params, int n_params,
                                                    const int64_t* __restrict__ term_ptr, int64_t n_terms,
                                                    float* __restrict__ energy, float* __restrict__ forces,
                                                    int32_t* __restrict__ status) {
  constexpr int ORDER = KIND == 0 || KIND == 2 ? 2 : (KIND == 1 ? 3 : 4);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t a0 = mol_ptr[m];
  const int n = (int)(mol_ptr[m + 1] - a0);
  const int64_t t0 = term_ptr[m], t1 = term_ptr[m + 1];
  if (t1 <= t0) return;  // uniform
  float* sp = smem;                       // [3n] positions
  float* g = smem + 3 * n + wv * 3 * n;   // this wave's [3n] dU/dpos
  __shared__ float esum[4];
  for (int i = tid; i < 3 * n; i += 256) sp[i] = pos[3 * a0 + i];
  for (int i = tid; i < 12 * n; i += 256) smem[3 * n + i] = 0.f;
  __syncthreads();
  auto P = [&](int i) { return V3{sp[3 * i], sp[3 * i + 1], sp[3 * i + 2]}; };
  auto add = [&](int i, V3 v) {
    atomicAdd(&g[3 * i], v.x);
    atomicAdd(&g[3 * i + 1], v.y);
    atomicAdd(&g[3 * i + 2], v.z);
  };
  const int64_t cnt = t1 - t0, chunk = (cnt + 3) / 4;
  const int64_t w0 = t0 + wv * chunk, w1 = min(t1, w0 + chunk);
  float e = 0.f;
  bool bad = false;
  for (int64_t t = w0 + lane; t < w1; t += 64) {
    int idx[ORDER];
#pragma unroll
    for (int o = 0; o < ORDER; ++o) {
      const int64_t b = mapping[o * n_terms + t] - a0;
      bad |= b < 0 || b >= n;
      idx[o] = (int)min(max(b, (int64_t)0), (int64_t)n - 1);
    }
    if (KIND == 0 || KIND == 2) {
      const V3 r = P(idx[1]) - P(idx[0]);
      const float d2 = dot(r, r), d = sqrtf(d2);
      float de;  // dU/dd
      if (KIND == 0) {
        const float k = params[t], x0 = params[n_terms + t];
        e += k * (d - x0) * (d - x0);
        de = 2.0f * k * (d - x0);
      } else {
        const float s = params[t] / d, s2 = s * s, u = s2 * s2 * s2;
        e += u;
        de = -6.0f * u / d;
      }
      const V3 gj = (de / d) * r;
      add(idx[1], gj);
      add(idx[0], -1.0f * gj);
    } else if (KIND == 1) {
      const V3 r1 = P(idx[0]) - P(idx[1]), r2 = P(idx[2]) - P(idx[1]);
      const float n1 = sqrtf(dot(r1, r1)), n2 = sqrtf(dot(r2, r2)), inv = 1.0f / (n1 * n2);
      const float c = dot(r1, r2) * inv;
      const float k = params[t], x0 = params[n_terms + t];
      e += k * (c - x0) * (c - x0);
      const float de = 2.0f * k * (c - x0);  // dU/dcos
      const V3 gi = de * (inv * r2 - (c / (n1 * n1)) * r1);
      const V3 gk = de * (inv * r1 - (c / (n2 * n2)) * r2);
      add(idx[0], gi);
      add(idx[2], gk);
      add(idx[1], -1.0f * (gi + gk));
    } else {
      const V3 ri = P(idx[0]), rj = P(idx[1]), rk = P(idx[2]), rl = P(idx[3]);
      // phi = atan2(-(n1 x b2^) . n2, n1 . n2), unit bond vectors (internal_coordinates.py:210-221)
      const V3 b1 = rj - ri, b2 = rk - rj, b3 = rl - rk;
      const V3 u1 = rsqrtf(dot(b1, b1)) * b1, u2 = rsqrtf(dot(b2, b2)) * b2, u3 = rsqrtf(dot(b3, b3)) * b3;
      const V3 n1 = cross(u1, u2), n2 = cross(u2, u3), m1 = cross(n1, u2);
      const float phi = atan2f(-dot(m1, n2), dot(n1, n2));
      const int n_deg = (n_params - 1) / 2;
      float u = params[t], du = 0.f;  // U and dU/dphi
      for (int q = 1; q <= n_deg; ++q) {
        float s, c;
        sincosf((float)q * phi, &s, &c);
        const float k1 = params[(int64_t)q * n_terms + t], k2 = params[(int64_t)(n_deg + q) * n_terms + t];
        u += k1 * s + k2 * c;
        du += (float)q * (k1 * c - k2 * s);
      }
      e += u;
      // dphi/dr with F = ri - rj, G = rj - rk, H = rl - rk, A = F x G, B = H x G
      const V3 Fv = ri - rj, Gv = rj - rk, Hv = rl - rk;
      const V3 A = cross(Fv, Gv), B = cross(Hv, Gv);
      const float nG = sqrtf(dot(Gv, Gv)), iA = 1.0f / dot(A, A), iB = 1.0f / dot(B, B);
      const float fg = dot(Fv, Gv) / nG, hg = dot(Hv, Gv) / nG;
      const V3 gi = (-du * nG * iA) * A;
      const V3 gl = (du * nG * iB) * B;
      const V3 s = (du * fg * iA) * A - (du * hg * iB) * B;
      add(idx[0], gi);
      add(idx[3], gl);
      add(idx[1], s - gi);
      add(idx[2], -1.0f * (s + gl));
    }
  }
  e = fmd_wave_sum(e);
  if (lane == 0) esum[wv] = e;
  if (bad) atomicOr(&status[3], 1);
  __syncthreads();
  if (tid == 0) energy[m] += ((esum[0] + esum[1]) + esum[2]) + esum[3];
  for (int i = tid; i < 3 * n; i += 256) {
    const float* g0 = smem + 3 * n;
    forces[3 * a0 + i] -= ((g0[i] + g0[3 * n + i]) + g0[6 * n + i]) + g0[9 * n + i];
  }
}

}  // namespace

extern "C" int fmd_prior_energy_forces(const float* pos, const int64_t* mol_ptr, int64_t n_mol,
                                       int64_t max_beads_per_mol, int kind, const int64_t* mapping,
                                       const float* params, int n_params, const int64_t* term_ptr,
                                       int64_t n_terms, float* energy, float* forces, int32_t* status,
                                       void* stream) {
  if (n_mol < 0 || n_terms < 0 || max_beads_per_mol < 1 || kind < 0 || kind > 3) return FMD_EINVAL;
  if (n_mol == 0 || n_terms == 0) return FMD_OK;
  if (!pos || !mol_ptr || !mapping || !params || !term_ptr || !energy || !forces || !status) return FMD_EINVAL;
  const int need = kind == 2 ? 1 : (kind == 3 ? 3 : 2);
  if (n_params < need || (kind == 3 && (n_params - 1) % 2 != 0)) return FMD_EINVAL;
  const size_t lds = (size_t)15 * max_beads_per_mol * sizeof(float);
  if (lds > 160 * 1024 - 64) return FMD_EUNSUPPORTED;
  hipStream_t s = fmd_stream(stream);
#define FMD_PRIOR(K)                                                                                    \
  {                                                                                                     \
    if (hipFuncSetAttribute((const void*)prior_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)lds) != hipSuccess)                                                    \
      return FMD_ELAUNCH;                                                                               \
    prior_kernel<K><<<(unsigned)n_mol, 256, lds, s>>>(pos, mol_ptr, mapping, params, n_params, term_ptr, \
                                                      n_terms, energy, forces, status);                 \
  }
  if (kind == 0) FMD_PRIOR(0)
  else if (kind == 1) FMD_PRIOR(1)
  else if (kind == 2) FMD_PRIOR(2)
  else FMD_PRIOR(3)
#undef FMD_PRIOR
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}
