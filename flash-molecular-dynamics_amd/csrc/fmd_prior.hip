// Row N2: classical prior terms of the coarse-grained force field, one fused kernel per term kind in the
// "Static" form (per-term parameter arrays; reference simulation/specialize_prior.py:76-207):
//   bonds      prior/harmonic.py:91-123,200-225       U = k (d - x0)^2
//   angles     prior/harmonic.py:228-262              U = k (cos(theta) - x0)^2
//   repulsion  prior/repulsion.py:80-122              U = (sigma / d)^6
//   dihedrals  prior/fourier_series.py:97-192         U = v0 + sum_n k1_n sin(n phi) + k2_n cos(n phi)
//   impropers  prior/harmonic.py:301-322 (Harmonic.compute :116-118 on compute_torsions)   U = k (phi - x0)^2
//   quartic angles     prior/polynomial.py:127-186    U = v0 + sum_n k_n cos(theta)^n
//   restricted angles  prior/restricted_bending.py:200-238   U = a c^4 + b c^3 + c c^2 + d c + k / sin(theta)^2 + v0, c = cos(theta)
//   raw angles         prior/harmonic.py:264-298 (HarmonicAnglesRaw)            U = k (theta - x0)^2, theta = atan2(|r1 x r2|, r1 . r2)
//   shifted impropers  prior/harmonic.py:325-400 (ShiftedPeriodicHarmonicImpropers)  U = k (phi' - x0)^2, phi' = (phi < 0 ? phi + 2 pi : phi) - pi
// with the features of geometry/internal_coordinates.py:72-223.  Energy and the analytic -dU/dpos are added to
// energy[B] / forces[N,3] in one pass, as a GATHER over the fixed topology: one workgroup per molecule keeps the
// molecule's positions in LDS, one wave per bead walks that bead's (term, slot) entries (a CSR built once on the
// host: bead_ptr, entries = term << 2 | slot), evaluates each term and keeps the gradient of its own slot; wave
// reductions and the per-molecule energy sum run in fixed order => reproducible, and no atomics (LDS float
// atomics cost ~3 cycles per lane here: the scatter form of this kernel ran 5x slower on the 35k repulsion
// pairs of a 269-bead chain).  A term of order k is evaluated k times; the arithmetic is free next to the
// term-list traffic, 4 + 8*order + 4*P bytes per entry.
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

// WAVE_PER_BEAD: lanes of a wave share one bead's entries (long lists: the repulsion pairs); otherwise one lane
// per bead walks its own few entries (bonded terms: <= 2..4 entries per bead) and the whole molecule is one pass.
// 0 bonds, 1 angles (cos), 2 repulsion, 3 dihedrals, 4 impropers, 5 quartic angles, 6 restricted angles, 7 raw angles,
// 8 shifted-periodic impropers
template <int KIND, bool WAVE_PER_BEAD>
__global__ __launch_bounds__(1024) void prior_kernel(const float* __restrict__ pos,
                                                     const int64_t* __restrict__ mol_ptr,
                                                     const int64_t* __restrict__ mapping,
                                                     const float* __restrict__ params, int n_params,
                                                     const int64_t* __restrict__ bead_ptr,
                                                     const int32_t* __restrict__ entries, int64_t n_terms,
                                                     float* __restrict__ energy, float* __restrict__ forces,
                                                     int32_t* __restrict__ status, int max_beads) {
  constexpr int ORDER = KIND == 0 || KIND == 2 ? 2 : (KIND == 3 || KIND == 4 || KIND == 8 ? 4 : 3);
  extern __shared__ __attribute__((aligned(16))) float sp[];  // [3n] positions of the molecule
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t a0 = mol_ptr[m];
  int n = (int)(mol_ptr[m + 1] - a0);
  if (n > max_beads) {  // wrong bound from the caller: stay inside the LDS sized for it, flag it
    if (threadIdx.x == 0) atomicOr(&status[3], 2);
    n = max_beads;
  }
  __shared__ float esum[16];
  for (int i = tid; i < 3 * n; i += 1024) sp[i] = pos[3 * a0 + i];
  __syncthreads();
  auto P = [&](int i) { return V3{sp[3 * i], sp[3 * i + 1], sp[3 * i + 2]}; };
  float e_wave = 0.f;
  bool bad = false;
  // fixed order everywhere: entries ascending, wave reductions as xor trees, energy partials summed 0..15
  for (int b = WAVE_PER_BEAD ? wv : tid; b < n; b += WAVE_PER_BEAD ? 16 : 1024) {
    const int64_t e0 = bead_ptr[a0 + b], e1 = bead_ptr[a0 + b + 1];
    V3 g = {0.f, 0.f, 0.f};  // dU/d(pos of bead b)
    float e = 0.f;
    for (int64_t q = e0 + (WAVE_PER_BEAD ? lane : 0); q < e1; q += WAVE_PER_BEAD ? 64 : 1) {
      const int32_t ent = entries[q];
      const int64_t t = ent >> 2;
      const int slot = ent & 3;
      int idx[ORDER];
#pragma unroll
      for (int o = 0; o < ORDER; ++o) {
        const int64_t bb = mapping[o * n_terms + t] - a0;
        bad |= bb < 0 || bb >= n;
        idx[o] = (int)min(max(bb, (int64_t)0), (int64_t)n - 1);
      }
      float u;       // energy of the term
      V3 gs[ORDER];  // its gradient on each slot
      if (KIND == 0 || KIND == 2) {
        const V3 r = P(idx[1]) - P(idx[0]);
        const float d = sqrtf(dot(r, r));
        float de;  // dU/dd
        if (KIND == 0) {
          const float k = params[t], x0 = params[n_terms + t];
          u = k * (d - x0) * (d - x0);
          de = 2.0f * k * (d - x0);
        } else {
          const float s = params[t] / d, s2 = s * s;
          u = s2 * s2 * s2;
          de = -6.0f * u / d;
        }
        gs[1] = (de / d) * r;
        gs[0] = -1.0f * gs[1];
      } else if (ORDER == 3) {
        const V3 r1 = P(idx[0]) - P(idx[1]), r2 = P(idx[2]) - P(idx[1]);
        const float n1 = sqrtf(dot(r1, r1)), n2 = sqrtf(dot(r2, r2)), inv = 1.0f / (n1 * n2);
        const float c = dot(r1, r2) * inv;
        float de;  // dU/dcos
        if (KIND == 1) {
          const float k = params[t], x0 = params[n_terms + t];
          u = k * (c - x0) * (c - x0);
          de = 2.0f * k * (c - x0);
        } else if (KIND == 5) {  // rows: v0, k_1 .. k_n (Polynomial.compute, prior/polynomial.py:127-146)
          u = 0.f;
          de = 0.f;
          for (int q2 = n_params - 1; q2 >= 1; --q2) {  // Horner, and the derivative alongside
            const float kq = params[(int64_t)q2 * n_terms + t];
            de = de * c + (u + kq);
            u = (u + kq) * c;
          }
          u += params[t];
        } else if (KIND == 7) {  // rows: k, x0 on the raw angle; d theta / d cos = -1 / sin
          const float k = params[t], x0 = params[n_terms + t];
          const V3 nx = cross(r1, r2);
          const float sn = sqrtf(dot(nx, nx)), th = atan2f(sn, dot(r1, r2));
          u = k * (th - x0) * (th - x0);
          de = -2.0f * k * (th - x0) * (n1 * n2) / sn;
        } else {  // rows: a, b, c, d, k, v0; theta = atan2(|r1 x r2|, r1 . r2) => sin^2 = |r1 x r2|^2 / (n1 n2)^2
          const float pa = params[t], pb = params[n_terms + t], pc = params[2 * n_terms + t];
          const float pd = params[3 * n_terms + t], pk = params[4 * n_terms + t], v0 = params[5 * n_terms + t];
          const V3 nx = cross(r1, r2);
          const float is2 = 1.0f / (dot(nx, nx) * inv * inv);  // 1 / sin^2
          u = (((pa * c + pb) * c + pc) * c + pd) * c + pk * is2 + v0;
          de = ((4.0f * pa * c + 3.0f * pb) * c + 2.0f * pc) * c + pd + 2.0f * pk * c * is2 * is2;
        }
        gs[0] = de * (inv * r2 - (c / (n1 * n1)) * r1);
        gs[2] = de * (inv * r1 - (c / (n2 * n2)) * r2);
        gs[1] = -1.0f * (gs[0] + gs[2]);
      } else {
        const V3 ri = P(idx[0]), rj = P(idx[1]), rk = P(idx[2]), rl = P(idx[3]);
        // phi = atan2(-(n1 x b2^) . n2, n1 . n2), unit bond vectors (internal_coordinates.py:210-221)
        const V3 b1 = rj - ri, b2 = rk - rj, b3 = rl - rk;
        const V3 u1 = rsqrtf(dot(b1, b1)) * b1, u2 = rsqrtf(dot(b2, b2)) * b2, u3 = rsqrtf(dot(b3, b3)) * b3;
        const V3 n1 = cross(u1, u2), n2 = cross(u2, u3), m1 = cross(n1, u2);
        const float phi = atan2f(-dot(m1, n2), dot(n1, n2));
        float du;  // dU/dphi
        if (KIND == 4 || KIND == 8) {
          const float k = params[t], x0 = params[n_terms + t];
          const float ps = KIND == 8 ? (phi < 0.f ? phi + 6.283185307179586f : phi) - 3.141592653589793f : phi;
          u = k * (ps - x0) * (ps - x0);
          du = 2.0f * k * (ps - x0);
        } else {
          const int n_deg = (n_params - 1) / 2;
          du = 0.f;
          u = params[t];
          for (int k = 1; k <= n_deg; ++k) {
            float sn, cs;
            sincosf((float)k * phi, &sn, &cs);
            const float k1 = params[(int64_t)k * n_terms + t], k2 = params[(int64_t)(n_deg + k) * n_terms + t];
            u += k1 * sn + k2 * cs;
            du += (float)k * (k1 * cs - k2 * sn);
          }
        }
        // dphi/dr with F = ri - rj, G = rj - rk, H = rl - rk, A = F x G, B = H x G
        const V3 Fv = ri - rj, Gv = rj - rk, Hv = rl - rk;
        const V3 A = cross(Fv, Gv), B = cross(Hv, Gv);
        const float nG = sqrtf(dot(Gv, Gv)), iA = 1.0f / dot(A, A), iB = 1.0f / dot(B, B);
        const float fg = dot(Fv, Gv) / nG, hg = dot(Hv, Gv) / nG;
        gs[0] = (-du * nG * iA) * A;
        gs[3] = (du * nG * iB) * B;
        const V3 sv = (du * fg * iA) * A - (du * hg * iB) * B;
        gs[1] = sv - gs[0];
        gs[2] = -1.0f * (sv + gs[3]);
      }
      V3 mine = gs[0];
#pragma unroll
      for (int o = 1; o < ORDER; ++o)
        if (slot == o) mine = gs[o];
      g = g + mine;
      if (slot == 0) e += u;  // every term has exactly one slot-0 entry
    }
    if (WAVE_PER_BEAD) {
      g.x = fmd_wave_sum(g.x);
      g.y = fmd_wave_sum(g.y);
      g.z = fmd_wave_sum(g.z);
      e = fmd_wave_sum(e);
    }
    e_wave += e;
    if ((!WAVE_PER_BEAD || lane == 0) && e1 > e0) {
      float* f = forces + 3 * (a0 + b);
      f[0] -= g.x;
      f[1] -= g.y;
      f[2] -= g.z;
    }
  }
  if (!WAVE_PER_BEAD) e_wave = fmd_wave_sum(e_wave);
  if (lane == 0) esum[wv] = e_wave;
  if (bad) atomicOr(&status[3], 1);
  __syncthreads();
  if (tid == 0) {
    float tot = 0.f;
    for (int k = 0; k < 16; ++k) tot += esum[k];
    energy[m] += tot;
  }
}

// Pair kinds (bonds, repulsion) in the partner-list form: the entries of bead b are (partner, parameters) records
// -- 4 + 4 P bytes instead of 4 + 16 + 4 P -- because a pair term's gradient on b depends only on r = pos[partner] -
// pos[b]; bit 31 of the partner word marks the entry that owns the term's energy (slot 0).
template <int KIND>  // 0 bonds, 2 repulsion
__global__ __launch_bounds__(1024) void prior_pair_kernel(const float* __restrict__ pos,
                                                          const int64_t* __restrict__ mol_ptr,
                                                          const int64_t* __restrict__ bead_ptr,
                                                          const uint32_t* __restrict__ partner,
                                                          const float* __restrict__ pparams, int64_t n_entries,
                                                          float* __restrict__ energy, float* __restrict__ forces,
                                                          int32_t* __restrict__ status, int max_beads) {
  extern __shared__ __attribute__((aligned(16))) float sp[];
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t a0 = mol_ptr[m];
  int n = (int)(mol_ptr[m + 1] - a0);
  if (n > max_beads) {  // wrong bound from the caller: stay inside the LDS sized for it, flag it
    if (threadIdx.x == 0) atomicOr(&status[3], 2);
    n = max_beads;
  }
  __shared__ float esum[16];
  for (int i = tid; i < 3 * n; i += 1024) sp[i] = pos[3 * a0 + i];
  __syncthreads();
  float e_wave = 0.f;
  bool bad = false;
  for (int b = wv; b < n; b += 16) {
    const int64_t e0 = bead_ptr[a0 + b], e1 = bead_ptr[a0 + b + 1];
    const V3 pb = {sp[3 * b], sp[3 * b + 1], sp[3 * b + 2]};
    V3 g = {0.f, 0.f, 0.f};
    float e = 0.f;
    for (int64_t q = e0 + lane; q < e1; q += 64) {
      const uint32_t w = partner[q];
      int p = (int)(w & 0x7fffffffu);
      bad |= p >= n;
      p = min(p, n - 1);
      const V3 r = V3{sp[3 * p], sp[3 * p + 1], sp[3 * p + 2]} - pb;
      const float d = sqrtf(dot(r, r));
      float u, de;
      if (KIND == 0) {
        const float k = pparams[q], x0 = pparams[n_entries + q];
        u = k * (d - x0) * (d - x0);
        de = 2.0f * k * (d - x0);
      } else {
        const float sg = pparams[q] / d, s2 = sg * sg;
        u = s2 * s2 * s2;
        de = -6.0f * u / d;
      }
      g = g + (-de / d) * r;  // dU/d(pos b) = -(dU/dd) r / d
      if (w >> 31) e += u;
    }
    g.x = fmd_wave_sum(g.x);
    g.y = fmd_wave_sum(g.y);
    g.z = fmd_wave_sum(g.z);
    e_wave += fmd_wave_sum(e);
    if (lane == 0 && e1 > e0) {
      float* f = forces + 3 * (a0 + b);
      f[0] -= g.x;
      f[1] -= g.y;
      f[2] -= g.z;
    }
  }
  if (lane == 0) esum[wv] = e_wave;
  if (bad) atomicOr(&status[3], 1);
  __syncthreads();
  if (tid == 0) {
    float tot = 0.f;
    for (int k = 0; k < 16; ++k) tot += esum[k];
    energy[m] += tot;
  }
}

}  // namespace

extern "C" int fmd_prior_pairs_energy_forces(const float* pos, const int64_t* mol_ptr, int64_t n_mol,
                                             int64_t max_beads_per_mol, int kind, const int64_t* bead_ptr,
                                             const uint32_t* partner, const float* pair_params,
                                             int64_t n_entries, float* energy, float* forces, int32_t* status,
                                             void* stream) {
  if (n_mol < 0 || n_entries < 0 || max_beads_per_mol < 1 || (kind != 0 && kind != 2)) return FMD_EINVAL;
  if (n_mol == 0 || n_entries == 0) return FMD_OK;
  if (!pos || !mol_ptr || !bead_ptr || !partner || !pair_params || !energy || !forces || !status) return FMD_EINVAL;
  const size_t lds = (size_t)3 * max_beads_per_mol * sizeof(float);
  if (lds > 160 * 1024 - 256) return FMD_EUNSUPPORTED;
  hipStream_t s = fmd_stream(stream);
  if (kind == 0) {
    if (hipFuncSetAttribute((const void*)prior_pair_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return FMD_ELAUNCH;
    prior_pair_kernel<0><<<(unsigned)n_mol, 1024, lds, s>>>(pos, mol_ptr, bead_ptr, partner, pair_params, n_entries,
                                                            energy, forces, status, (int)max_beads_per_mol);
  } else {
    if (hipFuncSetAttribute((const void*)prior_pair_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return FMD_ELAUNCH;
    prior_pair_kernel<2><<<(unsigned)n_mol, 1024, lds, s>>>(pos, mol_ptr, bead_ptr, partner, pair_params, n_entries,
                                                            energy, forces, status, (int)max_beads_per_mol);
  }
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}

extern "C" int fmd_prior_energy_forces(const float* pos, const int64_t* mol_ptr, int64_t n_mol,
                                       int64_t max_beads_per_mol, int kind, const int64_t* mapping,
                                       const float* params, int n_params, const int64_t* bead_ptr,
                                       const int32_t* entries, int64_t n_terms, float* energy, float* forces,
                                       int32_t* status, void* stream) {
  if (n_mol < 0 || n_terms < 0 || max_beads_per_mol < 1 || kind < 0 || kind > 8) return FMD_EINVAL;
  if (n_mol == 0 || n_terms == 0) return FMD_OK;
  if (n_terms >= ((int64_t)1 << 29)) return FMD_EUNSUPPORTED;  // entries pack (term << 2 | slot) in 32 bits
  if (!pos || !mol_ptr || !mapping || !params || !bead_ptr || !entries || !energy || !forces || !status)
    return FMD_EINVAL;
  const int need = kind == 2 ? 1 : (kind == 3 ? 3 : (kind == 6 ? 6 : 2));  // rows of params a kind reads at least
  if (n_params < need || (kind == 3 && (n_params - 1) % 2 != 0) || (kind == 6 && n_params != 6)) return FMD_EINVAL;
  const size_t lds = (size_t)3 * max_beads_per_mol * sizeof(float);
  if (lds > 160 * 1024 - 256) return FMD_EUNSUPPORTED;
  hipStream_t s = fmd_stream(stream);
#define FMD_PRIOR(K, W)                                                                                    \
  {                                                                                                        \
    if (hipFuncSetAttribute((const void*)prior_kernel<K, W>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                            (int)lds) != hipSuccess)                                                       \
      return FMD_ELAUNCH;                                                                                  \
    prior_kernel<K, W><<<(unsigned)n_mol, 1024, lds, s>>>(pos, mol_ptr, mapping, params, n_params, bead_ptr, \
                                                          entries, n_terms, energy, forces, status,        \
                                                          (int)max_beads_per_mol);                        \
  }
  // bonded terms: a handful of entries per bead -> lane per bead; pair lists -> wave per bead
  if (kind == 0) FMD_PRIOR(0, false)
  else if (kind == 1) FMD_PRIOR(1, false)
  else if (kind == 2) FMD_PRIOR(2, true)
  else if (kind == 3) FMD_PRIOR(3, false)
  else if (kind == 4) FMD_PRIOR(4, false)
  else if (kind == 5) FMD_PRIOR(5, false)
  else if (kind == 6) FMD_PRIOR(6, false)
  else if (kind == 7) FMD_PRIOR(7, false)
  else FMD_PRIOR(8, false)
#undef FMD_PRIOR
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}
