// Row N1: Langevin BAOAB elementwise halves around the force call (reference
// simulation/langevin.py:136-170), the overdamped step (:315-434) and row (e): in-place parallel-tempering swap of accepted replica
// pairs (reference simulation/parallel_tempering.py:415-481).  Pure HBM streaming, 12 B/bead/array.
#include "fmd_common.h"

namespace {

__global__ __launch_bounds__(256) void langevin_pre_kernel(float* __restrict__ x, float* __restrict__ v,
                                                           const float* __restrict__ f,
                                                           const float* __restrict__ masses,
                                                           const float* __restrict__ beta_node,
                                                           const float* __restrict__ noise,
                                                           int64_t n3, float dt, float vscale,
                                                           float noisescale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n3) return;
  const int64_t n = i / 3;
  const float m = masses[n];
  float vn = v[i] + 0.5f * dt * f[i] / m;             // B  (langevin.py:138)
  float xn = x[i] + vn * dt * 0.5f;                    // A  (:142)
  const float bm = sqrtf(1.0f / beta_node[n] / m);     // beta_mass_ratio (:211-215)
  vn = vn * vscale + noisescale * (bm * noise[i]);     // O  (:146-148)
  xn = xn + vn * dt * 0.5f;                            // A  (:156)
  x[i] = xn;
  v[i] = vn;
}

__global__ __launch_bounds__(256) void langevin_post_kernel(float* __restrict__ v,
                                                            const float* __restrict__ f,
                                                            const float* __restrict__ masses,
                                                            int64_t n3, float dt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n3) return;
  v[i] = v[i] + 0.5f * dt * f[i] / masses[i / 3];      // B  (:169)
}

// velocity Verlet first half (reference simulation/velocity_verlet.py:38-41): v += .5 dt f/m ; x += dt v
__global__ __launch_bounds__(256) void verlet_pre_kernel(float* __restrict__ x, float* __restrict__ v,
                                                         const float* __restrict__ f,
                                                         const float* __restrict__ masses, int64_t n3,
                                                         float dt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n3) return;
  const float vh = v[i] + 0.5f * dt * f[i] / masses[i / 3];
  x[i] = x[i] + dt * vh;
  v[i] = vh;
}

// overdamped Langevin / Brownian step (reference simulation/langevin.py:370-372,417-421):
// D = 1 / beta / friction, dtau = D dt, x <- x + f dtau + sqrt(2 dtau) noise
__global__ __launch_bounds__(256) void overdamped_kernel(float* __restrict__ x, const float* __restrict__ f,
                                                         const float* __restrict__ beta_node,
                                                         const float* __restrict__ noise, int64_t n3, float dt,
                                                         float friction) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n3) return;
  const float dtau = (1.0f / beta_node[i / 3] / friction) * dt;
  x[i] = x[i] + f[i] * dtau + sqrtf(2.0f * dtau) * noise[i];
}

// one workgroup per proposed pair; untouched when the pair was rejected
__global__ __launch_bounds__(256) void pt_swap_kernel(float* __restrict__ x, float* __restrict__ v,
                                                      const int64_t* __restrict__ mol_a,
                                                      const int64_t* __restrict__ mol_b,
                                                      const uint8_t* __restrict__ accept,
                                                      const float* __restrict__ beta_mol,
                                                      int64_t n_atoms) {
  const int64_t p = blockIdx.x;
  if (!accept[p]) return;
  const int64_t a = mol_a[p], b = mol_b[p];
  const float ba = beta_mol[a], bb = beta_mol[b];
  const float s_ab = sqrtf(ba / bb), s_ba = sqrtf(bb / ba);  // parallel_tempering.py:467-468
  const int64_t n3 = n_atoms * 3;
  for (int64_t i = threadIdx.x; i < n3; i += blockDim.x) {
    const int64_t ia = a * n3 + i, ib = b * n3 + i;
    const float xa = x[ia], xb = x[ib], va = v[ia], vb = v[ib];
    x[ia] = xb;
    x[ib] = xa;
    v[ia] = vb * s_ab;  // :470-472
    v[ib] = va * s_ba;  // :473-475
  }
}

}  // namespace

extern "C" int fmd_langevin_pre(float* x, float* v, const float* f, const float* masses,
                                const float* beta_node, const float* noise, int64_t n_nodes, float dt,
                                float vscale, float noisescale, void* stream) {
  if (n_nodes < 0) return FMD_EINVAL;
  if (n_nodes == 0) return FMD_OK;
  if (!x || !v || !f || !masses || !beta_node || !noise) return FMD_EINVAL;
  const int64_t n3 = n_nodes * 3;
  langevin_pre_kernel<<<(unsigned)((n3 + 255) / 256), 256, 0, fmd_stream(stream)>>>(
      x, v, f, masses, beta_node, noise, n3, dt, vscale, noisescale);
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}

extern "C" int fmd_langevin_post(float* v, const float* f, const float* masses, int64_t n_nodes,
                                 float dt, void* stream) {
  if (n_nodes < 0) return FMD_EINVAL;
  if (n_nodes == 0) return FMD_OK;
  if (!v || !f || !masses) return FMD_EINVAL;
  const int64_t n3 = n_nodes * 3;
  langevin_post_kernel<<<(unsigned)((n3 + 255) / 256), 256, 0, fmd_stream(stream)>>>(v, f, masses, n3,
                                                                                  dt);
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}

extern "C" int fmd_verlet_pre(float* x, float* v, const float* f, const float* masses, int64_t n_nodes,
                              float dt, void* stream) {
  if (n_nodes < 0) return FMD_EINVAL;
  if (n_nodes == 0) return FMD_OK;
  if (!x || !v || !f || !masses) return FMD_EINVAL;
  const int64_t n3 = n_nodes * 3;
  verlet_pre_kernel<<<(unsigned)((n3 + 255) / 256), 256, 0, fmd_stream(stream)>>>(x, v, f, masses, n3, dt);
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}

extern "C" int fmd_overdamped_step(float* x, const float* f, const float* beta_node, const float* noise,
                                   int64_t n_nodes, float dt, float friction, void* stream) {
  if (n_nodes < 0 || !(friction > 0.f)) return FMD_EINVAL;
  if (n_nodes == 0) return FMD_OK;
  if (!x || !f || !beta_node || !noise) return FMD_EINVAL;
  const int64_t n3 = n_nodes * 3;
  overdamped_kernel<<<(unsigned)((n3 + 255) / 256), 256, 0, fmd_stream(stream)>>>(x, f, beta_node, noise, n3, dt,
                                                                                  friction);
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}

extern "C" int fmd_pt_swap_local(float* x, float* v, const int64_t* mol_a, const int64_t* mol_b,
                                 const uint8_t* accept, const float* beta_mol, int64_t n_pairs,
                                 int64_t n_atoms, void* stream) {
  if (n_pairs < 0 || n_atoms < 0) return FMD_EINVAL;
  if (n_pairs == 0 || n_atoms == 0) return FMD_OK;
  if (!x || !v || !mol_a || !mol_b || !accept || !beta_mol) return FMD_EINVAL;
  pt_swap_kernel<<<(unsigned)n_pairs, 256, 0, fmd_stream(stream)>>>(x, v, mol_a, mol_b, accept,
                                                                   beta_mol, n_atoms);
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}
