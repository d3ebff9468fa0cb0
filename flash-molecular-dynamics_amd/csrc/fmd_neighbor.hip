// A1: radius graph inside each molecule (replaces torch_cluster.radius_graph as called from the
// reference's neighbor_list/torch_impl.py:216-224).  One wave per centre bead: the wave sweeps the
// molecule's beads 64 at a time from LDS, the strict test d2 < rc2 is made with unfused fp32
// (contract(off): separately rounded products and sums) so the edge set is bit-identical to the CPU oracle, and hits are compacted
// with a ballot so rows come out sorted by neighbour index.
#include "fmd_common.h"

namespace {

constexpr int NB_WAVES = 4;           // centres per workgroup
constexpr int NB_TILE = 1024;         // beads staged in LDS per sweep step

__device__ __forceinline__ bool within(float xi, float yi, float zi, float xj, float yj, float zj,
                                       float rc2) {
  // plain operators under contract(off): the __fmul_rn / __fadd_rn intrinsics are header functions whose * and + keep
  // the translation unit's default (contract = fast) and were fused into v_fma_f32 after inlining
#pragma clang fp contract(off)
  float dx = xj - xi, dy = yj - yi, dz = zj - zi;
  float d2 = (dx * dx + dy * dy) + dz * dz;
  return d2 < rc2;
}

// grid.x = chunk of NB_WAVES centres inside molecule blockIdx.y
template <bool FILL>
__global__ __launch_bounds__(NB_WAVES * 64) void radius_graph_kernel(
    const float* __restrict__ pos, const int64_t* __restrict__ mol_ptr, float rc2,
    int32_t* __restrict__ deg, const int64_t* __restrict__ row_ptr, int64_t* __restrict__ edge_src,
    int64_t* __restrict__ edge_dst) {
  __shared__ float sx[NB_TILE], sy[NB_TILE], sz[NB_TILE];
  const int mol = blockIdx.y;
  const int64_t a = mol_ptr[mol], b = mol_ptr[mol + 1];
  const int n = (int)(b - a);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ci = blockIdx.x * NB_WAVES + wave;  // centre index inside the molecule
  if (blockIdx.x * NB_WAVES >= n) return;       // whole workgroup beyond this molecule
  const bool active = ci < n;
  float xi = 0.f, yi = 0.f, zi = 0.f;
  if (active) {
    xi = pos[3 * (a + ci)];
    yi = pos[3 * (a + ci) + 1];
    zi = pos[3 * (a + ci) + 2];
  }
  int count = 0;
  int64_t base = 0;
  if (FILL && active) base = row_ptr[a + ci];
  for (int t0 = 0; t0 < n; t0 += NB_TILE) {
    const int tn = min(NB_TILE, n - t0);
    __syncthreads();
    for (int t = threadIdx.x; t < tn; t += NB_WAVES * 64) {
      sx[t] = pos[3 * (a + t0 + t)];
      sy[t] = pos[3 * (a + t0 + t) + 1];
      sz[t] = pos[3 * (a + t0 + t) + 2];
    }
    __syncthreads();
    if (active) {
      for (int j0 = 0; j0 < tn; j0 += 64) {
        const int j = j0 + lane;
        bool hit = false;
        if (j < tn) hit = within(xi, yi, zi, sx[j], sy[j], sz[j], rc2) && (t0 + j != ci);
        const unsigned long long m = __ballot(hit);
        if (FILL && hit) {
          const int rank = __popcll(m & ((1ull << lane) - 1ull));
          edge_src[base + count + rank] = a + ci;
          edge_dst[base + count + rank] = a + t0 + j;
        }
        count += __popcll(m);
      }
    }
  }
  if (!FILL && active && lane == 0) deg[a + ci] = count;
}

// single-workgroup exclusive scan int32 -> int64 (N is tens of thousands: a few microseconds)
__global__ __launch_bounds__(1024) void scan_deg_kernel(const int32_t* __restrict__ deg, int64_t n,
                                                        int64_t* __restrict__ row_ptr) {
  __shared__ int64_t wsum[16];
  __shared__ int64_t carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t i0 = 0; i0 < n; i0 += 1024) {
    const int64_t i = i0 + threadIdx.x;
    int64_t v = i < n ? (int64_t)deg[i] : 0;
    int64_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int64_t t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int64_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int64_t carry = carry_s;
    if (i < n) row_ptr[i] = carry + woff + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) row_ptr[n] = carry_s;
}

}  // namespace

// beads per molecule are only known on the device; the caller passes an upper bound
// (max_beads_per_mol <= 0: assume the worst case, one molecule holding all but n_mol-1 beads).
static inline int64_t per_mol_bound(int64_t n_mol, int64_t n_nodes, int64_t max_beads_per_mol) {
  int64_t worst = n_nodes - (n_mol - 1);
  if (max_beads_per_mol > 0 && max_beads_per_mol < worst) return max_beads_per_mol;
  return worst > 0 ? worst : 1;
}

extern "C" int fmd_radius_graph_count(const float* pos, const int64_t* mol_ptr, int64_t n_mol,
                                      int64_t n_nodes, int64_t max_beads_per_mol, float cutoff,
                                      int32_t* deg, int64_t* row_ptr, void* stream) {
  if (!pos || !mol_ptr || !deg || !row_ptr || n_mol < 0 || n_nodes < 0) return FMD_EINVAL;
  hipStream_t s = fmd_stream(stream);
  if (n_nodes > 0 && n_mol > 0) {
    const int64_t per = per_mol_bound(n_mol, n_nodes, max_beads_per_mol);
    dim3 grid((unsigned)((per + NB_WAVES - 1) / NB_WAVES), (unsigned)n_mol);
    radius_graph_kernel<false><<<grid, NB_WAVES * 64, 0, s>>>(pos, mol_ptr, cutoff * cutoff, deg,
                                                            nullptr, nullptr, nullptr);
    FMD_CHECK_LAUNCH();
  }
  scan_deg_kernel<<<1, 1024, 0, s>>>(deg, n_nodes, row_ptr);
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}

extern "C" int fmd_radius_graph_fill(const float* pos, const int64_t* mol_ptr, int64_t n_mol,
                                     int64_t n_nodes, int64_t max_beads_per_mol, float cutoff,
                                     const int64_t* row_ptr,
                                     int64_t* edge_src, int64_t* edge_dst, void* stream) {
  if (!pos || !mol_ptr || !row_ptr || n_mol < 0 || n_nodes < 0) return FMD_EINVAL;
  if (n_nodes == 0 || n_mol == 0) return FMD_OK;
  if (!edge_src || !edge_dst) return FMD_EINVAL;
  const int64_t per = per_mol_bound(n_mol, n_nodes, max_beads_per_mol);
  dim3 grid((unsigned)((per + NB_WAVES - 1) / NB_WAVES), (unsigned)n_mol);
  radius_graph_kernel<true><<<grid, NB_WAVES * 64, 0, fmd_stream(stream)>>>(
      pos, mol_ptr, cutoff * cutoff, nullptr, row_ptr, edge_src, edge_dst);
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}
