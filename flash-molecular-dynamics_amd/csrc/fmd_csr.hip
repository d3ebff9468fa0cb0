// A3: COO -> CSR bucket sort (replaces build_csr_index / build_src_csr_index,
// reference kernels/csr_kernels.py:88-169, 229-294).
//
// Integer work, HBM/atomic-bound.  Output is the STABLE counting sort:
//   1. histogram of keys (int32 atomics) + "already sorted?" flag
//   2. exclusive scan -> ptr (int64)
//   3. fill: if the keys are already non-decreasing (the src-keyed list of a centre-sorted
//      neighbour list) perm is the identity; otherwise atomic cursors scatter the edge ids ...
//   4. ... and each segment is put in ascending edge-id order by a rank sort (one wave per
//      segment), which makes the result independent of the atomic arrival order.
#include "fmd_common.h"

namespace {

struct CsrWs {
  int32_t* counts;   // [N]
  int32_t* cursor;   // [N]
  int32_t* flags;    // [4]: [0] = 1 when keys are not sorted, [1] = 1 when a key was out of range
  int64_t* tmp;      // [E] unsorted fill
};

__host__ __device__ inline int64_t align16(int64_t x) { return (x + 15) & ~(int64_t)15; }

__global__ void csr_hist_kernel(const int64_t* __restrict__ keys, int64_t n_edges, int64_t n_nodes,
                                int32_t* __restrict__ counts, int32_t* __restrict__ flags) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  const int64_t k = keys[e];
  if ((uint64_t)k >= (uint64_t)n_nodes) {  // out-of-range key: never touch memory with it
    flags[1] = 1;
    return;
  }
  atomicAdd(&counts[k], 1);
  if (e > 0 && keys[e - 1] > k) flags[0] = 1;  // benign race: every writer stores 1
}

__global__ __launch_bounds__(1024) void csr_scan_kernel(const int32_t* __restrict__ counts, int64_t n,
                                                        int64_t* __restrict__ ptr,
                                                        int32_t* __restrict__ cursor) {
  __shared__ int64_t wsum[16];
  __shared__ int64_t carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t i0 = 0; i0 < n; i0 += 1024) {
    const int64_t i = i0 + threadIdx.x;
    const int64_t v = i < n ? (int64_t)counts[i] : 0;
    int64_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int64_t t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int64_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int64_t carry = carry_s;
    if (i < n) {
      ptr[i] = carry + woff + incl - v;
      cursor[i] = 0;
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) ptr[n] = carry_s;
}

__global__ void csr_fill_kernel(const int64_t* __restrict__ keys, int64_t n_edges, int64_t n_nodes,
                                const int64_t* __restrict__ ptr, int32_t* __restrict__ cursor,
                                const int32_t* __restrict__ flags, int64_t* __restrict__ perm,
                                int64_t* __restrict__ tmp) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_edges) return;
  if (flags[0] == 0) {  // sorted input: stable sort is the identity
    perm[e] = e;
    return;
  }
  const int64_t k = keys[e];
  if ((uint64_t)k >= (uint64_t)n_nodes) return;
  const int32_t slot = atomicAdd(&cursor[k], 1);
  tmp[ptr[k] + slot] = e;
}

// one wave per segment: rank of an element = number of smaller edge ids in the segment
__global__ __launch_bounds__(256) void csr_segsort_kernel(const int64_t* __restrict__ ptr,
                                                          int64_t n_nodes,
                                                          const int32_t* __restrict__ flags,
                                                          const int64_t* __restrict__ tmp,
                                                          int64_t* __restrict__ perm) {
  if (flags[0] == 0) return;
  const int lane = threadIdx.x & 63;
  const int64_t node = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (node >= n_nodes) return;
  const int64_t a = ptr[node], b = ptr[node + 1];
  const int64_t len = b - a;
  for (int64_t i0 = 0; i0 < len; i0 += 64) {
    const int64_t i = i0 + lane;
    const int64_t mine = i < len ? tmp[a + i] : 0;
    int64_t rank = 0;
    for (int64_t j0 = 0; j0 < len; j0 += 64) {
      const int64_t j = j0 + lane;
      const int64_t other = j < len ? tmp[a + j] : INT64_MAX;
      const int cnt = (int)min((int64_t)64, len - j0);
      for (int t = 0; t < cnt; ++t) {
        const int64_t o = __shfl(other, t, 64);
        rank += (o < mine) ? 1 : 0;
      }
    }
    if (i < len) perm[a + rank] = mine;
  }
}

inline CsrWs carve(void* workspace, int64_t n_nodes, int64_t n_edges) {
  char* p = reinterpret_cast<char*>(workspace);
  CsrWs w;
  w.flags = reinterpret_cast<int32_t*>(p);
  p += 16;
  w.counts = reinterpret_cast<int32_t*>(p);
  p += align16(4 * n_nodes);
  w.cursor = reinterpret_cast<int32_t*>(p);
  p += align16(4 * n_nodes);
  w.tmp = reinterpret_cast<int64_t*>(p);
  return w;
}

}  // namespace

extern "C" int64_t fmd_build_csr_workspace_bytes(int64_t n_nodes, int64_t n_edges) {
  if (n_nodes < 0 || n_edges < 0) return FMD_EINVAL;
  return 16 + 2 * align16(4 * n_nodes) + align16(8 * n_edges) + 16;
}

extern "C" int fmd_build_csr(const int64_t* keys, int64_t n_edges, int64_t n_nodes, int64_t* ptr,
                             int64_t* perm, void* workspace, void* stream) {
  if (n_edges < 0 || n_nodes < 0 || !ptr) return FMD_EINVAL;
  hipStream_t s = fmd_stream(stream);
  if (n_edges == 0) {  // reference: zeros / empty (kernels/csr_kernels.py:132-137)
    if (hipMemsetAsync(ptr, 0, sizeof(int64_t) * (n_nodes + 1), s) != hipSuccess) return FMD_ELAUNCH;
    return FMD_OK;
  }
  if (!keys || !perm || !workspace || n_nodes == 0) return FMD_EINVAL;
  if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return FMD_EINVAL;
  CsrWs w = carve(workspace, n_nodes, n_edges);
  // flags + counts in one memset (contiguous at the start of the workspace)
  if (hipMemsetAsync(workspace, 0, 16 + align16(4 * n_nodes), s) != hipSuccess) return FMD_ELAUNCH;
  const unsigned eb = (unsigned)((n_edges + 255) / 256);
  csr_hist_kernel<<<eb, 256, 0, s>>>(keys, n_edges, n_nodes, w.counts, w.flags);
  FMD_CHECK_LAUNCH();
  csr_scan_kernel<<<1, 1024, 0, s>>>(w.counts, n_nodes, ptr, w.cursor);
  FMD_CHECK_LAUNCH();
  csr_fill_kernel<<<eb, 256, 0, s>>>(keys, n_edges, n_nodes, ptr, w.cursor, w.flags, perm, w.tmp);
  FMD_CHECK_LAUNCH();
  csr_segsort_kernel<<<(unsigned)((n_nodes + 3) / 4), 256, 0, s>>>(ptr, n_nodes, w.flags, w.tmp, perm);
  FMD_CHECK_LAUNCH();
  return FMD_OK;
}
