/* flashmd_hip.h -- C ABI of the MI355X (gfx950) CGSchNet force-evaluation path.
 *
 * Drop-in boundary for the hot path of UNITES-Lab/flash-molecular-dynamics ("flashmd"): every
 * entry point replaces one Triton-kernel host wrapper (or third-party native call) of the
 * reference; the reference-side symbol each one stands in for is cited as
 * <file>:<line> relative to /root/reference/src/flashmd.
 *
 * Conventions (all entry points):
 *   - plain pointers to DEVICE memory, sizes as int64_t, parameters as float/int; no torch types;
 *   - tensors are contiguous row-major; index tensors are int64 (the reference's dtype);
 *   - nothing is allocated inside: the caller passes outputs and (where named) a workspace;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue;
 *   - return 0 on success, a negative FMD_E* code otherwise (fmd_strerror gives the text);
 *     no exceptions, and no state a caller can observe: the library holds no buffers, reads no
 *     environment variables and keeps nothing between calls except a per-device "kernel attributes are
 *     set" flag (the dynamic-LDS ceiling of its own kernels, raised once, idempotent).
 */
#ifndef FLASHMD_HIP_H
#define FLASHMD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMD_OK 0
#define FMD_EINVAL (-1)     /* bad argument (null pointer, negative size, unsupported shape) */
#define FMD_ELAUNCH (-2)    /* HIP reported a launch error */
#define FMD_ECAPACITY (-3)  /* a fixed-capacity buffer was too small (reported by the *_status call) */
#define FMD_EUNSUPPORTED (-4)
#define FMD_ENONFINITE (-5) /* a force contribution was Inf / NaN / >= 2^19 (reported in the sticky status word) */

const char* fmd_strerror(int code);
/* ABI version of this library (bumped on any change a caller can observe).
 *   3: status word 3 of fmd_schnet_energy_forces is sticky and must be zeroed by the caller; FMD_ENONFINITE;
 *      fmd_schnet_workspace_bytes no longer depends on the current device (same signatures as 2).
 *   4: fmd_schnet_desc ends in `flags`.
 *   5: the only flag is FMD_FLAG_NO_FILTER_STASH (the opt-in tabulated-filter prototype of ABI 4, bit 2, left the
 *      product: measured at 0.63x of the MFMA path, DESIGN.md section 3 "Round 4"); desc.cutoff must be finite and
 *      > 0, desc.gamma finite, else FMD_EINVAL; num_rbf up to 128; fmd_overdamped_step; the packed weight layout
 *      changed (fmd_schnet_pack_bytes says how much).
 *   6: FMD_FLAG_NO_MOLECULE_GROUPS; fmd_schnet_workspace_bytes grew by the room of a partial group for small molecules;
 *      status word 1 (pair-list entries needed) counts a GROUP of molecules where the engine groups them. */
int fmd_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * A1  neighbour list: replaces torch_cluster.radius_graph(pos, rc, batch, flow="target_to_source",
 *     loop=False) as called from neighbor_list/torch_impl.py:216-224.
 *     Edges = ordered pairs (i,j), i!=j, same molecule, ((dx*dx+dy*dy)+dz*dz) < rc*rc (strict,
 *     unfused fp32), sorted by centre i then neighbour j.  Two passes so the caller can size E:
 *       fmd_radius_graph_count : deg[N] (int32) and row_ptr[N+1] (int64, exclusive scan of deg)
 *       fmd_radius_graph_fill  : edge_src[E]=i, edge_dst[E]=j (int64) using row_ptr
 *     mol_ptr[B+1] (int64): first bead of each molecule (data.ptr of the collated batch).
 *     max_beads_per_mol: upper bound on beads in one molecule (sizes the grid; <= 0 = unknown).
 * ------------------------------------------------------------------------------------------- */
int fmd_radius_graph_count(const float* pos, const int64_t* mol_ptr, int64_t n_mol, int64_t n_nodes,
                           int64_t max_beads_per_mol, float cutoff, int32_t* deg, int64_t* row_ptr,
                           void* stream);
int fmd_radius_graph_fill(const float* pos, const int64_t* mol_ptr, int64_t n_mol, int64_t n_nodes,
                          int64_t max_beads_per_mol, float cutoff, const int64_t* row_ptr,
                          int64_t* edge_src, int64_t* edge_dst, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A2  fused distance + Gaussian RBF + cosine cutoff.
 *     fwd replaces fused_distance_gaussian_rbf_cutoff   (kernels/cfconv_kernels.py:1578-1656)
 *     bwd replaces FusedDistanceGaussianRBFCutoffFunction.backward (:1680-1735; that code raises
 *         NameError in the reference -- the formula at :1696-1733 is the spec)
 *     d[e] = |pos[dst]-pos[src]|, rbf[e,k] = exp(gamma (d-c_k)^2) * C(d), C = .5(cos(pi d/rc)+1)[d<rc]
 *     bwd: grad_pos (zero-filled by the callee) += d(d, rbf)/d(pos) . (grad_d, grad_rbf);
 *          grad_d / grad_rbf may be NULL (treated as zero).
 * ------------------------------------------------------------------------------------------- */
int fmd_distance_rbf_cutoff_fwd(const float* pos, const int64_t* edge_src, const int64_t* edge_dst,
                                const float* centers, float gamma, float cutoff, int64_t n_edges,
                                int64_t n_rbf, float* dist, float* rbf, void* stream);
int fmd_distance_rbf_cutoff_bwd(const float* pos, const int64_t* edge_src, const int64_t* edge_dst,
                                const float* centers, float gamma, float cutoff, int64_t n_edges,
                                int64_t n_rbf, int64_t n_nodes, const float* dist,
                                const float* grad_dist, const float* grad_rbf, float* grad_pos,
                                void* stream);
/* The same gradient without float atomics, for callers that hold both CSR indices of the edge list (the reference's
 * SchNet.forward builds them anyway, models/schnet.py:309-334): an edge pass stores the per-edge vector
 * gd (r_dst - r_src) / d into edge_vec (n_edges x 16 bytes, 16-byte aligned), a node pass sums it over the dst- and
 * src-segments (ptr/perm as produced by fmd_build_csr).  Deterministic; grad_pos is fully overwritten. */
int fmd_distance_rbf_cutoff_bwd_csr(const float* pos, const int64_t* edge_src, const int64_t* edge_dst,
                                    const float* centers, float gamma, float cutoff, int64_t n_edges,
                                    int64_t n_rbf, int64_t n_nodes, const float* dist, const float* grad_dist,
                                    const float* grad_rbf, const int64_t* dst_ptr, const int64_t* dst_perm,
                                    const int64_t* src_ptr, const int64_t* src_perm, void* edge_vec,
                                    float* grad_pos, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A3  COO -> CSR bucket sort.  Replaces build_csr_index (kernels/csr_kernels.py:88-169) and
 *     build_src_csr_index (:229-294): ptr[N+1] = [0, cumsum(bincount(keys))], perm[E] = original
 *     edge ids grouped by key.  Output is the STABLE counting sort (ascending edge id inside a
 *     segment) -- deterministic, and a member of the reference's (atomic-order) output set.
 *     workspace: fmd_build_csr_workspace_bytes(N, E) bytes, 16-byte aligned.
 * ------------------------------------------------------------------------------------------- */
int64_t fmd_build_csr_workspace_bytes(int64_t n_nodes, int64_t n_edges);
int fmd_build_csr(const int64_t* keys, int64_t n_edges, int64_t n_nodes, int64_t* ptr, int64_t* perm,
                  void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A5/A6  CFConv message passing.  filt_is_f16: filter_out / grad_filter_out stored as fp16.
 *   fmd_csr_cfconv_fwd       : fused_csr_cfconv          (kernels/csr_kernels.py:727-810)
 *        out[n,:] = sum_{k in [ptr[n],ptr[n+1])} x[gather[perm[k]],:] * filt[perm[k],:] * C(d[perm[k]])
 *        With (gather=edge_src, ptr/perm = dst-CSR) this is the forward; with (x=grad_out,
 *        gather=edge_dst, ptr/perm = src-CSR) it is fused_src_csr_grad_x (:399-482).
 *   fmd_cfconv_scatter_atomic: fused_cutoff_gather_multiply_scatter (kernels/cfconv_kernels.py:93-170)
 *        out (zero-filled by the callee) [dst[e],:] += x[src[e],:] * filt[e,:] * C(d[e])  (float atomics)
 *   fmd_cfconv_grad_edge     : fused_grad_filter_out     (kernels/cfconv_kernels.py:261-337)
 *        gfilt[e,:] = x[src[e],:] * g[dst[e],:] * C(d[e])            (written in filt's dtype)
 *        and, which the reference drops but its PyTorch path has (models/schnet.py:710-713):
 *        gd[e] = C'(d[e]) * sum_f x[src[e],f] * filt[e,f] * g[dst[e],f]     (gd may be NULL)
 *   fmd_csr_segment_reduce   : csr_segment_reduce        (kernels/csr_kernels.py:559-617)
 * ------------------------------------------------------------------------------------------- */
int fmd_csr_cfconv_fwd(const float* x, const void* filt, int filt_is_f16, const float* dist,
                       const int64_t* gather, const int64_t* ptr, const int64_t* perm, int64_t n_nodes,
                       int64_t n_feat, float cutoff, float* out, void* stream);
int fmd_cfconv_scatter_atomic(const float* x, const void* filt, int filt_is_f16, const float* dist,
                              const int64_t* edge_src, const int64_t* edge_dst, int64_t n_edges,
                              int64_t n_nodes, int64_t n_feat, float cutoff, float* out, void* stream);
int fmd_cfconv_grad_edge(const float* x, const float* grad_out, const void* filt, int filt_is_f16,
                         const float* dist, const int64_t* edge_src, const int64_t* edge_dst,
                         int64_t n_edges, int64_t n_feat, float cutoff, void* grad_filt,
                         float* grad_dist, void* stream);
int fmd_csr_segment_reduce(const float* msg, const int64_t* ptr, const int64_t* perm, int64_t n_nodes,
                           int64_t n_feat, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A7 + node GEMMs (exact fp32 on v_mfma_f32_32x32x2_f32).
 *   C[M,N] = epi( pro(A[M,K]) @ B + bias[N] )
 *     b_is_nk   : 0 -> B is [K,N] row-major (the kernels' layout), 1 -> B is [N,K] (nn.Linear.weight)
 *     prologue  : FMD_PRO_NONE | FMD_PRO_TANH (tanh applied to A)
 *     epilogue  : FMD_EPI_NONE | FMD_EPI_TANH | FMD_EPI_MUL_DTANH (C *= 1 - aux^2, aux[M,N])
 *   fused_tanh_linear  (kernels/cfconv_kernels.py:1846-1898)  = (b_is_nk=0, PRO_TANH, EPI_NONE)
 *   its backward (:1920-1941) grad_x = (1 - tanh_x^2) * (g @ W^T) = (A=g, B=W [K,N] read as [N',K'],
 *                                                                    b_is_nk=1, EPI_MUL_DTANH)
 *   fused_linear_tanh  (:543-593)                              = (b_is_nk=0, PRO_NONE, EPI_TANH)
 *   nn.Linear lin1/lin2 (models/schnet.py:644,719)             = (b_is_nk=1)
 * ------------------------------------------------------------------------------------------- */
#define FMD_PRO_NONE 0
#define FMD_PRO_TANH 1
#define FMD_EPI_NONE 0
#define FMD_EPI_TANH 1
#define FMD_EPI_MUL_DTANH 2
int fmd_gemm_f32(const float* a, const float* b, const float* bias, const float* aux, float* c,
                 int64_t m, int64_t n, int64_t k, int b_is_nk, int prologue, int epilogue, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A4/A9  W16A16 GEMMs (fp16 storage, fp32 accumulate on v_mfma_f32_32x32x16_f16).
 *   fmd_linear_tanh_f16   : fused_linear_tanh_fp16 (kernels/cfconv_kernels.py:723-758)
 *        y_f16 = f16( tanh_c( f16(x) @ w_f16 + b_f16 ) ), tanh_c = (e^{2z}-1)/(e^{2z}+1), z clamped +-10 (:449-454)
 *        x_is_f16: input dtype (fp32 inputs are rounded to fp16 on load, :701)
 *   fmd_linear_f16        : linear_fp16 (:835-892)  y = x_f16 @ w_f16, stored fp32 (out_is_f16=0) or fp16
 *   fmd_tanh_bwd_grad_x_f16 : fused_tanh_backward_grad_x_v2 (:1038-1064)
 *        gx_f32[M,N] = f16( g * (1 - y^2) ) @ w_t_f16[K,N]       (g fp16 or fp32, y fp16)
 *   fmd_matmul_f32_f16_to_f16 : matmul_fp32_fp16_to_fp16_v2 (:1211-1226)  c_f16 = f16(a_f32) @ b_f16
 *   w / w_t / b are [K,N] row-major.
 * ------------------------------------------------------------------------------------------- */
int fmd_linear_tanh_f16(const void* x, int x_is_f16, const void* w, const void* bias, void* y,
                        int64_t m, int64_t n, int64_t k, void* stream);
int fmd_linear_f16(const void* x, const void* w, void* y, int out_is_f16, int64_t m, int64_t n,
                   int64_t k, void* stream);
int fmd_tanh_bwd_grad_x_f16(const void* g, int g_is_f16, const void* y, const void* w_t, float* gx,
                            int64_t m, int64_t n, int64_t k, void* stream);
int fmd_matmul_f32_f16_to_f16(const float* a, const void* b, void* c, int64_t m, int64_t n, int64_t k,
                              void* stream);

/* ---------------------------------------------------------------------------------------------
 * Whole-path engine: energy[B] and forces[N,3] of a batch in one call, no [E,F] tensor in HBM.
 * Replaces SchNet.forward (models/schnet.py:178-369) + GradientsOut.forward
 * (models/gradients.py:226-290) for StandardSchNet models (Gaussian basis, cosine cutoff with
 * lower bound 0, tanh).  The kernels are built for hidden = filters = 128: the caller passes weights zero-padded to
 * that width (narrower layers are exact under zero padding: a padded feature is 0 in every layer; the Python host
 * pads hidden, filters <= 128 by itself) and 2 <= num_rbf <= 128 (65..128: molecules of at most 283 beads, ABI 5).
 *   Weights are packed once by fmd_schnet_pack_bytes / fmd_schnet_pack (device buffer).
 *   precision: FMD_PREC_F32 (default: every operand of the filter-network contractions is carried as two fp16
 *   words, a.b = hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation -- relative error 2e-7,
 *   the same as the fp32 MFMA, at 3/16 of its MFMA cycles.  Operands of bounded magnitude (weights, Gaussians,
 *   C*tanh) carry an UNSCALED low word lo = f16(x - hi) -- fp16 denormals are honoured -- and all three products
 *   share one accumulator; only the unbounded gW operand of the backward u GEMM keeps the scaled form
 *   lo = f16((x - hi) * 2048) with a second accumulator);
 *   FMD_PREC_W16A16 = the reference's W16A16 mode
 *   (models/gptq.py:92-130,266-306): filter and output networks with fp16 operands (weights =
 *   Linear.weight.half(), activations rounded to fp16 where they are stored), fp32 accumulation.
 * ------------------------------------------------------------------------------------------- */
#define FMD_PREC_F32 0        /* fp32 results: fp16 hi/lo split operands on the fp16 MFMA, fp32 accumulate */
#define FMD_PREC_W16A16 1     /* filter and output networks: fp16 operands and stored activations, fp32 accumulate */

typedef struct fmd_schnet_desc {
  int32_t hidden;        /* = num_filters = 128: the kernels' width (narrower models: zero-padded weights, see above) */
  int32_t num_rbf;       /* R (<= 128; beyond 64 the feature-split mode for molecules above 283 beads is not available) */
  int32_t num_blocks;    /* L */
  int32_t num_out;       /* number of output-network linears (2 or 3) */
  int32_t out_width[4];  /* widths after each output linear, last = 1 */
  int32_t embedding_size;
  int32_t precision;
  int32_t n_split;       /* workgroups per molecule in the pair kernels; 0 = auto: floor(CUs / n_mol) in 1..32, one round
                            of workgroups for up to CUs molecules, with the tiles of large molecules handed to the
                            workgroups of small ones so that every workgroup walks about the same number (more than
                            CUs / 2 but fewer than CUs molecules: one workgroup per CU over equal chunks of all tiles);
                            > 0 = fixed static interleave */
  float cutoff;          /* rc of the cosine cutoff and of the neighbour search */
  float gamma;           /* GaussianBasis.coeff (reference models/radial_basis/gaussian.py:74) */
  int32_t flags;         /* 0 or an OR of the FMD_FLAG_* below; any other bit: FMD_EINVAL */
} fmd_schnet_desc;
/* By default (FMD_PREC_F32 only) the forward pair kernel of every block but the first keeps its filter values W = filt(d) C(d) (fp32, one per
 * list entry and feature: n_mol * max_pairs_per_mol * 512 bytes of workspace per such block, of which the entries a
 * batch really has are touched) and the backward kernel of that block reads them back instead of recomputing filter
 * layer 1.  With this flag the workspace stays small and the backward pass recomputes (same results to the last bit:
 * the recomputation runs the same arithmetic). */
#define FMD_FLAG_NO_FILTER_STASH 1
/* Batches of MORE molecules than the device has CUs, all of at most 64 beads (max_beads_per_mol), are evaluated over
 * groups of up to 32 consecutive molecules / 256 beads: one pair list, one aggregate staging and one work item per group
 * instead of per molecule (a 10-bead molecule is 45 pairs, two tiles: all fixed cost).  Pairs never cross molecules; a
 * molecule's sums are accumulated in another order than without groups (fp32 rounding), so with this flag the engine
 * keeps one list per molecule.  The device-side check of max_beads_per_mol then applies to a group's total. */
#define FMD_FLAG_NO_MOLECULE_GROUPS 2

/* Host pointers to the reference's parameter tensors (fp32, nn.Linear [out,in] layout). */
typedef struct fmd_schnet_block_weights {
  const float* lin1_w;            /* [F,H]  CFConv.lin1 (no bias), models/schnet.py:584 */
  const float* f0_w; const float* f0_b; /* [F,R],[F] filter MLP layer 0, models/schnet.py:813-817 */
  const float* f1_w;              /* [F,F]  filter MLP layer 1 (no bias) */
  const float* lin2_w; const float* lin2_b; /* [H,F],[H]  CFConv.lin2, :585 */
  const float* lin_w; const float* lin_b;   /* [H,H],[H]  InteractionBlock.lin, :479 */
} fmd_schnet_block_weights;

typedef struct fmd_schnet_weights {
  const float* embedding;               /* [embedding_size, H] */
  const fmd_schnet_block_weights* blocks; /* [num_blocks] */
  const float* out_w[4];                /* [out_width[k], in] */
  const float* out_b[4];                /* NULL when the layer has no bias */
} fmd_schnet_weights;

int64_t fmd_schnet_pack_bytes(const fmd_schnet_desc* desc);
/* Packs HOST weight tensors (and the num_rbf RBF centres, GaussianBasis.offset) into `packed_host`
 * (pack_bytes bytes) in the per-lane register order the kernels load; the caller copies that buffer
 * to the device once. */
int fmd_schnet_pack(const fmd_schnet_desc* desc, const fmd_schnet_weights* w, const float* centers,
                    void* packed_host);

int64_t fmd_schnet_workspace_bytes(const fmd_schnet_desc* desc, int64_t n_mol, int64_t n_nodes,
                                   int64_t max_pairs_per_mol);
/* status[0..3] (int32, device): [0] = 0 ok / FMD_ECAPACITY (pair list too small) / FMD_EINVAL (a molecule has more
 * beads than max_beads_per_mol: it was truncated to stay inside the LDS sized for that bound -- results are
 * meaningless), [1] = max pairs seen in one molecule, [2] = total pairs: these three are written by every call's last
 * kernel (nothing to zero).  [3] is STICKY: the most negative error code of any call since the caller last zeroed it
 * (the codes of [0], or FMD_ENONFINITE when a per-pair force contribution was Inf / NaN / >= 2^19 in magnitude -- the
 * 64-bit fixed-point force sum of a bead takes fewer than 2^12 addends, so below that threshold it cannot wrap; the
 * forces of that molecule are then NaN, never a finite wrong number).  The caller zeroes [3] once and after every read; it may then
 * read the words every few hundred calls without missing a failed call in between.  forces / energy are fully
 * overwritten.
 * The workspace size does not depend on the current device (it is computed for the largest split any gfx950
 * partition mode can choose), so a workspace sized under one device is valid under another. */
/* max_beads_per_mol: upper bound on beads of one molecule.  The per-molecule aggregate lives in LDS: up to 283
 * beads in one workgroup; 284..535 beads in feature-split mode (two workgroups per molecule, 64 features each).
 * Larger molecules return FMD_EUNSUPPORTED and take the operator path.
 * max_pairs_per_mol: capacity of the per-molecule pair list, whose per-centre runs are padded to
 *   multiples of 4 entries (plus at most one 4-entry group per run and 7 at the end); A(A-1)/2 + 8A can never
 *   overflow.
 * workspace: fmd_schnet_workspace_bytes bytes, 256-byte aligned. */
int fmd_schnet_energy_forces(const fmd_schnet_desc* desc, const void* packed_dev, const float* pos,
                             const int64_t* atom_types, const int64_t* mol_ptr, int64_t n_mol,
                             int64_t n_nodes, int64_t max_beads_per_mol, int64_t max_pairs_per_mol,
                             float* energy, float* forces, void* workspace, int32_t* status,
                             void* stream);

/* Measurement hooks (bench.py): the engine launches fmd_schnet_kernel_count(desc) kernels per call,
 * in a fixed order named by fmd_schnet_kernel_name.  The _timed variant is the same call that also
 * records events[2k] / events[2k+1] (hipEvent_t created by the caller with timing enabled) on `stream`
 * immediately before / after kernel k, so per-kernel durations come from the stream the kernels run on.
 * A NULL entry skips that record (each record costs a few microseconds of stream time: 30 records per
 * call are ~5 % of a 1ENH batch-128 force evaluation), so a caller can bracket only the kernels it needs. */
int fmd_schnet_kernel_count(const fmd_schnet_desc* desc);
const char* fmd_schnet_kernel_name(const fmd_schnet_desc* desc, int idx);
int fmd_schnet_energy_forces_timed(const fmd_schnet_desc* desc, const void* packed_dev, const float* pos,
                                   const int64_t* atom_types, const int64_t* mol_ptr, int64_t n_mol,
                                   int64_t n_nodes, int64_t max_beads_per_mol,
                                   int64_t max_pairs_per_mol, float* energy, float* forces,
                                   void* workspace, int32_t* status, void* stream, void** events,
                                   int64_t n_events);

/* ---------------------------------------------------------------------------------------------
 * Row N1: Langevin BAOAB elementwise halves around the force call (simulation/langevin.py:136-170).
 *   pre : v += .5 dt f/m ; x += .5 dt v ; v = vscale v + noisescale sqrt(1/(beta m)) * noise ; x += .5 dt v
 *   post: v += .5 dt f/m
 *   noise[N,3] is supplied by the caller (torch's generator, langevin.py:146), beta per node.
 * Row (e): parallel-tempering swap of accepted pairs (simulation/parallel_tempering.py:415-481):
 *   for each accepted pair p: x[a] <-> x[b], v[a] = v[b]*sqrt(beta_a/beta_b), v[b] = v[a]*sqrt(beta_b/beta_a)
 *   (molecule a/b = mol_a[p]/mol_b[p], n_atoms beads each, in place).
 * ------------------------------------------------------------------------------------------- */
int fmd_langevin_pre(float* x, float* v, const float* f, const float* masses, const float* beta_node,
                     const float* noise, int64_t n_nodes, float dt, float vscale, float noisescale,
                     void* stream);
int fmd_langevin_post(float* v, const float* f, const float* masses, int64_t n_nodes, float dt,
                      void* stream);
/* NVE velocity Verlet (simulation/velocity_verlet.py:38-45): pre: v += .5 dt f/m ; x += dt v.
 * The second half-kick after the force call is fmd_langevin_post. */
int fmd_verlet_pre(float* x, float* v, const float* f, const float* masses, int64_t n_nodes, float dt,
                   void* stream);
/* Overdamped Langevin / Brownian dynamics (simulation/langevin.py:315-434; ABI 5): in place
 * x <- x + f D dt + sqrt(2 D dt) noise with D = 1 / (beta friction) per bead; noise ~ N(0,1) [N,3]. */
int fmd_overdamped_step(float* x, const float* f, const float* beta_node, const float* noise, int64_t n_nodes,
                        float dt, float friction, void* stream);
int fmd_pt_swap_local(float* x, float* v, const int64_t* mol_a, const int64_t* mol_b,
                      const uint8_t* accept, const float* beta_mol, int64_t n_pairs, int64_t n_atoms,
                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * Row N2: classical prior terms in the condensed ("Static") form of simulation/specialize_prior.py:76-207.
 *   kind 0 bonds      (prior/harmonic.py:91-123,200-225)   U = k (d - x0)^2            params rows: k, x0
 *   kind 1 angles     (prior/harmonic.py:228-262)          U = k (cos(theta) - x0)^2   params rows: k, x0
 *   kind 2 repulsion  (prior/repulsion.py:80-122)          U = (sigma / d)^6           params rows: sigma
 *   kind 3 dihedrals  (prior/fourier_series.py:97-192)     U = v0 + sum_n k1_n sin(n phi) + k2_n cos(n phi)
 *                                                          params rows: v0, k1_1..k1_n, k2_1..k2_n
 *   kind 4 impropers  (prior/harmonic.py:301-322: Harmonic.compute on compute_torsions)
 *                                                          U = k (phi - x0)^2          params rows: k, x0
 *   kind 5 quartic angles (prior/polynomial.py:127-186)    U = v0 + sum_n k_n cos(theta)^n
 *                                                          params rows: v0, k_1..k_n (any n >= 1)
 *   kind 6 restricted angles (prior/restricted_bending.py:200-238; theta = atan2(|r1 x r2|, r1 . r2))
 *                                                          U = a c^4 + b c^3 + c c^2 + d c + k / sin(theta)^2 + v0
 *                                                          params rows: a, b, c, d, k, v0
 *   kind 7 raw angles (prior/harmonic.py:264-298)           U = k (theta - x0)^2, theta as in kind 6   params rows: k, x0
 *   kind 8 shifted-periodic impropers (prior/harmonic.py:325-400)
 *                                                          U = k (phi' - x0)^2, phi' = (phi < 0 ? phi + 2 pi : phi) - pi   rows: k, x0
 *   mapping [order][n_terms] i64 global bead indices (order 2/3/2/4/4/3/3/3/4); params [n_params][n_terms] f32;
 *   bead_ptr [N+1] i64 and entries [order * n_terms] i32 = the CSR of the fixed topology by bead: entry =
 *   term << 2 | slot, the entries of bead b are bead_ptr[b]..bead_ptr[b+1] in ascending (term, slot) order
 *   (built once per topology by the caller; n_terms < 2^29).  ADDS the per-molecule energies to energy[n_mol]
 *   and -dU/dpos to forces[N,3] (so it can follow fmd_schnet_energy_forces on the same buffers, which is what
 *   the reference's SumOut does, models/gradients.py:71-152).  Gather form, no atomics, fixed summation order.
 *   status[3] |= 1 when a term references a bead outside its molecule (that term's indices are clamped),
 *   |= 2 when a molecule has more beads than max_beads_per_mol (it is truncated to the LDS sized for the bound).
 * ------------------------------------------------------------------------------------------- */
int fmd_prior_energy_forces(const float* pos, const int64_t* mol_ptr, int64_t n_mol, int64_t max_beads_per_mol,
                            int kind, const int64_t* mapping, const float* params, int n_params,
                            const int64_t* bead_ptr, const int32_t* entries, int64_t n_terms, float* energy,
                            float* forces, int32_t* status, void* stream);

/* Pair kinds (0 bonds, 2 repulsion) in the partner-list form, for long pair lists: the by-bead CSR holds
 * (partner, parameters) records instead of term references -- partner [n_entries] u32 = index of the other bead
 * inside the molecule, bit 31 set on the entry that owns the term's energy; pair_params [n_params][n_entries]
 * f32 (k, x0 | sigma) -- 4 + 4 P bytes per entry.  Same results and conventions as fmd_prior_energy_forces. */
int fmd_prior_pairs_energy_forces(const float* pos, const int64_t* mol_ptr, int64_t n_mol,
                                  int64_t max_beads_per_mol, int kind, const int64_t* bead_ptr,
                                  const uint32_t* partner, const float* pair_params, int64_t n_entries,
                                  float* energy, float* forces, int32_t* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FLASHMD_HIP_H */
