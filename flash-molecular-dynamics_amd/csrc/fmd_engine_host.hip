// Fused engine, host side: weight packing, workspace carve-up, the launch sequence of one force evaluation and the
// C-ABI entry points fmd_schnet_* (include/flashmd_hip.h).  Replaces SchNet.forward + GradientsOut.forward of the
// reference (models/schnet.py:178-369, models/gradients.py:226-290) for StandardSchNet models whose weights the caller
// has zero-padded to the kernels' width, hidden = filters = 128.  The kernels live in
// fmd_engine_{pairs,edge_fwd,edge_bwd,node}.hip.
#include <atomic>
#include <cmath>
#include <vector>

#include "fmd_engine.h"

namespace {

// ---------------------------------------------------------------------------------------------
// packed weight layout (floats)
// ---------------------------------------------------------------------------------------------
struct BlockOff {
  int64_t b0, l1, l1t, l2, b2, l2t, l3, b3, l3t;
  int64_t q0h, q0l, q1h, q1l, q1th, q1tl;  // fp16 hi/lo split operands of the filter network
  int64_t q0ch, q0cl;                      // W0 with column k scaled by (centre_k - cutoff/2): backward only
};
struct PackLayout {
  int64_t emb, centers, blocks, block_stride, outs, out_stride, emb_x1, total;
};
constexpr int64_t BLOCK_FLOATS = 3 * F + 6 * (int64_t)SLICE + 4 * Q0_FLOATS + 4 * Q1_FLOATS;
constexpr int64_t OUT_FLOATS = 2 * (int64_t)SLICE + F;

PackLayout pack_layout(const fmd_schnet_desc* d) {
  PackLayout L;
  L.emb = 0;
  L.centers = (int64_t)d->embedding_size * F;
  L.blocks = L.centers + MAX_RBF;
  L.block_stride = BLOCK_FLOATS;
  L.outs = L.blocks + (int64_t)d->num_blocks * BLOCK_FLOATS;
  L.out_stride = OUT_FLOATS;
  L.emb_x1 = L.outs + (int64_t)d->num_out * OUT_FLOATS;  // [embedding_size][128]: embedding lin1(block 0)^T
  L.total = L.emb_x1 + (int64_t)d->embedding_size * F;
  return L;
}
BlockOff block_off(const PackLayout& L, int l) {
  BlockOff o;
  int64_t p = L.blocks + l * L.block_stride;
  o.b0 = p; p += F;
  o.l1 = p; p += SLICE;
  o.l1t = p; p += SLICE;
  o.l2 = p; p += SLICE;
  o.b2 = p; p += F;
  o.l2t = p; p += SLICE;
  o.l3 = p; p += SLICE;
  o.b3 = p; p += F;
  o.l3t = p; p += SLICE;
  o.q0h = p; p += Q0_FLOATS;
  o.q0l = p; p += Q0_FLOATS;
  o.q1h = p; p += Q1_FLOATS;
  o.q1l = p; p += Q1_FLOATS;
  o.q1th = p; p += Q1_FLOATS;
  o.q1tl = p; p += Q1_FLOATS;
  o.q0ch = p; p += Q0_FLOATS;
  o.q0cl = p; p += Q0_FLOATS;
  return o;
}

// fp16 hi/lo split of an operand for v_mfma_f32_32x32x16_f16: x ~ hi + lo / lo_scale with hi = f16(x),
// lo = f16((x - hi) * lo_scale).  lo_scale = 2048 for operands whose partner is a scaled activation split (two
// accumulators: node kernels, the u GEMM); lo_scale = 1 for the contractions that run on ONE accumulator (filter layers
// 0 and 1: weights and their bounded activations both carry an unscaled low word, fmd_engine.h split_pair_u).  M is [n_out][n_k] row-major (or its transpose when `transposed`): the operand that
// multiplies a row tile as out[row][o] = sum_k a[row][k] * M[o][k].  Record (w, s, lane) holds the 8 halves
// M[32w + (lane&31)][16s + 8(lane>>5) + j], zero outside the matrix (the A-operand and the B-operand fragment of that
// instruction index their matrix the same way).
void pack_split(float* dst_hi, float* dst_lo, const float* M, int n_out, int n_k, int n_steps,
                bool transposed, float lo_scale = 2048.0f) {
  _Float16* hi = reinterpret_cast<_Float16*>(dst_hi);
  _Float16* lo = reinterpret_cast<_Float16*>(dst_lo);
  for (int w = 0; w < 4; ++w)
    for (int s = 0; s < n_steps; ++s)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const int o = 32 * w + (l & 31), k = 16 * s + 8 * (l >> 5) + j;
          float v = 0.f;
          if (o < n_out && k < n_k) v = transposed ? M[(int64_t)k * n_out + o] : M[(int64_t)o * n_k + k];
          const _Float16 h = (_Float16)v;
          const int64_t idx = (((int64_t)w * n_steps + s) * 64 + l) * 8 + j;
          hi[idx] = h;
          lo[idx] = (_Float16)((v - (float)h) * lo_scale);
        }
}
// node-kernel operand: hi record followed by lo record (SLICE floats)
void pack_node(float* dst, const float* M, int n_out, int n_k, bool transposed) {
  pack_split(dst, dst + Q1_FLOATS, M, n_out, n_k, 8, transposed);
}

int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

struct Workspace {
  float *x, *gx, *gagg, *e_node, *pd, *pc, *pdc, *pgd;
  float* x1[8];
  float* t[8];
  float* slabs;
  float* wst[8];    // filter stash of block l (null: block 0, or stash disabled)
  uint32_t *poi, *poj;
  int32_t* npairs;
  int32_t* report;  // [n_mol][3] pairs_kernel -> finish_kernel
  int* items;       // [n_mol * n_split][2][ITEM_INTS] work items of the pair kernels (sched_body)
  int* extra_row;   // [N]
  int* queue;       // [16] item counters of the pair-kernel launches in queue mode (zeroed by pairs_kernel)
  int64_t total;
};

constexpr int MAX_SPLIT = 32;  // workgroups per molecule (small batches: 8 molecules x 32 = 256 workgroups)
constexpr int MAX_CUS = 256;   // gfx950: 256 CUs in SPX mode, fewer per device in the partitioned modes
int device_cus() {
  int cus = MAX_CUS;
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  return cus < MAX_CUS ? cus : MAX_CUS;
}
// `cus`: the device's CU count at launch; MAX_CUS when SIZING a workspace (the split falls with the CU count, so the
// size computed for MAX_CUS covers every device: fmd_schnet_workspace_bytes is device independent)
constexpr int TAIL_MAX_SPLIT = 4;  // batches larger than the chip: workgroups per molecule tried against the tail
// Molecule groups (fmd_engine.h MolView): batches of MORE molecules than CUs whose molecules are small run the pair list,
// the pair kernels and the final reduction over groups of `stride` consecutive molecules -- up to GROUP_BEADS beads and
// GROUP_MAX molecules per group.  2048 molecules of 10 beads: 82 groups of 25 instead of 2048 two-tile items.
constexpr int GROUP_MAX = 32;          // molecules per group at most
constexpr int GROUP_BEADS = 256;       // beads per group at most (the aggregate of a group stays in one workgroup's LDS)
constexpr int GROUP_MOL_BEADS = 64;    // only molecules up to this size are grouped
constexpr int64_t GROUP_CAP = 2560;    // pair-list capacity of such a molecule at most (64 * 63 / 2 + 8 * 64 = 2528)
int group_stride(const fmd_schnet_desc* d, int64_t n_mol, int64_t max_beads, int64_t cap, int cus) {
  if ((d->flags & FMD_FLAG_NO_MOLECULE_GROUPS) || max_beads > GROUP_MOL_BEADS || cap > GROUP_CAP || n_mol <= cus) return 1;
  int64_t k = GROUP_BEADS / max_beads;
  if (k > GROUP_MAX) k = GROUP_MAX;
  // ... and no larger than what brings the batch down to about one group per CU: a group's workgroup zeroes and flushes
  // the whole group's aggregate, which for a 250-bead group of 10-bead molecules cost more than its 14 tiles
  // (profiles/r06_small_molecules.txt: 2048 x 10 beads 0.45 ms with groups of 25, 1024 x 35 beads SLOWER than ungrouped
  // with groups of 7)
  const int64_t fill = (n_mol + cus - 1) / cus;
  if (k > fill) k = fill;
  return k < 2 ? 1 : (int)k;
}
// list arrays are carved per GROUP (stride x the per-molecule capacity): the last, partial group needs up to
// GROUP_MAX - 1 molecules' worth of room beyond n_mol
int64_t list_slots(int64_t n_mol, int64_t cap) { return cap <= GROUP_CAP ? n_mol + GROUP_MAX - 1 : n_mol; }
int resolve_split(const fmd_schnet_desc* d, int64_t n_mol, int cus) {
  if (d->n_split > 0) return d->n_split > MAX_SPLIT ? MAX_SPLIT : d->n_split;
  if (n_mol > cus) {
    // More molecules than CUs: the hardware deals workgroups to CUs as they free up, and what is lost is the tail --
    // 384 molecules on 256 CUs keep half the chip idle for the second half of the launch.  Cutting every molecule into
    // S pieces (static interleave of its tiles) makes the pieces finer than the tail: the launch lasts
    // ceil(n_mol S / cus) / S molecule-times, plus a few per cent per extra piece for its aggregate (zeroing, flush,
    // one more slab for the node kernels to add).  384 molecules: S = 2 -> 1.5 instead of 2; 512: S = 1 (2.0 either way).
    int best = 1;
    double best_t = 1e30;
    for (int s = 1; s <= TAIL_MAX_SPLIT; ++s) {
      const double t = (double)((n_mol * s + cus - 1) / cus) / s * (1.0 + 0.04 * (s - 1));
      if (t < best_t - 1e-9) { best_t = t; best = s; }
    }
    return best;
  }
  // as many workgroups per molecule as fit the chip in ONE round (floor, not ceil: 100 molecules x 3 would take a second
  // round for 44 workgroups -- 0.67 of a molecule's tiles per CU against 0.5 with x 2 -- and only a launch of one round
  // gets the balanced work items)
  int64_t s = n_mol > 0 ? cus / n_mol : 1;
  if (s < 1) s = 1;
  if (s > MAX_SPLIT) s = MAX_SPLIT;
  return (int)s;
}

Workspace carve_ws(const fmd_schnet_desc* d, void* base, int64_t n_mol, int64_t n_nodes, int64_t cap,
                   int n_split) {
  Workspace w;
  char* p = reinterpret_cast<char*>(base);
  auto take = [&](int64_t bytes) {
    char* q = p;
    p += align256(bytes);
    return q;
  };
  const int64_t nf = n_nodes * F * (int64_t)sizeof(float);
  // arrays the node kernels WRITE: whole 32-row tiles (their stores are unconditional, see rows_fetch)
  const int64_t nfp = (n_nodes + 31) / 32 * 32 * F * (int64_t)sizeof(float);
  w.x = (float*)take(nfp);
  w.gx = (float*)take(nfp);
  w.gagg = (float*)take(nfp);
  for (int l = 0; l < d->num_blocks; ++l) {
    w.x1[l] = (float*)take(nfp);
    w.t[l] = (float*)take(nfp);
  }
  w.slabs = (float*)take(nf * (n_split + 1));  // the last one: extra pieces of the balanced schedule (at most N rows)
  w.e_node = (float*)take(n_nodes * 4);
  w.poi = (uint32_t*)take(n_mol * cap * 4);
  w.poj = (uint32_t*)take(n_mol * cap * 4);
  w.pd = (float*)take(n_mol * cap * 4);
  w.pc = (float*)take(n_mol * cap * 4);
  w.pdc = (float*)take(n_mol * cap * 4);
  w.pgd = (float*)take(2 * n_mol * cap * 4);  // second slab: feature-split mode
  w.npairs = (int32_t*)take(n_mol * 4);
  w.report = (int32_t*)take(n_mol * 12);
  w.items = (int*)take(n_mol * n_split * 2 * ITEM_INTS * 4);
  w.extra_row = (int*)take(n_nodes * 4);
  w.queue = (int*)take(16 * 4);
  // filter stash (EdgeArgs::wst): one float per list entry and feature, for every block whose backward pass has a gx1
  // aggregate to update (all but the first).  Last in the carve: nothing else moves when it is switched off.
  // fp32-grade precision only: with a third of the MFMA work the W16A16 kernels gain less from the dropped layer than the
  // stash traffic costs them (same-box A/B: 108.2k against 110.5k timestep*mol/s).
  const bool stash = d->precision == FMD_PREC_F32 && !(d->flags & FMD_FLAG_NO_FILTER_STASH);
  for (int l = 0; l < d->num_blocks; ++l)
    w.wst[l] = (l > 0 && stash) ? (float*)take(n_mol * cap * F * (int64_t)sizeof(float)) : nullptr;
  w.total = p - reinterpret_cast<char*>(base);
  return w;
}

bool desc_ok(const fmd_schnet_desc* d) {
  if (!d) return false;
  if (d->hidden != F || d->num_rbf < 2 || d->num_rbf > MAX_RBF) return false;
  if (d->num_blocks < 1 || d->num_blocks > 8) return false;
  if (d->num_out < 2 || d->num_out > 3) return false;
  for (int k = 0; k < d->num_out; ++k)
    if (d->out_width[k] < 1 || d->out_width[k] > F) return false;
  if (d->out_width[d->num_out - 1] != 1) return false;
  if (d->precision != FMD_PREC_F32 && d->precision != FMD_PREC_W16A16) return false;
  if (d->embedding_size < 1) return false;
  if (d->flags & ~(FMD_FLAG_NO_FILTER_STASH | FMD_FLAG_NO_MOLECULE_GROUPS)) return false;
  // the struct arrives from C callers: a cutoff or gamma that is not a finite positive / finite number must not reach
  // the kernels
  if (!(d->cutoff > 0.f) || !std::isfinite(d->cutoff) || !std::isfinite(d->gamma)) return false;
  return true;
}


// one-time, per-device raise of the kernels' dynamic-LDS ceilings (idempotent; nothing a caller can observe)
int ensure_kernel_attrs() {
  static std::atomic<int> done[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return FMD_ELAUNCH;
  if (done[dev].load(std::memory_order_acquire)) return FMD_OK;
  int rc = setup_pairs_attrs();
  if (rc == FMD_OK) rc = setup_edge_fwd_attrs();
  if (rc == FMD_OK) rc = setup_edge_bwd_attrs();
  if (rc == FMD_OK) done[dev].store(1, std::memory_order_release);
  return rc;
}

#ifdef FMD_PHASE_STAMPS
unsigned long long* g_stamps = nullptr;   // diagnostic build only
int g_stamp_block = -1, g_stamp_fwd = 0, g_stamp_dbg = 0;  // which launch writes them; role-skip bits
#endif

}  // namespace

#ifdef FMD_PHASE_STAMPS
extern "C" int fmd_debug_set_stamps(void* device_buffer, int block, int fwd, int dbg) {
  g_stamps = reinterpret_cast<unsigned long long*>(device_buffer);
  g_stamp_block = block;
  g_stamp_fwd = fwd;
  g_stamp_dbg = dbg;
  return FMD_OK;
}
#endif

extern "C" int64_t fmd_schnet_pack_bytes(const fmd_schnet_desc* desc) {
  if (!desc_ok(desc)) return FMD_EINVAL;
  return pack_layout(desc).total * (int64_t)sizeof(float);
}

extern "C" int fmd_schnet_pack(const fmd_schnet_desc* desc, const fmd_schnet_weights* wt,
                               const float* centers, void* packed_host) {
  if (!desc_ok(desc) || !wt || !centers || !packed_host || !wt->embedding || !wt->blocks) return FMD_EINVAL;
  const PackLayout L = pack_layout(desc);
  float* out = reinterpret_cast<float*>(packed_host);
  // node operands in split (hi/lo fp16) form; W16A16 keeps lin1/lin2/lin fp32-grade like the reference
  memset(out, 0, L.total * sizeof(float));
  memcpy(out + L.emb, wt->embedding, (size_t)desc->embedding_size * F * sizeof(float));
  for (int k = 0; k < desc->num_rbf; ++k) out[L.centers + k] = centers[k];
  // The first block's x1 = lin1(embedding[type]) only depends on the bead type: one table row per type (fp64
  // accumulation, rounded once), so the first node launch of a timestep is a gather instead of a GEMM.
  if (desc->num_blocks > 0 && wt->blocks[0].lin1_w) {
    const float* l1 = reinterpret_cast<const float*>(wt->blocks[0].lin1_w);
    const float* em = reinterpret_cast<const float*>(wt->embedding);
    for (int t = 0; t < desc->embedding_size; ++t)
      for (int o = 0; o < F; ++o) {
        double acc = 0.0;
        for (int k = 0; k < F; ++k) acc += (double)em[(int64_t)t * F + k] * (double)l1[(int64_t)o * F + k];
        out[L.emb_x1 + (int64_t)t * F + o] = (float)acc;
      }
  }
  const int q0_steps = desc->num_rbf > 64 ? 8 : 4;   // k-steps of 16 per wave in the layer-0 records (q0_rec)
  for (int l = 0; l < desc->num_blocks; ++l) {
    const fmd_schnet_block_weights& b = wt->blocks[l];
    if (!b.lin1_w || !b.f0_w || !b.f0_b || !b.f1_w || !b.lin2_w || !b.lin2_b || !b.lin_w || !b.lin_b)
      return FMD_EINVAL;
    const BlockOff o = block_off(L, l);
    memcpy(out + o.b0, b.f0_b, F * sizeof(float));
    if (desc->precision == FMD_PREC_W16A16)  // the fp16 filter network carries its bias in fp16 (gptq.py:178-184)
      for (int f = 0; f < F; ++f) out[o.b0 + f] = (float)(_Float16)out[o.b0 + f];
    pack_node(out + o.l1, b.lin1_w, F, F, false);
    pack_node(out + o.l1t, b.lin1_w, F, F, true);
    pack_node(out + o.l2, b.lin2_w, F, F, false);
    memcpy(out + o.b2, b.lin2_b, F * sizeof(float));
    pack_node(out + o.l2t, b.lin2_w, F, F, true);
    pack_node(out + o.l3, b.lin_w, F, F, false);
    memcpy(out + o.b3, b.lin_b, F * sizeof(float));
    pack_node(out + o.l3t, b.lin_w, F, F, true);
    pack_split(out + o.q0h, out + o.q0l, b.f0_w, F, desc->num_rbf, q0_steps, false, 1.0f);
    {  // Wc[f][k] = (centre_k - cutoff/2) W0[f][k]:  W0 (raw (d - centre)) = (d - cutoff/2) W0 raw - Wc raw
      std::vector<float> wc((size_t)F * desc->num_rbf);
      const float c0 = 0.5f * desc->cutoff;
      for (int f = 0; f < F; ++f)
        for (int k = 0; k < desc->num_rbf; ++k)
          wc[(size_t)f * desc->num_rbf + k] = (centers[k] - c0) * b.f0_w[(size_t)f * desc->num_rbf + k];
      pack_split(out + o.q0ch, out + o.q0cl, wc.data(), F, desc->num_rbf, q0_steps, false, 1.0f);
    }
    pack_split(out + o.q1h, out + o.q1l, b.f1_w, F, F, 8, false, 1.0f);
    pack_split(out + o.q1th, out + o.q1tl, b.f1_w, F, F, 8, true);
  }
  int in_w = F;
  for (int k = 0; k < desc->num_out; ++k) {
    if (!wt->out_w[k]) return FMD_EINVAL;
    float* base = out + L.outs + k * L.out_stride;
    const int ow = desc->out_width[k];
    pack_node(base, wt->out_w[k], ow, in_w, false);
    if (wt->out_b[k]) memcpy(base + SLICE, wt->out_b[k], ow * sizeof(float));
    // the last layer (Linear(in_w, 1), no bias): its weight vector, plain, in the bias slot -- node_out evaluates that
    // layer and the first backward stage on registers
    if (k == desc->num_out - 1)
      for (int i = 0; i < in_w; ++i) base[SLICE + i] = wt->out_w[k][i];
    if (desc->precision == FMD_PREC_W16A16)
      for (int f = 0; f < F; ++f) base[SLICE + f] = (float)(_Float16)base[SLICE + f];
    pack_node(base + SLICE + F, wt->out_w[k], in_w, ow, true);  // (W^T)[o=in][k=out] = W[out][in]
    in_w = ow;
  }
  return FMD_OK;
}

extern "C" int64_t fmd_schnet_workspace_bytes(const fmd_schnet_desc* desc, int64_t n_mol,
                                              int64_t n_nodes, int64_t max_pairs_per_mol) {
  if (!desc_ok(desc) || n_mol < 0 || n_nodes < 0 || max_pairs_per_mol < 0) return FMD_EINVAL;
  const int64_t cap = (max_pairs_per_mol + 31) / 32 * 32;
  // the slabs scale with the split: the largest any device can choose (a partition with fewer CUs than molecules may
  // cut the molecules into up to TAIL_MAX_SPLIT pieces); >= 2 for the feature-split mode
  int split = resolve_split(desc, n_mol, MAX_CUS);
  if (desc->n_split == 0 && split < TAIL_MAX_SPLIT) split = TAIL_MAX_SPLIT;
  return carve_ws(desc, nullptr, list_slots(n_mol, cap), n_nodes, cap, split < 2 ? 2 : split).total + 256;
}

static int run_engine(const fmd_schnet_desc* desc, const void* packed_dev, const float* pos,
                      const int64_t* atom_types, const int64_t* mol_ptr, int64_t n_mol,
                      int64_t n_nodes, int64_t max_beads_per_mol, int64_t max_pairs_per_mol,
                      float* energy, float* forces, void* workspace, int32_t* status, void* stream,
                      hipEvent_t* ev) {
  if (!desc_ok(desc) || n_mol < 0 || n_nodes < 0) return FMD_EINVAL;
  if (n_mol == 0 || n_nodes == 0) return FMD_OK;
  if (!packed_dev || !pos || !atom_types || !mol_ptr || !energy || !forces || !workspace || !status)
    return FMD_EINVAL;
  if (max_beads_per_mol < 1 || max_beads_per_mol > MAX_LDS_ROWS_FS) return FMD_EUNSUPPORTED;
  // feature-split mode: the split kernels only (fp32-grade default and W16A16)
  const bool fs = max_beads_per_mol > MAX_LDS_ROWS;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return FMD_EINVAL;
  hipStream_t s = fmd_stream(stream);
  const int L = desc->num_blocks;
  const int64_t cap_mol = (max_pairs_per_mol + 31) / 32 * 32;
  if (cap_mol <= 0 || cap_mol > (int64_t)1 << 30) return FMD_EINVAL;
  const int cus = device_cus();
  // molecule groups: from here on "molecule" means a group of `stride` consecutive molecules (MolView); the caller's
  // arrays keep their meaning (energy[n_mol_real], mol_ptr[n_mol_real + 1])
  const int64_t n_mol_real = n_mol;
  const int stride = fs ? 1 : group_stride(desc, n_mol, max_beads_per_mol, cap_mol, cus);
  const MolView mv{mol_ptr, n_mol_real, stride};
  n_mol = (n_mol_real + stride - 1) / stride;
  max_beads_per_mol *= stride;
  const int64_t cap = cap_mol * stride;
  int split = fs ? 2 : resolve_split(desc, n_mol, cus);  // FS: the two feature halves
  if (stride > 1 && split > TAIL_MAX_SPLIT) split = TAIL_MAX_SPLIT;   // (what fmd_schnet_workspace_bytes sized the slabs for)
  // chained work items (sched_body): more than half but not all CUs would get a molecule of their own
  const bool chain = !fs && desc->n_split == 0 && split == 1 && 2 * n_mol > cus && n_mol < cus && cus <= 256;
  const int n_slabs = chain ? 2 : split;                       // regular slabs the node kernels add
  // queue mode (round 5): more (molecule, split) items than CUs -- e.g. a thousand small molecules -- run as ONE workgroup
  // per CU that keeps its weight fragments and draws items from a device counter, instead of rounds of one-item
  // workgroups (static items, no feature split; a caller-given n_split keeps the plain launch)
  const bool queue = !fs && !chain && desc->n_split == 0 && n_mol * split > cus;
  const int64_t pair_wgs = chain || queue ? cus : n_mol;       // grid.y of the pair kernels
  if ((int64_t)(n_slabs + 1) * n_nodes >= ((int64_t)1 << 31)) return FMD_EUNSUPPORTED;  // slab rows are 32-bit in the work items
  const Workspace ws = carve_ws(desc, workspace, n_mol, n_nodes, cap, n_slabs);
  // the carve of THIS launch configuration (groups, split) must fit what fmd_schnet_workspace_bytes told the caller to
  // allocate for the batch -- pure arithmetic, checked on every call
  if (ws.total + 256 > fmd_schnet_workspace_bytes(desc, n_mol_real, n_nodes, max_pairs_per_mol)) return FMD_EINVAL;
  const PackLayout PL = pack_layout(desc);
  const float* pk = reinterpret_cast<const float*>(packed_dev);

  int rc = ensure_kernel_attrs();
  if (rc != FMD_OK) return rc;

  int kidx = 0;  // kernel index in launch order (fmd_schnet_kernel_name)
  // a NULL slot skips that record: a caller may time a single kernel without paying for 2 records per launch
  auto ev_begin = [&]() { if (ev && ev[2 * kidx]) (void)hipEventRecord(ev[2 * kidx], s); };
  auto ev_end = [&]() { if (ev && ev[2 * kidx + 1]) (void)hipEventRecord(ev[2 * kidx + 1], s); ++kidx; };
  SchedArgs sa;
  {
    PairsArgs pa;
    pa.pos = pos; pa.mol = mv; pa.rc = desc->cutoff; pa.cap = (int)cap; pa.poi = ws.poi; pa.poj = ws.poj;
    pa.pd = ws.pd; pa.pc = ws.pc; pa.pdc = ws.pdc; pa.npairs = ws.npairs; pa.report = ws.report;
    pa.max_beads = (int)max_beads_per_mol; pa.forces = forces; pa.extra_row = ws.extra_row; pa.queue = ws.queue;
    // the pair kernels' work items (sched_body in fmd_engine_pairs.hip): balanced when the split is left to the library
    // and the whole launch is one round of workgroups; a caller-given n_split keeps the static tile interleave
    sa.npairs = ws.npairs; sa.mol = mv; sa.n_mol = (int)n_mol; sa.n_split = split; sa.fs = fs ? 1 : 0;
    sa.balanced = chain ? 2 : (!fs && n_mol >= 2 && n_mol <= 256 && n_mol * split <= cus && desc->n_split == 0) ? 1 : 0;
    sa.n_wg = (int)pair_wgs;
    sa.n_nodes = n_nodes; sa.items = ws.items; sa.extra_row = ws.extra_row;
    ev_begin();
    rc = launch_pairs(pa, n_mol, s);
    ev_end();
    if (rc != FMD_OK) return rc;
  }

  const bool half = desc->precision == FMD_PREC_W16A16;
  const int64_t n_tiles = (n_nodes + 31) / 32;
  // node kernels: one workgroup per CU walks 32-row tiles (the kernels keep their operands in registers)
  const unsigned res_grid = (unsigned)(n_tiles < cus ? n_tiles : cus);
  const int64_t slab_stride = n_nodes * F;

  auto edge = [&](bool fwd, int l) -> int {
    const BlockOff o = block_off(PL, l);
    EdgeArgs ea;
    ea.items = ws.items;
    ea.poi = ws.poi; ea.poj = ws.poj; ea.pd = ws.pd; ea.pc = ws.pc; ea.pdc = ws.pdc; ea.pgd = ws.pgd; ea.npairs = ws.npairs;
    ea.mol = mv; ea.x1 = ws.x1[l]; ea.gagg = ws.gagg;
    ea.b0 = pk + o.b0;
    ea.q0h = pk + o.q0h; ea.q0l = pk + o.q0l; ea.q1h = pk + o.q1h; ea.q1l = pk + o.q1l;
    ea.q1th = pk + o.q1th; ea.q1tl = pk + o.q1tl;
    ea.q0ch = pk + o.q0ch; ea.q0cl = pk + o.q0cl; ea.c0 = 0.5f * desc->cutoff; ea.nb = (desc->num_rbf + 7) / 8;
    ea.centers = pk + PL.centers; ea.slabs = ws.slabs; ea.slab_stride = slab_stride;
    ea.gamma = desc->gamma; ea.rc = desc->cutoff; ea.cap = (int)cap; ea.n_split = split;
    ea.pgd_stride = n_mol * cap; ea.fs = fs; ea.max_beads = (int)max_beads_per_mol;
    ea.wst = ws.wst[l];
    // (launch index: forward block l -> l, backward block l -> L + l; each launch has its own counter)
    ea.queue = queue ? ws.queue + (fwd ? l : L + l) : nullptr;
    ea.n_items = (int)(n_mol * split);
    if (queue) ea.n_split = 1;   // grid (1, CUs): the items carry molecule and split
#ifdef FMD_PHASE_STAMPS
    ea.stamps = (l == g_stamp_block && (fwd ? 1 : 0) == g_stamp_fwd) ? g_stamps : nullptr;
    ea.dbg = (ea.stamps || g_stamp_block == -2) ? g_stamp_dbg : 0;   // block -2: the switches apply to every launch
#endif
    if (fwd) return launch_edge_fwd(ea, desc->num_rbf, half, pair_wgs, s);
    return launch_edge_bwd(ea, desc->num_rbf, half, /*need_gx1=*/l > 0, /*first=*/l == L - 1, pair_wgs, s);
  };

  // ---------------- forward
  for (int l = 0; l < L; ++l) {
    const BlockOff o = block_off(PL, l);
    ev_begin();
    if (l == 0) {
      rc = launch_node_embed(atom_types, pk + PL.emb, pk + PL.emb_x1, ws.x, ws.x1[0], n_nodes, sa, s);
    } else {
      const BlockOff q = block_off(PL, l - 1);
      NodeFwdArgs na;
      na.x = ws.x; na.slabs = ws.slabs; na.slab_stride = slab_stride; na.n_slabs = n_slabs; na.extra_row = ws.extra_row;
      na.l1 = pk + o.l1; na.x1_out = ws.x1[l]; na.n_nodes = n_nodes;
      na.l2 = pk + q.l2; na.b2 = pk + q.b2; na.l3 = pk + q.l3; na.b3 = pk + q.b3; na.t_out = ws.t[l - 1];
      rc = launch_node_fwd(na, res_grid, s);
    }
    ev_end();
    if (rc != FMD_OK) return rc;
    ev_begin();
    rc = edge(true, l);
    ev_end();
    if (rc != FMD_OK) return rc;
  }
  {
    const BlockOff q = block_off(PL, L - 1);
    NodeOutArgs oa;
    oa.x = ws.x; oa.slabs = ws.slabs; oa.slab_stride = slab_stride; oa.n_slabs = n_slabs; oa.extra_row = ws.extra_row;
    oa.l2 = pk + q.l2; oa.b2 = pk + q.b2; oa.l3 = pk + q.l3; oa.b3 = pk + q.b3; oa.t_out = ws.t[L - 1];
    for (int k = 0; k < 3; ++k) {
      const int kk = k < desc->num_out ? k : desc->num_out - 1;
      const float* base = pk + PL.outs + kk * PL.out_stride;
      oa.po[k] = base; oa.bo[k] = base + SLICE; oa.pot[k] = base + SLICE + F;
    }
    oa.w_last = pk + PL.outs + (desc->num_out - 1) * PL.out_stride + SLICE;
    oa.e_node = ws.e_node; oa.gx = ws.gx; oa.n_nodes = n_nodes;
    oa.l3t = pk + q.l3t; oa.l2t = pk + q.l2t; oa.gagg = ws.gagg;   // the last block's node backward rides along
    ev_begin();
    rc = launch_node_out(oa, desc->num_out, half, res_grid, s);
    ev_end();
    if (rc != FMD_OK) return rc;
  }
  // ---------------- backward
  for (int l = L - 1; l >= 0; --l) {
    const BlockOff o = block_off(PL, l);
    NodeBwdArgs ba;
    ba.gx = ws.gx; ba.slabs = ws.slabs; ba.slab_stride = slab_stride; ba.n_slabs = n_slabs; ba.extra_row = ws.extra_row;
    ba.t = ws.t[l]; ba.l3t = pk + o.l3t; ba.l2t = pk + o.l2t; ba.gagg = ws.gagg; ba.n_nodes = n_nodes;
    ba.l1t = l == L - 1 ? nullptr : pk + block_off(PL, l + 1).l1t;
    if (l != L - 1) {   // (the last block's node backward is part of node_out)
      ev_begin();
      rc = launch_node_bwd(ba, res_grid, s);
      ev_end();
      if (rc != FMD_OK) return rc;
    }
    ev_begin();
    rc = edge(false, l);
    ev_end();
    if (rc != FMD_OK) return rc;
  }
  {
    FinishArgs fa;
    fa.pos = pos; fa.mol = mv; fa.poi = ws.poi; fa.poj = ws.poj; fa.pd = ws.pd; fa.pgd = ws.pgd;
    fa.pgd2 = fs ? ws.pgd + n_mol * cap : nullptr; fa.npairs = ws.npairs; fa.cap = (int)cap; fa.e_node = ws.e_node;
    fa.energy = energy; fa.forces = forces; fa.max_beads = (int)max_beads_per_mol; fa.report = ws.report;
    fa.status = status;
    ev_begin();
    rc = launch_finish(fa, n_mol, s);
    ev_end();
    if (rc != FMD_OK) return rc;
  }
  return FMD_OK;
}

extern "C" int fmd_schnet_energy_forces(const fmd_schnet_desc* desc, const void* packed_dev,
                                        const float* pos, const int64_t* atom_types,
                                        const int64_t* mol_ptr, int64_t n_mol, int64_t n_nodes,
                                        int64_t max_beads_per_mol, int64_t max_pairs_per_mol,
                                        float* energy, float* forces, void* workspace,
                                        int32_t* status, void* stream) {
  return run_engine(desc, packed_dev, pos, atom_types, mol_ptr, n_mol, n_nodes, max_beads_per_mol,
                    max_pairs_per_mol, energy, forces, workspace, status, stream, nullptr);
}

extern "C" int fmd_schnet_kernel_count(const fmd_schnet_desc* desc) {
  return desc_ok(desc) ? 4 * desc->num_blocks + 2 : FMD_EINVAL;
}

extern "C" const char* fmd_schnet_kernel_name(const fmd_schnet_desc* desc, int idx) {
  if (!desc_ok(desc) || idx < 0 || idx >= 4 * desc->num_blocks + 2) return "";
  const int L = desc->num_blocks;
  // pairs | [node_fwd, edge_fwd] x L | node_out (with the last block's node backward) | edge_bwd | [node_bwd, edge_bwd]
  // x (L - 1) | finish
  if (idx == 0) return "pairs_kernel";
  if (idx <= 2 * L) return (idx & 1) ? "node_fwd_kernel" : "edge_fwd_kernel";
  if (idx == 2 * L + 1) return "node_out_kernel";
  if (idx <= 4 * L) return (idx & 1) ? "node_bwd_kernel" : "edge_bwd_kernel";
  return "finish_kernel";
}

extern "C" int fmd_schnet_energy_forces_timed(const fmd_schnet_desc* desc, const void* packed_dev,
                                              const float* pos, const int64_t* atom_types,
                                              const int64_t* mol_ptr, int64_t n_mol, int64_t n_nodes,
                                              int64_t max_beads_per_mol, int64_t max_pairs_per_mol,
                                              float* energy, float* forces, void* workspace,
                                              int32_t* status, void* stream, void** events,
                                              int64_t n_events) {
  if (!desc_ok(desc) || !events || n_events < 2 * (4 * desc->num_blocks + 2)) return FMD_EINVAL;
  return run_engine(desc, packed_dev, pos, atom_types, mol_ptr, n_mol, n_nodes, max_beads_per_mol,
                    max_pairs_per_mol, energy, forces, workspace, status, stream,
                    reinterpret_cast<hipEvent_t*>(events));
}
