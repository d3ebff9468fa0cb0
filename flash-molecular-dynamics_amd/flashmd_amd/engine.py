"""Fused whole-path engine: energy[B] and forces[N,3] of a batch in one HIP pipeline of 14 launches
(csrc/fmd_engine_{pairs,edge_fwd,edge_bwd,node,host}.hip behind fmd_schnet_energy_forces).

Replaces, for qualifying models, what the reference does per timestep in SchNet.forward +
GradientsOut.forward (reference models/schnet.py:178-369, models/gradients.py:226-290).  A model
qualifies when it has the StandardSchNet shape the kernels are built for: Gaussian basis with
<= 128 functions, cosine cutoff with lower bound 0 used for both the basis and the CFConv envelope,
tanh, hidden <= 128 and filters <= 128 (the kernels are built for 128 x 128: narrower layers are zero-padded when the
weights are packed, which is exact -- a padded feature is 0 in every layer -- and costs the padded arithmetic), a 2- or
3-linear output MLP with widths <= 128, fp32 weights (or the W16A16 twins of models/gptq.py on both the filter and
the output networks), and no molecule larger than 535 beads (up to 283 a molecule's aggregate fits one workgroup's LDS; up to 535 the engine
switches to its feature-split mode, two workgroups holding 64 features each).  Anything else returns
None here and GradientsOut uses the operator path.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _ffi
from ._ffi import call, ptr, stream

MAX_BEADS = 535   # 283 with the whole aggregate in one workgroup's LDS, 535 in feature-split mode
MAX_BEADS_WHOLE = 283
MAX_RBF = 128     # csrc/fmd_engine.h MAX_RBF; 65..128 Gaussians run in the per-wave Gaussian form (round 6: feature-split mode too)
# fmd_schnet_desc.precision (include/flashmd_hip.h)
PRECISIONS = {"f32": 0,        # FMD_PREC_F32: split-operand f16 MFMA, fp32-grade (2e-7 per contraction), the default
              "w16a16": 1}     # FMD_PREC_W16A16: filter and output networks with fp16 operands, fp32 accumulate
STATUS_CHECK_EVERY = 1024      # force evaluations between two reads of the device status words (one sync each)
MAX_STASH_BYTES = 48 << 30     # a batch whose filter stash would exceed this runs without it (same results, slower)


def _linears(mlp):
    return [m for m in mlp.layers if isinstance(m, torch.nn.Linear)]


WIDTH = 128   # hidden = filters of the kernels (csrc/fmd_engine.h F)


def _pad(t, rows=None, cols=None):
    """fp32 host copy of a weight matrix / bias zero-padded to [rows, cols] / [rows] (None: keep)."""
    t = t.detach().to("cpu", torch.float32)
    shape = list(t.shape)
    if rows is not None:
        shape[0] = rows
    if cols is not None and t.dim() == 2:
        shape[1] = cols
    if shape == list(t.shape):
        return t.contiguous()
    out = torch.zeros(shape, dtype=torch.float32)
    out[tuple(slice(0, n) for n in t.shape)] = t
    return out


def _filter_weights(fn):
    """(f0_w [128,R], f0_b [128], f1_w [128,128]) in nn.Linear layout and whether the module is the W16A16
    twin (reference models/gptq.py:92-130, weights stored transposed in fp16)."""
    from .gptq import GPTQW16A16FilterNetwork

    if isinstance(fn, GPTQW16A16FilterNetwork):
        return fn.weight0.t().float(), fn.bias0.float(), fn.weight1.t().float(), True
    if not hasattr(fn, "layers"):
        raise ValueError("fused engine: filter network must be an MLP or its W16A16 twin")
    fl = _linears(fn)
    if len(fl) != 2 or fl[1].bias is not None or fl[0].bias is None:
        raise ValueError("fused engine: filter network must be Linear(R,128)+tanh+Linear(128,128,no bias)")
    return fl[0].weight, fl[0].bias, fl[1].weight, False


def _output_weights(on):
    """[(w [out,in], b | None), ...] and whether the module is the W16A16 twin (models/gptq.py:266-306)."""
    from .gptq import GPTQW16A16OutputNetwork

    if isinstance(on, GPTQW16A16OutputNetwork):
        return [(on.weight0.t().float(), on.bias0.float()), (on.weight1.t().float(), on.bias1.float()),
                (on.weight2.t().float(), None)], True
    if not hasattr(on, "layers"):
        raise ValueError("fused engine: output network must be an MLP or its W16A16 twin")
    return [(m.weight, m.bias) for m in _linears(on)], False


class FusedSchNetEngine:
    """Packed weights + workspace for one model on one device."""

    def __init__(self, model, n_split: int = 0, precision: Optional[str] = None, filter_stash: bool = True,
                 molecule_groups: bool = True):
        """``precision``: "f32" (default) or "w16a16"; a model whose filter and output networks
        were swapped by ``apply_gptq_w16a16_to_model`` selects "w16a16" by itself.  ``filter_stash=False``: the
        backward pair kernels recompute filter layer 1 instead of reading the values the forward kernels kept
        (FMD_FLAG_NO_FILTER_STASH: smaller workspace, slower, same bits).  ``molecule_groups=False``
        (FMD_FLAG_NO_MOLECULE_GROUPS): batches of more molecules than CUs whose molecules have at most 64 beads keep one
        pair list and one work item per molecule instead of per group of up to 32 molecules / 256 beads (slower; a
        molecule's fp32 sums then do not depend on its neighbours in the batch)."""
        from .schnet import CosineCutoff, GaussianBasis

        rbf = model.rbf_layer
        blocks = list(model.interaction_blocks)
        if not isinstance(rbf, GaussianBasis) or not isinstance(rbf.cutoff, CosineCutoff):
            raise ValueError("fused engine: needs GaussianBasis with CosineCutoff")
        if rbf.cutoff.cutoff_lower != 0 or rbf.num_rbf > MAX_RBF or rbf.num_rbf < 2:
            raise ValueError(f"fused engine: cutoff_lower must be 0 and 2 <= num_rbf <= {MAX_RBF}")
        hidden = model.embedding_layer.embedding_dim
        for blk in blocks:
            cut = blk.conv.cutoff
            if not isinstance(cut, CosineCutoff) or cut.cutoff_lower != 0 \
                    or cut.cutoff_upper != rbf.cutoff.cutoff_upper:
                raise ValueError("fused engine: CFConv cutoff must equal the basis cutoff")
            if not isinstance(blk.activation, torch.nn.Tanh):
                raise ValueError("fused engine: tanh activation required")
        filt = [_filter_weights(blk.conv.filter_network) for blk in blocks]
        n_filters = int(filt[0][0].shape[0])
        if hidden > WIDTH or n_filters > WIDTH or hidden < 1 or n_filters < 1:
            raise ValueError(f"fused engine: hidden ({hidden}) and filters ({n_filters}) must be <= {WIDTH}")
        for (f0_w, f0_b, f1_w, _), blk in zip(filt, blocks):
            if tuple(f0_w.shape) != (n_filters, rbf.num_rbf) or tuple(f1_w.shape) != (n_filters, n_filters) \
                    or tuple(blk.conv.lin1.weight.shape) != (n_filters, hidden) \
                    or tuple(blk.conv.lin2.weight.shape) != (hidden, n_filters) \
                    or tuple(blk.lin.weight.shape) != (hidden, hidden):
                raise ValueError("fused engine: a block must be lin1(hidden, filters) / Linear(R, filters)+tanh+"
                                 "Linear(filters, filters, no bias) / lin2(filters, hidden) / lin(hidden, hidden)")
        outs, out_half = _output_weights(model.output_network)
        if len(outs) not in (2, 3) or outs[-1][0].shape[0] != 1 or outs[-1][1] is not None \
                or any(w_.shape[0] > 128 for w_, _ in outs) or outs[0][0].shape[1] != hidden:
            raise ValueError("fused engine: output MLP must have 2-3 linears, widths <= 128, last -> 1 without bias")
        halves = [q for *_, q in filt] + [out_half]
        if any(halves):
            if not all(halves):
                raise ValueError("fused engine: W16A16 needs the filter AND output networks swapped")
            if precision not in (None, "w16a16"):
                raise ValueError("fused engine: the model carries W16A16 networks; precision must be 'w16a16'")
            precision = "w16a16"
        precision = precision or "f32"

        self.device = model.embedding_layer.weight.device
        self.cutoff = float(rbf.cutoff.cutoff_upper)
        # the largest molecule this engine takes (round 6: a basis of 65..128 Gaussians has a feature-split mode as well)
        self.max_beads = MAX_BEADS
        d = _ffi.SchnetDesc()
        d.hidden, d.num_rbf, d.num_blocks, d.num_out = 128, rbf.num_rbf, len(blocks), len(outs)
        for k in range(4):
            d.out_width[k] = outs[k][0].shape[0] if k < len(outs) else 0
        d.embedding_size = model.embedding_layer.num_embeddings
        if precision not in PRECISIONS:
            raise ValueError(f"fused engine: precision must be one of {sorted(PRECISIONS)}")
        d.precision, d.n_split = PRECISIONS[precision], int(n_split)
        d.cutoff, d.gamma = self.cutoff, float(rbf.coeff.item())
        d.flags = (0 if filter_stash else _ffi.FLAG_NO_FILTER_STASH) | (0 if molecule_groups else _ffi.FLAG_NO_MOLECULE_GROUPS)
        self._flags = d.flags   # as requested; _workspace() may add FMD_FLAG_NO_FILTER_STASH for a very large batch
        self.desc = d

        # host copies of the weights (fp32, contiguous) -> packed register-order buffer -> device
        keep = []

        def host(t):
            h = t.detach().to("cpu", torch.float32).contiguous()
            keep.append(h)
            return ctypes.c_void_p(h.data_ptr())

        # narrower layers are zero-padded to the kernels' 128 x 128: a padded feature is exactly 0 everywhere (zero
        # embedding column -> zero lin1 row -> tanh(0 + 0) = 0 in the filter -> zero message -> zero lin2 / lin rows and
        # biases), so energies and forces are those of the unpadded model
        bw = (_ffi.SchnetBlockWeights * len(blocks))()
        for l, blk in enumerate(blocks):
            bw[l].lin1_w = host(_pad(blk.conv.lin1.weight, WIDTH, WIDTH))
            bw[l].f0_w = host(_pad(filt[l][0], WIDTH, None))
            bw[l].f0_b = host(_pad(filt[l][1], WIDTH))
            bw[l].f1_w = host(_pad(filt[l][2], WIDTH, WIDTH))
            bw[l].lin2_w, bw[l].lin2_b = host(_pad(blk.conv.lin2.weight, WIDTH, WIDTH)), host(_pad(blk.conv.lin2.bias, WIDTH))
            bw[l].lin_w, bw[l].lin_b = host(_pad(blk.lin.weight, WIDTH, WIDTH)), host(_pad(blk.lin.bias, WIDTH))
        w = _ffi.SchnetWeights()
        w.embedding = host(_pad(model.embedding_layer.weight, None, WIDTH))
        w.blocks = ctypes.cast(bw, ctypes.POINTER(_ffi.SchnetBlockWeights))
        for k, (ow, ob) in enumerate(outs):
            w.out_w[k] = host(_pad(ow, None, WIDTH) if k == 0 else ow)
            w.out_b[k] = host(ob) if ob is not None else None
        centers = host(rbf.offset)
        nbytes = _ffi.lib().fmd_schnet_pack_bytes(ctypes.byref(d))
        if nbytes <= 0:
            raise ValueError("fused engine: model rejected by fmd_schnet_pack_bytes")
        packed = torch.empty(nbytes // 4, dtype=torch.float32)
        call("fmd_schnet_pack", ctypes.byref(d), ctypes.byref(w), centers, ctypes.c_void_p(packed.data_ptr()))
        self.packed = packed.to(self.device)
        self._ws = None
        self._ws_key = None
        self.status = torch.zeros(4, dtype=torch.int32, device=self.device)   # [3] is sticky: zeroed here and on read
        self._bound_key = None      # layout whose caller-supplied bead bound has been validated ...
        self._bound_ref = None      # ... and the tensor itself: while it is alive its address cannot be handed out again
        self._since_check = 0       # force evaluations since the status words were last read
        self.next_events = None     # measurement hook (bench.py): events for the NEXT evaluation, whatever entry point makes it

    @classmethod
    def from_state_dict(cls, sd, device="cuda", prefix: Optional[str] = None, cutoff: Optional[float] = None, **kw):
        """Engine from a ``state_dict`` keyed by the reference's parameter names (a dict, an ``.npz`` or a
        weights-only ``torch.load`` file; ``prefix=None`` finds the SchNet inside a ``GradientsOut`` / ``SumOut``
        container; ``cutoff``: see ``schnet.model_from_state_dict`` -- hyper-parameters the weights do not encode).
        Raises ValueError when the model is outside the engine's envelope (hidden, filters <= 128,
        2 <= num_rbf <= 128, 2-3 output linears)."""
        from .schnet import model_from_state_dict

        return cls(model_from_state_dict(sd, prefix, cutoff).to(device).eval(), **kw)

    def _workspace(self, n_mol, n_nodes, cap):
        key = (n_mol, n_nodes, cap)
        if self._ws_key != key:
            # the filter stash is sized for the worst-case pair list (n_mol * cap * 512 bytes per block but the first):
            # batches for which that would be unreasonable fall back to recomputation in the backward pass
            # (a fixed ceiling, and never more than half of what the device has free right now, the current workspace --
            # about to be released -- counted as free; an allocation that fails all the same is retried once without)
            self.desc.flags = self._flags
            self._ws, self._ws_key = None, None   # (a failed allocation must leave the engine able to allocate again)
            wants_stash = self.desc.precision == PRECISIONS["f32"] and not (self._flags & _ffi.FLAG_NO_FILTER_STASH)
            stash = (self.desc.num_blocks - 1) * n_mol * cap * 512
            # (free = what the driver reports plus what torch's caching allocator holds unused: the old workspace is there)
            free = torch.cuda.mem_get_info(self.device)[0] + torch.cuda.memory_reserved(self.device) \
                - torch.cuda.memory_allocated(self.device)
            if wants_stash and stash > min(MAX_STASH_BYTES, free // 2):
                self.desc.flags = self._flags | _ffi.FLAG_NO_FILTER_STASH

            def alloc():
                nbytes = _ffi.lib().fmd_schnet_workspace_bytes(ctypes.byref(self.desc), n_mol, n_nodes, cap)
                if nbytes <= 0:
                    raise RuntimeError("fmd_schnet_workspace_bytes failed")
                return torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)

            try:
                self._ws = alloc()
            except torch.cuda.OutOfMemoryError:
                if not wants_stash or (self.desc.flags & _ffi.FLAG_NO_FILTER_STASH):
                    raise
                self.desc.flags = self._flags | _ffi.FLAG_NO_FILTER_STASH
                self._ws = alloc()
            self._ws_key = key
        off = (-self._ws.data_ptr()) % 256
        return ctypes.c_void_p(self._ws.data_ptr() + off)

    def kernel_names(self):
        lib = _ffi.lib()
        n = lib.fmd_schnet_kernel_count(ctypes.byref(self.desc))
        return [lib.fmd_schnet_kernel_name(ctypes.byref(self.desc), k).decode() for k in range(n)]

    def energy_forces(self, pos, atom_types, mol_ptr, max_beads_per_mol: int = 0, events=None):
        """energy[B] f32, forces[N,3] f32 of the batch; everything enqueued on the current stream.
        ``events``: optional ctypes array of 2*len(kernel_names()) hipEvent_t handles recorded
        before/after each kernel (measurement only)."""
        _ffi.require_gpu(pos, atom_types, mol_ptr)
        pos = pos.detach()
        assert pos.dtype == torch.float32 and pos.is_contiguous() and atom_types.dtype == torch.int64
        n_nodes, n_mol = pos.shape[0], mol_ptr.numel() - 1
        # a caller-supplied bound is validated once per batch layout (one host sync); without a bound the true
        # maximum is computed on every call (one sync each: pass the bound in a time loop).  The validated tensor is
        # kept alive, so an equal address means the same memory, and its version counter (absent under
        # inference_mode: such tensors are re-validated every call) catches in-place edits; what neither sees -- a raw
        # write through another view -- is caught by the device-side check, whose verdict lands in the sticky status
        # word and is never lost between two reads.
        try:
            version = mol_ptr._version
        except RuntimeError:
            version = None
        key = (mol_ptr.data_ptr(), mol_ptr.numel(), version, int(max_beads_per_mol))
        fresh = version is None or key != self._bound_key or self._bound_ref is None
        if max_beads_per_mol <= 0 or fresh:
            true_max = int((mol_ptr[1:] - mol_ptr[:-1]).max().item()) if n_mol else 1
            if max_beads_per_mol <= 0:
                max_beads_per_mol = true_max
            elif true_max > max_beads_per_mol:
                raise ValueError(f"fused engine: max_beads_per_mol={max_beads_per_mol} but the batch holds a molecule "
                                 f"of {true_max} beads")
            self._bound_key, self._bound_ref = key, mol_ptr
        if max_beads_per_mol > self.max_beads:
            raise RuntimeError(f"fused engine: molecule of {max_beads_per_mol} beads > {self.max_beads}")
        # every pair plus <= 7 pad entries per centre (8-entry segments): can never overflow
        a_ = max_beads_per_mol
        cap = max(32, (a_ * (a_ - 1) // 2 + 8 * a_ + 31) // 32 * 32)
        energy = torch.empty(n_mol, dtype=torch.float32, device=pos.device)
        forces = torch.empty(n_nodes, 3, dtype=torch.float32, device=pos.device)
        args = (ctypes.byref(self.desc), ptr(self.packed), ptr(pos), ptr(atom_types), ptr(mol_ptr), n_mol,
                n_nodes, max_beads_per_mol, cap, ptr(energy), ptr(forces),
                self._workspace(n_mol, n_nodes, cap), ptr(self.status), stream())
        if events is None and self.next_events is not None:
            events, self.next_events = self.next_events, None
        if events is None:
            call("fmd_schnet_energy_forces", *args)
        else:
            call("fmd_schnet_energy_forces_timed", *args, events, len(events))
        # the device status words: read on the first evaluation of a batch layout and then every
        # STATUS_CHECK_EVERY evaluations (callers with a natural sync point -- save / export / exchange
        # intervals -- call check_status() there as well)
        if fresh or self._since_check >= STATUS_CHECK_EVERY:
            self.check_status()
        else:
            self._since_check += 1
        return energy, forces

    def check_status(self):
        """Read status[0..3] (one device sync) and raise if the last evaluation -- or, through the sticky word [3], ANY
        evaluation since the previous read -- flagged a problem (include/flashmd_hip.h: FMD_ECAPACITY = pair list too
        small, FMD_EINVAL = a molecule larger than max_beads_per_mol was truncated, FMD_ENONFINITE = a force
        contribution was Inf / NaN: the results of that call are meaningless)."""
        self._since_check = 0
        st = self.status.tolist()
        if st[3] != 0:
            self.status[3] = 0
        code = st[0] or st[3]
        if code != 0:
            when = "the last evaluation" if st[0] else "an evaluation since the previous status read"
            raise RuntimeError(f"fused engine: device status {code} ({_ffi.strerror(code)}) in {when}; max list entries "
                               f"needed by one molecule (or molecule group) {st[1]}, total pairs {st[2]}: energies and forces "
                               "of that call are invalid")
        return st


def maybe_engine(model, data):
    """Engine for ``model`` if it qualifies (cached on the model), else None (operator path)."""
    if not data.pos.is_cuda:
        raise RuntimeError(
            "flashmd_amd: the HIP gates are on but the batch is on the CPU. There is no CPU fallback "
            "for the optimized path; run with --disable_optim (all MLCG_USE_* = 0).")
    eng = getattr(model, "_engine", None)
    if eng is False:
        return None
    if eng is None:
        try:
            eng = FusedSchNetEngine(model)
        except ValueError:
            model._engine = False
            return None
        model._engine = eng
    mb = int(getattr(data, "max_beads_per_mol", 0))
    if mb <= 0:
        ptr_ = data.ptr
        mb = int((ptr_[1:] - ptr_[:-1]).max().item())
        data.max_beads_per_mol = mb
    if mb > eng.max_beads:
        return None
    # the engine always builds the full within-cutoff graph: a model whose max_num_neighbors could truncate it, or
    # a batch that brings its own compatible neighbour list (reference models/schnet.py:216-245, 371-380), goes to
    # the operator path, which honours both
    if int(getattr(model, "max_num_neighbors", 1000)) < mb - 1:
        return None
    nl = getattr(data, "neighbor_list", None)
    if nl and model.is_nl_compatible(nl.get(model.name)):
        return None
    return eng
