"""CPU, world_size 2 over gloo: the N > 1 paths -- replica sharding (no collective in the force
path) and the parallel-tempering exchange (point-to-point send/recv) -- with every MLCG_USE_* gate off
so the PyTorch path supplies the forces."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys, math, torch
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "flash-molecular-dynamics_amd")]
import torch.distributed as dist
from oracle import schnet_oracle as so
import flashmd_amd.schnet as S
from flashmd_amd import simulate

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.manual_seed(0)
p = so.make_params(num_interactions=1, out_widths=(32,), seed=1, bias_scale=0.1)
model = S.GradientsOut(S.model_from_params(p)).eval()
A, B = 12, 4                      # beads per molecule, molecules in the global batch
pos, types, ptr = so.synth_batch(B, A, 5.0)
masses = torch.full((B * A,), 12.0)

def force_fn_for(types_, ptr_):
    def f(x):
        data = model(S.Batch(x.clone(), types_, ptr_))
        return data.out["SchNet"]["energy"].detach(), data.out["SchNet"]["forces"].detach()
    return f

mode = %(mode)r
if mode == "shard":
    # every rank integrates its own contiguous block of molecules; rank 0 also integrates all of
    # them and the blocks must agree exactly (replicas never interact)
    lo, hi = simulate.shard_range(B, rank, world)
    sl = slice(lo * A, hi * A)
    ptr_l = torch.arange(hi - lo + 1) * A
    sim = simulate.LangevinSimulation(force_fn_for(types[sl], ptr_l), pos[sl], masses[sl],
                                      torch.full((hi - lo,), 1.67), A, seed=7)
    sim.v.zero_(); noise_free = True
    sim.friction = 1e-9            # ~ deterministic dynamics: noise scale ~ 0
    for _ in range(3): sim.timestep()
    gathered = [torch.zeros(0)] * world
    xs = [torch.empty((simulate.shard_range(B, r, world)[1] - simulate.shard_range(B, r, world)[0]) * A, 3) for r in range(world)]
    dist.all_gather(xs, sim.x) if len(set(t.shape for t in xs)) == 1 else None
    if rank == 0:
        full = simulate.LangevinSimulation(force_fn_for(types, ptr), pos, masses, torch.full((B,), 1.67), A, seed=7)
        full.v.zero_(); full.friction = 1e-9
        for _ in range(3): full.timestep()
        err = (torch.cat(xs) - full.x).abs().max().item()
        print("SHARD_ERR", err)
elif mode == "pt":
    betas = [1.67, 1.16]
    n_indep = 2
    sl = slice(0, n_indep * A)     # both ranks start from the same two molecules
    ptr_l = torch.arange(n_indep + 1) * A
    ff = force_fn_for(types[sl], ptr_l)
    sim = simulate.LangevinSimulation(ff, pos[sl] + 0.05 * rank, masses[sl], torch.full((n_indep,), betas[rank]), A, seed=11 + rank)
    ex = simulate.DistributedReplicaExchange(betas, n_indep, A, seed=5)
    x0, v0, u0 = sim.x.clone(), sim.v.clone(), sim.energy.clone()
    allx = [torch.empty_like(x0) for _ in range(world)]; dist.all_gather(allx, x0)
    allv = [torch.empty_like(v0) for _ in range(world)]; dist.all_gather(allv, v0)
    allu = [torch.empty_like(u0) for _ in range(world)]; dist.all_gather(allu, u0)
    ex(sim)                        # attempt 0: even pairs -> (0, 1)
    # expected result from the gathered pre-exchange state, same random numbers
    g = torch.Generator().manual_seed((5 * 1000003 + 0) * 4099 + 0)
    rand_u = torch.rand(n_indep, generator=g)
    acc = so.pt_exchange_decisions(allu[0], allu[1], betas[0], betas[1], rand_u)
    peer = 1 - rank
    mask = acc.repeat_interleave(A)[:, None]
    x_exp = torch.where(mask, allx[peer], allx[rank])
    v_exp = torch.where(mask, allv[peer] * math.sqrt(betas[rank] / betas[peer]), allv[rank])
    ok = torch.equal(sim.x, x_exp) and torch.allclose(sim.v, v_exp)
    a_, t_ = ex.acceptance()
    # one write per rank: the two ranks share a pipe and print() of several arguments can interleave
    sys.stdout.write(f"PT_OK {int(ok)} {int(acc.sum())} {int(a_[0])} {int(t_[0])}\n")
    sys.stdout.flush()
    ex(sim)                        # attempt 1: odd pairs -> nobody has a partner with 2 ranks
    assert ex.partner(False) is None
dist.barrier()
dist.destroy_process_group()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(mode):
    env = dict(os.environ, MLCG_USE_TRITON_MESSAGE_PASSING="0", MLCG_USE_FUSED_RBF="0",
               MLCG_USE_FUSED_TANH_LINEAR="0", MLCG_USE_CSR="0", MLCG_USE_SRC_CSR_GRAD_X="0",
               OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "-c",
           _WORKER % {"root": ROOT, "mode": mode}]
    # torch.distributed.run has no -c: write the worker to a temp file instead
    import tempfile

    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as fh:
        fh.write(_WORKER % {"root": ROOT, "mode": mode})
        path = fh.name
    cmd = cmd[:-2] + [path]
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    finally:
        os.unlink(path)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    return out.stdout


def test_replica_sharding_two_ranks_matches_single_process():
    out = _run("shard")
    err = float(out.split("SHARD_ERR")[1].split()[0])
    # same molecules, same dynamics; only the fp32 GEMM blocking differs with the batch size
    assert err < 1e-5


def test_parallel_tempering_exchange_two_ranks():
    out = _run("pt")
    import re

    recs = re.findall(r"PT_OK (\d+) (\d+) (\d+) (\d+)", out)
    assert len(recs) == 2, out[-500:]
    for rec in recs:
        ok, n_acc, acc0, att0 = (int(v) for v in rec)
        assert ok == 1 and acc0 == n_acc and att0 == 2
