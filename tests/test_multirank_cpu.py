"""CPU, world_size 2 and 4 over gloo: the N > 1 paths -- replica sharding (no collective in the force
path) and the parallel-tempering exchange (point-to-point send/recv; even phase, odd phase with idle end
ranks, the reference-shaped acceptance matrix) -- with every MLCG_USE_* gate off so the PyTorch path supplies
the forces."""
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
    ex = simulate.DistributedReplicaExchange(betas, n_indep, A, seed=5, forces="moved")
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
    e_chk, f_chk = sim.force_fn(sim.x)   # forces and energies moved with the configurations
    ok = ok and torch.allclose(sim.f, f_chk, rtol=1e-4, atol=1e-5) and torch.allclose(sim.energy, e_chk, rtol=1e-5, atol=1e-5)
    a_, t_ = ex.acceptance()
    # one write per rank: the two ranks share a pipe and print() of several arguments can interleave
    sys.stdout.write(f"PT_OK {int(ok)} {int(acc.sum())} {int(a_[0])} {int(t_[0])}\n")
    sys.stdout.flush()
    # two temperatures: there is no odd pair and the reference proposes (0, 1) again (parallel_tempering.py:262-263)
    assert ex.partner(False) == peer
elif mode in ("pt4", "pt8"):
    # four temperatures, one per rank: attempt 0 pairs (0,1),(2,3); attempt 1 pairs (1,2) while ranks 0 and 3 sit out
    # (reference parallel_tempering.py:256-284,349-366); attempt 2 is even again.  Every attempt is checked against
    # the expectation computed from the gathered pre-exchange state, and the acceptance matrix against the
    # reference's bookkeeping (:99-104,407-412): accepted above the diagonal, rejected below.
    # pt8: BASELINE configs[4]'s shape -- the ladder of EIGHT temperatures bench.py --pt --gpus 8 uses, one per rank,
    # four attempts (even, odd, even, odd): (0,1)(2,3)(4,5)(6,7) then (1,2)(3,4)(5,6) with ranks 0 and 7 idle.
    if mode == "pt8":
        sys.path.insert(0, %(root)r)
        import bench
        betas = bench.beta_ladder(8)
        assert world == 8 and abs(betas[1] - 1.42) < 0.03 and abs(betas[2] - 1.16) < 1e-6
    else:
        betas = [1.67, 1.42, 1.16, 0.95]
    n_attempts = 4 if mode == "pt8" else 3
    n_indep = 3
    sl = slice(0, n_indep * A)
    ptr_l = torch.arange(n_indep + 1) * A
    ff = force_fn_for(types[sl], ptr_l)
    sim = simulate.LangevinSimulation(ff, pos[sl] + 0.03 * rank, masses[sl], torch.full((n_indep,), betas[rank]), A, seed=11 + rank)
    ex = simulate.DistributedReplicaExchange(betas, n_indep, A, seed=5, forces="moved")
    expected = torch.zeros(world, world, dtype=torch.int64)
    ok, evals = True, 0
    counted = [0]
    inner = sim.force_fn
    def counting(x):
        counted[0] += 1
        return inner(x)
    sim.force_fn = counting
    for attempt in range(n_attempts):
        for _ in range(2): sim.timestep()
        x0, v0, u0 = sim.x.clone(), sim.v.clone(), sim.energy.clone()
        allx = [torch.empty_like(x0) for _ in range(world)]; dist.all_gather(allx, x0)
        allv = [torch.empty_like(v0) for _ in range(world)]; dist.all_gather(allv, v0)
        allu = [torch.empty_like(u0) for _ in range(world)]; dist.all_gather(allu, u0)
        before = counted[0]
        ex(sim)
        pairs = simulate.pt_pairs(world, attempt %% 2 == 0)
        assert pairs == ([(k, k + 1) for k in range(0, world - 1, 2)] if attempt %% 2 == 0 else
                         [(k, k + 1) for k in range(1, world - 1, 2)])
        x_exp, v_exp, moved = allx[rank], allv[rank], False
        for a, b in pairs:
            g = torch.Generator().manual_seed((5 * 1000003 + attempt) * 4099 + a)
            acc = so.pt_exchange_decisions(allu[a], allu[b], betas[a], betas[b], torch.rand(n_indep, generator=g))
            expected[a, b] += int(acc.sum()); expected[b, a] += n_indep - int(acc.sum())
            if rank in (a, b):
                peer = b if rank == a else a
                mask = acc.repeat_interleave(A)[:, None]
                x_exp = torch.where(mask, allx[peer], allx[rank])
                v_exp = torch.where(mask, allv[peer] * math.sqrt(betas[rank] / betas[peer]), allv[rank])
                moved = bool(acc.any())
        ok = ok and torch.equal(sim.x, x_exp) and torch.allclose(sim.v, v_exp)
        # an exchange never evaluates forces: they (and the energies) travel with their configuration ...
        ok = ok and (counted[0] - before == 0)
        # ... and are the forces AT the new positions
        e_chk, f_chk = inner(sim.x)
        ok = ok and torch.allclose(sim.f, f_chk, rtol=1e-4, atol=1e-5) and torch.allclose(sim.energy, e_chk, rtol=1e-5, atol=1e-5)
        ok = ok and (moved or torch.equal(sim.x, allx[rank]))
    m = ex.acceptance_matrix()
    n_even, n_odd = (n_attempts + 1) // 2, n_attempts // 2
    ok = ok and torch.equal(m, expected) and int(m.sum()) == n_indep * (n_even * (world // 2) + n_odd * ((world - 1) // 2))
    ok = ok and int(torch.triu(m, 2).sum() + torch.tril(m, -2).sum()) == 0   # only ladder neighbours ever exchange
    sys.stdout.write(f"PT4_OK {int(ok)} {rank} {int(m[0, 1] + m[1, 0])} {int(m[1, 2] + m[2, 1])}\n")
    sys.stdout.flush()
dist.barrier()
dist.destroy_process_group()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(mode, nproc=2):
    env = dict(os.environ, MLCG_USE_TRITON_MESSAGE_PASSING="0", MLCG_USE_FUSED_RBF="0",
               MLCG_USE_FUSED_TANH_LINEAR="0", MLCG_USE_CSR="0", MLCG_USE_SRC_CSR_GRAD_X="0",
               OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
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


def test_parallel_tempering_exchange_four_ranks_odd_phase_and_matrix():
    out = _run("pt4", nproc=4)
    import re

    recs = re.findall(r"PT4_OK (\d+) (\d+) (\d+) (\d+)", out)
    assert len(recs) == 4, out[-800:]
    for ok, rank, n01, n12 in recs:
        assert int(ok) == 1, f"rank {rank}"
        assert int(n01) == 6 and int(n12) == 3   # (0,1): two even attempts x 3 replicas; (1,2): one odd attempt


def test_parallel_tempering_exchange_eight_ranks_the_bench_ladder():
    """BASELINE configs[4] without the hardware: eight ranks over gloo, the ladder of eight temperatures that
    `bench.py --pt --gpus 8` builds, both phases twice; every attempt against the expectation from the gathered
    pre-exchange state, the acceptance matrix against the reference's bookkeeping
    (simulation/parallel_tempering.py:99-104,407-412), idle end ranks in the odd phase, no force evaluation in an exchange
    (forces and energies travel with their configuration)."""
    out = _run("pt8", nproc=8)
    import re

    recs = re.findall(r"PT4_OK (\d+) (\d+) (\d+) (\d+)", out)
    assert len(recs) == 8, out[-800:]
    assert sorted(int(r[1]) for r in recs) == list(range(8))
    for ok, rank, n01, n12 in recs:
        assert int(ok) == 1, f"rank {rank}"
        assert int(n01) == 6 and int(n12) == 6   # two even attempts on (0,1), two odd attempts on (1,2), 3 replicas each
