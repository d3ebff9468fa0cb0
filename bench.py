"""Benchmark of the hot path: timestep*mol/s of CGSchNet Langevin dynamics on synthetic 1ENH-shaped
batches (BASELINE.json metric), one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: launched by torch.distributed.run)

A step is one BAOAB Langevin timestep of the rank's batch: fused elementwise half, the fused HIP
force engine (neighbour pairs -> SchNet energy -> forces), second half-kick.  Inputs are resident in
HBM before the timed region.  Replicas shard across ranks with no collective in the data path (weak
scaling: every rank owns --batch molecules).  Rank 0 prints ONE JSON line.

  --pt   BASELINE configs[4]: parallel tempering, one beta per rank (geometric ladder down from 1.67, reference
         examples/parallel_tempering.yaml:19), an exchange attempt every --exchange-interval steps over
         point-to-point RCCL send/recv; with --gpus 1 the --pt-betas temperatures share the batch and the swap is
         the in-batch HIP kernel (the reference's own layout).  The line then also carries exchange time and the
         acceptance per temperature pair.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "flash-molecular-dynamics_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

PUBLISHED_BASELINE = 2861.0  # timestep*mol/s, reference README.md:76-81 (1ENH, batch 128, RTX PRO 6000)
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters (fp32 matrix, dense)
PEAK_F16_MFMA_TFLOPS = 16 * PEAK_F32_MFMA_TFLOPS  # same table: fp32 matrix rate = 1/16 of the f16/bf16 rate (~2.5 PF dense)
# The pair kernels compute fp32-grade products as three f16 MFMA products of split operands
# (hi.hi + hi.lo + lo.hi, fp32 accumulate): the roof for ALGORITHMIC flops is a third of the f16 peak.
PEAK_SPLIT3_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128, help="molecules per GPU")
    ap.add_argument("--beads", type=int, default=269)
    ap.add_argument("--num-rbf", type=int, default=50)
    ap.add_argument("--blocks", type=int, default=3)
    ap.add_argument("--precision", default="f32", choices=["f32", "w16a16"],
                    help="f32 = BASELINE configs[1] (default, the judged line); w16a16 = configs[2]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reset-every", type=int, default=0,
                    help="restore positions, velocities and forces of the start of the timed region every this many "
                         "steps (inside the timed region): a long run then repeats the specified workload instead of "
                         "following the synthetic system's relaxation; 0 = never (the default)")
    ap.add_argument("--settle-seconds", type=float, default=0.75,
                    help="untimed force evaluations at fixed positions for this long (on top of --warmup) before the "
                         "timed region: a GPU that has just been woken holds lower clocks for the first tenths of a "
                         "second; 0 = none")
    ap.add_argument("--pt", action="store_true", help="parallel tempering (BASELINE configs[4])")
    ap.add_argument("--exchange-interval", type=int, default=100, help="reference examples/parallel_tempering.yaml:11")
    ap.add_argument("--pt-betas", type=int, default=8, help="temperatures in the batch when --pt runs on one GPU")
    return ap.parse_args()


def beta_ladder(n):
    """examples/parallel_tempering.yaml:19 is [1.67, 1.42, 1.16]: a geometric ladder (ratio 0.8334) continued to n"""
    r = (1.16 / 1.67) ** 0.5
    return [1.67 * r ** k for k in range(n)]


class HipEvents:
    """Timing events created through the HIP runtime (recorded by the C ABI on torch's stream)."""

    def __init__(self, n):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.ev = (ctypes.c_void_p * n)()
        for i in range(n):
            h = ctypes.c_void_p()
            assert self.hip.hipEventCreate(ctypes.byref(h)) == 0
            self.ev[i] = h

    def elapsed_ms(self, i, j):
        ms = ctypes.c_float()
        rc = self.hip.hipEventElapsedTime(ctypes.byref(ms), ctypes.c_void_p(self.ev[i]), ctypes.c_void_p(self.ev[j]))
        assert rc == 0, rc
        return ms.value


def cpu_baseline(model, args):
    """The CPU oracle (a port of the reference's --disable_optim PyTorch path) timed on this node's host cores at
    BASELINE configs[0]'s size (B = 4) and at the GPU line's size (B = --batch), >= 3 force evaluations each after a
    warm-up, same throughput definition."""
    from oracle import schnet_oracle as so
    from flashmd_amd import synthetic

    p = {"hidden": 128, "filters": 128, "num_rbf": args.num_rbf, "cutoff": 5.0,
         "emb": model.embedding_layer.weight.detach().cpu(), "blocks": [], "out": []}
    for blk in model.interaction_blocks:
        fl = [m for m in blk.conv.filter_network.layers if isinstance(m, torch.nn.Linear)]
        p["blocks"].append({k: v.detach().cpu() for k, v in dict(
            lin1_w=blk.conv.lin1.weight, f0_w=fl[0].weight, f0_b=fl[0].bias, f1_w=fl[1].weight,
            lin2_w=blk.conv.lin2.weight, lin2_b=blk.conv.lin2.bias, lin_w=blk.lin.weight,
            lin_b=blk.lin.bias).items()})
    for m in model.output_network.layers:
        if isinstance(m, torch.nn.Linear):
            p["out"].append({"w": m.weight.detach().cpu(),
                             "b": None if m.bias is None else m.bias.detach().cpu()})
    # the GPU box gives one GPU a 16-core share of the host (os.cpu_count() reports the whole machine)
    threads = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(threads)

    def run(n_mol, reps):
        pos, types, ptr, masses = synthetic.batch(n_mol, args.beads, 5.0)
        t0 = time.perf_counter()
        for _ in range(reps):
            e, f = so.schnet_energy_forces(p, pos, types, ptr)
            so.langevin_baoab_post(f, f, masses, 0.004)  # integrator cost is negligible; keep a token of it
        return reps * n_mol / (time.perf_counter() - t0), time.perf_counter() - t0

    run(4, 1)  # warm-up (thread pool, allocator)
    v4, t4 = run(4, 5)
    vb, tb = run(args.batch, 3)
    return {"value": vb, "unit": "timestep*mol/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"3 force evaluations of {args.batch} molecules x {args.beads} beads after a warm-up "
                      f"(oracle/schnet_oracle.py, fp32, {tb:.1f} s)",
            "b4": {"value": v4, "sample": f"5 force evaluations of 4 molecules x {args.beads} beads ({t4:.1f} s): "
                                          "the size of BASELINE configs[0]"}}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs the GPU (the HIP path has no CPU fallback)"
    # FMD_BENCH_REHEARSE=1: rehearsal of the N > 1 plumbing on a box with fewer GPUs than ranks (ranks share
    # devices, gloo instead of RCCL); never used for a reported number
    rehearse = os.environ.get("FMD_BENCH_REHEARSE") == "1"
    if rehearse:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist

        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import flashmd_amd.schnet as S
    from flashmd_amd import engine, simulate, synthetic

    torch.manual_seed(0)
    rc = 5.0
    model = S.StandardSchNet(S.GaussianBasis(S.CosineCutoff(0.0, rc), args.num_rbf), S.CosineCutoff(0.0, rc),
                             [128, 64], num_interactions=args.blocks)
    with torch.no_grad():  # the reference zero-initialises biases; use small non-zero ones
        for n_, prm in model.named_parameters():
            if n_.endswith("bias"):
                prm.uniform_(-0.1, 0.1)
    model = model.to(dev).eval()
    eng = engine.FusedSchNetEngine(model, n_split=int(os.environ.get("FMD_NSPLIT", "0")),
                                   precision=args.precision)
    names = eng.kernel_names()

    # every rank owns its own molecules (different seeds per rank): embarrassingly parallel shards
    pos, types, ptr, masses = synthetic.batch(args.batch, args.beads, rc, first_seed=1000 + rank * args.batch)
    pos, types, ptr, masses = pos.to(dev), types.to(dev), ptr.to(dev), masses.to(dev)
    beta = torch.full((args.batch,), 1.67, device=dev)  # reference examples/langevin.yaml:20
    exchange = None
    if args.pt and world > 1:      # one temperature per rank
        betas = beta_ladder(world)
        beta = torch.full((args.batch,), betas[rank], device=dev)
        exchange = simulate.DistributedReplicaExchange(betas, args.batch, args.beads, seed=5)
    elif args.pt:                  # all temperatures in one batch (reference layout, beta-major)
        assert args.batch % args.pt_betas == 0
        betas = beta_ladder(args.pt_betas)
        beta = torch.tensor(betas, device=dev).repeat_interleave(args.batch // args.pt_betas)
        exchange = simulate.LocalReplicaExchange(betas, args.batch // args.pt_betas, args.beads, seed=5)

    # Event plan.  Recording an event pair around all 15 kernels of every step costs ~0.1 ms per step (5 % of the
    # step: measured, tools/event_overhead.py), so: (1) a few untimed profile steps with every kernel bracketed
    # give the per-kernel table and name the dominant kernel; (2) in the timed region only the dominant kernel's
    # launches are bracketed (its live duration is what `roofline.achieved` uses).
    nk = len(names)
    n_prof = 5
    prof_events = HipEvents(2 * nk * n_prof)
    step_no = {"i": -1, "mode": "plain", "mask": None}
    timed_events = None

    def force_fn(x):
        i = step_no["i"]
        if step_no["mode"] == "profile" and 0 <= i < n_prof:
            ev = (ctypes.c_void_p * (2 * nk)).from_buffer(prof_events.ev, i * 2 * nk * ctypes.sizeof(ctypes.c_void_p))
            return eng.energy_forces(x, types, ptr, args.beads, events=ev)
        if step_no["mode"] == "timed" and 0 <= i < args.steps:
            ev = (ctypes.c_void_p * (2 * nk))()
            for j, slot in enumerate(step_no["mask"]):
                ev[2 * slot], ev[2 * slot + 1] = timed_events.ev[(i * n_dom + j) * 2], timed_events.ev[(i * n_dom + j) * 2 + 1]
            return eng.energy_forces(x, types, ptr, args.beads, events=ev)
        return eng.energy_forces(x, types, ptr, args.beads)

    sim = simulate.LangevinSimulation(force_fn, pos, masses, beta, args.beads, dt=0.004, friction=1.0)
    for _ in range(args.warmup):
        sim.timestep()
    if exchange is not None:   # untimed first attempts: RCCL sets up its point-to-point channels on first use
        exchange(sim)
        exchange(sim)
    # (1) untimed profile steps: every kernel bracketed
    step_no["mode"] = "profile"
    for i in range(n_prof):
        step_no["i"] = i
        sim.timestep()
    torch.cuda.synchronize()
    per = {}
    for k, name in enumerate(names):
        per.setdefault(name, []).extend(prof_events.elapsed_ms((i * nk + k) * 2, (i * nk + k) * 2 + 1)
                                        for i in range(1, n_prof))
    launches = {n: len(v) / (n_prof - 1) for n, v in per.items()}
    prof_ms = {n: sum(v) / len(v) for n, v in per.items()}
    dom = max(prof_ms, key=lambda n: prof_ms[n] * launches[n])
    dom_slots = [k for k, name in enumerate(names) if name == dom]
    n_dom = len(dom_slots)
    timed_events = HipEvents(2 * n_dom * args.steps)
    # clocks settle: untimed force evaluations at the CURRENT positions for --settle-seconds (nothing is integrated:
    # the synthetic system is not stationary -- its pair count falls by 0.8 % in 500 steps, 7 % in 3600, 19 % in
    # 20000 -- so steps here would hand the timed region a lighter workload)
    step_no.update(mode="plain", i=-1)
    n_settle = 0
    t_s = time.perf_counter()
    while time.perf_counter() - t_s < args.settle_seconds:
        for _ in range(20):
            force_fn(sim.x)
        torch.cuda.synchronize()
        n_settle += 20
    pairs_start = eng.check_status()[2]   # workload at the start of the timed region
    step_no.update(mode="timed", mask=dom_slots, i=-1)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # sustained-rate check (runs of >= 200 steps): device-synchronised marks after the first and before the last
    # tenth of the timed region (two extra syncs: nothing next to hundreds of steps)
    marks = {}
    tenth = args.steps // 10 if args.steps >= 200 else 0
    t_exchange, n_exchange = 0.0, 0
    if args.reset_every:
        x0, v0, f0, e0 = sim.x.clone(), sim.v.clone(), sim.f.clone(), sim.energy.clone()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step_no["i"] = i
        sim.timestep()
        if args.reset_every and (i + 1) % args.reset_every == 0:   # (three device copies of 0.4 MB, timed)
            sim.x.copy_(x0); sim.v.copy_(v0); sim.f.copy_(f0); sim.energy.copy_(e0)
        if exchange is not None and (i + 1) % args.exchange_interval == 0:
            step_no["i"] = -1
            torch.cuda.synchronize()
            te = time.perf_counter()
            exchange(sim)
            torch.cuda.synchronize()
            t_exchange += time.perf_counter() - te
            n_exchange += 1
        if tenth and i + 1 in (tenth, args.steps - tenth):
            torch.cuda.synchronize()
            marks[i + 1] = time.perf_counter() - t0
    step_no["i"] = -1
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(sim.x).all() and torch.isfinite(sim.f).all(), "simulation diverged"
    st = eng.check_status()
    n_pairs = st[2]
    # the reference-shaped entry point users call: GradientsOut(model)(Batch) picks the same engine; a short
    # run through it right after the timed region (its result is reported beside the line, not as `value`)
    via_go = None
    if not args.pt:
        go = S.GradientsOut(model).eval()
        model._engine = eng

        batch_obj = S.Batch(pos, types, ptr, max_beads_per_mol=args.beads)  # collated once, like the reference's
        # simulation keeps one AtomicData and updates its positions (simulation/base.py:991-1020)

        def force_go(x):
            batch_obj.pos, batch_obj.out = x, {}
            d = go(batch_obj)
            return d.out["SchNet"]["energy"], d.out["SchNet"]["forces"]

        sim.force_fn = force_go
        for _ in range(3):
            sim.timestep()
        torch.cuda.synchronize()
        tg = time.perf_counter()
        n_go = max(10, min(args.steps, 40))
        for _ in range(n_go):
            sim.timestep()
        torch.cuda.synchronize()
        via_go = 1e3 * (time.perf_counter() - tg) / n_go
    pt_info = None
    if exchange is not None:
        m = exchange.acceptance_matrix() if world > 1 else exchange.acceptance_matrix
        nb = m.shape[0]
        pt_info = {"betas": [round(b, 4) for b in (beta_ladder(world) if world > 1 else beta_ladder(args.pt_betas))],
                   "exchange_interval": args.exchange_interval, "attempts_timed": n_exchange,
                   "exchange_ms_per_attempt": 1e3 * t_exchange / max(1, n_exchange),
                   "accepted_per_pair": [int(m[k, k + 1]) for k in range(nb - 1)],
                   "rejected_per_pair": [int(m[k + 1, k]) for k in range(nb - 1)],
                   "layout": "one beta per rank, point-to-point RCCL exchange" if world > 1 else
                             f"{args.pt_betas} betas x {args.batch // args.pt_betas} replicas in one batch, in-batch HIP swap"}

    if rank == 0:
        # (2) the dominant kernel's device time over the timed region, events recorded on the launch stream
        dom_ms = [timed_events.elapsed_ms((i * n_dom + j) * 2, (i * n_dom + j) * 2 + 1)
                  for i in range(args.steps) for j in range(n_dom)]
        avg_ms = dict(prof_ms)
        avg_ms[dom] = sum(dom_ms) / len(dom_ms)
        # directed edges of the reference's graph: the mean of the counts at the two ends of the timed region (the
        # synthetic system relaxes while it is integrated; over the default 40 steps the count moves by 0.02 %)
        n_edges = n_pairs + pairs_start
        F, R = 128, args.num_rbf
        # algorithmic FLOPs per launch: filter nets of one block, forward
        # (edge_fwd) or input-gradient backward (edge_bwd): 2*E*F*(R+F) each
        alg_flops = 2.0 * n_edges * F * (R + F)
        achieved = alg_flops / (avg_ms[dom] * 1e-3) / 1e12
        # HBM-side bytes per launch of the dominant kernel: a STORED rocprofv3 --pmc result (FETCH_SIZE / WRITE_SIZE
        # passes of tools/profile_round.sh, corrected as MI355X_MICROARCH.md prescribes), not measured in this run
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            traffic = tj.get(dom, {}).get("hbm_bytes_per_launch")
            traffic_src = "stored profile: profiles/pmc_traffic.json (" + str(tj.get("_source", "rocprofv3 --pmc passes")) + ")"
        value = args.steps * args.batch * world / elapsed
        prec = {
            "f32": dict(dtype="f32", peak=PEAK_SPLIT3_TFLOPS, cfg="Langevin fp32 (BASELINE configs[1])",
                        arith="fp32 storage and accumulation; filter-network and node GEMMs as split-operand "
                              "f16 MFMA (x = hi + lo/2048, 3 products), error 2e-7 = fp32 MFMA grade",
                        note="dense f16 MFMA peak (16 x 157.3) / 3 products per fp32-grade product; "
                             "the exact-fp32 MFMA peak is 157.3"),
            "w16a16": dict(dtype="f16", peak=PEAK_F16_MFMA_TFLOPS, cfg="W16A16 (BASELINE configs[2])",
                           arith="filter and output networks: fp16 operands, fp32 accumulation (reference "
                                 "W16A16 mode); node linears fp32-grade", note="dense f16 MFMA peak (16 x 157.3)"),
        }[args.precision]
        line = {
            "metric": "timestep*mol/s, CGSchNet 1ENH batch=128",
            "value": value, "unit": "timestep*mol/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            **({"rehearsal": "ranks share GPUs, gloo: not a measurement"} if rehearse else {}),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": value / PUBLISHED_BASELINE,
            "dtype": prec["dtype"], "data": "synthetic",
            "config": {"workload": "CGSchNet 1ENH batch=128 " + (prec["cfg"] if not args.pt else
                                   "parallel tempering, one beta per GPU (BASELINE configs[4])" if world > 1 else
                                   "parallel tempering, all betas in one batch (BASELINE configs[4] on one GPU)"),
                       "arithmetic": prec["arith"],
                       "molecules_per_gpu": args.batch, "beads_per_molecule": args.beads,
                       "directed_edges_per_gpu": n_edges, "num_rbf": R, "interaction_blocks": args.blocks,
                       "hidden": 128, "output_widths": [128, 64, 1], "cutoff": rc, "dt": 0.004,
                       "sharding": f"replica-sharded x{world}, no collective in the force path",
                       "directed_edges_at_start_and_end_of_timed_region": [2 * pairs_start, 2 * n_pairs],
                       "reset_every": args.reset_every,
                       "untimed": {"warmup_steps": args.warmup, "kernel_profile_steps": n_prof,
                                   "clock_settle_force_evaluations": n_settle,
                                   "note": "all before the first barrier of the timed region; the settle phase evaluates "
                                           "forces at fixed positions (no integration); --settle-seconds 0 disables it"},
                       "baseline_note": "2861 = reference README (W16A16, real 1ENH, RTX PRO 6000)"},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": prec["peak"],
                         "unit": "TFLOP/s", "frac": achieved / prec["peak"],
                         "traffic": traffic if args.precision == "f32" else None,
                         "traffic_source": traffic_src if args.precision == "f32" else None,
                         "peak_note": prec["note"],
                         "algorithmic_flops_per_launch": alg_flops, "avg_launch_ms": avg_ms[dom],
                         "launches_per_step": launches[dom]},
            "kernel_ms_per_launch": {n: round(v, 4) for n, v in avg_ms.items()},
            "kernel_ms_per_step": {n: round(avg_ms[n] * launches[n], 4) for n in avg_ms},
            "kernel_timing_note": f"{dom}: HIP events over the timed region; the other kernels: events over "
                                  f"{n_prof - 1} untimed profile steps just before it (bracketing all 15 launches "
                                  "of every step costs ~5 % of the step)",
        }
        if via_go is not None:
            line["ms_per_step_via_GradientsOut"] = via_go
        if pt_info is not None:
            line["parallel_tempering"] = pt_info
        if marks:
            first = marks[tenth] / tenth
            last = (elapsed - marks[args.steps - tenth]) / tenth
            line["sustained"] = {"ms_per_step_first_tenth": 1e3 * first, "ms_per_step_last_tenth": 1e3 * last,
                                 "seconds_timed": elapsed}
        if world == 1 and not args.no_cpu_baseline and not args.pt:
            line["cpu_baseline"] = cpu_baseline(model, args)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
