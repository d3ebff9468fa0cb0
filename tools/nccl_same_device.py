"""Does RCCL accept two ranks on ONE GPU?  (The PT exchange's nccl branch cannot otherwise run on a one-GPU box.)
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/nccl_same_device.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flash-molecular-dynamics_amd")]
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
rank = dist.get_rank()
try:
    a = torch.full((4,), float(rank), device="cuda")
    b = torch.empty_like(a)
    ops = [dist.P2POp(dist.isend, a, 1 - rank), dist.P2POp(dist.irecv, b, 1 - rank)]
    for r in dist.batch_isend_irecv(ops):
        r.wait()
    torch.cuda.synchronize()
    print(f"rank {rank}: point-to-point exchange over RCCL on one device ok, received {b.tolist()}", flush=True)
    # the exchange class itself
    from flashmd_amd import simulate

    class Sim:
        pass

    sim = Sim()
    n_indep, A = 4, 5
    sim.energy = torch.arange(n_indep, device="cuda", dtype=torch.float32) * (1 + rank)
    sim.x = torch.full((n_indep * A, 3), float(rank), device="cuda")
    sim.v = torch.ones(n_indep * A, 3, device="cuda")
    sim.force_fn = lambda x: (sim.energy, torch.zeros_like(x))
    ex = simulate.DistributedReplicaExchange([1.67, 1.16], n_indep, A, seed=3)
    ex(sim)
    m = ex.acceptance_matrix()
    print(f"rank {rank}: DistributedReplicaExchange over nccl ok, acceptance matrix {m.tolist()}, x[0] {sim.x[0].tolist()}", flush=True)
except Exception as exc:  # noqa: BLE001
    print(f"rank {rank}: RCCL with two ranks on one device failed: {type(exc).__name__}: {str(exc)[:300]}", flush=True)
dist.destroy_process_group()
