"""The collective calls `bench.py --gpus N` and DistributedReplicaExchange make, over RCCL with ONE rank (the boxes of
this build have one GPU and RCCL refuses two ranks on one device, tools/nccl_same_device.py): process-group start-up
with a bound device, barrier, MAX all-reduce of a float64 on the device, all_gather_object, the int64 all-reduce of the
acceptance matrix, shutdown.  Point-to-point transfers need a second rank and stay unrehearsed.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 tools/nccl_one_rank.py"""
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flash-molecular-dynamics_amd")]
import torch
import torch.distributed as dist

import bench

local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
t0 = time.perf_counter()
dist.init_process_group("nccl", device_id=dev)
rank, world = dist.get_rank(), dist.get_world_size()
dist.barrier()
torch.cuda.synchronize()
print(f"rank {rank}/{world}: nccl process group + barrier in {time.perf_counter() - t0:.2f} s", flush=True)
t = torch.tensor([1.25 + rank], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.25 + world - 1
recs = [None] * world
dist.all_gather_object(recs, bench.device_identity(rank, local))
print("all_gather_object:", recs, flush=True)
from flashmd_amd import simulate

ex = simulate.DistributedReplicaExchange(bench.beta_ladder(world), 4, 5, seed=3)
assert ex.partner(True) is None and ex.partner(False) is None
print("acceptance matrix over nccl:", ex.acceptance_matrix().tolist(), "backend", dist.get_backend(), flush=True)
dist.barrier()
dist.destroy_process_group()
print("ok", socket.gethostname(), flush=True)
