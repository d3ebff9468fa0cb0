"""CPU: the C-ABI library loads and exports every symbol include/flashmd_hip.h declares (no compute
calls), the ctypes table matches the header, and the host mirror's PyTorch (--disable_optim) path
agrees with the oracle and the golden fixtures."""
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import schnet_oracle as so

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "flashmd_hip.h")


def _declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(fmd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from flashmd_amd import _ffi

    assert os.path.exists(_ffi.LIB_PATH), "run __graft_entry__.build() first"
    lib = _ffi.lib()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert sorted(_ffi.SIGNATURES) == declared, "ctypes table and header disagree"
    assert lib.fmd_abi_version() >= 1
    assert lib.fmd_strerror(-1).decode() == "invalid argument"


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device (safe without a GPU)."""
    from flashmd_amd import _ffi

    lib = _ffi.lib()
    assert lib.fmd_build_csr(None, -1, 4, None, None, None, None) == -1
    assert lib.fmd_gemm_f32(None, None, None, None, None, 4, 0, 4, 0, 0, 0, None) == -1
    assert lib.fmd_build_csr_workspace_bytes(10, 100) >= 16 + 80 + 800


def test_hip_operators_refuse_cpu_tensors():
    import flashmd_amd.kernels as K

    x = torch.zeros(4, 3)
    with pytest.raises((RuntimeError, AssertionError)):
        K.radius_graph(x, 5.0)
    with pytest.raises((RuntimeError, AssertionError)):
        K.fused_tanh_linear(torch.zeros(4, 8), torch.zeros(8, 8))
    with pytest.raises((RuntimeError, AssertionError)):
        K.build_csr_index(torch.zeros(3, dtype=torch.int64), 4)


def test_operator_names_match_reference_surface():
    import flashmd_amd.kernels as K

    # reference src/flashmd/kernels/__init__.py:47-75 (__all__)
    for name in ["fused_cutoff_gather_multiply_scatter_kernel", "fused_cutoff_gather_multiply_scatter",
                 "fused_cutoff_gather_multiply_scatter_autograd", "fused_linear_tanh_kernel",
                 "fused_linear_tanh", "fused_linear_tanh_autograd", "fused_linear_tanh_fp16_kernel",
                 "fused_linear_tanh_fp16", "fused_linear_tanh_fp16_autograd", "linear_fp16_kernel",
                 "linear_fp16", "linear_fp16_autograd", "fused_distance_gaussian_rbf_cutoff_kernel",
                 "fused_distance_gaussian_rbf_cutoff", "fused_distance_gaussian_rbf_cutoff_autograd",
                 "fused_tanh_linear_kernel", "fused_tanh_linear", "fused_tanh_linear_autograd",
                 "build_csr_index", "build_src_csr_index", "fused_csr_cfconv",
                 "fused_csr_cfconv_autograd", "fused_src_csr_grad_x"]:
        assert hasattr(K, name), name


_CPU_PATH = r"""
import os, sys, numpy as np, torch
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "flash-molecular-dynamics_amd")]
from oracle import schnet_oracle as so
import flashmd_amd.schnet as S
p = so.params_from_npz(os.path.join(%(root)r, "tests/golden/weights_L3_R50.npz"))
z = np.load(os.path.join(%(root)r, "tests/golden/%(case)s.npz"))
pos, types, ptr = (torch.from_numpy(z[k]) for k in ("pos", "types", "ptr"))
model = S.GradientsOut(S.model_from_params(p)).eval()
data = model(S.Batch(pos.clone(), types, ptr))
e, f = data.out["SchNet"]["energy"].detach(), data.out["SchNet"]["forces"]
re = float((e.double() - torch.from_numpy(z["energy_ref64"])).abs().max() / np.abs(z["energy_ref64"]).max())
rf = float(np.linalg.norm(f.double().numpy() - z["forces_ref64"]) / np.linalg.norm(z["forces_ref64"]))
print("REL", re, rf)
# gates on + CPU tensors must fail loudly (no silent fallback)
"""


@pytest.mark.parametrize("case", ["case_small_3x37", "case_ragged", "case_1enh_4x269"])
def test_disable_optim_pytorch_path_on_cpu_matches_reference(case):
    """BASELINE config 1: all MLCG_USE_* = 0 -> plain PyTorch ops on the CPU, checked against the
    reference's fp64 vectors (budget 1e-4; fp32 lands near 1e-6).  ``case_1enh_4x269`` is the configuration at its
    stated size: 4 molecules x 269 beads."""
    env = dict(os.environ, MLCG_USE_TRITON_MESSAGE_PASSING="0", MLCG_USE_FUSED_RBF="0",
               MLCG_USE_FUSED_TANH_LINEAR="0", MLCG_USE_CSR="0", MLCG_USE_SRC_CSR_GRAD_X="0")
    out = subprocess.run([sys.executable, "-c", _CPU_PATH % {"root": ROOT, "case": case}], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    re_, rf = (float(v) for v in out.stdout.split("REL")[1].split())
    assert re_ < 1e-5 and rf < 1e-5


def test_gates_on_with_cpu_tensors_fails_loudly():
    code = _CPU_PATH % {"root": ROOT, "case": "case_small_3x37"}
    env = {k: v for k, v in os.environ.items() if not k.startswith("MLCG_")}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "no CPU" in out.stderr.replace("There is no CPU", "no CPU")


def test_shard_range_partitions_molecules():
    from flashmd_amd.simulate import shard_range

    for n, w in [(128, 8), (1024, 8), (10, 3), (5, 8)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_trajectory_writer_layout_and_checkpoint_resume(tmp_path):
    """Row N3: file names, array layouts and checkpoint keys of the reference (simulation/base.py:1366-1418,
    langevin.py:273-291), and a resume from the checkpoint, on the PyTorch path with a harmonic toy force."""
    import numpy as np
    import torch

    from flashmd_amd import simulate

    n_sims, n_atoms = 3, 5
    g = torch.Generator().manual_seed(1)
    pos = torch.randn(n_sims * n_atoms, 3, generator=g)
    masses = torch.full((n_sims * n_atoms,), 12.0)
    beta = torch.full((n_sims,), 1.67)

    def ff(x):  # U = 0.5 |x|^2 per molecule
        return 0.5 * (x * x).sum(-1).view(n_sims, n_atoms).sum(-1), -x

    sim = simulate.LangevinSimulation(ff, pos, masses, beta, n_atoms, dt=0.004, use_hip=False)
    base = str(tmp_path / "traj")
    wr = simulate.TrajectoryWriter(base, sim, save_interval=2, export_interval=10, save_forces=True,
                                   save_energies=True, create_checkpoints=True)
    frames = []

    def spy(s):
        frames.append((s.x.clone(), s.f.clone(), s.energy.clone(), s.kinetic_energy().clone(), s.v.clone()))

    sim.simulate(20, subroutine=spy, interval=2, writer=wr)
    for chunk in (0, 1):
        c = np.load(f"{base}_coords_{chunk:04d}.npy")
        f = np.load(f"{base}_forces_{chunk:04d}.npy")
        u = np.load(f"{base}_potential_{chunk:04d}.npy")
        k = np.load(f"{base}_kineticenergy_{chunk:04d}.npy")
        assert c.shape == (n_sims, 5, n_atoms, 3) and f.shape == c.shape and u.shape == (n_sims, 5) and k.shape == u.shape
        for j in range(5):
            x_, f_, u_, k_, _ = frames[chunk * 5 + j]
            assert np.array_equal(c[:, j], x_.view(n_sims, n_atoms, 3).numpy())
            assert np.array_equal(f[:, j], f_.view(n_sims, n_atoms, 3).numpy())
            assert np.array_equal(u[:, j], u_.numpy()) and np.allclose(k[:, j], k_.numpy())
    ck = torch.load(f"{base}_checkpoint_0001.pt", weights_only=True)
    assert set(ck) == {"pos", "velocities", "current_timestep", "export_interval", "save_interval", "log_interval"}
    assert ck["current_timestep"] == 2 and ck["export_interval"] == 10 and ck["save_interval"] == 2
    sim2 = simulate.LangevinSimulation(ff, pos, masses, beta, n_atoms, dt=0.004, use_hip=False)
    assert sim2.load_checkpoint(f"{base}_checkpoint_0001.pt") == 2
    assert torch.equal(sim2.x, frames[-1][0]) and torch.equal(sim2.v, frames[-1][4])
    # a resumed run (t_init != 0) with a FRESH writer must leave {filename}_checkpoint_init.pt alone: the reference
    # writes it only when t_init == 0 (simulation/base.py:538) -- mid-run state labelled timestep 0 would be a lie
    init_before = torch.load(f"{base}_checkpoint_init.pt", weights_only=True)
    assert init_before["current_timestep"] == 0 and torch.equal(init_before["pos"], pos)
    wr2 = simulate.TrajectoryWriter(base, sim2, save_interval=2, export_interval=10, create_checkpoints=True)
    sim2.simulate(10, writer=wr2)
    assert os.path.exists(f"{base}_coords_0002.npy")   # the resumed run continues the chunk numbering
    init_after = torch.load(f"{base}_checkpoint_init.pt", weights_only=True)
    assert torch.equal(init_after["pos"], init_before["pos"]) and torch.equal(init_after["velocities"], init_before["velocities"])


def test_state_dict_loaders_row_n4(tmp_path, golden_dir):
    """Row N4, the buildable part: a model rebuilt from a state_dict keyed by the reference's parameter names -- from
    a dict, from an .npz and from a weights-only torch.load file, bare or inside a wrapper ("model." prefix) -- is the
    model the fixtures were produced with (same energies / forces on the CPU path as the reference's vectors); a
    pickled module is refused (the published model_and_prior.pt is one: out of scope)."""
    import numpy as np
    import torch

    from oracle import schnet_oracle as so
    import flashmd_amd.schnet as S

    S_gates = (S.USE_TRITON_MESSAGE_PASSING, S.USE_FUSED_RBF, S.USE_FUSED_TANH_LINEAR, S.USE_CSR, S.USE_SRC_CSR_GRAD_X)
    p = so.params_from_npz(os.path.join(golden_dir, "weights_L3_R50.npz"))
    ref_model = S.model_from_params(p)
    sd = ref_model.state_dict()
    assert "interaction_blocks.2.conv.filter_network.layers.2.weight" in sd and "rbf_layer.offset" in sd
    np.savez(tmp_path / "sd.npz", **{k: v.numpy() for k, v in sd.items()})
    torch.save(sd, tmp_path / "sd.pt")
    torch.save({"state_dict": {"model." + k: v for k, v in sd.items()}}, tmp_path / "wrapped.pt")
    torch.save(ref_model, tmp_path / "module.pt")
    models = [S.model_from_state_dict(sd), S.model_from_state_dict(str(tmp_path / "sd.npz")),
              S.model_from_state_dict(str(tmp_path / "sd.pt")),
              S.model_from_state_dict(str(tmp_path / "wrapped.pt"), prefix="model.")]
    for m in models:
        assert m.rbf_layer.cutoff.cutoff_upper == 5.0 and m.rbf_layer.num_rbf == 50 and len(m.interaction_blocks) == 3
        for (ka, va), (kb, vb) in zip(sd.items(), m.state_dict().items()):
            assert ka == kb and torch.equal(va, vb), ka
    with pytest.raises(Exception):   # weights_only refuses a pickled nn.Module
        S.model_from_state_dict(str(tmp_path / "module.pt"))
    with pytest.raises(KeyError):
        S.model_from_state_dict({"foo.weight": torch.zeros(1)})
    if not any(S_gates[:1] + S_gates[2:]) and S_gates[1] != "1":   # gates off in this process: evaluate on the CPU path
        z = np.load(os.path.join(golden_dir, "case_small_3x37.npz"))
        pos, types, ptr = (torch.from_numpy(z[k]) for k in ("pos", "types", "ptr"))
        data = S.GradientsOut(models[1]).eval()(S.Batch(pos.clone(), types, ptr))
        f = data.out["SchNet"]["forces"].double().numpy()
        assert np.linalg.norm(f - z["forces_ref64"]) / np.linalg.norm(z["forces_ref64"]) < 1e-5


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` (no RANK in the environment: how the driver starts the single-GPU line) must start N
    ranks itself -- as a child `python -m torch.distributed.run ... bench.py <same args>`, rendezvous on 127.0.0.1 --
    and relay rank 0's line; under torch.distributed.run (RANK set) it must not launch again."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    a = bench.parse(["--gpus", "8", "--pt"])
    assert bench.needs_self_launch(a, {}) and not bench.needs_self_launch(a, {"RANK": "3"})
    assert not bench.needs_self_launch(bench.parse([]), {})
    argv = bench.launcher_argv(8, 29512, ["--gpus", "8", "--pt", "--steps", "40"])
    assert argv[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in argv and "--nnodes=1" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29512"
    assert argv[-5:] == ["--gpus", "8", "--pt", "--steps", "40"] and argv[-6].endswith("bench.py")
    # the real thing, two ranks on the CPU: the launch path only (FMD_BENCH_LAUNCH_CHECK stops before any GPU use)
    env = dict(os.environ, FMD_BENCH_LAUNCH_CHECK="1")
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "7"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert {k: line[k] for k in ("launch_check", "n_gpus", "rank_sum", "gpus_arg")} == \
        {"launch_check": True, "n_gpus": 2, "rank_sum": 1, "gpus_arg": 2}
    # every rank reports who it is (on a GPU run: PCI bus id / UUID of its device, `devices` on the bench line)
    assert sorted(d["rank"] for d in line["devices"]) == [0, 1] and line["distinct_processes"] == 2


@pytest.mark.parametrize("extra", [[], ["--pt"]])
def test_bench_launch_path_with_eight_ranks(extra):
    """The driver's exact multi-GPU command lines, `python bench.py --gpus 8 [--pt] --steps 20 --warmup 5`, as far as
    they go without GPUs: the script starts eight ranks itself (child torch.distributed.run, rendezvous on 127.0.0.1),
    all eight rendezvous (gloo) and rank 0 reports who was there."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FMD_BENCH_LAUNCH_CHECK="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5",
                          *extra], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert {k: line[k] for k in ("launch_check", "n_gpus", "rank_sum", "gpus_arg")} == \
        {"launch_check": True, "n_gpus": 8, "rank_sum": 28, "gpus_arg": 8}
    # the keys the first 8-GPU run will carry: one identity record per rank, gathered over the process group
    assert sorted(d["rank"] for d in line["devices"]) == list(range(8)) and line["distinct_processes"] == 8
    assert all(set(d) >= {"rank", "local_rank", "host", "pid", "name", "uuid", "pci_bus_id"} for d in line["devices"])


def test_bench_exchange_attempts_fall_inside_a_short_timed_region():
    """--pt must time at least one even- and one odd-phase attempt whatever --steps is (the driver runs 40 steps against
    an exchange interval of 100); long regions keep the reference's schedule (parallel_tempering.py:349-366)."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    assert bench.attempt_steps(40, 100) == [13, 26]
    assert bench.attempt_steps(199, 100) == [66, 132]
    assert bench.attempt_steps(200, 100) == [100, 200]
    assert bench.attempt_steps(1000, 100) == list(range(100, 1001, 100))
    for k in range(2, 60):
        a = bench.attempt_steps(k, 100)
        assert len(a) == 2 and 1 <= a[0] < a[1] <= k
    assert bench.attempt_steps(1, 100) == [1] and bench.attempt_steps(0, 100) == []


def test_engine_descriptor_is_validated_without_a_gpu():
    """fmd_schnet_pack_bytes only validates and sizes: unknown flag bits (the tabulated-filter prototype's bit of
    ABI 4 included), a non-finite or non-positive cutoff and a non-finite gamma are FMD_EINVAL (a C caller's
    uninitialised struct must not reach the kernels)."""
    import ctypes

    from flashmd_amd import _ffi

    def desc(**kw):
        d = _ffi.SchnetDesc()
        d.hidden, d.num_rbf, d.num_blocks, d.num_out = 128, 50, 3, 3
        d.out_width[0], d.out_width[1], d.out_width[2] = 128, 64, 1
        d.embedding_size, d.precision, d.n_split, d.cutoff, d.gamma, d.flags = 100, 0, 0, 5.0, -48.0, 0
        for k, v in kw.items():
            setattr(d, k, v)
        return d

    lib = _ffi.lib()
    assert lib.fmd_abi_version() == _ffi.ABI_VERSION == 6
    assert lib.fmd_schnet_pack_bytes(ctypes.byref(desc())) > 0
    assert lib.fmd_schnet_pack_bytes(ctypes.byref(desc(flags=_ffi.FLAG_NO_FILTER_STASH))) > 0
    assert lib.fmd_schnet_pack_bytes(ctypes.byref(desc(flags=_ffi.FLAG_NO_FILTER_STASH | _ffi.FLAG_NO_MOLECULE_GROUPS))) > 0
    # the workspace of a batch of small molecules has room for a partial group (ABI 6); large molecules: unchanged rule
    wb = lib.fmd_schnet_workspace_bytes
    assert wb(ctypes.byref(desc()), 2048, 20480, 128) > 0
    for bad in (dict(flags=4), dict(flags=7), dict(flags=1 << 20), dict(cutoff=0.0), dict(cutoff=-1.0),
                dict(cutoff=float("inf")), dict(cutoff=float("nan")), dict(gamma=float("nan")), dict(gamma=float("inf"))):
        assert lib.fmd_schnet_pack_bytes(ctypes.byref(desc(**bad))) == -1, bad
