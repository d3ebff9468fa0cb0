"""When does a timestep run slow?  Per-step device time (one torch event per step) of the bench's BAOAB loop from a cold
start, after host pauses of several lengths, and around the bench's own prologue (profile steps, settle phase, event
creation) -- with the sclk / power the SMI library reports, if it imports.  Diagnostic for the driver line's slow first
region.  python tools/clock_timeline.py [--steps 1500]"""
import argparse, ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flash-molecular-dynamics_amd")]
from flashmd_amd import engine, simulate, synthetic
import flashmd_amd.schnet as S

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=1500)
ap.add_argument("--precision", default="f32")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = S.StandardSchNet(S.GaussianBasis(S.CosineCutoff(0.0, 5.0), 50), S.CosineCutoff(0.0, 5.0), [128, 64],
                         num_interactions=3).to(dev).eval()
eng = engine.FusedSchNetEngine(model, precision=args.precision)
pos, types, ptr, masses = synthetic.batch(128, 269, 5.0)
pos, types, ptr, masses = pos.to(dev), types.to(dev), ptr.to(dev), masses.to(dev)
beta = torch.full((128,), 1.67, device=dev)
sim = simulate.LangevinSimulation(lambda x: eng.energy_forces(x, types, ptr, 269), pos, masses, beta, 269)
x0, v0, f0 = sim.x.clone(), sim.v.clone(), sim.f.clone()


def smi():
    try:
        import amdsmi
        if not getattr(smi, "h", None):
            amdsmi.amdsmi_init()
            smi.h = amdsmi.amdsmi_get_processor_handles()[0]
        c = amdsmi.amdsmi_get_clock_info(smi.h, amdsmi.AmdSmiClkType.GFX)
        try:
            p = amdsmi.amdsmi_get_power_info(smi.h)
            pw = p.get("current_socket_power", p.get("average_socket_power"))
        except Exception:
            pw = None
        return f"sclk {c.get('clk')} MHz power {pw} W"
    except Exception as e:  # noqa: BLE001
        return f"(smi unavailable: {type(e).__name__})"


def reset():
    sim.x.copy_(x0); sim.v.copy_(v0); sim.f.copy_(f0)


def run(n, label, bucket=50):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(n):
        sim.timestep()
        ev[i + 1].record()
        if (i + 1) % 40 == 0:
            reset()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    out = [f"{sum(ms[i:i + bucket]) / len(ms[i:i + bucket]):.4f}" for i in range(0, n, bucket)]
    print(f"{label}: wall {1e3 * wall / n:.4f} ms/step; per {bucket} steps: {' '.join(out)}   [{smi()}]", flush=True)
    return ms


print("idle:", smi())
run(args.steps, "cold start")
for pause in (0.002, 0.01, 0.05, 0.2, 1.0):
    time.sleep(pause)
    run(200, f"after {1e3 * pause:.0f} ms pause", bucket=20)
# the bench prologue: force-only settle at fixed positions (another launch cadence), then 600 event creations
t = time.perf_counter()
while time.perf_counter() - t < 0.75:
    for _ in range(20):
        sim.force_fn(sim.x)
    torch.cuda.synchronize()
hip = ctypes.CDLL("libamdhip64.so")
hs = []
t = time.perf_counter()
for _ in range(600):
    h = ctypes.c_void_p()
    hip.hipEventCreate(ctypes.byref(h))
    hs.append(h)
print(f"600 hipEventCreate: {1e3 * (time.perf_counter() - t):.2f} ms")
run(200, "after force-only settle + event creation", bucket=20)
# the alternative: settle with the integrated loop itself
t = time.perf_counter()
while time.perf_counter() - t < 0.75:
    for _ in range(20):
        sim.timestep()
    reset()
    torch.cuda.synchronize()
run(200, "after integrated settle", bucket=20)
