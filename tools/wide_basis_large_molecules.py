"""Engine envelope, round 6: a basis of 65..128 Gaussians with molecules of 284..535 beads (feature-split mode) -- time of
one force evaluation on the engine beside the operator path such a model took before, and beside the 64-Gaussian
feature-split engine.  python tools/wide_basis_large_molecules.py [--beads 400] [--batch 64]"""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flash-molecular-dynamics_amd")]
import flashmd_amd.schnet as S
from flashmd_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--beads", type=int, default=400)
ap.add_argument("--batch", type=int, default=64)
args = ap.parse_args()
dev = "cuda:0"
pos, types, ptr, _ = synthetic.batch(args.batch, args.beads, 5.0)
pos, types, ptr = pos.to(dev), types.to(dev), ptr.to(dev)
for num_rbf in (64, 100, 128):
    torch.manual_seed(0)
    model = S.StandardSchNet(S.GaussianBasis(S.CosineCutoff(0.0, 5.0), num_rbf), S.CosineCutoff(0.0, 5.0), [128, 64]).to(dev)
    go = S.GradientsOut(model).eval()
    for prec in ("f32", "w16a16"):
        if prec == "w16a16":
            from flashmd_amd.gptq import apply_gptq_w16a16_to_model
            apply_gptq_w16a16_to_model(model)
            model._engine = None
        for engine_on in (True, False):
            S.USE_FUSED_ENGINE = engine_on

            def step():
                d = go(S.Batch(pos.clone(), types, ptr, max_beads_per_mol=args.beads))
                return d.out["SchNet"]["forces"]

            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10 if engine_on else 3
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            print(f"R = {num_rbf:3d} {prec:6s} {args.batch} x {args.beads} beads: {'fused engine ' if engine_on else 'operator path'} "
                  f"{1e3 * dt:8.2f} ms per evaluation", flush=True)
S.USE_FUSED_ENGINE = True
