"""Pad entries of the engine's pair list at the bench size: status word 1 (largest per-molecule entry count) against
the mean pairs per molecule (status word 2 / B).  python tools/pad_stats.py [--batch B]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flash-molecular-dynamics_amd")]
from flashmd_amd import engine, synthetic
import flashmd_amd.schnet as S

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--beads", type=int, default=269)
args = ap.parse_args()
dev = torch.device("cuda:0")
model = S.StandardSchNet(S.GaussianBasis(S.CosineCutoff(0.0, 5.0), 50), S.CosineCutoff(0.0, 5.0), [128, 64],
                         num_interactions=3).to(dev).eval()
eng = engine.FusedSchNetEngine(model)
worst, tot_pairs = 0, 0
for b in range(args.batch):   # one molecule at a time: word 1 is then that molecule's entry count
    pos, types, ptr, _ = synthetic.batch(b + 1, args.beads, 5.0)
    sl = slice(b * args.beads, (b + 1) * args.beads)
    p1, t1 = pos[sl].to(dev), types[sl].to(dev)
    eng.energy_forces(p1, t1, torch.tensor([0, args.beads], device=dev), args.beads)
    st = eng.status.tolist()
    worst += st[1]; tot_pairs += st[2]
    if b >= 15: break
print(f"{b + 1} molecules: pairs {tot_pairs}, entries {worst}, pad {100.0 * (worst - tot_pairs) / tot_pairs:.2f} %")
