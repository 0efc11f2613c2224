"""Randomised sweep of the drop-in Opt_* path (GPU): random sizes, masks, generic or pixel-grid UrShape, targets with
negative / duplicate entries; Opt_ProblemSolve through the ten reference symbols must equal the float32 CPU oracle bit for
bit (generic UrShape -> two-kernel path, pixel grid -> resident kernel).   python tools/fuzz_dropin.py [N] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers
from arap_flow_amd import opt
from oracle import oracle as orc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
st = opt.State()
bad = res = 0
t0 = time.time()
for it in range(N):
    W, H = int(rng.integers(1, 300)), int(rng.integers(1, 180))
    generic = bool(rng.integers(0, 2))
    pb = helpers.random_problem(W, H, seed=int(rng.integers(0, 1 << 30)), generic_urshape=generic, ncons=int(rng.integers(0, 60)))
    nIter, lIter = int(rng.integers(1, 4)), int(rng.integers(1, 40))
    dev = {k: torch.from_numpy(pb[k].copy()).cuda() for k in "OAUCM"}
    s = opt.OptSolver(st, (W, H))
    pp = opt.NamedParameters()
    for n, k in (("Offset", "O"), ("Angle", "A"), ("UrShape", "U"), ("Constraints", "C"), ("Mask", "M")):
        pp.set(n, dev[k])
    pp.set("w_fitSqrt", 10.0); pp.set("w_regSqrt", 0.1)
    sp = opt.NamedParameters(); sp.set("nIterations", nIter); sp.set("lIterations", lIter)
    cost = s.solve(sp, pp)
    res += int(s.resident_launches() > 0)
    s.close()
    Or, Ar, cs = orc.solve(pb["O"], pb["A"], pb["U"], pb["C"], pb["M"], 10.0, 0.1, nIter, lIter, dtype=np.float32, mode=1, trig=1)
    ok = np.array_equal(dev["O"].cpu().numpy(), Or, equal_nan=True) and np.array_equal(dev["A"].cpu().numpy(), Ar, equal_nan=True) \
        and (cost == cs[-1] or (np.isnan(cost) and np.isnan(cs[-1])))
    if not ok:
        bad += 1
        print("MISMATCH it", it, "W,H", W, H, "generic", generic, "nIter,lIter", nIter, lIter, "cost", cost, cs[-1])
print("drop-in fuzz: %d cases (%d on the resident kernel), %d mismatches, %.1f s" % (N, res, bad, time.time() - t0))
sys.exit(1 if bad else 0)
