"""Randomised sweep of the "LMGPU" solver kind (GPU) against the CPU restatement oracle_solve_lm: random sizes, radii,
iteration counts, generic / pixel-grid UrShape.   python tools/fuzz_lm.py [N] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers
from arap_flow_amd import opt
from oracle import oracle as orc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
st = opt.State()
bad = exact = 0
t0 = time.time()
for it in range(N):
    W, H = int(rng.integers(2, 160)), int(rng.integers(2, 120))
    generic = bool(rng.integers(0, 2))
    pb = helpers.random_problem(W, H, seed=int(rng.integers(0, 1 << 30)), generic_urshape=generic, ncons=int(rng.integers(1, 60)))
    nIter, lIter = int(rng.integers(1, 6)), int(rng.integers(1, 30))
    radius = float(10.0 ** rng.uniform(-2, 5))
    dev = {k: torch.from_numpy(pb[k].copy()).cuda() for k in "OAUCM"}
    s = opt.OptSolver(st, (W, H), opt.BUILTIN_PLAN, b"LMGPU")
    pp = opt.NamedParameters()
    for n, k in (("Offset", "O"), ("Angle", "A"), ("UrShape", "U"), ("Constraints", "C"), ("Mask", "M")):
        pp.set(n, dev[k])
    pp.set("w_fitSqrt", 10.0); pp.set("w_regSqrt", 0.1)
    sp = opt.NamedParameters(); sp.set("nIterations", nIter); sp.set("lIterations", lIter); sp.set("trust_region_radius", radius)
    cost = s.solve(sp, pp)
    s.close()
    Or, Ar, costs, steps, rad = orc.solve_lm(pb["O"], pb["A"], pb["U"], pb["C"], pb["M"], 10.0, 0.1, nIter, lIter, trust_region_radius=radius)
    O = dev["O"].cpu().numpy(); A = dev["A"].cpu().numpy()
    same = np.array_equal(O, Or, equal_nan=True) and np.array_equal(A, Ar, equal_nan=True)
    exact += int(same)
    close = helpers.rel_l2(O - pb["O"], Or - pb["O"]) < 1e-4 and helpers.rel_l2(A - pb["A"], Ar - pb["A"]) < 1e-4
    if not (same or close):
        bad += 1
        print("MISMATCH it", it, "W,H", W, H, "generic", generic, "nIter,lIter", nIter, lIter, "radius %g" % radius,
              "rel", helpers.rel_l2(O - pb["O"], Or - pb["O"]), "cost", cost, costs[-1], "steps", steps)
print("LM fuzz: %d cases, %d bit-identical, %d mismatches (> 1e-4), %.1f s" % (N, exact, bad, time.time() - t0))
sys.exit(1 if bad else 0)
