"""Full-schedule (19/8/400) twin check: HIP path vs the float32 CPU oracle (double-accumulated dots, spec cos/sin)."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers
from arap_flow_amd import opt, synth
from oracle import oracle as orc
st = opt.State()
cases = []
f = synth.make_frame(160, 96, seed=3, K=1, fd=2)
cases.append(("synth160x96", f["mask_red"], f["constraints"]))
cat = helpers.load_cat512(os.path.join(ROOT, "tests", "golden"))
cases.append(("cat512", cat["mask_red"], cat["constraints"]))
for name, mask, cons in cases:
    H, W = mask.shape
    fs = opt.FrameSolver(st, W, H, batch=1)
    fs.set_frame(0, mask, cons)
    t = time.time(); fs.solve(1, 19, 8, 400); r = fs.results(0, want_rgb=False); tg = time.time() - t
    print(name, "gpu", tg, "s", fs.stats())
    fs.close()
    t = time.time(); O, A, costs = orc.frame(mask, cons, dtype=np.float32, mode=1, trig=1); to = time.time() - t
    ys, xs = np.mgrid[0:H, 0:W]; grid = np.stack([xs, ys], -1).astype(np.float32)
    print(name, "oracle", to, "s; mismatching floats O:", int((r["offset"] != O).sum()), "A:", int((r["angle"] != A).sum()),
          "rel-L2 flow", helpers.rel_l2(r["offset"] - grid, O - grid), "cost", r["cost"], costs[-1])
