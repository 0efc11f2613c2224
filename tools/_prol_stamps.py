import os, sys
os.environ["ARAPOPT_STAMPS"] = "1"
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from arap_flow_amd import opt, synth
B, W, H, L = 8, 854, 480, int(sys.argv[1]) if len(sys.argv) > 1 else 400
st = opt.State()
fs = opt.FrameSolver(st, W, H, batch=B)
for b in range(B):
    f = synth.make_frame(W, H, seed=b)
    fs.set_frame(b, f["mask_red"], f["constraints"])
fs.solve(B, 1, 2, L); torch.cuda.synchronize()
fs.solve(B, 1, 4, L); torch.cuda.synchronize()
out = np.zeros((1024, 16), np.uint64)
assert st.lib.ArapFlow_SolverStamps(fs.h, out.ctypes.data) == 0
P = out[512:, :9].astype(np.float64)
used = P[:, 0] > 0
P = P[used]
t00 = P[:, 0].min()
names = ["start(rel. to first WG)", "startup+issue loads", "lds zero+loads arrive", "halo list", "decode table", "drain", "xcc exchange", "neighbour check", "loop"]
print("workgroups", used.sum(), " (100 MHz ticks -> us)")
print("%-26s mean %.2f max %.2f" % (names[0], ((P[:, 0] - t00) * 0.01).mean(), ((P[:, 0] - t00) * 0.01).max()))
for k in range(1, 9):
    d = (P[:, k] - P[:, k - 1]) * 0.01
    print("%-26s mean %.2f  min %.2f max %.2f" % (names[k], d.mean(), d.min(), d.max()))
print("kernel span (first start -> last loop end) %.2f us; last start -> its loop start: %.2f" % ((P[:, 8].max() - t00) * 0.01, ((P[:, 7] - P[:, 0]) * 0.01).mean()))
