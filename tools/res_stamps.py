"""Diagnostic: where does a PCG iteration of the resident kernel spend its time?
   ARAPOPT_STAMPS=1 python tools/res_stamps.py [batch]   (instrumented build; never quote its run time)"""
import os, sys, time
os.environ["ARAPOPT_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from arap_flow_amd import opt, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
full = len(sys.argv) > 2 and sys.argv[2] == "full"
W, H, L = 854, 480, 400
st = opt.State()
fs = opt.FrameSolver(st, W, H, batch=B)
for b in range(B):
    f = synth.make_frame(W, H, seed=b, full_mask=full)
    fs.set_frame(b, f["mask_red"], f["constraints"])
fs.solve(B, 1, 2, L)
torch.cuda.synchronize()
t = time.perf_counter(); fs.solve(B, 1, 4, L); torch.cuda.synchronize(); dt = time.perf_counter() - t
out = np.zeros((512, 16), np.uint64)
assert st.lib.ArapFlow_SolverStamps(fs.h, out.ctypes.data) == 0
o = out.astype(np.float64)
used = o[:, 0] > 0
us = o[used, :5] * 0.01 / L          # 100 MHz ticks -> us per iteration
print("frames", B, "wall us/iter (4 GN steps incl. launches):", dt / (4 * L) * 1e6, fs.stats())
fl = out[:, 7][used]
print("of", int(used.sum()), "workgroups:", int((fl & 1).astype(bool).sum()), "in a group on one XCD,", int((fl & 4).astype(bool).sum()), "in a group with two-level sums,", int((fl & 2).astype(bool).sum()), "keep their z in L2 (plain stores)")
print("workgroups active", used.sum(), "tiles/WG", o[used, 5].min(), o[used, 5].max(), "halo cells", (out[:, 6][used] & 0xffffffff).min(), (out[:, 6][used] & 0xffffffff).max())
for n, col in zip(["phaseA", "wait1", "phaseB", "wait2", "update"], us.T):
    print("%-14s mean %.2f  min %.2f  max %.2f us" % (n, col.mean(), col.min(), col.max()))
print("sum of means %.2f us" % us.mean(0).sum())
# inside the two group sums of an iteration (wave 0, shader clocks -> us at the clock the launch ran at, from the stamps)
clk = o[used, 8:12] / (2.0 * L * 4)           # per sum; 4 launches... (the table holds the LAST launch only: / L)
clk = o[used, 8:12] / (2.0 * L)
print("per group sum, shader clocks: block sum %.0f  publish %.0f  poll %.0f  tail %.0f   sweeps per sum %.2f" % (
    clk[:, 0].mean(), clk[:, 1].mean(), clk[:, 2].mean(), clk[:, 3].mean(), (o[used, 12] / (2.0 * L)).mean()))
print("repeated looks at the neighbours' z tags per iteration (wave 0 of each workgroup): mean %.3f  max %.3f" % (
    (o[used, 13] / float(L)).mean(), (o[used, 13] / float(L)).max()))
# per-workgroup view of one group (the workgroups of XCD 0: blockIdx & 7 == 0), sorted by tiles then phase A time
idx = np.arange(512)
g0 = used & ((idx & 7) == 0)
rows = sorted(zip(o[g0, 5], (out[:, 6] & 0xffffffff).astype(np.float64)[g0], *(o[g0, c] * 0.01 / L for c in range(5)), idx[g0] >> 3), key=lambda r: (r[0], r[2]))
print("XCD 0 workgroups: tiles halo | phaseA wait1 phaseB wait2 update | local index")
for r in rows:
    print("%3d %5d | %5.2f %5.2f %5.2f %5.2f %5.2f | %2d" % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]))
