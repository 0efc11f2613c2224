"""Diagnostic for groups that span XCDs (854x480, every vertex active): which workgroups keep their z in L2, how many
halo neighbours / remote neighbours they see, and whether the resident path equals the two-kernel path.
   ARAPOPT_STAMPS=1 python tools/zfast_probe.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from arap_flow_amd import opt, synth
st = opt.State()
W, H = 854, 480
frames = [synth.make_frame(W, H, seed=10 + s, full_mask=True) for s in range(2)]
outs = []
for resident in (True, False):
    st.set_resident(resident)
    fs = opt.FrameSolver(st, W, H, batch=2)
    for b, f in enumerate(frames):
        fs.set_frame(b, f["mask_red"], f["constraints"])
    fs.solve(2, 1, 2, 60)
    outs.append([fs.results(b, want_rgb=False) for b in range(2)])
    if resident and os.environ.get("ARAPOPT_STAMPS") == "1":
        out = np.zeros((512, 16), np.uint64)          # ArapFlow_SolverStamps copies RES_WGS x 16 u64
        st.lib.ArapFlow_SolverStamps(fs.h, out.ctypes.data)
        fl = out[:, 7]
        print("flags (1 granules fast, 2 z fast, 4 two-level sums, 8 first level plain):", {int(k): int((fl == k).sum()) for k in np.unique(fl)})
        print("neighbour counts", np.unique((out[:, 6] >> 32) & 0xffff, return_counts=True), "nremote", np.unique(out[:, 6] >> 48, return_counts=True))
        xs = np.arange(512) & 7
        for x in range(8):
            print("xcd slot", x, "zfast", int(((fl & 2) != 0)[xs == x].sum()), "of 64")
    fs.close()
for a, b in zip(*outs):
    d = a["offset"] != b["offset"]
    print("mismatching floats", int(d.sum()), "max abs", float(np.abs(a["offset"] - b["offset"]).max()))
    ys, xs_ = np.nonzero(d.any(-1))
    if len(ys):
        print("rows", ys.min(), ys.max(), "cols", xs_.min(), xs_.max())
