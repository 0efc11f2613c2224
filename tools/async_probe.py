"""Diagnostic (GPU box): does enqueueing a solve on one solver object wait for the other object's running solve?
   python tools/async_probe.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from arap_flow_amd import opt, synth
st = opt.State()
st.use_own_stream()
W, H = 854, 480
lanes = [opt.FrameSolver(st, W, H, batch=8) for _ in range(2)]
frames = [synth.make_frame(W, H, seed=s) for s in range(40)]
def fill(k, off):
    for b in range(8):
        f = frames[(off + b) % 40]
        lanes[k].set_frame(b, f["mask_red"], f["constraints"], rgb=f["rgb"])
for rnd in range(4):
    fill(0, 16 * rnd); fill(1, 16 * rnd + 8)          # new frames in both lanes every round: new deals, new tables
    t0 = time.perf_counter(); lanes[0].solve_async(8, 19, 8, 400, warp=True, download=True)
    t1 = time.perf_counter(); lanes[1].solve_async(8, 19, 8, 400, warp=True, download=True)
    t2 = time.perf_counter(); lanes[0].wait()
    t3 = time.perf_counter(); lanes[1].wait()
    t4 = time.perf_counter()
    print("round %d: enqueue A %.1f ms, enqueue B (A in flight) %.1f ms, wait A %.1f ms, wait B %.1f ms, total %.1f ms" % (rnd, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3), 1e3*(t4-t0)))
