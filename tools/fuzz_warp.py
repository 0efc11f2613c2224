"""Randomised sweep of the rasteriser (GPU) vs the CPU oracle: random sizes, masks, flows with folds, large
displacements that leave the frame, NaN / Inf entries.   python tools/fuzz_warp.py [N] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from arap_flow_amd import opt
from oracle import oracle as orc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
st = opt.State()
bad = 0
t0 = time.time()
for it in range(N):
    W, H = int(rng.integers(1, 400)), int(rng.integers(1, 300))
    rgb = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    mask = np.where(rng.random((H, W)) < rng.uniform(0, 0.6), 255, 0).astype(np.uint8)
    amp = float(10.0 ** rng.uniform(-1, 2.5))
    fl = (rng.normal(size=(H, W, 2)) * amp).astype(np.float32)
    if rng.random() < 0.3:                                   # smooth field with a strong fold
        ys, xs = np.mgrid[0:H, 0:W]
        fl = np.stack([np.sin(xs / 7.0) * amp, np.cos(ys / 5.0) * amp], -1).astype(np.float32)
    if rng.random() < 0.2:
        fl[rng.random((H, W)) < 0.01] = np.float32(rng.choice([np.nan, np.inf, -np.inf, 1e30]))
    fl[mask != 0] = 0
    wrgb, wmsk = opt.warp_image(st, rgb, mask, fl)
    o_rgb, o_msk = orc.warp(rgb, mask, fl)
    if not (np.array_equal(wmsk, o_msk) and np.array_equal(wrgb, o_rgb)):
        bad += 1
        print("MISMATCH it", it, "W,H", W, H, "amp %g" % amp, "mask px", int((wmsk != o_msk).sum()), "rgb px", int((wrgb != o_rgb).any(-1).sum()))
print("warp fuzz: %d cases, %d mismatches, %.1f s" % (N, bad, time.time() - t0))
sys.exit(1 if bad else 0)
