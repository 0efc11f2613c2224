"""Soak run (GPU): many back-to-back full-schedule solves of mixed batches; reports whether any resident launch ever
gave up (which would silently move the state to the two-kernel path).   python tools/soak.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from arap_flow_amd import opt, synth
T = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
st = opt.State()
W, H = 854, 480
davis = [synth.make_frame(W, H, seed=s) for s in range(8)]
segs = [sg for s in range(7) for sg in synth.segment_masks(synth.make_frame(W, H, seed=s, K=3, fd=2))]
full = [synth.make_frame(W, H, seed=s, full_mask=True) for s in range(2)]
mixes = [("davis x8", davis), ("segments x21", segs), ("full x2", full), ("mixed", [full[0]] + davis[:4])]
fs = opt.FrameSolver(st, W, H, batch=24)
t0 = time.time(); n = 0; frames = 0
while time.time() - t0 < T:
    name, batch = mixes[n % len(mixes)]
    for b, f in enumerate(batch):
        fs.set_frame(b, f["mask_red"], f["constraints"])
    fs.solve(len(batch), 19, 8, 400)
    r = fs.results(0, want_rgb=False)
    assert np.isfinite(r["offset"]).all(), name
    failed = st.lib.ArapFlow_ResidentFailed(st.handle)
    n += 1; frames += len(batch)
    print("%6.1f s  solve %3d  %-13s cost %.4f  resident gave up: %d  launches so far %d" % (time.time() - t0, n, name, r["cost"], failed, fs.stats()["resident_launches"]), flush=True)
    assert failed == 0, "a resident launch gave up"
print("soak ok: %d solve calls, %d solves, %.1f s, resident path throughout" % (n, frames, time.time() - t0))
