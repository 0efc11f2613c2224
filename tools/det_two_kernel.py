"""GPU box: are the kernel-per-phase paths (streaming two-kernel path of the frame solver, generic two-kernel path of
the drop-in API) bit-reproducible run to run?  Their sums are order-fixed since round 3 (arap_device.h:
block_reduce_fixed); before, per-workgroup partials were added with float64 atomics in arrival order.

    ARAPOPT_NO_RESIDENT=1 python tools/det_two_kernel.py [repeats]

Solves the same batches `repeats` times, with a different amount of unrelated work in flight each time (a second
solver object on the same state), and compares sha256(Offset + Angle) of every frame and the final costs."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from arap_flow_amd import opt, synth

R = int(sys.argv[1]) if len(sys.argv) > 1 else 6
st = opt.State()
cases = [("davis 854x480 x8 (3,4,400)", 854, 480, [synth.make_frame(854, 480, seed=s) for s in range(8)], (3, 4, 400)),
         ("full 1920x1080 x1 (1,2,400)", 1920, 1080, [synth.make_frame(1920, 1080, seed=1, full_mask=True)], (1, 2, 400)),
         ("full 854x480 x2 (2,3,400)", 854, 480, [synth.make_frame(854, 480, seed=s, full_mask=True) for s in range(2)], (2, 3, 400))]
bad = 0
for name, W, H, frames, sched in cases:
    fs = opt.FrameSolver(st, W, H, batch=len(frames))
    for b, f in enumerate(frames):
        fs.set_frame(b, f["mask_red"], f["constraints"])
    ref = None
    t0 = time.time()
    for r in range(R):
        fs.solve(len(frames), *sched)
        res = [fs.results(b, want_rgb=False) for b in range(len(frames))]
        h = [hashlib.sha256(x["offset"].tobytes() + x["angle"].tobytes()).hexdigest()[:16] for x in res]
        c = [x["cost"] for x in res]
        if ref is None:
            ref = (h, c)
        same = (h, c) == ref
        bad += not same
        print("%-32s run %d  resident launches %d  cost[0] %.9g  %s" % (name, r, fs.stats()["resident_launches"], c[0],
                                                                       "same bits" if same else "DIFFERENT"), flush=True)
    print("   %.1f s" % (time.time() - t0))
    fs.close()
print("deterministic" if bad == 0 else "NOT deterministic: %d runs differed" % bad)
sys.exit(1 if bad else 0)
