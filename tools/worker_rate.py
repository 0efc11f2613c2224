"""Diagnostic (GPU box): steady-state rate of the persistent C++ worker alone.  Feeds N synthetic 854x480 frames to
`arap_deform --serve` at once and timestamps its `Done` lines:   python tools/worker_rate.py [N]"""
import os, subprocess, sys, tempfile, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                  # noqa: E402
from PIL import Image                               # noqa: E402
from arap_flow_amd import pipeline, synth           # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 160
W, H = 854, 480
d = tempfile.mkdtemp(prefix="arap_rate_")
try:
    lines = []
    for i in range(N):
        f = synth.make_frame(W, H, seed=i % 32)
        p = lambda s: os.path.join(d, "%03d_%s" % (i, s))          # noqa: E731
        Image.fromarray(f["rgb"]).save(p("rgb.png"))
        Image.fromarray(np.stack([f["mask_red"]] * 3, -1)).save(p("msk.png"))
        pipeline.write_constraints(p("c.txt"), [tuple(c) for c in f["constraints"]])
        lines.append(" ".join([p("rgb.png"), p("msk.png"), p("c.txt"), p("o.flo"), p("o_rgb.png"), p("o_msk.png")]))
    pr = subprocess.Popen([os.path.join(ROOT, "arap_flow_amd", "bin", "arap_deform"), "--serve"], stdin=subprocess.PIPE,
                          stdout=subprocess.PIPE, text=True, bufsize=1)
    assert pr.stdout.readline().strip() == "Ready"
    t0 = time.time()
    pr.stdin.write("\n".join(lines) + "\n"); pr.stdin.close()
    done = []
    for ln in pr.stdout:
        if ln.startswith("Done "):
            done.append(time.time() - t0)
    pr.wait()
    done = np.array(done)
    print("frames", len(done), "total %.2f s -> %.2f frames/s" % (done[-1], len(done) / done[-1]))
    k = len(done) // 5
    print("middle 60%%: %.2f frames/s" % ((len(done) - 2 * k) / (done[-k - 1] - done[k - 1])))
    b = done[7::8]
    print("time between every 8th Done (ms):", " ".join("%.0f" % (1e3 * x) for x in np.diff(b)))
finally:
    shutil.rmtree(d, ignore_errors=True)
