"""End-to-end timing of the host programs on synthetic 854x480 PNG inputs (GPU): C++ arap_deform, Python arap_deform.py.
   python tools/bench_hosts.py [frames]"""
import os, sys, time, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image
from arap_flow_amd import synth, pipeline

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
d = tempfile.mkdtemp(prefix="arap_hosts_")
lines = []
for i in range(n):
    f = synth.make_frame(854, 480, seed=i)
    p = lambda s: os.path.join(d, "%03d_%s" % (i, s))
    Image.fromarray(f["rgb"]).save(p("rgb.png"))
    Image.fromarray(np.stack([f["mask_red"]] * 3, -1)).save(p("msk.png"))
    pipeline.write_constraints(p("c.txt"), [tuple(c) for c in f["constraints"]])
    lines.append(" ".join([p("rgb.png"), p("msk.png"), p("c.txt"), p("o.flo"), p("o_rgb.png"), p("o_msk.png")]))
lst = os.path.join(d, "list.txt")
open(lst, "w").write("\n".join(lines) + "\n")
env = {k: v for k, v in os.environ.items() if k != "ARAP_PLAN"}
for name, cmd in (("C++ bin/arap_deform", [os.path.join(ROOT, "arap_flow_amd", "bin", "arap_deform"), lst]),
                  ("python arap_deform.py", [sys.executable, os.path.join(ROOT, "arap_deform.py"), lst])):
    t = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)
    dt = time.time() - t
    print("%-24s rc %d  %d frames in %.2f s = %.2f frames/s  (%s)" % (name, r.returncode, n, dt, n / dt, (r.stdout + r.stderr).strip().splitlines()[-1][:80] if (r.stdout + r.stderr).strip() else ""))
