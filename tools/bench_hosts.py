"""End-to-end throughput of the host pipeline on synthetic DAVIS-shaped 854x480 inputs (GPU box):

    python tools/bench_hosts.py [pairs] [--multseg] [--out JSON]

  1. the real CLI: `para_gen.py --matches` over a synthetic input tree of `pairs` frame pairs (PNG frames, label masks,
     precomputed match files) with the C++ driver as a persistent --serve worker: wall time from process start to
     exit, time since the GPU worker was ready, batch sizes handed to the GPU (OUT/arap_stats.json);
  2. the C++ driver and its Python twin on a list file of the same frames (process start to exit).
One JSON record on stdout (and into --out): what profiles/r02_hosts*.json hold.

    python tools/bench_hosts.py 512 --standin 8 [--jobs J]          (no GPU needed)

Host capacity of a node: `para_gen.py --gpu 0 1 .. 7` with a STAND-IN in place of the GPU worker -- a process per GPU id
that speaks the --serve protocol and does the worker's file work (decode the two PNGs of a line, write a .flo and two
PNGs) but no solve.  What is measured is whether one host can prepare, hand out and post-process pairs fast enough for
8 GPUs at 28 pairs/s each (224 pairs/s): pairs/s, the CPUs this process may use, --jobs."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                      # noqa: E402
from PIL import Image                   # noqa: E402
from arap_flow_amd import pipeline, synth   # noqa: E402


def make_moving_tree(d, pairs, W, H):
    """pairs for the built-in matcher: textured frames, one elliptical object label; frame 2 = frame 1 moved by (dx, dy)"""
    inp = os.path.join(d, "in")
    rng = np.random.default_rng(0)
    for n in range(pairs):
        name = "seq%04d" % n
        for sub in ("orgRGB", "orgMasks"):
            os.makedirs(os.path.join(inp, sub, name))
        big = np.clip(synth.make_rgb(W + 64, H + 64, n).astype(np.int32) + rng.integers(-40, 41, (H + 64, W + 64, 1)), 0, 255).astype(np.uint8)
        yy, xx = np.mgrid[0:H + 64, 0:W + 64]
        lab = (((xx - 32 - W / 2) / (W * 0.28)) ** 2 + ((yy - 32 - H / 2) / (H * 0.30)) ** 2 < 1.0).astype(np.uint8)
        dx, dy = int(rng.integers(-9, 10)) or 3, int(rng.integers(-9, 10)) or -2
        for k in range(2):
            y0, x0 = 32 - k * dy, 32 - k * dx
            Image.fromarray(big[y0:y0 + H, x0:x0 + W]).save(os.path.join(inp, "orgRGB", name, "%05d.png" % k))
            Image.fromarray(lab[y0:y0 + H, x0:x0 + W]).save(os.path.join(inp, "orgMasks", name, "%05d.png" % k))
    return inp


def make_tree(d, pairs, W, H, K):
    """one two-frame sequence per pair (frame seed = pair index, DAVIS-shaped blob(s), lattice matches): both frames of
    a pair carry the same labels so that para_gen's match filter (same label at both ends, para_gen.py:216-223) keeps
    the synthetic matches"""
    inp, mdir = os.path.join(d, "in"), os.path.join(d, "matches")
    for n in range(pairs):
        name = "seq%04d" % n
        for sub in ("orgRGB", "orgMasks"):
            os.makedirs(os.path.join(inp, sub, name))
        os.makedirs(os.path.join(mdir, name))
        f = synth.make_frame(W, H, seed=n, K=K, fd=1)
        rgb, lab = Image.fromarray(f["rgb"]), Image.fromarray(f["labels"].astype(np.uint8))
        for k in range(2):
            rgb.save(os.path.join(inp, "orgRGB", name, "%05d.png" % k))
            lab.save(os.path.join(inp, "orgMasks", name, "%05d.png" % k))
        open(os.path.join(mdir, name, "00000.txt"), "w").write(
            "\n".join("%d %d %d %d 1.0 0" % tuple(c) for c in f["constraints"]))
    return inp, mdir


STANDIN = r'''
import sys
import numpy as np
from PIL import Image
sys.path.insert(0, %r)
from arap_flow_amd import flo as F
# stand-in for `arap_deform --serve` on one GPU id: the worker's file work, no solve (tools/bench_hosts.py --standin)
print("Ready", flush=True)
for line in sys.stdin:
    rgb, msk, cst, flo, wrgb, wmsk = line.split()
    im = np.array(Image.open(rgb).convert("RGB"))
    m = np.array(Image.open(msk).convert("RGB"))[..., 0]
    open(cst).read()
    F.flow_write(flo, np.zeros(m.shape + (2,), np.float32))
    Image.fromarray(im).save(wrgb)
    Image.fromarray(m == 0).save(wmsk)
    print("Batch 1", flush=True)
    print("Done " + flo, flush=True)
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pairs", type=int, nargs="?", default=128)
    ap.add_argument("--standin", type=int, default=0, metavar="NGPU",
                    help="host capacity run: NGPU stand-in workers (no GPU), see the module docstring")
    ap.add_argument("--jobs", type=int, default=0)
    ap.add_argument("--builtin-matcher", action="store_true",
                    help="no precomputed matches: `para_gen.py --dm_bin builtin` (libarapmatch.so) on pairs whose second frame "
                         "is the first moved by a few pixels (so that the matcher has something to find)")
    ap.add_argument("--multseg", action="store_true")
    ap.add_argument("--size", type=int, nargs=2, default=[854, 480])
    ap.add_argument("--out", default=None)
    ap.add_argument("--skip-list-mode", action="store_true")
    a = ap.parse_args()
    W, H = a.size
    d = tempfile.mkdtemp(prefix="arap_hosts_")
    rec = {"pairs": a.pairs, "size": [W, H], "multseg": bool(a.multseg)}
    try:
        t = time.time()
        if a.builtin_matcher:
            inp, mdir = make_moving_tree(d, a.pairs, W, H), None
        else:
            inp, mdir = make_tree(d, a.pairs, W, H, 3 if a.multseg else 1)
        rec["tree_seconds"] = time.time() - t
        out = os.path.join(d, "out")
        cmd = [sys.executable, os.path.join(ROOT, "para_gen.py"), "--input", inp, "--output", out, "--gpu", "0",
               "--matches", str(mdir)] + (["--multseg"] if a.multseg else [])
        if a.builtin_matcher:
            cmd = [sys.executable, os.path.join(ROOT, "para_gen.py"), "--input", inp, "--output", out, "--gpu", "0",
                   "--dm_bin", "builtin"]
            a.skip_list_mode = True
            rec["matcher"] = "builtin (libarapmatch.so), two phases: match all pairs, then solve"
        if a.standin:
            fake = os.path.join(d, "standin_worker.py")
            open(fake, "w").write(STANDIN % ROOT)
            cmd = [sys.executable, os.path.join(ROOT, "para_gen.py"), "--input", inp, "--output", out, "--gpu"] + \
                  [str(g) for g in range(a.standin)] + ["--matches", mdir, "--worker", "serve", "--arap_bin",
                                                        "%s %s" % (sys.executable, fake)] + (["--multseg"] if a.multseg else [])
            a.skip_list_mode = True
            try:
                rec["cpus_usable"] = len(os.sched_getaffinity(0))
            except AttributeError:
                rec["cpus_usable"] = os.cpu_count()
            try:                                                   # cgroup v2 CPU quota (a 1-GPU box: 16 of the host's cores)
                quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
                rec["cpu_quota"] = None if quota == "max" else float(quota) / float(period)
            except (OSError, ValueError):
                rec["cpu_quota"] = None
            rec["standin_workers"] = a.standin
        if a.jobs:
            cmd += ["--jobs", str(a.jobs)]
        env = {k: v for k, v in os.environ.items() if k != "ARAP_PLAN"}
        t = time.time()
        r = subprocess.run(cmd, env=env, cwd=d, capture_output=True, text=True)
        dt = time.time() - t
        if r.returncode != 0:
            print(r.stdout[-2000:], r.stderr[-2000:])
            raise SystemExit("para_gen failed")
        st = json.load(open(os.path.join(out, "arap_stats.json")))
        done = len(open(os.path.join(out, "all_files.list")).read().splitlines())
        rec["para_gen"] = {"frames": st["frames"], "solves": st["solves"], "listed": done, "wall_seconds": dt,
                           "frames_per_s_wall": st["frames"] / dt,
                           "frames_per_s_since_worker_ready": st["frames"] / st["seconds_since_workers_ready"]
                           if st.get("seconds_since_workers_ready") else None,
                           "mean_batch": st["mean_batch"],
                           "batches": st["batches"] if not a.standin else "1 each (the stand-in reports every line)",
                           "jobs": st["jobs"],
                           "narap": st["narap"], "worker": st["worker"]}
        if not a.skip_list_mode:
            n = min(a.pairs, 64)
            lines = []
            for i in range(n):
                f = synth.make_frame(W, H, seed=i)
                p = lambda s: os.path.join(d, "%03d_%s" % (i, s))          # noqa: E731
                Image.fromarray(f["rgb"]).save(p("rgb.png"))
                Image.fromarray(np.stack([f["mask_red"]] * 3, -1)).save(p("msk.png"))
                pipeline.write_constraints(p("c.txt"), [tuple(c) for c in f["constraints"]])
                lines.append(" ".join([p("rgb.png"), p("msk.png"), p("c.txt"), p("o.flo"), p("o_rgb.png"), p("o_msk.png")]))
            lst = os.path.join(d, "list.txt")
            open(lst, "w").write("\n".join(lines) + "\n")
            for name, c in (("cpp_arap_deform_list", [os.path.join(ROOT, "arap_flow_amd", "bin", "arap_deform"), lst]),
                            ("python_arap_deform_list", [sys.executable, os.path.join(ROOT, "arap_deform.py"), lst])):
                t = time.time()
                r = subprocess.run(c, env=env, capture_output=True, text=True)
                dt = time.time() - t
                rec[name] = {"rc": r.returncode, "frames": n, "wall_seconds": dt, "frames_per_s_wall": n / dt}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    s = json.dumps(rec)
    print(s)
    if a.out:
        open(a.out, "w").write(s + "\n")


if __name__ == "__main__":
    main()
