#!/usr/bin/env python3
"""run_arap -- batch driver (reference run_arap.py:10-15,58-80, whose paths were hard-coded): builds the
list-file lines `rgb mask constraints flow warped_rgb warped_mask` for every frame under --root and hands
them to arap_deform on the selected GPUs (one process per GPU, frames dealt round-robin, no collective).

  python run_arap.py --root DATA --gpu 0 1 2 3
Layout under --root (para_gen.py:18-26): inpRGB/ inpMasks/ tmpCnstr/ -> Flow/ wRGB/ wMasks/
"""
import argparse
import os
import os.path as osp
import subprocess
import sys
import time

HERE = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, HERE)


def collect(root):
    from arap_flow_amd import pipeline  # noqa: F401  (fails early when the package is broken)
    lines = []
    rgb_root = osp.join(root, "inpRGB")
    for d, _, files in sorted(os.walk(rgb_root)):
        for f in sorted(files):
            if not f.lower().endswith(".png"):
                continue
            rel = osp.relpath(osp.join(d, f), rgb_root)
            stem = osp.splitext(rel)[0]
            p = dict(rgb=osp.join(rgb_root, rel), msk=osp.join(root, "inpMasks", rel),
                     cst=osp.join(root, "tmpCnstr", stem + ".txt"), flo=osp.join(root, "Flow", stem + ".flo"),
                     wrgb=osp.join(root, "wRGB", rel), wmsk=osp.join(root, "wMasks", rel))
            if not (osp.exists(p["msk"]) and osp.exists(p["cst"])):
                continue
            for k in ("flo", "wrgb", "wmsk"):
                os.makedirs(osp.dirname(p[k]), exist_ok=True)
            lines.append(" ".join(osp.abspath(p[k]) for k in ("rgb", "msk", "cst", "flo", "wrgb", "wmsk")))
    return lines


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", required=True)
    ap.add_argument("--gpu", nargs="*", type=int, default=[0])
    ap.add_argument("--arap_bin", default=None, help="default: this repo's arap_deform.py")
    a = ap.parse_args()
    from arap_flow_amd import shard
    lines = collect(a.root)
    os.makedirs("tmp", exist_ok=True)
    procs = []
    begin = time.time()
    for k, gpu in enumerate(a.gpu):
        mine = shard.shard_lines(lines, k, len(a.gpu))
        if not mine:
            continue
        lf = osp.abspath(osp.join("tmp", "gpu-%d_%s.txt" % (gpu, str(time.time()).replace(".", "_"))))
        open(lf, "w").write("\n".join(mine))
        cmd = [a.arap_bin, lf] if a.arap_bin else [sys.executable, osp.join(HERE, "arap_deform.py"), lf]
        env = dict(os.environ, HIP_VISIBLE_DEVICES=str(gpu))           # para_gen.py:190 CUDA_VISIBLE_DEVICES
        procs.append((subprocess.Popen(cmd, env=env), lf))
    rc = 0
    for p, lf in procs:
        status = p.wait()
        os.remove(lf)
        assert status == 0, "ARAP exited with code %d" % status
        rc |= status
    print("Finish run_arap: %d frames | Elapsed %.3fs" % (len(lines), time.time() - begin))
    return rc


if __name__ == "__main__":
    sys.exit(main())
