#!/usr/bin/env python3
"""run_warp -- batch driver for warp_image (reference run_warp.py:9-19,65-67, paths hard-coded there):
warps every frame under --root with its flow.   python run_warp.py --root DATA [--flow-dir Flow]
inpRGB/ inpMasks/ <flow-dir>/ -> wRGB/ wMasks/"""
import argparse
import os
import os.path as osp
import sys

sys.path.insert(0, osp.dirname(osp.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", required=True)
    ap.add_argument("--flow-dir", default="Flow")
    a = ap.parse_args()
    from arap_flow_amd import opt, pipeline
    state = opt.State()
    rgb_root = osp.join(a.root, "inpRGB")
    n = 0
    for d, _, files in sorted(os.walk(rgb_root)):
        for f in sorted(files):
            if not f.lower().endswith(".png"):
                continue
            rel = osp.relpath(osp.join(d, f), rgb_root)
            fl = osp.join(a.root, a.flow_dir, osp.splitext(rel)[0] + ".flo")
            msk = osp.join(a.root, "inpMasks", rel)
            if not (osp.exists(fl) and osp.exists(msk)):
                continue
            o1, o2 = osp.join(a.root, "wRGB", rel), osp.join(a.root, "wMasks", rel)
            os.makedirs(osp.dirname(o1), exist_ok=True)
            os.makedirs(osp.dirname(o2), exist_ok=True)
            pipeline.warp_files(state, osp.join(rgb_root, rel), msk, fl, o1, o2)
            n += 1
    state.close()
    print("Saved %d" % n)
    return 0


if __name__ == "__main__":
    sys.exit(main())
