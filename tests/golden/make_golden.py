"""Regenerates the committed golden fixtures.  Run in the build container (needs /root/reference for the
warp part):   python tests/golden/make_golden.py

 1. solve_*.npz : short-schedule trajectories (tier T2) from the CPU oracle, float32, float64
    accumulation of the dot products (mode 1), spec cos/sin (trig 1).  The oracle itself is pinned by
    tests/test_oracle.py against a finite-difference Jacobian and the reference's own golden
    ARAP/warping/cat512_iFlo.flo.
 2. warp_synth/ : inputs (PNG, .flo) and the outputs of the REFERENCE's own warp_image, compiled from
    /root/reference/ARAP/warping/src by oracle/Makefile into oracle/_ref/warp_image, on a synthetic
    flow with fold-overs and out-of-frame targets.
"""
import os
import subprocess
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as orc          # noqa: E402
from arap_flow_amd import flo             # noqa: E402
import helpers                            # noqa: E402


def solve_case(name, mask_red, cons, schedule):
    numIter, nIter, lIter = schedule
    O, A, costs = orc.frame(mask_red, cons, numIter=numIter, nIterations=nIter, lIterations=lIter,
                            dtype=np.float32, mode=1, trig=1)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), mask_red=mask_red, constraints=cons,
                        schedule=np.asarray(schedule, np.int32), offset=O, angle=A, costs=costs)
    print(name, "final costs", costs)


def main():
    # ---- config 1 of BASELINE.json: 64x64, mask == 0, 4 handles -------------------------------
    m64 = np.zeros((64, 64), np.uint8)
    c64 = np.asarray([(16, 16, 19, 19), (48, 16, 45, 19), (16, 48, 19, 45), (48, 48, 51, 51)], np.int32)
    solve_case("solve_64_1x2x50", m64, c64, (1, 2, 50))
    solve_case("solve_64_1x10x400", m64, c64, (1, 10, 400))
    # ---- 128x128 crop of the cat512 mask with three handles ---------------------------------
    cat = helpers.load_cat512(HERE)
    y0, x0 = 150, 200
    m128 = np.ascontiguousarray(cat["mask_red"][y0:y0 + 128, x0:x0 + 128])
    ys, xs = np.nonzero(m128 == 0)
    rng = np.random.default_rng(3)
    pick = rng.choice(len(xs), 3, replace=False)
    c128 = np.asarray([(xs[k], ys[k], xs[k] + dx, ys[k] + dy)
                       for k, (dx, dy) in zip(pick, [(6, -4), (-5, 3), (2, 7)])], np.int32)
    solve_case("solve_cat128_1x1x100", m128, c128, (1, 1, 100))
    solve_case("solve_cat128_1x4x50", m128, c128, (1, 4, 50))
    solve_case("solve_cat128_2x1x100", m128, c128, (2, 1, 100))
    solve_case("solve_cat128_1x1x200", m128, c128, (1, 1, 200))

    # ---- warp: reference binary on a synthetic folded flow -----------------------------------
    ref = orc.ref_warp_binary()
    if ref is None:
        print("oracle/_ref/warp_image not built (no /root/reference?) - warp fixtures not regenerated")
        return
    d = os.path.join(HERE, "warp_synth")
    os.makedirs(d, exist_ok=True)
    W, H = 96, 80
    rng = np.random.default_rng(11)
    rgb = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    ys, xs = np.mgrid[0:H, 0:W]
    mask = np.where(((xs - 50) / 34.0) ** 2 + ((ys - 40) / 30.0) ** 2 <= 1.0, 0, 255).astype(np.uint8)
    mask[20:28, 30:36] = 255                                   # a hole
    fl = np.zeros((H, W, 2), np.float32)
    fl[..., 0] = 9.0 * np.sin(ys / 7.0) + 0.37 * (xs - 48) * np.cos(xs / 5.0)     # folds in x
    fl[..., 1] = -6.0 * np.cos(xs / 9.0) + 14.0 * np.exp(-((xs - 60) ** 2 + (ys - 45) ** 2) / 90.0)
    fl[30:36, 70:80, 0] += 40.0                                # pushes some vertices out of frame
    fl[mask != 0] = 0
    fl = fl.astype(np.float32)
    Image.fromarray(rgb).save(os.path.join(d, "iRGB.png"))
    Image.fromarray(np.stack([mask] * 3, -1)).save(os.path.join(d, "iMsk.png"))
    flo.flow_write(os.path.join(d, "iFlo.flo"), fl)
    subprocess.check_call([ref, os.path.join(d, "iRGB.png"), os.path.join(d, "iMsk.png"),
                           os.path.join(d, "iFlo.flo"), os.path.join(d, "wRGB.png"), os.path.join(d, "wMsk.png")])
    print("warp_synth written with", ref)


if __name__ == "__main__":
    main()
