"""Calibration table for parity tier T4: the CPU oracle's arithmetic variants on the reference's
only solver fixture (cat512, full 19/8/400 schedule) against the reference golden ARAP/warping/cat512_iFlo.flo.

    python tests/golden/make_t4_variants.py        (build container, CPU only, ~5 min on 8 cores)

writes tests/golden/t4_variants.json: per variant the final cost, rel-L2 / median / handle error against the
golden flow, the number of inverted quads, and the mutual rel-L2 distances of the variants.  The schedule runs
float32 PCG far past stability and far short of convergence, so every variant is one more rounding trajectory
of the same algorithm; tests/test_oracle.py and tests/test_gpu_solve.py derive their T4 bands from this table
instead of from the value the product happens to give.

Variants (the product = HIP path's bit-level twin is "f32_sum64_spec_fma"):
    dtype f32/f64 x dot products accumulated sequentially in REAL (mode 0) or in float64 (mode 1)
    x cos/sin from libm (trig 0) or arap_sincos_spec (trig 1) x fused multiply-add sites on/off (liboracle_nofma.so)
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as orc          # noqa: E402
import helpers                            # noqa: E402

VARIANTS = [
    # name                     dtype        mode trig fma
    ("f64_libm_fma",           np.float64,  1,   0,   True),
    ("f32_seq_libm_fma",       np.float32,  0,   0,   True),
    ("f32_sum64_libm_fma",     np.float32,  1,   0,   True),
    ("f32_sum64_spec_fma",     np.float32,  1,   1,   True),     # the product's arithmetic
    ("f32_seq_libm_nofma",     np.float32,  0,   0,   False),
    ("f32_sum64_libm_nofma",   np.float32,  1,   0,   False),
    ("f32_sum64_spec_nofma",   np.float32,  1,   1,   False),
    ("f64_libm_nofma",         np.float64,  1,   0,   False),
]


def main():
    cat = helpers.load_cat512(HERE)
    gold = cat["golden_flow"]
    act = cat["mask_red"] == 0
    flows, rows = {}, {}
    for name, dt, mode, trig, fma in VARIANTS:
        orc.use_variant(None if fma else "nofma")
        t = time.time()
        O, A, costs = orc.frame(cat["mask_red"], cat["constraints"], dtype=dt, mode=mode, trig=trig)
        flow = orc.flow_from_offset(O.astype(np.float32))
        flows[name] = flow
        err = np.linalg.norm(flow - gold, axis=-1)[act]
        hand = max(float(np.abs(flow[y1, x1] - gold[y1, x1]).max()) for x1, y1, _, _ in cat["constraints"])
        rows[name] = {"dtype": np.dtype(dt).name, "sums": "float64" if mode == 1 else "sequential in dtype",
                      "trig": "spec" if trig else "libm", "fma_sites": bool(fma),
                      "final_cost": float(costs[-1]), "rel_l2_vs_golden": float(helpers.rel_l2(flow[act], gold[act])),
                      "median_px_vs_golden": float(np.median(err)), "p99_px_vs_golden": float(np.percentile(err, 99)),
                      "max_px_vs_golden": float(err.max()), "max_handle_px_vs_golden": hand,
                      "neg_det_quads": helpers.neg_det_quads(flow, act), "seconds": round(time.time() - t, 1)}
        print(name, json.dumps(rows[name]), flush=True)
    orc.use_variant(None)
    names = list(flows)
    mutual = {a: {b: float(helpers.rel_l2(flows[a][act], flows[b][act])) for b in names if b != a} for a in names}
    out = {"fixture": "tests/golden/cat512 (= ARAP/deformation/cat512_i{RGB,Msk}.png, cat512_iCstr.txt; golden "
                      "ARAP/warping/cat512_iFlo.flo), schedule 19/8/400",
           "golden_neg_det_quads": helpers.neg_det_quads(gold, act), "variants": rows, "mutual_rel_l2": mutual,
           "product_variant": "f32_sum64_spec_fma"}
    with open(os.path.join(HERE, "t4_variants.json"), "w") as f:
        json.dump(out, f, indent=1)
    costs = [r["final_cost"] for r in rows.values()]
    print("cost range %.3f .. %.3f" % (min(costs), max(costs)))


if __name__ == "__main__":
    main()
