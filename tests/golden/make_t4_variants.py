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


def frame_schedule(mask_red, cons, A0, numIter=19, nIter=8, lIter=400, trig=1):
    """the arap_deform schedule (CombinedSolverBase.h:99-120) from Python over oracle.solve, product arithmetic, with
    a chosen initial Angle image (oracle.frame itself always starts from Angle = 0, CombinedSolver.h:207-221)"""
    H, W = mask_red.shape
    ys, xs = np.mgrid[0:H, 0:W]
    U = np.stack([xs, ys], -1).astype(np.float32)
    O, A = U.copy(), A0.astype(np.float32)
    M = mask_red.astype(np.float32)
    allc = np.concatenate([np.asarray(cons, np.int32).reshape(-1, 4), orc.border_pins(W, H)])
    wf, wr = np.sqrt(np.float32(100.0)), np.sqrt(np.float32(0.01))
    costs = None
    for i in range(numIter):
        Cn = orc.constraint_image(mask_red, allc, np.float32(i + 1) / np.float32(numIter))
        O, A, costs = orc.solve(O, A, U, Cn, M, wf, wr, nIter, lIter, dtype=np.float32, mode=1, trig=trig)
    return O, A, costs


def perturbed_starts(cat, flows_ref, nseeds=6, amp=1e-6, trig=1, have=None):
    """How far does a perturbation far below anything physical move the answer?  Same arithmetic as the product (or, with
    trig = 0 and the no-FMA build selected by the caller, the variant f32_sum64_libm_nofma), the initial Angle image
    N(0, amp^2) rad instead of exactly 0 (amp = 1e-6 rad: 1e-4 px over a 100 px lever).  `have`: rows already made."""
    gold = cat["golden_flow"]
    act = cat["mask_red"] == 0
    H, W = act.shape
    # sanity: zero perturbation reproduces oracle.frame bit for bit
    O0, A0, c0 = frame_schedule(cat["mask_red"], cat["constraints"], np.zeros((H, W), np.float32), numIter=2, nIter=2, lIter=30, trig=trig)
    Of, Af, cf = orc.frame(cat["mask_red"], cat["constraints"], numIter=2, nIterations=2, lIterations=30, dtype=np.float32, mode=1, trig=trig)
    assert np.array_equal(O0, Of) and np.array_equal(A0, Af)
    rows = dict(have or {})
    for sd in range(nseeds):
        if "seed%d" % sd in rows:
            continue
        rng = np.random.default_rng(4242 + sd)
        A_init = (rng.normal(size=(H, W)) * amp).astype(np.float32) * act
        t = time.time()
        O, A, costs = frame_schedule(cat["mask_red"], cat["constraints"], A_init, trig=trig)
        flow = orc.flow_from_offset(O)
        err = np.linalg.norm(flow - gold, axis=-1)[act]
        rows["seed%d" % sd] = {"initial_angle_sigma_rad": amp, "final_cost": float(costs[-1]),
                               "rel_l2_vs_golden": float(helpers.rel_l2(flow[act], gold[act])),
                               "rel_l2_vs_unperturbed_product": float(helpers.rel_l2(flow[act], flows_ref[act])),
                               "median_px_vs_golden": float(np.median(err)),
                               "max_handle_px_vs_golden": max(float(np.abs(flow[y1, x1] - gold[y1, x1]).max())
                                                              for x1, y1, _, _ in cat["constraints"]),
                               "neg_det_quads": helpers.neg_det_quads(flow, act), "seconds": round(time.time() - t, 1)}
        print("perturbed", sd, json.dumps(rows["seed%d" % sd]), flush=True)
    return rows


def more_perturbed(n_product=20, n_nofma=12):
    """--perturbed-only: keep the variant rows of the committed table and (re)make only the perturbed starts -- n_product
    seeds of the product's arithmetic and n_nofma seeds of f32_sum64_libm_nofma (a second arithmetic, so that the
    empirical band is not the scatter of one rounding recipe alone).  ~1 minute per run on 8 cores."""
    cat = helpers.load_cat512(HERE)
    path = os.path.join(HERE, "t4_variants.json")
    tab = json.load(open(path))
    act = cat["mask_red"] == 0
    orc.use_variant(None)
    O, A, c = orc.frame(cat["mask_red"], cat["constraints"], dtype=np.float32, mode=1, trig=1)
    ref = orc.flow_from_offset(O)
    assert float(c[-1]) == tab["variants"]["f32_sum64_spec_fma"]["final_cost"]
    tab["perturbed_starts_product_arithmetic"] = perturbed_starts(
        cat, ref, nseeds=n_product, trig=1, have=tab.get("perturbed_starts_product_arithmetic"))
    json.dump(tab, open(path, "w"), indent=1)
    orc.use_variant("nofma")
    O, A, c = orc.frame(cat["mask_red"], cat["constraints"], dtype=np.float32, mode=1, trig=0)
    ref2 = orc.flow_from_offset(O)
    assert float(c[-1]) == tab["variants"]["f32_sum64_libm_nofma"]["final_cost"]
    tab["perturbed_starts_f32_sum64_libm_nofma"] = perturbed_starts(
        cat, ref2, nseeds=n_nofma, trig=0, have=tab.get("perturbed_starts_f32_sum64_libm_nofma"))
    orc.use_variant(None)
    json.dump(tab, open(path, "w"), indent=1)


def main():
    if "--perturbed-only" in sys.argv:
        return more_perturbed()
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
    pert = perturbed_starts(cat, flows["f32_sum64_spec_fma"])
    out = {"perturbed_starts_product_arithmetic": pert, "fixture": "tests/golden/cat512 (= ARAP/deformation/cat512_i{RGB,Msk}.png, cat512_iCstr.txt; golden "
                      "ARAP/warping/cat512_iFlo.flo), schedule 19/8/400",
           "golden_neg_det_quads": helpers.neg_det_quads(gold, act), "variants": rows, "mutual_rel_l2": mutual,
           "product_variant": "f32_sum64_spec_fma"}
    with open(os.path.join(HERE, "t4_variants.json"), "w") as f:
        json.dump(out, f, indent=1)
    costs = [r["final_cost"] for r in rows.values()] + [r["final_cost"] for r in pert.values()]
    print("cost range %.3f .. %.3f" % (min(costs), max(costs)))


if __name__ == "__main__":
    main()
