"""BASELINE.json configs[0] (64x64 mesh, mask == 0, 4 handles displaced by (+-3, +-3), schedule 1 ramp step x 10
Gauss-Newton steps x 400 PCG iterations): the record the survey asks for -- cost trajectory, handle
error, wall time of the CPU solver of the same energy (the oracle; the reference has no Ceres path, SURVEY preamble 2),
and, on a GPU box, the same solve through the C ABI with its time and the bit comparison.

    python tools/config0_record.py [--out profiles/r03_config0_64x64.json]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as orc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "profiles", "r03_config0_64x64.json"))
    a = ap.parse_args()
    W = H = 64
    mask = np.zeros((H, W), np.uint8)
    cons = np.asarray([(16, 16, 19, 19), (48, 16, 45, 19), (16, 48, 19, 45), (48, 48, 51, 51)], np.int32)
    sched = (1, 10, 400)
    rec = {"config": "BASELINE.json configs[0]: 64x64 mesh, 4 pinned handles, 10 GN iterations x 400 PCG iterations, mask == 0",
           "schedule": list(sched), "constraints": cons.tolist()}
    orc.frame(mask, cons, numIter=1, nIterations=1, lIterations=10, dtype=np.float32, mode=1, trig=1)      # (library load)
    t = time.time()
    O, A, costs = orc.frame(mask, cons, numIter=sched[0], nIterations=sched[1], lIterations=sched[2], dtype=np.float32, mode=1, trig=1)
    dt = time.time() - t
    herr = [float(np.hypot(O[y1, x1, 0] - x2, O[y1, x1, 1] - y2)) for x1, y1, x2, y2 in cons]
    # the cost after Init and after each of the 10 Gauss-Newton steps: the same solve through oracle.solve (one ramp step at
    # alpha = 1: the constraint image with the border pins, CombinedSolver.h:223-242), same bits
    ys, xs = np.mgrid[0:H, 0:W]
    U = np.stack([xs, ys], -1).astype(np.float32)
    allc = np.concatenate([cons, orc.border_pins(W, H)])
    Cn = orc.constraint_image(mask, allc, np.float32(1.0))
    O2, A2, costs = orc.solve(U.copy(), np.zeros((H, W), np.float32), U, Cn, mask.astype(np.float32), np.sqrt(np.float32(100.0)),
                              np.sqrt(np.float32(0.01)), sched[1], sched[2], dtype=np.float32, mode=1, trig=1)
    assert np.array_equal(O2, O) and np.array_equal(A2, A)
    rec["cpu_oracle"] = {"wall_seconds": dt, "threads": int(os.environ.get("OMP_NUM_THREADS", "0")) or None,
                         "cost_trajectory": [float(c) for c in costs], "handle_error_px": herr,
                         "kind": "port (CPU restatement of the same energy, float32, float64 sums)"}
    gold = np.load(os.path.join(ROOT, "tests", "golden", "solve_64_1x10x400.npz"))
    rec["cpu_oracle"]["equals_committed_golden"] = bool(np.array_equal(gold["offset"], O) and np.array_equal(gold["angle"], A))
    try:
        import torch
        if torch.cuda.is_available():
            from arap_flow_amd import opt
            st = opt.State()
            fs = opt.FrameSolver(st, W, H, batch=1)
            fs.set_frame(0, mask, cons)
            fs.solve(1, *sched)                                   # warm-up (graph capture)
            torch.cuda.synchronize()
            t = time.time()
            fs.solve(1, *sched)
            torch.cuda.synchronize()
            g = time.time() - t
            r = fs.results(0, want_rgb=False)
            rec["gpu"] = {"wall_seconds": g, "final_cost": float(r["cost"]),
                          "bit_equal_to_cpu_oracle": bool(np.array_equal(r["offset"], O) and np.array_equal(r["angle"], A)),
                          "resident_launches": fs.stats()["resident_launches"],
                          "note": "one 64x64 solve occupies one group of the resident launch; time = 10 launches of 400 iterations"}
            fs.close(); st.close()
    except Exception as e:                                        # the CPU part stands without a GPU
        rec["gpu"] = {"error": str(e)[:200]}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    open(a.out, "w").write(json.dumps(rec, indent=1) + "\n")
    print(json.dumps(rec)[:1500])


if __name__ == "__main__":
    main()
