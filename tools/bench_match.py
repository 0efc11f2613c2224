"""GPU box: the matcher (libarapmatch.so) on synthetic 854x480 pairs -- pairs/s and, under rocprofv3 --kernel-trace
--stats, the time per kernel.   python tools/bench_match.py [pairs] [W H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arap_flow_amd import match, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (854, 480)
rng = np.random.default_rng(0)
big = np.clip(synth.make_rgb(W + 128, H + 128, 1).astype(np.int32) + rng.integers(-40, 41, (H + 128, W + 128, 1)), 0, 255).astype(np.uint8)
a = np.ascontiguousarray(big[64:64 + H, 64:64 + W])
b = np.ascontiguousarray(big[64 + 9:64 + 9 + H, 64 - 17:64 - 17 + W])
mt = match.Matcher(W, H, 100)
m = mt.run(a, b)
t = time.time(); ms = []; cms = []
for _ in range(n):
    m = mt.run(a, b); ms.append(mt.last_ms()); cms.append(mt.last_corr_ms())
dt = time.time() - t
d = m[:, 2:4] - m[:, 0:2]
print("%dx%d: %d matches, median displacement %s; %.2f ms device per pair (median), %.1f pairs/s incl. host copies"
      % (W, H, len(m), np.median(d, axis=0), float(np.median(ms)), n / dt))
# useful work of the bottom-level correlation: patches x placements x 144 multiply-adds
gh, gw, S, c = mt.levels()[0]
print("bottom level: %d patches x %d placements: %.1f GFLOP useful" % (gh * gw, S * S, gh * gw * S * S * 144 * 2 / 1e9))
# MFMA roofline of the dominant kernel: executed = workgroups x displacements x N-tiles x 72 MFMAs x (32 x 32 x 2 x 2 flop)
import json
r = (S - 1) // 2
executed = ((gw + 31) // 32) * gh * S * ((128 + 2 * r + 31) // 32) * 72 * 4096.0
useful = gh * gw * S * S * 144 * 2.0
c = float(np.median(cms)) * 1e-3
print(json.dumps({"metric": "matcher pairs/s at %dx%d (-ngh_rad 100)" % (W, H), "value": n / dt, "unit": "pairs/s",
                  "device_ms_per_pair": float(np.median(ms)),
                  "roofline": {"bound": "mfma", "kernel": "k_corr0", "dtype": "f32", "achieved": executed / c / 1e12, "peak": 157.3,
                               "unit": "TFLOP/s", "frac": executed / c / 1e12 / 157.3, "useful_TFLOPs": useful / c / 1e12,
                               "avg_launch_us": c * 1e6, "traffic": None,
                               "note": "flop of the v_mfma_f32_32x32x2_f32 issued / kernel time (HIP events); 44 % of the products lie "
                                       "inside a patch's own window at r = 50 (useful); peak: f32 MFMA, MI355X_MICROARCH.md"}}))
mt.close()
