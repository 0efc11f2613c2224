"""BASELINE config 5: LDS tile sweep of phase A of the two-kernel path + HBM roofline report.
   python tools/tile_sweep.py            (prints a table; copy it under profiles/)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from arap_flow_amd import opt, synth
st = opt.State()
st.set_resident(False)
L = 100
print("# phase A (k_pcg_a direct / k_pcg_a_lds<TX,TY> tiled, both with k_pcg_b4; 'march' = the default for the frame solver:")
print("# k_pcg_a_march<5> 64-column strips + k_pcg_b4_lean) of the two-kernel path, HIP events around every launch, %d PCG iterations" % L)
print("# algorithmic bytes per active vertex: A 64 B, B 96 B (SURVEY 8d); peak 8000 GB/s")
print("%-26s %-8s %10s %10s %10s %10s" % ("workload", "tile", "A us", "A GB/s", "B us", "B GB/s"))
for (W, H, B, K, full, name) in [(1920, 1080, 1, 1, True, "1920x1080 mask==0 x1"), (1920, 1080, 1, 3, False, "1920x1080 K=3 fd=5 x1"),
                                 (854, 480, 4, 1, True, "854x480 mask==0 x4"), (854, 480, 8, 1, False, "854x480 davis x8")]:
    fr = [synth.make_frame(W, H, seed=s, K=K, fd=5 if K > 1 else 1, full_mask=full) for s in range(B)]
    fs = opt.FrameSolver(st, W, H, batch=B)
    for b, f in enumerate(fr):
        fs.set_frame(b, f["mask_red"], f["constraints"])
    nact = sum(int((f["mask_red"] == 0).sum()) for f in fr)
    for tile in [(0, 0), (16, 16), (32, 8), (64, 4), (32, 16), (64, 8), (-1, -1)]:
        st.set_tile(*tile)
        fs.solve(B, 1, 1, 10); torch.cuda.synchronize()
        st.set_kernel_timing(True)
        fs.solve(B, 1, 1, L); torch.cuda.synchronize()
        ta, na = st.kernel_time("PCGStepA"); tb, nb_ = st.kernel_time("PCGStepB")
        st.set_kernel_timing(False)
        ua, ub = ta / na * 1e3, tb / nb_ * 1e3
        print("%-26s %-8s %10.2f %10.0f %10.2f %10.0f" % (name, "direct" if tile == (0, 0) else ("march" if tile[0] < 0 else "%dx%d" % tile), ua,
                                                        64.0 * nact / ua / 1e3, ub, 96.0 * nact / ub / 1e3))
    fs.close()
st.set_tile(-1, -1)
