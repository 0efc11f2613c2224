"""Time the reference's own host loop (ramp on the host, 19 x Opt_ProblemSolve, CombinedSolverBase.h:99-120) over the
ten drop-in Opt_* symbols, one 854x480 DAVIS-shaped frame at a time -- what a maintainer gets by relinking only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from arap_flow_amd import opt, synth
st = opt.State()
W, H = 854, 480
f = synth.make_frame(W, H, seed=0)
cons = np.concatenate([f["constraints"], opt.border_pins(W, H)])
for resident in (True, False):
    st.set_resident(resident)
    cs = opt.CombinedSolver(st, W, H)
    cs.add_image(f["mask_red"], cons)
    cs.solve_all(); torch.cuda.synchronize()
    t = time.perf_counter(); costs = cs.solve_all(); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("drop-in Opt_* path, resident=%s: %.3f s/frame = %.2f frames/s, final cost %.4f, resident launches %d"
          % (resident, dt, 1 / dt, costs[-1], cs.solver.resident_launches()))
    cs.close()
