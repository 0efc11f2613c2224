"""Host-side mirror of the reference's C++ driver layer, over the C ABI of libarapopt.so.

  OptSolver        ARAP/shared/OptSolver.h:43-91        (Opt_NewState/ProblemDefine/ProblemPlan, solve)
  NamedParameters  ARAP/shared/NamedParameters.h:34-87  (name -> pointer table, insertion order = void**)
  CombinedSolver   ARAP/deformation/src/CombinedSolver.h:99-390 + ARAP/shared/CombinedSolverBase.h:23-120
                   (the reference's own host loop: ramp on the host, one Opt_ProblemSolve per ramp step)
  FrameSolver      the device-resident batched counterpart (ArapFlow_Solver*, include/arap_opt.h part 2)
  warp_image       ARAP/warping/src/main.cpp:145-225 through ArapFlow_Warp

torch is used only to own device memory (tensor.data_ptr()) and streams.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import capi

BUILTIN_PLAN = b"builtin:arap"


def _dev_ptr(t):
    assert t.is_cuda and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


class State:
    """Opt_State (Opt.h:35).  One per process/device; shared by solvers."""

    def __init__(self, verbosity=0, timing=False):
        self.lib = capi.load()
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU: libarapopt has no CPU path")
        torch.cuda.init()
        torch.zeros(1, device="cuda")  # make sure the primary context of the current device exists
        ip = capi.Opt_InitializationParameters(0, int(verbosity), int(bool(timing)), 0)
        self.handle = self.lib.Opt_NewState(ip)
        if not self.handle:
            raise RuntimeError("Opt_NewState failed")

    def use_own_stream(self):
        """non-blocking compute stream owned by the library (hosts that overlap copies with solves)"""
        self.lib.ArapFlow_UseOwnStream(self.handle)

    def set_stream(self, stream=None):
        self.lib.ArapFlow_SetStream(self.handle, C.c_void_p(stream.cuda_stream if stream is not None else 0))

    def timer_begin(self):
        self.lib.ArapFlow_TimerBegin(self.handle)

    def timer_end(self):
        return float(self.lib.ArapFlow_TimerEnd(self.handle))

    def set_resident(self, on):
        """allow (default) / forbid the on-chip resident PCG kernel of the frame solver"""
        self.lib.ArapFlow_SetResident(self.handle, int(bool(on)))

    def set_tile(self, tx, ty):
        """phase-A variant of the two-kernel path: (0, 0) direct loads, or an LDS tile shape"""
        if self.lib.ArapFlow_SetTile(self.handle, int(tx), int(ty)) != 0:
            raise ValueError("unsupported tile %dx%d" % (tx, ty))

    def set_kernel_timing(self, on):
        self.lib.ArapFlow_SetKernelTiming(self.handle, int(bool(on)))

    def kernel_time(self, name):
        """(total_ms, launches) of the named kernel since timing was switched on, or None"""
        t, n = C.c_double(), C.c_uint64()
        if self.lib.ArapFlow_KernelTime(self.handle, name.encode(), C.byref(t), C.byref(n)) != 0:
            return None
        return t.value, n.value

    def close(self):
        if self.handle:
            self.lib.ArapFlow_FreeState(self.handle)
            self.handle = None


class NamedParameters:
    """NamedParameters.h:34-87: ordered name -> value table; data() is the void** of the C API.
    Device images are torch CUDA tensors; scalars are kept in host ctypes floats/ints."""

    def __init__(self):
        self._names, self._vals, self._keep = [], [], []

    def set(self, name, value):
        if isinstance(value, torch.Tensor):
            holder, ptr = value, C.c_void_p(value.data_ptr())
        elif isinstance(value, int):
            holder = C.c_int(value)
            ptr = C.cast(C.pointer(holder), C.c_void_p)
        else:
            holder = C.c_float(float(value))
            ptr = C.cast(C.pointer(holder), C.c_void_p)
        if name in self._names:
            k = self._names.index(name)
            self._vals[k], self._keep[k] = ptr, holder
        else:
            self._names.append(name)
            self._vals.append(ptr)
            self._keep.append(holder)

    def names(self):
        return list(self._names)

    def items(self):
        return list(zip(self._names, self._vals))

    def data(self):
        arr = (C.c_void_p * len(self._vals))(*self._vals)
        return arr


class OptSolver:
    """OptSolver.h:43-91."""

    def __init__(self, state, dims, plan_file=BUILTIN_PLAN, solverkind=b"gaussNewtonGPU"):
        self.state, self.lib = state, state.lib
        if isinstance(plan_file, str):
            plan_file = plan_file.encode()
        if isinstance(solverkind, str):
            solverkind = solverkind.encode()
        self.problem = self.lib.Opt_ProblemDefine(state.handle, plan_file, solverkind)
        assert self.problem, "Opt_ProblemDefine returned NULL"       # OptSolver.h:55
        d = (C.c_uint * 2)(int(dims[0]), int(dims[1]))
        self.plan = self.lib.Opt_ProblemPlan(state.handle, self.problem, d)
        assert self.plan, "Opt_ProblemPlan returned NULL"            # OptSolver.h:56
        self.final_cost = float("nan")

    def set_solver_parameters(self, solver_params):
        for name, ptr in solver_params.items():                      # OptUtils.h:104-108
            self.lib.Opt_SetSolverParameter(self.state.handle, self.plan, name.encode(), ptr)

    def solve(self, solver_params, problem_params, profiled=False, iters=None):
        """OptSolver.h:72-91.  profiled=True is launchProfiledSolve (OptUtils.h:47-64): Init, then Step by Step,
        appending a (cost, milliseconds) record per Gauss-Newton iteration to `iters` (SolverIteration.h)."""
        self.set_solver_parameters(solver_params)
        if profiled:
            import time
            import torch
            recs = iters if iters is not None else []
            self.lib.Opt_ProblemInit(self.state.handle, self.plan, problem_params.data())
            torch.cuda.synchronize()
            recs.append((self.lib.Opt_ProblemCurrentCost(self.state.handle, self.plan), 0.0))
            while True:
                t0 = time.perf_counter()
                more = self.lib.Opt_ProblemStep(self.state.handle, self.plan, problem_params.data())
                if not more:
                    break
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3
                recs.append((self.lib.Opt_ProblemCurrentCost(self.state.handle, self.plan), ms))
        else:
            self.lib.Opt_ProblemSolve(self.state.handle, self.plan, problem_params.data())
        self.final_cost = self.lib.Opt_ProblemCurrentCost(self.state.handle, self.plan)
        return self.final_cost

    def init(self, problem_params):
        self.lib.Opt_ProblemInit(self.state.handle, self.plan, problem_params.data())

    def step(self, problem_params):
        return self.lib.Opt_ProblemStep(self.state.handle, self.plan, problem_params.data())

    def current_cost(self):
        return self.lib.Opt_ProblemCurrentCost(self.state.handle, self.plan)

    def resident_launches(self):
        return int(self.lib.ArapFlow_PlanResidentLaunches(self.plan))

    def close(self):
        if self.plan:
            self.lib.Opt_PlanFree(self.state.handle, self.plan)
            self.plan = None
        if self.problem:
            self.lib.Opt_ProblemDelete(self.state.handle, self.problem)
            self.problem = None


def load_constraints(path):
    """main.cpp:26-50: first token n, then n x 4 ints."""
    with open(path) as f:
        tok = f.read().split()
    n = int(tok[0])
    vals = [int(t) for t in tok[1:1 + 4 * n]]
    return np.asarray(vals, np.int32).reshape(n, 4)


def border_pins(W, H):
    """main.cpp:130-136"""
    ys, xs = np.mgrid[0:H, 0:W]
    sel = (ys == 0) | (xs == 0) | (ys == H - 1) | (xs == W - 1)
    x, y = xs[sel], ys[sel]                  # row-major order, as the reference's nested loops
    return np.stack([x, y, x, y], -1).astype(np.int32)


class CombinedSolver:
    """The reference's host loop, kept as it is in the reference (ramp built on the host and uploaded,
    one Opt_ProblemSolve per ramp step), driving the drop-in Opt_* symbols only.

    CombinedSolver.h:105-112,139-170 (ctor/addImage), :172-189 (combinedSolveInit), :207-221
    (resetGPU), :223-242 (setConstraintImage); CombinedSolverBase.h:23-31,99-120 (solveAll)."""

    def __init__(self, state, width, height, plan_file=BUILTIN_PLAN, num_iter=19, non_linear_iter=8,
                 linear_iter=400):
        self.state = state
        self.W, self.H = int(width), int(height)
        self.num_iter, self.non_linear_iter, self.linear_iter = num_iter, non_linear_iter, linear_iter
        self.solver = OptSolver(state, (self.W, self.H), plan_file, b"gaussNewtonGPU")
        self.mask_red = None
        self.constraints = None
        self.final_costs = []

    def add_image(self, mask_red, constraints):
        H, W = self.H, self.W
        assert mask_red.shape == (H, W)
        self.mask_red = np.ascontiguousarray(mask_red, np.uint8)
        self.constraints = np.asarray(constraints, np.int32).reshape(-1, 4)
        dev = "cuda"
        self.urshape = torch.empty(H, W, 2, dtype=torch.float32, device=dev)
        self.warp_field = torch.empty(H, W, 2, dtype=torch.float32, device=dev)
        self.warp_angles = torch.empty(H, W, dtype=torch.float32, device=dev)
        self.constraint_image = torch.empty(H, W, 2, dtype=torch.float32, device=dev)
        self.mask = torch.empty(H, W, dtype=torch.float32, device=dev)
        self.reset_gpu()

    def reset_gpu(self):
        H, W = self.H, self.W
        ys, xs = np.mgrid[0:H, 0:W]
        h_ur = np.stack([xs, ys], -1).astype(np.float32)
        self.set_constraint_image(1.0)
        self.urshape.copy_(torch.from_numpy(h_ur))
        self.warp_field.copy_(torch.from_numpy(h_ur))
        self.mask.copy_(torch.from_numpy(self.mask_red.astype(np.float32)))
        self.warp_angles.zero_()

    def set_constraint_image(self, alpha):
        H, W = self.H, self.W
        alpha = np.float32(alpha)
        h = np.full((H, W, 2), -1.0, np.float32)
        one = np.float32(1.0)
        for x, y, tx, ty in self.constraints:          # later entries overwrite earlier ones
            if self.mask_red[y, x] == 0:
                h[y, x, 0] = (one - alpha) * np.float32(x) + alpha * np.float32(tx)
                h[y, x, 1] = (one - alpha) * np.float32(y) + alpha * np.float32(ty)
        self.constraint_image.copy_(torch.from_numpy(h))

    def solve_all(self):
        w_fit_sqrt = math.sqrt(np.float32(100.0))
        w_reg_sqrt = float(np.sqrt(np.float32(0.01)))
        pp = NamedParameters()
        pp.set("Offset", self.warp_field)
        pp.set("Angle", self.warp_angles)
        pp.set("UrShape", self.urshape)
        pp.set("Constraints", self.constraint_image)
        pp.set("Mask", self.mask)
        pp.set("w_fitSqrt", float(np.float32(w_fit_sqrt)))
        pp.set("w_regSqrt", float(np.float32(w_reg_sqrt)))
        sp = NamedParameters()
        sp.set("nIterations", int(self.non_linear_iter))
        sp.set("lIterations", int(self.linear_iter))
        self.reset_gpu()                                   # preSingleSolve
        self.final_costs = []
        for i in range(self.num_iter):
            self.set_constraint_image(np.float32(i + 1) / np.float32(self.num_iter))
            self.final_costs.append(self.solver.solve(sp, pp))
        return self.final_costs

    def warp_field_as_flow(self):
        """warpField(): CombinedSolver.h:352-366"""
        H, W = self.H, self.W
        o = self.warp_field.cpu().numpy()
        ys, xs = np.mgrid[0:H, 0:W]
        o[..., 0] -= xs.astype(np.float32)
        o[..., 1] -= ys.astype(np.float32)
        return o

    def close(self):
        self.solver.close()


class FrameSolver:
    """Batched, device-resident CombinedSolver (ArapFlow_Solver)."""

    def __init__(self, state, width, height, batch=1):
        self.state, self.lib = state, state.lib
        self.W, self.H, self.batch = int(width), int(height), int(batch)
        self.h = self.lib.ArapFlow_SolverCreate(state.handle, self.W, self.H, self.batch)
        if not self.h:
            raise RuntimeError("ArapFlow_SolverCreate failed")

    def set_frame(self, slot, mask_red, constraints, rgb=None, border_pins=True):
        mask_red = np.ascontiguousarray(mask_red, np.uint8)
        assert mask_red.shape == (self.H, self.W)
        cons = np.ascontiguousarray(np.asarray(constraints, np.int32).reshape(-1, 4))
        rgbp = None
        if rgb is not None:
            rgb = np.ascontiguousarray(rgb, np.uint8)
            assert rgb.shape == (self.H, self.W, 3)
            rgbp = rgb.ctypes.data_as(C.c_void_p)
        rc = self.lib.ArapFlow_SolverSetFrame(self.h, slot, rgbp, mask_red.ctypes.data_as(C.c_void_p),
                                              cons.ctypes.data_as(C.c_void_p), len(cons), int(border_pins))
        if rc != 0:
            raise ValueError("ArapFlow_SolverSetFrame: bad arguments")

    def launches_for(self, nframes):
        """resident launches per Gauss-Newton step a solve of slots [0, nframes) would take (0: two-kernel path)"""
        return int(self.lib.ArapFlow_SolverLaunchesFor(self.h, int(nframes)))

    def solve(self, nframes=None, num_iter=19, non_linear_iter=8, linear_iter=400):
        n = self.batch if nframes is None else nframes
        rc = self.lib.ArapFlow_SolverSolve(self.h, n, num_iter, non_linear_iter, linear_iter)
        if rc != 0:
            raise ValueError("ArapFlow_SolverSolve: bad arguments")

    def solve_async(self, nframes=None, num_iter=19, non_linear_iter=8, linear_iter=400, warp=True, download=True):
        """ArapFlow_SolverSolveAsync: enqueue schedule (+ warp, + download into pinned host buffers), do not wait"""
        n = self.batch if nframes is None else nframes
        rc = self.lib.ArapFlow_SolverSolveAsync(self.h, n, num_iter, non_linear_iter, linear_iter, int(warp), int(download))
        if rc != 0:
            raise ValueError("ArapFlow_SolverSolveAsync: bad arguments")

    def wait(self):
        rc = self.lib.ArapFlow_SolverWait(self.h)
        if rc != 0:
            raise RuntimeError("ArapFlow_SolverWait failed: %d" % rc)

    def host_results(self, slot):
        """views (no copy) of the pinned result buffers of a `download` solve: valid until this solver's next solve"""
        H, W = self.H, self.W
        pf, pr, pm = C.c_void_p(), C.c_void_p(), C.c_void_p()
        rc = self.lib.ArapFlow_SolverHostResults(self.h, slot, C.byref(pf), C.byref(pr), C.byref(pm))
        if rc != 0:
            raise ValueError("ArapFlow_SolverHostResults: no downloaded results for slot %d" % slot)
        def view(ptr, ctype, shape):
            n = int(np.prod(shape))
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(n,)).reshape(shape)
        return dict(flow=view(pf, C.c_float, (H, W, 2)),
                    warped_rgb=view(pr, C.c_uint8, (H, W, 3)) if pr.value else None,
                    warped_mask=view(pm, C.c_uint8, (H, W)))

    def warp(self, nframes=None):
        n = self.batch if nframes is None else nframes
        rc = self.lib.ArapFlow_SolverWarp(self.h, n)
        if rc != 0:
            raise ValueError("ArapFlow_SolverWarp: bad arguments")

    def results(self, slot, want_rgb=True):
        H, W = self.H, self.W
        flow = np.empty((H, W, 2), np.float32)
        wrgb = np.empty((H, W, 3), np.uint8) if want_rgb else None
        wmsk = np.empty((H, W), np.uint8)
        off = np.empty((H, W, 2), np.float32)
        ang = np.empty((H, W), np.float32)
        cost = C.c_double(0.0)
        p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        rc = self.lib.ArapFlow_SolverGetResults(self.h, slot, p(flow), p(wrgb), p(wmsk), p(off), p(ang),
                                                C.byref(cost))
        if rc != 0:
            raise ValueError("ArapFlow_SolverGetResults: bad arguments")
        return dict(flow=flow, warped_rgb=wrgb, warped_mask=wmsk, offset=off, angle=ang, cost=cost.value)

    def stats(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.lib.ArapFlow_SolverStats(self.h, C.byref(a), C.byref(b), C.byref(c))
        ls, fl = C.c_int(0), C.c_int(0)
        self.lib.ArapFlow_SolverResidentLayout(self.h, C.byref(ls), C.byref(fl))
        return dict(pcg_iterations_per_frame=a.value, active_vertices=b.value, grid_vertices=c.value,
                    resident_launches=int(self.lib.ArapFlow_SolverResidentLaunches(self.h)),
                    resident_launches_per_step=ls.value, resident_solves_in_flight=fl.value,
                    lean_stream=bool(self.lib.ArapFlow_SolverLeanStream(self.h)))

    def close(self):
        if self.h:
            self.lib.ArapFlow_SolverFree(self.h)
            self.h = None


def warp_image(state, rgb, mask_red, flow):
    """warp_image (ARAP/warping/src/main.cpp:302-336 minus file I/O) on the GPU.
    rgb u8[H,W,3], mask_red u8[H,W], flow f32[H,W,2] (numpy) -> (warped_rgb, warped_mask)."""
    lib = state.lib
    H, W = mask_red.shape
    d_rgb = torch.from_numpy(np.ascontiguousarray(rgb, np.uint8)).cuda()
    d_msk = torch.from_numpy(np.ascontiguousarray(mask_red, np.uint8)).cuda()
    d_flow = torch.from_numpy(np.ascontiguousarray(flow, np.float32)).cuda()
    o_rgb = torch.empty(H, W, 3, dtype=torch.uint8, device="cuda")
    o_msk = torch.empty(H, W, dtype=torch.uint8, device="cuda")
    scratch = torch.empty(int(lib.ArapFlow_WarpScratchBytes(W, H)), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    rc = lib.ArapFlow_Warp(state.handle, W, H, _dev_ptr(d_rgb), _dev_ptr(d_msk), _dev_ptr(d_flow), _dev_ptr(o_rgb),
                           _dev_ptr(o_msk), _dev_ptr(scratch))
    if rc != 0:
        raise RuntimeError("ArapFlow_Warp failed: %d" % rc)
    torch.cuda.synchronize()
    return o_rgb.cpu().numpy(), o_msk.cpu().numpy()
