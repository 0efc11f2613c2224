"""ctypes loader for the CPU oracle (oracle/liboracle.so).

ORACLE -- TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by the product package arap_flow_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(ref=True):
    """Compile liboracle.so (and oracle/_ref/warp_image when the reference tree is present)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    if ref and os.path.isdir("/root/reference/ARAP/warping/src"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"], stderr=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        _LIB = C.CDLL(path)
    return _LIB


def use_variant(name=None):
    """Switch the loaded library: None = liboracle.so (the HIP path's bit-level twin), "nofma" = the same code with
    every fused multiply-add split into multiply + add (T4 calibration only)."""
    global _LIB
    if name is None:
        _LIB = None
        return lib()
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle_%s.so" % name])
    _LIB = C.CDLL(os.path.join(_HERE, "liboracle_%s.so" % name))
    return _LIB


def ref_warp_binary():
    p = os.path.join(_HERE, "_ref", "warp_image")
    return p if os.path.exists(p) else None


def set_trig(t):
    """cos/sin of the standalone kernel-level functions: 0 libm (default), 1 spec routine"""
    lib().oracle_set_trig(C.c_int(int(t)))


def _ct(dtype):
    return C.c_float if dtype == np.float32 else C.c_double


def _suf(dtype):
    return "_f32" if dtype == np.float32 else "_f64"


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _prep(dtype, *arrs):
    return [np.ascontiguousarray(a, dtype=dtype) for a in arrs]


def residuals(O, A, U, Cn, M, wf, wr, dtype=np.float64):
    H, W = A.shape
    O, A, U, Cn, M = _prep(dtype, O, A, U, Cn, M)
    out = np.zeros((H, W, 10), dtype)
    f = getattr(lib(), "oracle_residuals" + _suf(dtype))
    f.restype = None
    f(C.c_int(W), C.c_int(H), _p(O), _p(A), _p(U), _p(Cn), _p(M), _ct(dtype)(wf), _ct(dtype)(wr), _p(out))
    return out


def cost(O, A, U, Cn, M, wf, wr, dtype=np.float64, mode=1):
    H, W = A.shape
    O, A, U, Cn, M = _prep(dtype, O, A, U, Cn, M)
    f = getattr(lib(), "oracle_cost" + _suf(dtype))
    f.restype = C.c_double
    return f(C.c_int(W), C.c_int(H), _p(O), _p(A), _p(U), _p(Cn), _p(M), _ct(dtype)(wf), _ct(dtype)(wr),
             C.c_int(mode))


def evalJTF(O, A, U, Cn, M, wf, wr, dtype=np.float64):
    """returns (g[H,W,3], diag[H,W,3])"""
    H, W = A.shape
    O, A, U, Cn, M = _prep(dtype, O, A, U, Cn, M)
    g = np.zeros((H, W, 3), dtype)
    d = np.zeros((H, W, 3), dtype)
    f = getattr(lib(), "oracle_evalJTF" + _suf(dtype))
    f.restype = None
    f(C.c_int(W), C.c_int(H), _p(O), _p(A), _p(U), _p(Cn), _p(M), _ct(dtype)(wf), _ct(dtype)(wr), _p(g), _p(d))
    return g, d


def applyJTJ(A, U, Cn, M, wf, wr, P, dtype=np.float64):
    H, W = A.shape
    A, U, Cn, M, P = _prep(dtype, A, U, Cn, M, P)
    out = np.zeros((H, W, 3), dtype)
    f = getattr(lib(), "oracle_applyJTJ" + _suf(dtype))
    f.restype = None
    f(C.c_int(W), C.c_int(H), _p(A), _p(U), _p(Cn), _p(M), _ct(dtype)(wf), _ct(dtype)(wr), _p(P), _p(out))
    return out


def solve(O, A, U, Cn, M, wf, wr, nIterations, lIterations, dtype=np.float32, mode=1, trig=0):
    """One Opt_ProblemSolve.  Returns (O, A, costs[nIterations+1]); inputs are not modified."""
    H, W = A.shape
    O, A, U, Cn, M = _prep(dtype, O, A, U, Cn, M)
    O = O.copy()
    A = A.copy()
    costs = np.zeros(nIterations + 1, np.float64)
    f = getattr(lib(), "oracle_solve" + _suf(dtype))
    f.restype = C.c_double
    f(C.c_int(W), C.c_int(H), _p(O), _p(A), _p(U), _p(Cn), _p(M), _ct(dtype)(wf), _ct(dtype)(wr),
      C.c_int(nIterations), C.c_int(lIterations), C.c_int(mode), C.c_int(trig), _p(costs))
    return O, A, costs


LM_DEFAULTS = dict(min_relative_decrease=1e-3, min_trust_region_radius=1e-32, max_trust_region_radius=1e16,
                   q_tolerance=1e-4, function_tolerance=1e-6, trust_region_radius=1e4, radius_decrease_factor=2.0,
                   min_lm_diagonal=1e-6, max_lm_diagonal=1e32)      # solverGPUGaussNewton.t:26-39


def solve_lm(O, A, U, Cn, M, wf, wr, nIterations, lIterations, residual_reset_period=10, dtype=np.float32, trig=1,
             **lm):
    """One Opt_ProblemSolve with solver kind "LMGPU".  Returns (O, A, costs[0..steps], steps, final radius)."""
    H, W = A.shape
    O, A, U, Cn, M = _prep(dtype, O, A, U, Cn, M)
    O, A = O.copy(), A.copy()
    pars = dict(LM_DEFAULTS); pars.update(lm)
    order = ["min_relative_decrease", "min_trust_region_radius", "max_trust_region_radius", "q_tolerance",
             "function_tolerance", "trust_region_radius", "radius_decrease_factor", "min_lm_diagonal", "max_lm_diagonal"]
    lmv = np.asarray([pars[k] for k in order], dtype)
    costs = np.zeros(nIterations + 2, np.float64)
    rad = np.zeros(1, dtype)
    f = getattr(lib(), "oracle_solve_lm" + _suf(dtype))
    f.restype = C.c_int
    steps = f(C.c_int(W), C.c_int(H), _p(O), _p(A), _p(U), _p(Cn), _p(M), _ct(dtype)(wf), _ct(dtype)(wr),
              C.c_int(nIterations), C.c_int(lIterations), C.c_int(residual_reset_period), _p(lmv), C.c_int(trig),
              _p(costs), _p(rad))
    return O, A, costs[:steps + 1], steps, float(rad[0])


def frame(mask_red, cons, numIter=19, nIterations=8, lIterations=400, dtype=np.float32, mode=1, trig=0,
          border_pins=True):
    """Full arap_deform schedule for one frame.  mask_red u8[H,W]; cons int[n,4].
    Returns (Offset[H,W,2], Angle[H,W], final_costs[numIter])."""
    mask_red = np.ascontiguousarray(mask_red, np.uint8)
    H, W = mask_red.shape
    cons = np.ascontiguousarray(np.asarray(cons, np.int32).reshape(-1, 4))
    O = np.zeros((H, W, 2), dtype)
    A = np.zeros((H, W), dtype)
    costs = np.zeros(numIter, np.float64)
    f = getattr(lib(), "oracle_frame" + _suf(dtype))
    f.restype = None
    f(C.c_int(W), C.c_int(H), _p(mask_red), _p(cons), C.c_int(len(cons)), C.c_int(int(border_pins)),
      C.c_int(numIter), C.c_int(nIterations), C.c_int(lIterations), C.c_int(mode), C.c_int(trig),
      _p(O), _p(A), _p(costs))
    return O, A, costs


def constraint_image(mask_red, cons, alpha, dtype=np.float32):
    mask_red = np.ascontiguousarray(mask_red, np.uint8)
    H, W = mask_red.shape
    cons = np.ascontiguousarray(np.asarray(cons, np.int32).reshape(-1, 4))
    Cn = np.zeros((H, W, 2), dtype)
    f = getattr(lib(), "oracle_constraint_image" + _suf(dtype))
    f.restype = None
    f(C.c_int(W), C.c_int(H), _p(mask_red), _p(cons), C.c_int(len(cons)), C.c_float(alpha), _p(Cn))
    return Cn


def border_pins(W, H):
    """ARAP/deformation/src/main.cpp:130-136: (x,y,x,y) for every border pixel, row-major order."""
    out = []
    for y in range(H):
        for x in range(W):
            if y == 0 or x == 0 or y == H - 1 or x == W - 1:
                out.append((x, y, x, y))
    return np.asarray(out, np.int32).reshape(-1, 4)


def warp(rgb, mask_red, flow):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    mask_red = np.ascontiguousarray(mask_red, np.uint8)
    flow = np.ascontiguousarray(flow, np.float32)
    H, W = mask_red.shape
    out_rgb = np.zeros((H, W, 3), np.uint8)
    out_msk = np.zeros((H, W), np.uint8)
    f = lib().oracle_warp
    f.restype = None
    f(C.c_int(W), C.c_int(H), _p(rgb), _p(mask_red), _p(flow), _p(out_rgb), _p(out_msk))
    return out_rgb, out_msk


def warp_offset(rgb, mask_red, O):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    mask_red = np.ascontiguousarray(mask_red, np.uint8)
    O = np.ascontiguousarray(O, np.float32)
    H, W = mask_red.shape
    out_rgb = np.zeros((H, W, 3), np.uint8)
    out_msk = np.zeros((H, W), np.uint8)
    f = lib().oracle_warp_offset
    f.restype = None
    f(C.c_int(W), C.c_int(H), _p(rgb), _p(mask_red), _p(O), _p(out_rgb), _p(out_msk))
    return out_rgb, out_msk


def flow_from_offset(O):
    O = np.ascontiguousarray(O, np.float32)
    H, W = O.shape[:2]
    fl = np.zeros((H, W, 2), np.float32)
    f = lib().oracle_flow_from_offset
    f.restype = None
    f(C.c_int(W), C.c_int(H), _p(O), _p(fl))
    return fl


def sincos_spec(a):
    a = np.atleast_1d(np.asarray(a, np.float64))
    c = np.zeros_like(a)
    s = np.zeros_like(a)
    f = lib().arap_sincos_spec
    f.restype = None
    cd, sd = C.c_double(), C.c_double()
    for i, v in enumerate(a):
        f(C.c_double(v), C.byref(cd), C.byref(sd))
        c[i], s[i] = cd.value, sd.value
    return c, s
