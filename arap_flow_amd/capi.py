"""ctypes binding of libarapopt.so (include/arap_opt.h).

There is no CPU fallback: if the HIP library is missing or no GPU is present, loading/creating a
state raises.  Nothing here imports the oracle.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ARAPOPT_LIB") or os.path.join(_HERE, "lib", "libarapopt.so")     # (ARAPOPT_LIB: kernel experiments)


class Opt_InitializationParameters(C.Structure):
    """Opt.h:10-30"""
    _fields_ = [("doublePrecision", C.c_int), ("verbosityLevel", C.c_int),
                ("collectPerKernelTimingInfo", C.c_int), ("threadsPerBlock", C.c_int)]


# every symbol include/arap_opt.h declares: (name, restype, argtypes)
_VP, _U, _I = C.c_void_p, C.c_uint, C.c_int
SYMBOLS = [
    ("Opt_NewState", _VP, [Opt_InitializationParameters]),
    ("Opt_ProblemDefine", _VP, [_VP, C.c_char_p, C.c_char_p]),
    ("Opt_ProblemDelete", None, [_VP, _VP]),
    ("Opt_ProblemPlan", _VP, [_VP, _VP, C.POINTER(C.c_uint)]),
    ("Opt_PlanFree", None, [_VP, _VP]),
    ("Opt_SetSolverParameter", None, [_VP, _VP, C.c_char_p, _VP]),
    ("Opt_ProblemSolve", None, [_VP, _VP, C.POINTER(_VP)]),
    ("Opt_ProblemInit", None, [_VP, _VP, C.POINTER(_VP)]),
    ("Opt_ProblemStep", _I, [_VP, _VP, C.POINTER(_VP)]),
    ("Opt_ProblemCurrentCost", C.c_double, [_VP, _VP]),
    ("ArapFlow_Version", C.c_char_p, []),
    ("ArapFlow_FreeState", None, [_VP]),
    ("ArapFlow_SetStream", None, [_VP, _VP]),
    ("ArapFlow_UseOwnStream", _I, [_VP]),
    ("ArapFlow_TimerBegin", None, [_VP]),
    ("ArapFlow_TimerEnd", C.c_float, [_VP]),
    ("ArapFlow_SetKernelTiming", None, [_VP, _I]),
    ("ArapFlow_KernelTime", _I, [_VP, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    ("ArapFlow_EvalJTF", _I, [_VP, _U, _U, _VP, _VP, _VP, _VP, _VP, C.c_float, C.c_float, _VP, _VP, _VP, _VP]),
    ("ArapFlow_ApplyJTJ", _I, [_VP, _U, _U, _VP, _VP, _VP, _VP, C.c_float, C.c_float, _VP, _VP, _VP, _VP]),
    ("ArapFlow_Cost", _I, [_VP, _U, _U, _VP, _VP, _VP, _VP, _VP, C.c_float, C.c_float, C.POINTER(C.c_double)]),
    ("ArapFlow_SolverCreate", _VP, [_VP, _U, _U, _U]),
    ("ArapFlow_SolverFree", None, [_VP]),
    ("ArapFlow_SolverSetFrame", _I, [_VP, _U, _VP, _VP, _VP, _U, _I]),
    ("ArapFlow_SolverSolve", _I, [_VP, _U, _U, _U, _U]),
    ("ArapFlow_SolverSolveAsync", _I, [_VP, _U, _U, _U, _U, _I, _I]),
    ("ArapFlow_SolverWait", _I, [_VP]),
    ("ArapFlow_SolverHostResults", _I, [_VP, _U, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP)]),
    ("ArapFlow_SolverWarp", _I, [_VP, _U]),
    ("ArapFlow_SolverGetResults", _I, [_VP, _U, _VP, _VP, _VP, _VP, _VP, C.POINTER(C.c_double)]),
    ("ArapFlow_SolverStats", _I, [_VP, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("ArapFlow_SetResident", None, [_VP, _I]),
    ("ArapFlow_SetTile", _I, [_VP, _I, _I]),
    ("ArapFlow_SolverResidentLaunches", C.c_uint64, [_VP]),
    ("ArapFlow_PlanResidentLaunches", C.c_uint64, [_VP]),
    ("ArapFlow_SolverLaunchesFor", _I, [_VP, C.c_uint]),
    ("ArapFlow_ResidentDeal", _I, [C.POINTER(C.c_int), C.c_uint, C.POINTER(C.c_int), C.c_uint]),
    ("ArapFlow_ResidentTiles", _I, [_VP, _U, _U, _I, C.POINTER(C.c_int), _U, C.POINTER(C.c_int)]),
    ("ArapFlow_SolverResidentLayout", _I, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("ArapFlow_SolverLeanStream", _I, [_VP]),
    ("ArapFlow_ResidentFailed", _I, [_VP]),
    ("ArapFlow_SolverStamps", _I, [_VP, _VP]),
    ("ArapFlow_WarpScratchBytes", C.c_uint64, [_U, _U]),
    ("ArapFlow_Warp", _I, [_VP, _U, _U, _VP, _VP, _VP, _VP, _VP, _VP]),
]

_LIB = None


def load():
    """Load libarapopt.so and type every exported entry point.  Raises if it was not built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libarapopt.so not built (%s): run `python -m arap_flow_amd.build` or "
                "__graft_entry__.build(); there is no CPU fallback" % LIB_PATH)
        # One HIP runtime per process: libarapopt.so links /opt/rocm's libamdhip64, torch (which owns the device
        # buffers in the tests and the bench) bundles its own copy.  Whichever is loaded first serves both, but if
        # this library came first and torch second, HIP reports no device to this library.  Load torch first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)     # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB
