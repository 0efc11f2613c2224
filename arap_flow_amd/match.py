"""ctypes binding of libarapmatch.so (include/arap_match.h): dense matching of a frame pair on the GPU, the stage the
reference delegates to the external DeepMatching binary (/root/reference/para_gen.py:227-240).

No CPU fallback: load() raises when the library is not built, Matcher() when there is no GPU.  Nothing here imports
the oracle."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ARAPMATCH_LIB") or os.path.join(_HERE, "lib", "libarapmatch.so")

_VP, _U, _I = C.c_void_p, C.c_uint, C.c_int
SYMBOLS = [
    ("ArapMatch_Create", _VP, [_U, _U, _U]),
    ("ArapMatch_Free", None, [_VP]),
    ("ArapMatch_Run", _I, [_VP, _VP, _VP, _VP, _U]),
    ("ArapMatch_Levels", _I, [_VP]),
    ("ArapMatch_LevelInfo", _I, [_VP, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    ("ArapMatch_GetLevel", _I, [_VP, _I, _VP]),
    ("ArapMatch_GetDescriptors", _I, [_VP, _I, _VP]),
    ("ArapMatch_LastRunMs", C.c_float, [_VP]),
    ("ArapMatch_LastCorrMs", C.c_float, [_VP]),
]
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libarapmatch.so is not built (python -m arap_flow_amd.build); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
        _lib = lib
    return _lib


class Matcher:
    """matcher for W x H frames; ngh_rad as the reference passes it to the binary (-ngh_rad 100)"""

    def __init__(self, W, H, ngh_rad=100):
        self.lib = load()
        self.W, self.H = int(W), int(H)
        self.h = self.lib.ArapMatch_Create(self.W, self.H, int(ngh_rad))
        if not self.h:
            raise RuntimeError("ArapMatch_Create failed (no GPU?)")
        self.cap = (self.W // 8 + 1) * (self.H // 8 + 1)

    def run(self, rgb1, rgb2):
        """uint8 [H][W][3] x 2 -> float32 [n][6]: x1 y1 x2 y2 score index"""
        a = np.ascontiguousarray(rgb1, np.uint8)
        b = np.ascontiguousarray(rgb2, np.uint8)
        if a.shape != (self.H, self.W, 3) or b.shape != (self.H, self.W, 3):
            raise ValueError("frames must be uint8 [%d][%d][3]" % (self.H, self.W))
        out = np.zeros((self.cap, 6), np.float32)
        n = self.lib.ArapMatch_Run(self.h, a.ctypes.data, b.ctypes.data, out.ctypes.data, self.cap)
        if n < 0:
            raise RuntimeError("ArapMatch_Run failed")
        return out[:min(n, self.cap)].copy()

    def last_ms(self):
        return float(self.lib.ArapMatch_LastRunMs(self.h))

    def last_corr_ms(self):
        return float(self.lib.ArapMatch_LastCorrMs(self.h))

    def levels(self):
        out = []
        for l in range(self.lib.ArapMatch_Levels(self.h)):
            nh, nw, S, c = _I(), _I(), _I(), _I()
            self.lib.ArapMatch_LevelInfo(self.h, l, C.byref(nh), C.byref(nw), C.byref(S), C.byref(c))
            out.append((nh.value, nw.value, S.value, c.value))
        return out

    def level_maps(self, level):
        nh, nw, S, c = self.levels()[level]
        a = np.zeros((nh, nw, S, S), np.float32)
        self.lib.ArapMatch_GetLevel(self.h, level, a.ctypes.data)
        return a

    def descriptors(self, which):
        a = np.zeros((self.H // 2, self.W // 2, 9), np.float32)
        self.lib.ArapMatch_GetDescriptors(self.h, which, a.ctypes.data)
        return a

    def close(self):
        if self.h:
            self.lib.ArapMatch_Free(self.h)
            self.h = None


def format_lines(m):
    """the binary's output lines `x1 y1 x2 y2 score index` (what /root/reference/para_gen.py:468-479 parses: the first
    four fields are read with int())"""
    return ["%d %d %d %d %g %d" % (r[0], r[1], r[2], r[3], r[4], r[5]) for r in m]
