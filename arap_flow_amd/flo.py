"""Middlebury .flo files: "PIEH", int32 W, int32 H, H rows of W interleaved (u, v) float32, little
endian.  Reference: writer ARAP/deformation/src/main.cpp:53-75, reader ARAP/warping/src/main.cpp:228-274,
Python twin sintel_io.py:26-73 (flow_read / flow_write)."""
import numpy as np

TAG_FLOAT = 202021.25      # main.h:7  (the bytes "PIEH" read as a little-endian float)
TAG_STRING = b"PIEH"       # main.h:8


def flow_write(filename, flow):
    """flow: float array [H, W, 2]."""
    flow = np.asarray(flow)
    assert flow.ndim == 3 and flow.shape[2] == 2
    h, w = flow.shape[:2]
    with open(filename, "wb") as f:
        f.write(TAG_STRING)
        np.array([w, h], dtype="<i4").tofile(f)
        np.ascontiguousarray(flow, dtype="<f4").tofile(f)


def flow_read(filename):
    """Returns float32 [H, W, 2].  Same checks as the reference reader (tag, 1 <= W,H <= 99999,
    exact length)."""
    with open(filename, "rb") as f:
        raw = f.read()
    if len(raw) < 12:
        raise ValueError("flow_read(%s): file is too short" % filename)
    tag = np.frombuffer(raw[:4], "<f4")[0]
    if tag != np.float32(TAG_FLOAT):
        raise ValueError("flow_read(%s): wrong tag (possibly due to big-endian machine?)" % filename)
    w, h = (int(v) for v in np.frombuffer(raw[4:12], "<i4"))
    if w < 1 or w > 99999:
        raise ValueError("flow_read(%s): illegal width %d" % (filename, w))
    if h < 1 or h > 99999:
        raise ValueError("flow_read(%s): illegal height %d" % (filename, h))
    need = 12 + 8 * w * h
    if len(raw) < need:
        raise ValueError("flow_read(%s): file is too short" % filename)
    if len(raw) > need:
        raise ValueError("flow_read(%s): file is too long" % filename)
    return np.frombuffer(raw[12:], "<f4").reshape(h, w, 2).copy()
