"""What ties a record under profiles/ to the run that may quote it: the hash of the kernel sources and the signature
of the workload.  Used by bench.py (reader) and tools/collect_profile.py (writer)."""
import glob
import hashlib
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _code_only(text):
    """C++ source without comments and without whitespace: what the compiler sees, so that editing a comment does not
    orphan the measured records"""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r"\s+", "", text)


# the sources a kernel's code comes from: a record stays valid while THOSE are unchanged
KERNEL_SOURCES = {
    "k_pcg_resident": ("arap_resident.h", "arap_device.h"),
    "k_pcg_a": ("arap_stream.h", "arap_kernels.h", "arap_tiled.h", "arap_device.h"),
    "k_pcg_b": ("arap_stream.h", "arap_kernels.h", "arap_device.h"),
}


def source_hash(kernel=None):
    """sha256 over the code (comments and whitespace stripped) of the HIP sources the dominant kernel of a record is
    made of (all sources of libarapopt.so when no kernel is named), first 16 hex digits"""
    h = hashlib.sha256()
    only = KERNEL_SOURCES.get(kernel)
    for f in sorted(glob.glob(os.path.join(ROOT, "arap_flow_amd", "csrc", "*"))):
        if f.endswith((".h", ".hip")) and (only is None or os.path.basename(f) in only):
            h.update(os.path.basename(f).encode())
            h.update(_code_only(open(f, "r", errors="replace").read()).encode())
    return h.hexdigest()[:16]


def signature(workload, W, H, solves, K, fd, frames):
    """the workload of one bench step, as far as the kernels' counters depend on it"""
    return {"workload": workload, "size": [int(W), int(H)], "solves_per_step": int(solves), "segments_per_frame": int(K),
            "fd": int(fd), "frames_per_step": int(frames)}


def find_counters(sig):
    """the newest profiles/*_counters.json made for this signature AND for the sources of its dominant kernel as they
    are now"""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters.json"))):
        try:
            rec = json.load(open(f))
        except (OSError, ValueError):
            continue
        if rec.get("signature") == sig and rec.get("source_hash") == source_hash(rec.get("dominant_kernel")):
            if best is None or rec.get("unix_time", 0) >= best[1].get("unix_time", 0):
                best = (os.path.relpath(f, ROOT), rec)
    return best


def ceilings():
    """the newest profiles/*_ceilings.json (tools/calibrate_ceilings.py): measured ceilings of the VALU and LDS pipes"""
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_ceilings.json")))
    if not fs:
        return None
    try:
        return os.path.relpath(fs[-1], ROOT), json.load(open(fs[-1]))
    except (OSError, ValueError):
        return None
