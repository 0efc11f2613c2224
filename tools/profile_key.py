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


def source_hash():
    """sha256 over the code (comments and whitespace stripped) of the HIP sources of libarapopt.so, first 16 hex digits"""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "arap_flow_amd", "csrc", "*"))):
        if f.endswith((".h", ".hip")):
            h.update(os.path.basename(f).encode())
            h.update(_code_only(open(f, "r", errors="replace").read()).encode())
    return h.hexdigest()[:16]


def signature(workload, W, H, solves, K, fd, frames):
    """the workload of one bench step, as far as the kernels' counters depend on it"""
    return {"workload": workload, "size": [int(W), int(H)], "solves_per_step": int(solves), "segments_per_frame": int(K),
            "fd": int(fd), "frames_per_step": int(frames)}


def find_counters(sig):
    """the newest profiles/*_counters.json made for this signature AND for the kernel sources as they are now"""
    best = None
    sh = source_hash()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters.json"))):
        try:
            rec = json.load(open(f))
        except (OSError, ValueError):
            continue
        if rec.get("signature") == sig and rec.get("source_hash") == sh:
            if best is None or rec.get("unix_time", 0) >= best[1].get("unix_time", 0):
                best = (os.path.relpath(f, ROOT), rec)
    return best
