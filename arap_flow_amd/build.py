"""Build libarapopt.so (HIP, gfx950 only) in-tree: arap_flow_amd/lib/libarapopt.so.

hipcc cross-compiles without a GPU.  Flags that matter for results:
  -ffp-contract=off     every float operator is one IEEE operation (parity tier T3)
  -munsafe-fp-atomics   float64 atomic add is the hardware global_atomic_add_f64, not a CAS loop
  -fno-slp-vectorize    the SLP vectoriser pairs unrelated scalars of the resident kernel's register arrays and
                        spills ~300 VGPRs; packed-f32 math is written explicitly on float2 instead
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "arapopt.hip")
DEPS = [os.path.join(HERE, "csrc", f) for f in ("arapopt.hip", "arap_device.h", "arap_kernels.h", "arap_warp.h", "arap_resident.h", "arap_lm.h", "arap_tiled.h", "arap_stream.h")]
DEPS.append(os.path.join(HERE, "..", "include", "arap_opt.h"))
OUT_DIR = os.path.join(HERE, "lib")
OUT = os.path.join(OUT_DIR, "libarapopt.so")

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-munsafe-fp-atomics", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


def up_to_date():
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(d) <= t for d in DEPS)


def build(force=False, verbose=False):
    if not force and up_to_date():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + os.environ.get("ARAPOPT_EXTRA_FLAGS", "").split() + ["-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


# libarapmatch.so: the matching stage in front of the solve (include/arap_match.h), its own library and sources
MATCH_SRC = os.path.join(HERE, "csrc_dm", "arapmatch.hip")
MATCH_OUT = os.path.join(OUT_DIR, "libarapmatch.so")


def build_match(force=False, verbose=False):
    deps = [MATCH_SRC, os.path.join(HERE, "..", "include", "arap_match.h")]
    if not force and os.path.exists(MATCH_OUT) and all(os.path.getmtime(d) <= os.path.getmtime(MATCH_OUT) for d in deps):
        return MATCH_OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
           "-Wno-unused-function", "-o", MATCH_OUT, MATCH_SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return MATCH_OUT


HOST_DIR = os.path.join(HERE, "host")
BIN_DIR = os.path.join(HERE, "bin")
HOST_PROGRAMS = {"arap_deform": ["arap_deform.cpp", "png_io.cpp"], "warp_image": ["warp_image.cpp", "png_io.cpp"],
                 "png_tool": ["png_tool.cpp", "png_io.cpp"]}


def build_host(force=False, verbose=False):
    """C++ host programs (the reference's drivers are C++: ARAP/deformation/src/main.cpp, ARAP/warping/src/main.cpp)
    linked against libarapopt.so: arap_flow_amd/bin/arap_deform, arap_flow_amd/bin/warp_image."""
    build(force=False)
    os.makedirs(BIN_DIR, exist_ok=True)
    outs = []
    for name, srcs in HOST_PROGRAMS.items():
        out = os.path.join(BIN_DIR, name)
        deps = [os.path.join(HOST_DIR, f) for f in srcs + ["png_io.h", "flo_io.h"]] + [OUT]
        if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
            outs.append(out)
            continue
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"] + \
              [os.path.join(HOST_DIR, f) for f in srcs] + \
              ["-o", out, "-L" + OUT_DIR, "-larapopt", "-Wl,-rpath,$ORIGIN/../lib", "-L/opt/rocm/lib", "-lamdhip64",
               "-Wl,-rpath,/opt/rocm/lib", "-lz", "-pthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        outs.append(out)
    return outs


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    build_match(force="--force" in sys.argv, verbose=True)
    build_host(force="--force" in sys.argv, verbose=True)
