"""The PCG loop of the resident kernel as the compiler emitted it (CPU only: hipcc -S cross-compiles).

    python tools/loop_diff.py asm OUT.s            device assembly of arap_flow_amd/csrc/arapopt.hip as it is now
    python tools/loop_diff.py stats A.s [NS]       instruction mix of the main loop of k_pcg_resident<false, NS> (default 7)
    python tools/loop_diff.py diff A.s B.s [NS]    the two loops side by side with register NAMES normalised

Why: the loop runs at 253 of 256 VGPRs and ~100 SGPRs, and any change elsewhere in the kernel can change its register
allocation and instruction order.  Round 3: a build whose loop differed from its predecessor's by ONE s_waitcnt (the second
of two LDS reads of the block sum issued after the first had returned) was 2 % slower.  Diffing the normalised loops of a
build before and after a change shows such things without a GPU."""
import collections
import difflib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def emit_asm(out):
    from arap_flow_amd import build as b
    flags = [f for f in b.FLAGS if f not in ("-shared",)]
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["--cuda-device-only", "-S", "-o", out, b.SRC]
    subprocess.check_call(cmd)


def kernel_lines(path, ns):
    name = "_ZN4arap14k_pcg_residentILb0ELi%dEEEvNS_7PlanDevENS_6ResDevEi" % ns
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def main_loop(fn):
    """the outermost loop that comes last in the function (the PCG loop), header to the branch back to its latch"""
    hdr = [i for i, l in enumerate(fn) if re.search(r"=>This Loop Header: Depth=1", l)][-1]
    latch = [fn[i].split(":")[0] for i in range(max(0, hdr - 12), hdr + 1) if re.match(r"^\.LBB\d+_\d+:", fn[i])]
    end = -1
    for c in latch:
        for i, l in enumerate(fn):
            if i > hdr and re.search(r"s_c?branch\w*\s+" + re.escape(c) + r"\b", l):
                end = max(end, i)
    return fn[hdr:end + 1]


def normalise(l):
    l = re.sub(r";.*", "", l)
    l = re.sub(r"\.LBB\d+_\d+", "L", l)
    l = re.sub(r"v\[\d+:\d+\]", "V2", l)
    l = re.sub(r"\bv\d+\b", "V", l)
    l = re.sub(r"s\[\d+:\d+\]", "S2", l)
    l = re.sub(r"\bs\d+\b", "S", l)
    return l.rstrip()


def stats(body):
    ins = [l.strip().split()[0] for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter(ins)
    fam = lambda p: sum(v for k, v in c.items() if k.startswith(p))
    return {"instructions": len(ins), "valu": fam("v_"), "salu": fam("s_"), "lds": fam("ds_"), "global": fam("global_"),
            "scratch": fam("scratch_"), "s_load": fam("s_load"), "v_readlane (SGPR reloads)": c.get("v_readlane_b32", 0),
            "v_writelane": c.get("v_writelane_b32", 0), "s_nop": c.get("s_nop", 0), "s_waitcnt": c.get("s_waitcnt", 0)}


def main():
    a = sys.argv[1:]
    if len(a) >= 2 and a[0] == "asm":
        emit_asm(a[1])
    elif len(a) >= 2 and a[0] == "stats":
        print(stats(main_loop(kernel_lines(a[1], int(a[2]) if len(a) > 2 else 7))))
    elif len(a) >= 3 and a[0] == "diff":
        ns = int(a[3]) if len(a) > 3 else 7
        la, lb = main_loop(kernel_lines(a[1], ns)), main_loop(kernel_lines(a[2], ns))
        print(a[1], stats(la))
        print(a[2], stats(lb))
        na = [normalise(l) for l in la if normalise(l).strip()]
        nb = [normalise(l) for l in lb if normalise(l).strip()]
        n = 0
        for l in difflib.unified_diff(na, nb, a[1], a[2], n=2, lineterm=""):
            print(l)
            n += 1
        print("(%d diff lines)" % n)
    else:
        print(__doc__)
        return 2
    return 0


if __name__ == "__main__":
    sys.exit(main())
