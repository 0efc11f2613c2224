"""CPU-only checks of the host side: the C-ABI library loads and exports every declared symbol,
.flo I/O, constraint files, synthetic inputs.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_every_declared_symbol():
    from arap_flow_amd import build, capi
    path = build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    hdr = open(os.path.join(ROOT, "include", "arap_opt.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b((?:Opt|ArapFlow)_[A-Za-z]+)\s*\(", hdr))
    assert {"Opt_NewState", "Opt_ProblemDefine", "Opt_ProblemDelete", "Opt_ProblemPlan", "Opt_PlanFree",
            "Opt_SetSolverParameter", "Opt_ProblemSolve", "Opt_ProblemInit", "Opt_ProblemStep",
            "Opt_ProblemCurrentCost"} <= declared                     # the reference's Opt.h:35-71
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == {s[0] for s in capi.SYMBOLS}                   # binding covers the header exactly
    capi.load()
    lib.ArapFlow_Version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.ArapFlow_Version()


def test_flo_roundtrip_and_format(tmp_path):
    from arap_flow_amd import flo
    rng = np.random.default_rng(0)
    f = rng.normal(size=(5, 7, 2)).astype(np.float32)
    p = str(tmp_path / "a.flo")
    flo.flow_write(p, f)
    raw = open(p, "rb").read()
    assert raw[:4] == b"PIEH" and len(raw) == 12 + 8 * 35
    assert np.frombuffer(raw[:4], "<f4")[0] == np.float32(202021.25)
    assert tuple(np.frombuffer(raw[4:12], "<i4")) == (7, 5)
    assert np.array_equal(flo.flow_read(p), f)
    open(p, "ab").write(b"\0")
    with pytest.raises(ValueError):
        flo.flow_read(p)                                               # "file is too long"
    open(p, "wb").write(raw[:-4])
    with pytest.raises(ValueError):
        flo.flow_read(p)                                               # "file is too short"


def test_flo_reads_reference_golden(golden_dir):
    from arap_flow_amd import flo
    f = flo.flow_read(os.path.join(golden_dir, "cat512", "cat512_iFlo.flo"))
    assert f.shape == (512, 512, 2) and f.dtype == np.float32


def test_constraint_file_and_border_pins(golden_dir, oracle):
    from arap_flow_amd import opt
    c = opt.load_constraints(os.path.join(golden_dir, "cat512", "cat512_iCstr.txt"))
    assert c.shape == (9, 4) and tuple(c[0]) == (30, 132, 59, 44)
    b = opt.border_pins(7, 5)
    assert len(b) == 2 * (7 + 5) - 4
    assert np.array_equal(b, oracle.border_pins(7, 5))                 # same order as the reference loop


def test_synth_frames():
    from arap_flow_amd import synth
    fr = synth.make_frame(214, 120, seed=2, K=3, fd=2)
    lab = fr["labels"]
    assert fr["rgb"].shape == (120, 214, 3) and set(np.unique(fr["mask_red"])) <= {0, 255}
    assert lab[0].max() == 0 and lab[-1].max() == 0 and lab[:, 0].max() == 0 and lab[:, -1].max() == 0
    assert 0.15 < (lab != 0).mean() < 0.35
    c = fr["constraints"]
    assert len(c) > 10
    d = np.hypot(c[:, 2] - c[:, 0], c[:, 3] - c[:, 1])
    assert d.min() > 0 and d.max() < 60
    assert np.all(lab[c[:, 1], c[:, 0]] != 0)
    segs = synth.segment_masks(fr)
    assert len(segs) == 3 and sum(len(s["constraints"]) for s in segs) == len(c)
    full = synth.make_frame(64, 48, seed=1, full_mask=True)
    assert np.all(full["mask_red"] == 0)
    assert np.array_equal(synth.make_frame(64, 48, seed=5)["mask_red"], synth.make_frame(64, 48, seed=5)["mask_red"])


def _deal(tiles):
    """ArapFlow_ResidentDeal (pure host function of the library): launches and the [launches][512][4] table"""
    from arap_flow_amd import build
    lib = ctypes.CDLL(build.build())
    lib.ArapFlow_ResidentDeal.restype = ctypes.c_int
    t = np.ascontiguousarray(tiles, np.int32)
    n = lib.ArapFlow_ResidentDeal(t.ctypes.data_as(ctypes.c_void_p), len(t), None, 0)
    assert n >= 1
    tab = np.full((n, 512, 4), -7, np.int32)
    assert lib.ArapFlow_ResidentDeal(t.ctypes.data_as(ctypes.c_void_p), len(t), tab.ctypes.data_as(ctypes.c_void_p), n) == n
    return n, tab


def _check_deal(tiles, n, tab):
    """every solve appears in exactly one launch with one contiguous rank range 0..wgs-1, enough workgroups for its
    tiles (<= 9 each), granule blocks that do not overlap (a group of several runs owns whole blocks of 64 units);
    a group of <= 64 workgroups sits on one XCD slot (blockIdx & 7); one of 65..128 has ranks 0..63 on a slot of its own
    (its home, rank = local index) and the rest on ONE other slot; a wider one 4 or 8 whole aligned slots with ranks
    64 q + j on its q-th slot"""
    seen = {}
    for s in range(n):
        slot, rank, wgs, gran = tab[s, :, 0], tab[s, :, 1], tab[s, :, 2], tab[s, :, 3]
        used = slot >= 0
        gran_used = np.zeros(2 * 2 * 1024 + 1, bool)
        for b in np.unique(slot[used]):
            assert b not in seen, "solve dealt twice"
            seen[int(b)] = s
            idx = np.nonzero(slot == b)[0]
            w = int(wgs[idx[0]])
            assert (wgs[idx] == w).all() and len(idx) == w and sorted(rank[idx].tolist()) == list(range(w))
            assert w * 9 >= max(int(tiles[b]), 1)
            g = int(gran[idx[0]])
            units = w if w <= 64 else -(-w // 64) * 64
            assert (gran[idx] == g).all() and g % 4 == 0 and g + 4 * units <= 4 * 1024 and not gran_used[g:g + 4 * units].any()
            gran_used[g:g + 4 * units] = True
            xs = idx & 7
            if w <= 64:
                assert len(set(xs.tolist())) == 1
            elif w <= 128:
                hm = rank[idx] < 64
                hx = set(xs[hm].tolist())
                assert len(hx) == 1 and ((idx[hm] >> 3) == rank[idx][hm]).all()           # the home: a whole slot
                assert (slot[(np.arange(512) & 7) == hx.pop()] == b).all()                 # ... that nobody shares
                px = set(xs[~hm].tolist())
                assert len(px) == 1 and px != set(xs[hm].tolist())                         # the piece: one other slot
                loc = np.sort(idx[~hm] >> 3)
                assert (np.diff(loc) == 1).all()                                           # consecutive local indices
            else:
                k = w // 64
                assert w % 64 == 0 and k in (4, 8)
                a = int(xs.min())
                assert a % k == 0 and set(xs.tolist()) == set(range(a, a + k))
                assert ((rank[idx] >> 6) == (xs - a)).all() and ((rank[idx] & 63) == (idx >> 3)).all()
    assert sorted(seen) == list(range(len(tiles)))


def test_resident_deal_packs_narrow_and_wide_solves():
    """host logic of the resident path (no GPU): how solves are dealt to launches and workgroups"""
    rng = np.random.default_rng(0)
    # the benchmark shapes: 8 DAVIS frames -> one launch, one per XCD; 21 segments -> one launch; full frames -> two per launch
    n, tab = _deal([490] * 8)
    assert n == 1 and sorted(set((np.nonzero(tab[0, :, 0] == b)[0] & 7).tolist()).pop() for b in range(8)) == list(range(8))
    assert (tab[0, :, 2][tab[0, :, 0] >= 0] == 64).all()                  # widened to the whole XCD
    _check_deal([490] * 8, n, tab)
    n, tab = _deal([185] * 21)                                             # 21 workgroups each: three per XCD
    assert n == 1
    _check_deal([185] * 21, n, tab)
    n, tab = _deal([1680, 1680, 1680])
    assert n == 2
    _check_deal([1680, 1680, 1680], n, tab)
    n, tab = _deal([1680, 840, 500, 500, 500, 500, 500])                   # width 4 + home-and-piece + narrow ones
    assert n == 2
    _check_deal([1680, 840, 500, 500, 500, 500, 500], n, tab)
    # 1920x1080 --multseg segments (~716 tiles = 80 workgroups): a home XCD each plus a piece in a shared bin -- six
    # to a launch (whole pairs of bins held four)
    n, tab = _deal([716] * 6)
    assert n == 1
    _check_deal([716] * 6, n, tab)
    assert (np.unique(tab[0, :, 2][tab[0, :, 0] >= 0]) >= 80).all() and (tab[0, :, 0] >= 0).sum() >= 6 * 80
    n, tab = _deal([716] * 4)                                              # four of them: 128 workgroups each again
    assert n == 1 and (tab[0, :, 2][tab[0, :, 0] >= 0] == 128).all()
    _check_deal([716] * 4, n, tab)
    n, tab = _deal([716] * 7)
    assert n == 2
    _check_deal([716] * 7, n, tab)
    n, tab = _deal([716, 700, 650, 600, 590, 580, 300, 150, 20])           # mediums, their pieces and narrow solves mixed
    _check_deal([716, 700, 650, 600, 590, 580, 300, 150, 20], n, tab)
    n, tab = _deal([4608])                                                 # the largest solve: all 512 workgroups
    assert n == 1 and (tab[0, :, 2] == 512).all()
    _check_deal([4608], n, tab)
    n, tab = _deal([0, 1, 9, 10])                                          # empty and tiny solves still get a group
    _check_deal([0, 1, 9, 10], n, tab)
    for _ in range(200):                                                   # random mixes
        k = int(rng.integers(1, 40))
        tiles = rng.choice([0, 3, 40, 150, 200, 480, 577, 900, 1200, 2400, 4608], size=k).astype(np.int32)
        tiles = np.where(rng.random(k) < 0.5, rng.integers(0, 600, k), tiles).astype(np.int32)
        n, tab = _deal(tiles)
        _check_deal(tiles, n, tab)
        need = np.maximum((tiles + 8) // 9, 1)
        assert n >= int(np.ceil(need.sum() / 512))                         # cannot beat the capacity bound ...
        assert n <= len(tiles)                                             # ... and never worse than one solve per launch


def test_c_program_against_the_reference_header_links_with_the_library(tmp_path):
    """A C file that includes the REFERENCE's own ARAP/API/release/include/Opt.h and calls its ten functions must
    compile and link against libarapopt.so unchanged (the drop-in boundary).  Run here it gets NULL from
    Opt_NewState (no HIP device, no CPU fallback) and stops there.  Skipped where /root/reference is absent."""
    import subprocess
    from arap_flow_amd import build
    inc = "/root/reference/ARAP/API/release/include"
    if not os.path.exists(os.path.join(inc, "Opt.h")):
        pytest.skip("reference tree not present")
    lib = build.build()
    src = tmp_path / "dropin.c"
    src.write_text(r'''
#include <stdio.h>
#include "Opt.h"
int main(void)
{
    Opt_InitializationParameters ip = {0, 0, 0, 0};
    Opt_State* st = Opt_NewState(ip);
    if (!st) { printf("no state\n"); return 0; }
    unsigned dims[2] = {64, 64};
    int n = 2, l = 10;
    void* params[7] = {0};
    Opt_Problem* pr = Opt_ProblemDefine(st, "arap_plan.t", "gaussNewtonGPU");
    Opt_Plan* pl = Opt_ProblemPlan(st, pr, dims);
    Opt_SetSolverParameter(st, pl, "nIterations", &n);
    Opt_SetSolverParameter(st, pl, "lIterations", &l);
    Opt_ProblemInit(st, pl, params);
    while (Opt_ProblemStep(st, pl, params)) {}
    Opt_ProblemSolve(st, pl, params);
    printf("%f\n", Opt_ProblemCurrentCost(st, pl));
    Opt_PlanFree(st, pl);
    Opt_ProblemDelete(st, pr);
    return 0;
}
''')
    exe = tmp_path / "dropin"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + inc, str(src), "-o", str(exe),
                           "-L" + os.path.dirname(lib), "-larapopt", "-Wl,-rpath," + os.path.dirname(lib),
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "no state" in r.stdout, r.stdout + r.stderr


def test_resident_tile_lists_cover_every_active_vertex_once():
    """ArapFlow_ResidentTiles (pure host function): the aligned 32x8 tiling the resident kernel works on.  Every active
    vertex lies in exactly one listed tile, no listed tile is empty, tiles of a band start at the band's first active
    column and do not overlap, and the aligned form never needs more tiles than the fixed grid."""
    from arap_flow_amd import build, synth
    lib = ctypes.CDLL(build.build())
    lib.ArapFlow_ResidentTiles.restype = ctypes.c_int
    rng = np.random.default_rng(3)
    cases = [synth.make_frame(854, 480, seed=s)["mask_red"] for s in range(3)]
    cases += [synth.segment_masks(synth.make_frame(300, 200, seed=5, K=3))[1]["mask_red"], np.zeros((37, 70), np.uint8),
              np.full((20, 50), 255, np.uint8), np.where(rng.random((61, 131)) < 0.3, 0, 255).astype(np.uint8)]
    for m in cases:
        m = np.ascontiguousarray(m)
        H, W = m.shape
        counts = {}
        for aligned in (1, 0):
            cap = ((W + 31) // 32 + 1) * ((H + 7) // 8)
            org = np.zeros(cap, np.int32)
            bx = np.zeros((H + 7) // 8, np.int32)
            n = lib.ArapFlow_ResidentTiles(m.ctypes.data_as(ctypes.c_void_p), W, H, aligned, org.ctypes.data_as(ctypes.c_void_p),
                                           cap, bx.ctypes.data_as(ctypes.c_void_p))
            assert 0 <= n <= cap
            counts[aligned] = n
            cover = np.zeros((H, W), np.int32)
            prev = {}
            for o in org[:n]:
                y0, x0 = divmod(int(o), W)
                assert y0 % 8 == 0 and (x0 - bx[y0 // 8]) % 32 == 0 and x0 >= bx[y0 // 8]
                if not aligned:
                    assert x0 % 32 == 0
                assert prev.get(y0, -32) + 32 <= x0                                  # ascending, no overlap inside a band
                prev[y0] = x0
                blk = m[y0:y0 + 8, x0:x0 + 32]
                assert (blk == 0).any()                                              # no empty tile
                cover[y0:y0 + 8, x0:x0 + 32] += 1
            assert np.all(cover[m == 0] == 1)                                        # every active vertex exactly once
            if aligned:
                for b in range((H + 7) // 8):
                    cols = np.nonzero((m[8 * b:8 * b + 8] == 0).any(0))[0]
                    if len(cols):
                        assert bx[b] == cols[0]
        assert counts[1] <= counts[0]
