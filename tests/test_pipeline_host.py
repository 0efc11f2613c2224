"""CPU-only tests of the Python host pipeline (no GPU, no oracle): the para_gen helpers keep the
reference's rules (para_gen.py line numbers in arap_flow_amd/pipeline.py)."""
import os
import os.path as osp
import sys

import numpy as np
import pytest
from PIL import Image

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))


def test_valid_cnstr_and_filter():
    from arap_flow_amd import pipeline
    m1 = np.zeros((50, 80), np.uint8); m1[10:40, 10:60] = 1; m1[20:30, 62:70] = 2
    m2 = m1.copy()
    assert pipeline.valid_cnstr(15, 15, 20, 18, m1, m2)
    assert not pipeline.valid_cnstr(15, 15, 15, 15, m1, m2)             # zero displacement
    assert not pipeline.valid_cnstr(11, 11, 59, 39, m1, m1 * 0)         # other label at the target
    assert not pipeline.valid_cnstr(2, 2, 5, 5, m1, m2)                 # background
    assert not pipeline.valid_cnstr(15, 15, 80, 20, m1, m2)             # x2 out of range
    assert not pipeline.valid_cnstr(12, 12, 12 + 60, 12, m1, np.ones_like(m1))     # |d| = 60 is not < 60
    lines = ["15 15 20 18 0.9 1", "2 2 5 5 0.1 3", "65 25 66 26 1 1", "junk"]
    c, v = pipeline.filter_matches(lines, m1, m2)
    assert c == [(15, 15, 20, 18), (65, 25, 66, 26)] and v == [1, 2]


def test_constraint_file_roundtrip(tmp_path):
    from arap_flow_amd import opt, pipeline
    p = str(tmp_path / "c.txt")
    pipeline.write_constraints(p, [(1, 2, 3, 4), (5, 6, 7, 8)])
    assert open(p).read() == "2\n1\t2\t3\t4\n5\t6\t7\t8"
    assert opt.load_constraints(p).tolist() == [[1, 2, 3, 4], [5, 6, 7, 8]]


def test_scale_rotate(tmp_path):
    from arap_flow_amd import pipeline
    rng = np.random.default_rng(0)
    im = rng.integers(0, 255, (120, 90, 3)).astype(np.uint8)            # portrait
    mk = np.zeros((120, 90), np.uint8); mk[30:80, 20:60] = 3
    Image.fromarray(im).save(tmp_path / "a.jpg"); Image.fromarray(mk).save(tmp_path / "a.png")
    pre, i2, m2 = pipeline.scale_rotate(str(tmp_path / "a.jpg"), str(tmp_path / "a.png"), size=(64, 48))
    assert pre and i2.size == (64, 48) and m2.size == (64, 48)
    assert set(np.unique(np.array(m2))) <= {0, 3}                        # NEAREST keeps labels
    Image.fromarray(im[:48, :64]).save(tmp_path / "b.png"); Image.fromarray(mk[:48, :64]).save(tmp_path / "bm.png")
    pre, i3, _ = pipeline.scale_rotate(str(tmp_path / "b.png"), str(tmp_path / "bm.png"), size=(64, 48))
    assert not pre and i3.size == (64, 48)


def test_read_list_and_make_path(tmp_path):
    from arap_flow_amd import pipeline
    p = dict(rgb1_gen="a/r.png", msk1_gen="a/m.png", cstr_tmp="a/c.txt", flow_gen="a/f.flo", rgb2_gen="a/w.png",
             msk2_gen="a/wm.png")
    line = pipeline.make_arap_path(p)
    assert len(line.split(" ")) == 6 and all(osp.isabs(q) for q in line.split(" "))
    lf = tmp_path / "l.txt"
    lf.write_text(line + "\n\n" + line + "\n")
    assert len(pipeline.read_list(str(lf))) == 2
    lf.write_text("only three paths here\n")
    with pytest.raises(ValueError):
        pipeline.read_list(str(lf))
    q = pipeline.replace_ext(p, 2, keep_orgs=["rgb1_gen", "cstr_tmp"])
    assert q["rgb1_gen"] == "a/r.png" and q["msk1_gen"] == "a/m_seg2.png" and q["flow_gen"] == "a/f_seg2.flo"


def test_split_segments_and_flatten(tmp_path):
    from arap_flow_amd import flo, pipeline
    H, W = 20, 30
    mk1 = np.zeros((H, W), np.uint8); mk1[2:10, 2:12] = 1; mk1[8:18, 15:28] = 2; mk1[0:2, 0:2] = 5
    segs = pipeline.split_segments(mk1, [1, 2, 2, 0])                    # label 5 has no constraint: not solved
    assert [s for s, _ in segs] == [1, 2]
    assert np.array_equal(segs[0][1] == 0, mk1 == 1) and set(np.unique(segs[1][1])) == {0, 255}
    base = dict(rgb1_gen=str(tmp_path / "r.png"), msk1_gen=str(tmp_path / "m.png"), cstr_tmp=str(tmp_path / "c.txt"),
                flow_gen=str(tmp_path / "f.flo"), rgb2_gen=str(tmp_path / "w.png"), msk2_gen=str(tmp_path / "wm.png"))
    seg_lines = []
    for s, _ in segs:
        p_ = pipeline.replace_ext(base, s, keep_orgs=["rgb1_gen", "cstr_tmp"])
        wm = np.zeros((H, W), bool)
        wm[(3, 9)[s - 1]:(11, 19)[s - 1], (3, 14)[s - 1]:(13, 27)[s - 1]] = True      # warped masks overlap a little
        fl = np.zeros((H, W, 2), np.float32); fl[wm] = s
        rgb = np.zeros((H, W, 3), np.uint8); rgb[wm] = 50 * s
        flo.flow_write(p_["flow_gen"], fl); Image.fromarray(rgb).save(p_["rgb2_gen"]); pipeline.save_mask(wm, p_["msk2_gen"])
        seg_lines.append(pipeline.make_arap_path(p_))
    out = pipeline.flatten([(pipeline.make_arap_path(base), seg_lines)])
    assert out == [pipeline.make_arap_path(base)]
    fl = flo.flow_read(base["flow_gen"]); rgb = np.array(Image.open(base["rgb2_gen"])); wm = np.array(Image.open(base["msk2_gen"]))
    assert fl[10, 15, 0] == 2 and rgb[10, 15, 0] == 100                  # overlap: the later segment wins
    assert fl[5, 5, 0] == 1 and rgb[5, 5, 0] == 50 and fl[0, 0, 0] == 0
    assert set(np.unique(wm)) == {0, 1}                                  # 0/1 valued, as the reference writes it
    assert not osp.exists(seg_lines[0].split(" ")[3])                    # per-segment files are removed


def test_add_bg_and_fit_bg():
    import random
    from arap_flow_amd import pipeline
    im = np.full((40, 60, 3), 7, np.uint8); mk = np.zeros((40, 60), np.uint8); mk[10:20, 10:20] = 1
    bg = np.full((30, 30, 3), 200, np.uint8)
    fitted = pipeline.fit_bg(bg, im, rng=random.Random(1))
    assert fitted.shape == im.shape
    out = pipeline.add_bg(im, mk, fitted)
    assert np.all(out[mk == 1] == 7) and np.all(out[mk == 0] == 200)


def test_para_gen_scan_pairs_and_resume(tmp_path):
    sys.path.insert(0, ROOT)
    import para_gen
    inp, outp = tmp_path / "in", tmp_path / "out"
    for seq in ("bear", "camel"):
        os.makedirs(inp / "orgRGB" / seq); os.makedirs(inp / "orgMasks" / seq)
        for n in range(4):
            Image.new("RGB", (8, 8)).save(inp / "orgRGB" / seq / ("%05d.jpg" % n))
            if not (seq == "camel" and n == 3):
                Image.new("L", (8, 8)).save(inp / "orgMasks" / seq / ("%05d.png" % n))
    flags = para_gen.parse(["--input", str(inp), "--output", str(outp), "--fd", "2", "--matches", str(tmp_path)])
    e = para_gen.scan(flags, str(inp), str(outp))
    assert sorted((x["_seq"], x["_stem"]) for x in e) == [("bear", "00000"), ("bear", "00001"), ("camel", "00000")]
    assert e[0]["flow_gen"].endswith(osp.join("Flow", e[0]["_seq"], e[0]["_stem"] + ".flo"))
    os.makedirs(osp.dirname(e[0]["flow_gen"])); open(e[0]["flow_gen"], "w").close()
    flags.resume = True
    assert len(para_gen.scan(flags, str(inp), str(outp))) == 2          # --resume skips finished pairs


def _tiny_tree(tmp_path, nframes=7, W=64, H=40, seqs=("a", "b")):
    """DAVIS-shaped input tree with precomputed matches: nframes frames per sequence -> (nframes-1) pairs each"""
    from arap_flow_amd import synth
    inp, mdir = tmp_path / "in", tmp_path / "matches"
    for si, seq in enumerate(seqs):
        os.makedirs(inp / "orgRGB" / seq); os.makedirs(inp / "orgMasks" / seq); os.makedirs(mdir / seq)
        fr = synth.make_frame(W, H, seed=20 + si, K=2)
        for n in range(nframes):
            Image.fromarray(fr["rgb"]).save(inp / "orgRGB" / seq / ("%05d.png" % n))
            Image.fromarray(fr["labels"].astype(np.uint8)).save(inp / "orgMasks" / seq / ("%05d.png" % n))
            (mdir / seq / ("%05d.txt" % n)).write_text("\n".join("%d %d %d %d 1.0 0" % tuple(c) for c in fr["constraints"]))
    return inp, mdir


FAKE_WORKER = r'''
import sys, shutil
# stand-in for an ARAP executable (no GPU): "solves" a line by writing the flow / warped files para_gen expects
def solve(line):
    rgb, msk, cst, flo, wrgb, wmsk = line.split()
    import numpy as np
    from PIL import Image
    sys.path.insert(0, %r)
    from arap_flow_amd import flo as F
    m = np.array(Image.open(msk).convert("RGB"))[..., 0]
    F.flow_write(flo, np.zeros(m.shape + (2,), np.float32))
    shutil.copy(rgb, wrgb)
    Image.fromarray(m == 0).save(wmsk)
if sys.argv[1] == "--serve":
    print("Ready", flush=True)
    batch = []
    for line in sys.stdin:
        if FAIL_AFTER >= 0 and len(batch) >= FAIL_AFTER: sys.exit(3)
        batch.append(line)
        solve(line); print("Batch 1", flush=True); print("Done " + line.split()[3], flush=True)
else:
    lines = [l for l in open(sys.argv[1]).read().splitlines() if l.strip()]
    if FAIL_AFTER >= 0: sys.exit(3)
    for l in lines: solve(l)
'''


@pytest.mark.parametrize("worker,multseg", [("serve", False), ("serve", True), ("batch", True)])
def test_para_gen_with_a_stand_in_worker(tmp_path, worker, multseg):
    """para_gen's own machinery without a GPU: pool front end, one worker per GPU id fed over a pipe (serve) or one
    child per hand-out of <= --narap lines (batch), completion tracking per segment, flatten, all_files.list, stats"""
    sys.path.insert(0, ROOT)
    import json
    import para_gen
    inp, mdir = _tiny_tree(tmp_path)
    fake = tmp_path / "fake_arap.py"
    fake.write_text("FAIL_AFTER = -1\n" + FAKE_WORKER % ROOT)
    outp = tmp_path / "out"
    argv = ["--input", str(inp), "--output", str(outp), "--gpu", "0", "1", "--matches", str(mdir), "--worker", worker,
            "--arap_bin", "%s %s" % (sys.executable, fake), "--narap", "5", "--jobs", "2"] + (["--multseg"] if multseg else [])
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        out = para_gen.main(para_gen.parse(argv))
    finally:
        os.chdir(cwd)
    assert len(out) == 12                                                   # 2 sequences x 6 pairs
    st = json.load(open(outp / "arap_stats.json"))
    assert st["frames"] == 12 and st["solves"] == (24 if multseg else 12) and st["worker"] == worker
    assert sum(st["batches"]) == st["solves"] and max(st["batches"]) <= 5   # --narap bounds a hand-out
    for ln in out:
        assert all(osp.exists(q) for q in ln.split(" "))
    if multseg:                                                             # per-segment files were merged and removed
        assert not [f for f in os.listdir(outp / "Flow" / "a") if "_seg" in f]


@pytest.mark.parametrize("worker", ["serve", "batch"])
def test_para_gen_fails_promptly_when_the_arap_worker_dies(tmp_path, worker):
    """a worker that exits non-zero must fail the run at once -- the reference hangs here (the GPU id never returns
    to its queue, para_gen.py:193-214,560-567)"""
    sys.path.insert(0, ROOT)
    import time
    import para_gen
    inp, mdir = _tiny_tree(tmp_path, nframes=4, seqs=("a",))
    fake = tmp_path / "fake_arap.py"
    fake.write_text("FAIL_AFTER = 1\n" + FAKE_WORKER % ROOT)
    argv = ["--input", str(inp), "--output", str(tmp_path / "out"), "--gpu", "0", "--matches", str(mdir), "--worker", worker,
            "--arap_bin", "%s %s" % (sys.executable, fake), "--jobs", "2"]
    cwd = os.getcwd()
    os.chdir(tmp_path)
    t0 = time.time()
    try:
        with pytest.raises(AssertionError, match="exited with code 3"):
            para_gen.main(para_gen.parse(argv))
    finally:
        os.chdir(cwd)
    assert time.time() - t0 < 60


# ---------------------------------------------------------------------------------------------------------------------
# hand-computed fixtures for the helpers around the hot path (tests/golden/host/cases.json)
# ---------------------------------------------------------------------------------------------------------------------
from arap_flow_amd import pipeline  # noqa: E402


def _host_cases():
    import json
    return json.load(open(osp.join(ROOT, "tests", "golden", "host", "cases.json")))


def test_resize_crop_geometry_hand_cases():
    for c in _host_cases()["resize_crop_geometry"]:
        new, box = pipeline.resize_crop_geometry(tuple(c["in"]), tuple(c["out"]))
        assert list(new) == c["new"] and list(box) == c["box"], c
        assert box[2] - box[0] == c["out"][0] and box[3] - box[1] == c["out"][1]


def test_cover_scale_hand_cases():
    for c in _host_cases()["cover_scale_hw"]:
        assert list(pipeline.cover_scale(tuple(c["bg"]), tuple(c["im"]), c["u"])) == c["new"], c


def test_match_filter_hand_cases():
    mc = _host_cases()["match_ok"]
    W, H = mc["labels_wh"]
    lab = np.zeros((H, W), np.uint8)
    for bx in mc["label_boxes"]:
        lab[bx["y"][0]:bx["y"][1], bx["x"][0]:bx["x"][1]] = bx["label"]
    for c in mc["cases"]:
        assert pipeline.valid_cnstr(*c["m"], lab, lab) is c["ok"], c
    ones = np.ones((200, 200), np.uint8)
    for c in mc["distance_cases_on_all_ones_200x200"]:
        assert pipeline.valid_cnstr(*c["m"], ones, ones) is c["ok"], c
    # the vectorised form gives the same answers, in order, and filter_matches keeps the matcher's order
    ms = np.asarray([c["m"] for c in mc["cases"]])
    assert pipeline.match_ok(ms[:, :2], ms[:, 2:], lab, lab).tolist() == [c["ok"] for c in mc["cases"]]
    lines = ["%d %d %d %d 0.9 %d" % (*c["m"], i) for i, c in enumerate(mc["cases"]) if min(c["m"]) >= 0]
    rows, labels = pipeline.filter_matches(lines, lab, lab)
    assert rows == [tuple(c["m"]) for c in mc["cases"] if c["ok"]] and labels == [1, 2]


def test_merge_segments_hand_case(tmp_path):
    mc = _host_cases()["merge_segments"]
    masks = np.asarray(mc["masks"], np.uint8)
    S, H, W = masks.shape
    flows = np.stack([np.broadcast_to(np.asarray(f, np.float32), (H, W, 2)) for f in mc["flow_of_segment"]])
    rgbs = np.stack([np.full((H, W, 3), v, np.uint8) for v in mc["rgb_of_segment"]])
    flow, rgb, mask = pipeline.merge_segments(flows, rgbs, masks)
    win = np.asarray(mc["winner"])
    assert np.array_equal(flow[..., 0], win + 1.0) and np.array_equal(flow[..., 1], -(win + 1.0))
    assert np.array_equal(rgb[..., 0], 10 * (win + 1)) and np.array_equal(mask, np.asarray(mc["mask_out"]))
    # 1-bit mask files (what the C++ driver and LodePNG write) read back as bool: same selection
    f2, r2, m2 = pipeline.merge_segments(flows, rgbs, masks.astype(bool))
    assert np.array_equal(f2, flow) and np.array_equal(r2, rgb) and np.array_equal(m2.astype(np.uint8), mask)
    # a single segment is its own result
    f1, r1, m1 = pipeline.merge_segments(flows[:1], rgbs[:1], masks[:1])
    assert np.array_equal(f1, flows[0]) and np.array_equal(m1, masks[0])


def test_fit_bg_draw_order_and_window():
    """three draws, in this order: uniform(1, 2), then the window's top row, then its left column (both ends included)"""
    calls = []

    class Rng:
        def uniform(self, a, b):
            calls.append(("uniform", a, b)); return 1.5
        def randint(self, a, b):
            calls.append(("randint", a, b)); return b            # the far end is a legal position

    bg = np.zeros((10, 20, 3), np.uint8)
    bg[:, :, 0] = 200                                          # constant colour: resizing keeps it
    im = np.zeros((30, 30, 3), np.uint8)
    out = pipeline.fit_bg(bg, im, rng=Rng())
    assert calls == [("uniform", 1, 2), ("randint", 0, 45 - 30), ("randint", 0, 90 - 30)]
    assert out.shape == (30, 30, 3) and (out[..., 0] == 200).all() and (out[..., 1:] == 0).all()


def test_add_bg_selects_by_mask_value():
    im = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)
    bg = im + 100
    mk = np.asarray([[0, 255, 0], [7, 0, 255]], np.uint8)
    out = pipeline.add_bg(im, mk, bg)                          # bgval = 0: mask 0 -> background
    assert out.dtype == np.uint8 and np.array_equal(out[mk == 0], bg[mk == 0]) and np.array_equal(out[mk != 0], im[mk != 0])
    out = pipeline.add_bg(im, mk, bg, bgval=255)
    assert np.array_equal(out[mk == 255], bg[mk == 255]) and np.array_equal(out[mk != 255], im[mk != 255])
    with pytest.raises(AssertionError):
        pipeline.add_bg(im, mk[:, :2], bg)
    with pytest.raises(AssertionError):
        pipeline.add_bg(im, mk, bg[:, :2])


def test_scale_rotate_portrait_transpose_and_nearest_mask(tmp_path):
    """portrait -> landscape is a transpose (pixel (x, y) comes from (y, x)); the label mask is resized with nearest
    neighbour (no new label values) and cut by the same box as the image"""
    rgb = np.random.default_rng(3).integers(0, 256, (50, 20, 3), dtype=np.uint8)       # h = 50 > w = 20
    lab = np.zeros((50, 20), np.uint8); lab[10:30, 5:15] = 3; lab[35:45, 2:8] = 9
    Image.fromarray(rgb).save(tmp_path / "p.png"); Image.fromarray(lab).save(tmp_path / "pm.png")
    changed, im, mk = pipeline.scale_rotate(str(tmp_path / "p.png"), str(tmp_path / "pm.png"))
    assert changed and im.size == (50, 20)
    assert np.array_equal(np.asarray(im), rgb.transpose(1, 0, 2)) and np.array_equal(np.asarray(mk), lab.T)
    changed, im, mk = pipeline.scale_rotate(str(tmp_path / "p.png"), str(tmp_path / "pm.png"), size=(40, 30))
    new, box = pipeline.resize_crop_geometry((50, 20), (40, 30))                        # (100, 40), (30, 5, 70, 35)
    assert new == (100, 40) and box == (30, 5, 70, 35) and im.size == (40, 30) and mk.size == (40, 30)
    assert set(np.unique(np.asarray(mk))) <= {0, 3, 9}
    # nearest neighbour at scale 2: output pixel (x, y) of the resized mask is source pixel (x // 2, y // 2)
    big = np.repeat(np.repeat(lab.T, 2, axis=0), 2, axis=1)
    assert np.array_equal(np.asarray(mk), big[5:35, 30:70])
    # a PNG pair that already has the target size is left alone
    Image.fromarray(rgb.transpose(1, 0, 2)).save(tmp_path / "l.png"); Image.fromarray(lab.T).save(tmp_path / "lm.png")
    changed, im, mk = pipeline.scale_rotate(str(tmp_path / "l.png"), str(tmp_path / "lm.png"), size=(50, 20))
    assert not changed and np.array_equal(np.asarray(im), rgb.transpose(1, 0, 2))
