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
