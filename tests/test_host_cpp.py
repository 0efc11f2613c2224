"""CPU-only: the C++ host programs build, and their PNG codec (arap_flow_amd/host/png_io.cpp, independent of the
reference's vendored LodePNG) decodes every PNG flavour the pipeline meets exactly as PIL does."""
import os
import os.path as osp
import subprocess

import numpy as np
import pytest
from PIL import Image

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))


@pytest.fixture(scope="module")
def bins():
    from arap_flow_amd import build
    outs = build.build_host()
    return {osp.basename(o): o for o in outs}


def test_host_programs_build_and_print_usage(bins):
    for name in ("arap_deform", "warp_image"):
        assert osp.exists(bins[name])
    # argument errors are handled before any GPU call (there is no GPU here): same message + exit code 1 as the
    # reference (main.cpp:193-197, warping main.cpp:312-316)
    for name in ("arap_deform", "warp_image"):
        r = subprocess.run([bins[name], "just", "two"], capture_output=True, text=True)
        assert r.returncode == 1 and "Invalid Input!" in r.stdout and "Usage" in r.stdout


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "LA", "P", "1", "I;16", "L2", "L4"])
def test_png_codec_matches_pil(bins, tmp_path, mode, golden_dir):
    rng = np.random.default_rng(3)
    H, W = 37, 53                                         # odd sizes: sub-byte rows need padding
    src = str(tmp_path / "in.png")
    if mode == "RGB":
        Image.fromarray(rng.integers(0, 256, (H, W, 3)).astype(np.uint8)).save(src)
    elif mode == "RGBA":
        Image.fromarray(rng.integers(0, 256, (H, W, 4)).astype(np.uint8)).save(src)
    elif mode == "L":
        Image.fromarray(rng.integers(0, 256, (H, W)).astype(np.uint8)).save(src)
    elif mode == "LA":
        Image.fromarray(rng.integers(0, 256, (H, W, 2)).astype(np.uint8), "LA").save(src)
    elif mode == "P":
        im = Image.fromarray(rng.integers(0, 7, (H, W)).astype(np.uint8), "P")
        im.putpalette(list(rng.integers(0, 256, 21)))
        im.save(src)
    elif mode == "1":
        Image.fromarray(rng.random((H, W)) < 0.5).save(src)
    elif mode == "I;16":
        Image.fromarray((rng.integers(0, 65536, (H, W))).astype(np.uint16)).save(src)
    else:                                                  # 2- and 4-bit greyscale via PIL's bits option
        bits = int(mode[1])
        Image.fromarray(rng.integers(0, 1 << bits, (H, W)).astype(np.uint8), "P").save(src, bits=bits)
    r = subprocess.run([bins["png_tool"], src, str(tmp_path / "o.png"), str(tmp_path / "m.png")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert r.stdout.split() == [str(W), str(H)]
    got = np.array(Image.open(tmp_path / "o.png"))
    if mode == "I;16":
        ref = np.stack([(np.array(Image.open(src)) >> 8).astype(np.uint8)] * 3, -1)       # high byte
    else:
        ref = np.array(Image.open(src).convert("RGB"))
    assert got.shape == (H, W, 3) and np.array_equal(got, ref)
    m = Image.open(tmp_path / "m.png")
    assert m.mode == "1" and np.array_equal(np.array(m), ref[..., 0] != 0)


def test_png_codec_reads_reference_fixtures(bins, tmp_path, golden_dir):
    for f in ("cat512_iRGB.png", "cat512_iMsk.png", "cat512_wMsk.png"):
        src = osp.join(golden_dir, "cat512", f)
        r = subprocess.run([bins["png_tool"], src, str(tmp_path / "o.png"), str(tmp_path / "m.png")], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout
        assert np.array_equal(np.array(Image.open(tmp_path / "o.png")), np.array(Image.open(src).convert("RGB")))
