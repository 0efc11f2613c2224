"""Tier T5 (GPU): fused rasteriser + flow emission vs the CPU oracle (bit exact), vs the outputs of the
reference's own warp_image (bit exact) and vs the reference's committed PNGs (<= 1 LSB on < 0.5 %)."""
import os

import numpy as np
import pytest
from PIL import Image

import helpers
from arap_flow_amd import flo, opt

pytestmark = pytest.mark.gpu


def test_warp_cat512_vs_reference_goldens(gpu_state, oracle, golden_dir):
    cat = helpers.load_cat512(golden_dir)
    wrgb, wmsk = opt.warp_image(gpu_state, cat["rgb"], cat["mask_red"], cat["golden_flow"])
    o_rgb, o_msk = oracle.warp(cat["rgb"], cat["mask_red"], cat["golden_flow"])
    assert np.array_equal(wmsk, o_msk) and np.array_equal(wrgb, o_rgb)        # bit exact vs oracle
    assert np.array_equal(wmsk, cat["golden_wmsk"])                           # mask bit exact vs reference
    diff = np.abs(wrgb.astype(int) - cat["golden_wrgb"].astype(int)).max(-1)
    assert diff.max() <= 1 and (diff > 0).mean() < 0.005


def test_warp_synth_vs_reference_binary_outputs(gpu_state, golden_dir):
    d = os.path.join(golden_dir, "warp_synth")
    rgb = np.array(Image.open(os.path.join(d, "iRGB.png")).convert("RGB"))
    mred = np.array(Image.open(os.path.join(d, "iMsk.png")).convert("RGB"))[..., 0]
    fl = flo.flow_read(os.path.join(d, "iFlo.flo"))
    wrgb, wmsk = opt.warp_image(gpu_state, rgb, mred, fl)
    assert np.array_equal(wrgb, np.array(Image.open(os.path.join(d, "wRGB.png")).convert("RGB")))
    assert np.array_equal(wmsk, np.array(Image.open(os.path.join(d, "wMsk.png")).convert("L")))


@pytest.mark.parametrize("W,H,amp", [(70, 50, 3.0), (129, 65, 8.0), (64, 4, 1.0), (2, 2, 0.5), (1, 5, 1.0), (854, 480, 2.0)])
def test_warp_random_folded_flows_vs_oracle(gpu_state, oracle, W, H, amp):
    rng = np.random.default_rng(W * H)
    rgb = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    mask = np.where(rng.random((H, W)) < 0.1, 255, 0).astype(np.uint8)
    fl = (rng.normal(size=(H, W, 2)) * amp).astype(np.float32)
    fl[mask != 0] = 0
    wrgb, wmsk = opt.warp_image(gpu_state, rgb, mask, fl)
    o_rgb, o_msk = oracle.warp(rgb, mask, fl)
    assert np.array_equal(wmsk, o_msk)
    assert np.array_equal(wrgb, o_rgb)


def test_frame_solver_warp_uses_offset_field(gpu_state, oracle):
    """arap_deform rasterises with the solved Offset itself (CombinedSolver.h:280-342), not (x,y)+flow."""
    from arap_flow_amd import synth
    W, H = 120, 80
    f = synth.make_frame(W, H, seed=4, K=1, fd=3)
    fs = opt.FrameSolver(gpu_state, W, H, batch=1)
    fs.set_frame(0, f["mask_red"], f["constraints"], rgb=f["rgb"])
    fs.solve(1, 2, 2, 30)
    fs.warp(1)
    r = fs.results(0)
    fs.close()
    o_rgb, o_msk = oracle.warp_offset(f["rgb"], f["mask_red"], r["offset"])
    assert np.array_equal(r["warped_mask"], o_msk) and np.array_equal(r["warped_rgb"], o_rgb)
    assert np.array_equal(r["flow"], oracle.flow_from_offset(r["offset"]))
