"""Tier T1 (GPU): the HIP kernels vs the float64 CPU oracle on random inputs, through the C ABI
(ArapFlow_EvalJTF / ArapFlow_ApplyJTJ / ArapFlow_Cost).  Tolerance: float32 rounding only,
rel-L2 <= 1e-5."""
import ctypes as C

import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu

SIZES = [(9, 7), (64, 4), (65, 5), (130, 37), (200, 150)]


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _ptr(t):
    return C.c_void_p(t.data_ptr())


@pytest.mark.parametrize("W,H", SIZES)
@pytest.mark.parametrize("generic", [True, False])
def test_evaljtf_applyjtj_cost(gpu_state, oracle, W, H, generic):
    pb = helpers.random_problem(W, H, seed=W * 31 + H, generic_urshape=generic, ncons=max(4, W * H // 40))
    lib, st = gpu_state.lib, gpu_state.handle
    d = {k: _dev(pb[k]) for k in "OAUCM"}
    wf, wr = float(pb["wf"]), float(pb["wr"])
    gO = torch.zeros(H, W, 2, device="cuda"); gA = torch.zeros(H, W, device="cuda")
    dO = torch.zeros(H, W, 2, device="cuda"); dA = torch.zeros(H, W, device="cuda")
    rc = lib.ArapFlow_EvalJTF(st, W, H, _ptr(d["O"]), _ptr(d["A"]), _ptr(d["U"]), _ptr(d["C"]), _ptr(d["M"]),
                              wf, wr, _ptr(gO), _ptr(gA), _ptr(dO), _ptr(dA))
    assert rc == 0
    g64, d64 = oracle.evalJTF(pb["O"], pb["A"], pb["U"], pb["C"], pb["M"], wf, wr, dtype=np.float64)
    g = np.concatenate([gO.cpu().numpy(), gA.cpu().numpy()[..., None]], -1)
    dd = np.concatenate([dO.cpu().numpy(), dA.cpu().numpy()[..., None]], -1)
    assert helpers.rel_l2(g, g64) < 1e-5
    assert helpers.rel_l2(dd, d64) < 1e-5
    assert np.all(g[pb["M"] != 0] == 0)

    rng = np.random.default_rng(7)
    P = rng.normal(size=(H, W, 3)).astype(np.float32)
    P[pb["M"] != 0] = 0
    pO, pA = _dev(P[..., :2]), _dev(P[..., 2])
    oO = torch.zeros(H, W, 2, device="cuda"); oA = torch.zeros(H, W, device="cuda")
    rc = lib.ArapFlow_ApplyJTJ(st, W, H, _ptr(d["A"]), _ptr(d["U"]), _ptr(d["C"]), _ptr(d["M"]), wf, wr,
                               _ptr(pO), _ptr(pA), _ptr(oO), _ptr(oA))
    assert rc == 0
    j64 = oracle.applyJTJ(pb["A"], pb["U"], pb["C"], pb["M"], wf, wr, P, dtype=np.float64)
    j = np.concatenate([oO.cpu().numpy(), oA.cpu().numpy()[..., None]], -1)
    assert helpers.rel_l2(j, j64) < 1e-5

    cost = C.c_double()
    rc = lib.ArapFlow_Cost(st, W, H, _ptr(d["O"]), _ptr(d["A"]), _ptr(d["U"]), _ptr(d["C"]), _ptr(d["M"]), wf, wr,
                           C.byref(cost))
    assert rc == 0
    c64 = oracle.cost(pb["O"], pb["A"], pb["U"], pb["C"], pb["M"], wf, wr, dtype=np.float64)
    assert abs(cost.value - c64) <= 1e-5 * c64


def test_all_masked_and_single_vertex(gpu_state, oracle):
    """edge cases: every vertex excluded; 1x1 and 1xN grids"""
    lib, st = gpu_state.lib, gpu_state.handle
    for (W, H, frac) in [(17, 9, 1.1), (1, 1, 0.0), (1, 40, 0.2), (70, 1, 0.2)]:
        pb = helpers.random_problem(W, H, seed=5, mask_frac=frac, ncons=2)
        d = {k: _dev(pb[k]) for k in "OAUCM"}
        cost = C.c_double(-1)
        lib.ArapFlow_Cost(st, W, H, _ptr(d["O"]), _ptr(d["A"]), _ptr(d["U"]), _ptr(d["C"]), _ptr(d["M"]), 10.0, 0.1,
                          C.byref(cost))
        c64 = oracle.cost(pb["O"], pb["A"], pb["U"], pb["C"], pb["M"], 10.0, 0.1, dtype=np.float64)
        assert abs(cost.value - c64) <= 1e-5 * max(c64, 1e-30)
        if frac > 1:
            assert cost.value == 0.0
