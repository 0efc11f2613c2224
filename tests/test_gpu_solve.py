"""Tiers T2-T4 (GPU): the Opt_* drop-in API and the batched frame solver vs the float32 CPU oracle and
the reference's golden vector.  Everything goes through the C ABI (arap_flow_amd.opt is ctypes only)."""
import os

import numpy as np
import pytest
import torch

import helpers
from arap_flow_amd import opt

pytestmark = pytest.mark.gpu

T2_TOL = 1e-4        # BASELINE.json north_star: 1e-4 relative L2 (meaningful for short schedules)


def _solve_opt(state, pb, nIter, lIter, plan=opt.BUILTIN_PLAN):
    H, W = pb["A"].shape
    dev = {k: torch.from_numpy(pb[k].copy()).cuda() for k in "OAUCM"}
    s = opt.OptSolver(state, (W, H), plan)
    pp = opt.NamedParameters()
    pp.set("Offset", dev["O"]); pp.set("Angle", dev["A"]); pp.set("UrShape", dev["U"])
    pp.set("Constraints", dev["C"]); pp.set("Mask", dev["M"])
    pp.set("w_fitSqrt", float(pb["wf"])); pp.set("w_regSqrt", float(pb["wr"]))
    sp = opt.NamedParameters()
    sp.set("nIterations", nIter); sp.set("lIterations", lIter)
    cost = s.solve(sp, pp)
    out = dev["O"].cpu().numpy(), dev["A"].cpu().numpy(), cost
    s.close()
    return out


@pytest.mark.parametrize("W,H,nIter,lIter", [(9, 7, 1, 5), (64, 64, 2, 50), (130, 37, 4, 50), (200, 150, 1, 200),
                                             (97, 61, 3, 0)])
def test_T2_opt_solve_vs_oracle_f32(gpu_state, oracle, W, H, nIter, lIter):
    pb = helpers.random_problem(W, H, seed=W + H, generic_urshape=(W % 2 == 0), ncons=max(4, W * H // 60))
    O, A, cost = _solve_opt(gpu_state, pb, nIter, lIter)
    Or, Ar, costs = oracle.solve(pb["O"], pb["A"], pb["U"], pb["C"], pb["M"], pb["wf"], pb["wr"], nIter, lIter,
                                 dtype=np.float32, mode=1, trig=1)
    dO, dA = O - pb["O"], A - pb["A"]                       # compare the update, not the (large) positions
    dOr, dAr = Or - pb["O"], Ar - pb["A"]
    if lIter > 0:
        assert helpers.rel_l2(dO, dOr) < T2_TOL
        assert helpers.rel_l2(dA, dAr) < T2_TOL
    else:
        assert np.array_equal(O, pb["O"]) and np.array_equal(A, pb["A"])     # zero PCG iterations: delta = 0
    assert abs(cost - costs[-1]) <= 1e-4 * abs(costs[-1])
    ex = pb["M"] != 0
    assert np.array_equal(O[ex], pb["O"][ex]) and np.array_equal(A[ex], pb["A"][ex])


def test_opt_init_step_protocol(gpu_state, oracle):
    """Opt_ProblemInit / Opt_ProblemStep / Opt_ProblemCurrentCost (Opt.h:60-71): Step returns 1 for each
    of nIterations steps, then 0; cost after Init is the cost of the inputs; costs match the oracle's."""
    W, H = 48, 40
    pb = helpers.random_problem(W, H, seed=11, generic_urshape=False)
    dev = {k: torch.from_numpy(pb[k].copy()).cuda() for k in "OAUCM"}
    s = opt.OptSolver(gpu_state, (W, H))
    pp = opt.NamedParameters()
    for n, k in [("Offset", "O"), ("Angle", "A"), ("UrShape", "U"), ("Constraints", "C"), ("Mask", "M")]:
        pp.set(n, dev[k])
    pp.set("w_fitSqrt", 10.0); pp.set("w_regSqrt", 0.1)
    sp = opt.NamedParameters()
    sp.set("nIterations", 3); sp.set("lIterations", 20)
    s.set_solver_parameters(sp)
    _, _, costs = oracle.solve(pb["O"], pb["A"], pb["U"], pb["C"], pb["M"], 10.0, 0.1, 3, 20, dtype=np.float32,
                               mode=1, trig=1)
    s.init(pp)
    got = [s.current_cost()]
    rets = []
    for _ in range(5):
        rets.append(s.step(pp))
        got.append(s.current_cost())
    assert rets == [1, 1, 1, 0, 0]
    assert np.allclose(got[:4], costs, rtol=1e-4)
    assert got[4] == got[3] == got[5]
    s.close()


def test_problem_define_accepts_only_the_arap_energy(gpu_state, tmp_path):
    lib, st = gpu_state.lib, gpu_state.handle
    p_lm = lib.Opt_ProblemDefine(st, b"builtin:arap", b"LMGPU")                     # the other kind of o.t:122
    assert p_lm is not None
    lib.Opt_ProblemDelete(st, p_lm)
    assert lib.Opt_ProblemDefine(st, b"builtin:arap", b"bogus") is None
    assert lib.Opt_ProblemDefine(st, str(tmp_path / "missing.t").encode(), b"gaussNewtonGPU") is None
    bad = tmp_path / "other.t"
    bad.write_text('local W,H = Dim("W",0), Dim("H",1)\nlocal X = Unknown("X",opt_float,{W,H},0)\nEnergy(X(0,0))\n')
    assert lib.Opt_ProblemDefine(st, str(bad).encode(), b"gaussNewtonGPU") is None
    ref_plan = "/root/reference/arap_plan.t"                                       # build container only
    if os.path.exists(ref_plan):
        p = lib.Opt_ProblemDefine(st, ref_plan.encode(), b"gaussNewtonGPU")
        assert p is not None
        lib.Opt_ProblemDelete(st, p)


@pytest.mark.parametrize("name", ["solve_64_1x2x50", "solve_64_1x10x400", "solve_cat128_1x1x100",
                                  "solve_cat128_1x4x50", "solve_cat128_2x1x100", "solve_cat128_1x1x200"])
def test_T2_frame_solver_vs_committed_goldens(gpu_state, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    H, W = g["mask_red"].shape
    numIter, nIter, lIter = (int(v) for v in g["schedule"])
    fs = opt.FrameSolver(gpu_state, W, H, batch=1)
    fs.set_frame(0, g["mask_red"], g["constraints"])
    fs.solve(1, numIter, nIter, lIter)
    fs.warp(1)
    r = fs.results(0, want_rgb=False)
    ys, xs = np.mgrid[0:H, 0:W]
    grid = np.stack([xs, ys], -1).astype(np.float32)
    assert helpers.rel_l2(r["offset"] - grid, g["offset"] - grid) < T2_TOL
    assert helpers.rel_l2(r["angle"], g["angle"]) < T2_TOL
    assert np.array_equal(r["flow"], r["offset"] - grid)                # flow = Offset - grid, exactly
    assert abs(r["cost"] - g["costs"][-1]) <= 1e-4 * g["costs"][-1]
    fs.close()


def test_host_ramp_and_device_ramp_agree_and_batch_slots_are_independent(gpu_state):
    """The reference's host loop over the ten Opt_* symbols (opt.CombinedSolver) and the device-resident
    batched FrameSolver run the same arithmetic: identical output.  Frames in a batch do not interact."""
    from arap_flow_amd import synth
    W, H = 160, 96
    frames = [synth.make_frame(W, H, seed=s, K=1 + s % 2, fd=1) for s in range(3)]
    sched = (3, 2, 40)
    fs = opt.FrameSolver(gpu_state, W, H, batch=3)
    for b, f in enumerate(frames):
        fs.set_frame(b, f["mask_red"], f["constraints"])
    fs.solve(3, *sched)
    batch_out = [fs.results(b, want_rgb=False) for b in range(3)]
    fs.close()
    for b, f in enumerate(frames):
        cs = opt.CombinedSolver(gpu_state, W, H, num_iter=sched[0], non_linear_iter=sched[1], linear_iter=sched[2])
        cs.add_image(f["mask_red"], np.concatenate([f["constraints"], opt.border_pins(W, H)]))
        costs = cs.solve_all()
        o = cs.warp_field.cpu().numpy()
        a = cs.warp_angles.cpu().numpy()
        cs.close()
        ys, xs = np.mgrid[0:H, 0:W]
        grid = np.stack([xs, ys], -1).astype(np.float32)
        assert helpers.rel_l2(batch_out[b]["offset"] - grid, o - grid) < 1e-5
        assert helpers.rel_l2(batch_out[b]["angle"], a) < 1e-5
        assert abs(batch_out[b]["cost"] - costs[-1]) <= 1e-5 * costs[-1]


def test_T4_full_schedule_cat512_vs_reference_golden(gpu_state, golden_dir):
    """The reference's own known answer (ARAP/warping/cat512_iFlo.flo, schedule 19/8/400)."""
    cat = helpers.load_cat512(golden_dir)
    H, W = cat["mask_red"].shape
    fs = opt.FrameSolver(gpu_state, W, H, batch=1)
    fs.set_frame(0, cat["mask_red"], cat["constraints"], rgb=cat["rgb"])
    fs.solve(1, 19, 8, 400)
    fs.warp(1)
    r = fs.results(0)
    fs.close()
    flow, gold = r["flow"], cat["golden_flow"]
    act = cat["mask_red"] == 0
    band = helpers.t4_bands(golden_dir)        # from the committed table of oracle variants, not from this result
    assert np.all(flow[~act] == 0)
    for x1, y1, x2, y2 in cat["constraints"]:
        assert np.abs(flow[y1, x1] - gold[y1, x1]).max() < band["handle_px"]
    assert helpers.rel_l2(flow[act], gold[act]) < band["rel_l2"]
    assert np.median(np.linalg.norm(flow - gold, axis=-1)[act]) < band["median_px"]
    assert abs(helpers.neg_det_quads(flow, act) - helpers.neg_det_quads(gold, act)) <= band["quads"]
    assert band["cost"][0] < r["cost"] < band["cost"][1]
    # and it IS the table's product row: the HIP path is the bit-level twin of that oracle variant (tier T3)
    assert r["cost"] == band["table"]["variants"][band["table"]["product_variant"]]["final_cost"]
    # the warped outputs of that solve vs the reference's committed PNGs: the fields differ by the
    # rounding-trajectory noise above, so compare coverage, not pixels
    assert (r["warped_mask"] != cat["golden_wmsk"]).mean() < 0.01


def test_resident_and_two_kernel_paths_are_bit_identical(gpu_state):
    """The on-chip resident PCG kernel and the two-kernels-per-iteration path perform the same float32
    operation list: identical Offset/Angle bits.  Covers 8, 4 and 1 frames in flight and a frame that is
    too large for eight groups (so the groups widen)."""
    from arap_flow_amd import synth
    cases = [(854, 480, 5, 1, (2, 2, 60)), (320, 200, 2, 3, (3, 2, 40)), (200, 120, 8, 1, (1, 3, 25)),
             (854, 480, 2, 0, (1, 2, 60)),      # K = 0: every vertex active, 1680 tiles: groups of 256 workgroups on 4 XCDs
             (1920, 1080, 3, 1, (1, 2, 40))]    # ~2000 tiles: groups spanning XCDs with sparse masks
    for W, H, nfr, K, sched in cases:
        frames = [synth.make_frame(W, H, seed=10 + s, K=max(K, 1), fd=2, full_mask=(K == 0)) for s in range(nfr)]
        outs = []
        for resident in (True, False):
            gpu_state.set_resident(resident)
            fs = opt.FrameSolver(gpu_state, W, H, batch=nfr)
            for b, f in enumerate(frames):
                fs.set_frame(b, f["mask_red"], f["constraints"])
            fs.solve(nfr, *sched)
            outs.append([fs.results(b, want_rgb=False) for b in range(nfr)])
            st = fs.stats()
            assert (st["resident_launches"] > 0) == resident
            fs.close()
        gpu_state.set_resident(True)
        for a, b in zip(*outs):
            assert np.array_equal(a["offset"], b["offset"]) and np.array_equal(a["angle"], b["angle"])
            assert a["cost"] == b["cost"]


def test_frame_solver_step_sequencing_quiet_vs_verbose(gpu_state, capfd):
    """A quiet frame solve runs a Gauss-Newton step as [k_gn_init_resf, resident launch that applies the step itself], all
    steps of a ramp step in one graph; a verbose one (the state prints the cost after every step) as [k_gn_prep, k_gn_init,
    resident launch, k_gn_update] step by step.  Same operation list: identical bits, and both on the resident kernel."""
    from arap_flow_amd import synth
    W, H, nfr, sched = 320, 200, 3, (3, 3, 40)
    frames = [synth.make_frame(W, H, seed=70 + s, K=1, fd=2) for s in range(nfr)]
    outs = []
    loud = opt.State(verbosity=1)
    try:
        for st in (gpu_state, loud):
            fs = opt.FrameSolver(st, W, H, batch=nfr)
            for b, f in enumerate(frames):
                fs.set_frame(b, f["mask_red"], f["constraints"])
            fs.solve(nfr, *sched)
            outs.append([fs.results(b, want_rgb=False) for b in range(nfr)])
            assert fs.stats()["resident_launches"] == sched[0] * sched[1]
            fs.close()
    finally:
        loud.close()
    assert capfd.readouterr().out.count("cost:") == sched[0] * sched[1]      # (the verbose solve did step one by one)
    for a, b in zip(*outs):
        assert np.array_equal(a["offset"], b["offset"]) and np.array_equal(a["angle"], b["angle"])
        assert a["cost"] == b["cost"]


def test_resident_kernel_longest_pcg_loop_and_the_limit_above_it(gpu_state):
    """The border-z granules of the resident kernel carry 16-bit iteration tags (2 l + 3): launches of up to
    RES_MAX_L = 32 000 PCG iterations run on it (tags up to 64 003, bit-identical to the two-kernel path); one
    iteration more and the solve takes the two-kernel path (arapopt.hip: plan_resident_eligible)."""
    from arap_flow_amd import synth
    W, H = 96, 64
    frames = [synth.make_frame(W, H, seed=40 + s, K=1, fd=2) for s in range(3)]
    outs = {}
    for L, resident in ((32000, True), (32000, False), (32001, True)):
        gpu_state.set_resident(resident)
        fs = opt.FrameSolver(gpu_state, W, H, batch=len(frames))
        for b, f in enumerate(frames):
            fs.set_frame(b, f["mask_red"], f["constraints"])
        fs.solve(len(frames), 1, 1, L)
        outs[(L, resident)] = [fs.results(b, want_rgb=False) for b in range(len(frames))]
        assert (fs.stats()["resident_launches"] > 0) == (resident and L <= 32000), (L, resident)
        assert gpu_state.lib.ArapFlow_ResidentFailed(gpu_state.handle) == 0        # no wait gave up (no stale tag)
        fs.close()
    gpu_state.set_resident(True)
    # (a PCG loop this far past convergence may end in non-finite values: the comparison is on the bit patterns)
    bits = lambda r: (r["offset"].view(np.uint32), r["angle"].view(np.uint32))
    for a, b in zip(outs[(32000, True)], outs[(32000, False)]):
        assert all(np.array_equal(x, y) for x, y in zip(bits(a), bits(b)))


@pytest.mark.parametrize("case", ["synth160x96", "cat512"])
def test_T3_full_schedule_deterministic_twin(gpu_state, oracle, golden_dir, case):
    """Tier T3: the full 19/8/400 schedule, HIP vs the float32 CPU oracle that performs the same operation
    list (dot products accumulated in float64, spec cos/sin).  SURVEY 8c asks for 1e-4 "if reachable";
    measured: 0 differing floats (tools/dbg_t3.py), so the test demands 1e-6 and reports bit equality."""
    from arap_flow_amd import synth
    if case == "cat512":
        cat = helpers.load_cat512(golden_dir)
        mask, cons = cat["mask_red"], cat["constraints"]
    else:
        f = synth.make_frame(160, 96, seed=3, K=1, fd=2)
        mask, cons = f["mask_red"], f["constraints"]
    H, W = mask.shape
    fs = opt.FrameSolver(gpu_state, W, H, batch=1)
    fs.set_frame(0, mask, cons)
    fs.solve(1, 19, 8, 400)
    r = fs.results(0, want_rgb=False)
    fs.close()
    O, A, costs = oracle.frame(mask, cons, dtype=np.float32, mode=1, trig=1)
    ys, xs = np.mgrid[0:H, 0:W]
    grid = np.stack([xs, ys], -1).astype(np.float32)
    assert helpers.rel_l2(r["offset"] - grid, O - grid) < 1e-6
    assert helpers.rel_l2(r["angle"], A) < 1e-6
    assert abs(r["cost"] - costs[-1]) <= 1e-6 * costs[-1]
    print("T3 %s: differing floats Offset %d Angle %d" % (case, int((r["offset"] != O).sum()), int((r["angle"] != A).sum())))


def test_multseg_segments_batched_equal_separate_solves(gpu_state):
    """--multseg (para_gen.py:518-540): each label of a frame is its own ARAP solve; batching the segment
    solves of one frame on one GPU (BASELINE config 3) must equal solving them one at a time."""
    from arap_flow_amd import synth
    W, H = 214, 120
    frame = synth.make_frame(W, H, seed=21, K=3, fd=2)
    segs = synth.segment_masks(frame)
    sched = (2, 2, 50)
    fs = opt.FrameSolver(gpu_state, W, H, batch=len(segs))
    for b, s in enumerate(segs):
        fs.set_frame(b, s["mask_red"], s["constraints"])
    fs.solve(len(segs), *sched)
    together = [fs.results(b, want_rgb=False) for b in range(len(segs))]
    fs.close()
    for b, s in enumerate(segs):
        one = opt.FrameSolver(gpu_state, W, H, batch=1)
        one.set_frame(0, s["mask_red"], s["constraints"])
        one.solve(1, *sched)
        r = one.results(0, want_rgb=False)
        one.close()
        assert np.array_equal(r["offset"], together[b]["offset"]) and np.array_equal(r["angle"], together[b]["angle"])
        assert np.all(together[b]["flow"][s["mask_red"] != 0] == 0)


def test_drop_in_api_takes_resident_kernel_only_for_grid_urshape(gpu_state, oracle):
    """Opt_ProblemSolve analyses Mask/UrShape at Init: pixel-grid UrShape -> resident kernel, generic UrShape ->
    two-kernel path; both equal the float32 oracle bit for bit."""
    W, H = 150, 90
    for generic in (False, True):
        pb = helpers.random_problem(W, H, seed=77, generic_urshape=generic, ncons=60)
        dev = {k: torch.from_numpy(pb[k].copy()).cuda() for k in "OAUCM"}
        s = opt.OptSolver(gpu_state, (W, H))
        pp = opt.NamedParameters()
        for n, k in [("Offset", "O"), ("Angle", "A"), ("UrShape", "U"), ("Constraints", "C"), ("Mask", "M")]:
            pp.set(n, dev[k])
        pp.set("w_fitSqrt", 10.0); pp.set("w_regSqrt", 0.1)
        sp = opt.NamedParameters()
        sp.set("nIterations", 2); sp.set("lIterations", 40)
        cost = s.solve(sp, pp)
        assert (s.resident_launches() > 0) == (not generic)
        s.close()
        Or, Ar, costs = oracle.solve(pb["O"], pb["A"], pb["U"], pb["C"], pb["M"], 10.0, 0.1, 2, 40, dtype=np.float32,
                                     mode=1, trig=1)
        assert np.array_equal(dev["O"].cpu().numpy(), Or) and np.array_equal(dev["A"].cpu().numpy(), Ar)
        assert cost == costs[-1]


@pytest.mark.parametrize("W,H,nIter,lIter,radius", [(64, 48, 4, 25, 1e4), (130, 37, 6, 40, 1e4), (90, 70, 5, 30, 1e-3)])
def test_LM_solver_kind_vs_oracle(gpu_state, oracle, W, H, nIter, lIter, radius):
    """Solver kind "LMGPU"  through Opt_ProblemSolve vs the CPU restatement of the same branch:
    trust-region bookkeeping (accept / reject / revert), CtC, model cost, zeta break, residual reset every 10th
    iteration.  The reference holds no LM output, so this parity is pinned to the restatement only."""
    pb = helpers.random_problem(W, H, seed=W * 3 + H, generic_urshape=(W % 2 == 0), ncons=max(6, W * H // 50))
    dev = {k: torch.from_numpy(pb[k].copy()).cuda() for k in "OAUCM"}
    s = opt.OptSolver(gpu_state, (W, H), opt.BUILTIN_PLAN, b"LMGPU")
    pp = opt.NamedParameters()
    for n, k in [("Offset", "O"), ("Angle", "A"), ("UrShape", "U"), ("Constraints", "C"), ("Mask", "M")]:
        pp.set(n, dev[k])
    pp.set("w_fitSqrt", 10.0); pp.set("w_regSqrt", 0.1)
    sp = opt.NamedParameters()
    sp.set("nIterations", nIter); sp.set("lIterations", lIter); sp.set("trust_region_radius", float(radius))
    cost = s.solve(sp, pp)
    s.close()
    Or, Ar, costs, steps, rad = oracle.solve_lm(pb["O"], pb["A"], pb["U"], pb["C"], pb["M"], 10.0, 0.1, nIter, lIter,
                                                trust_region_radius=radius)
    dO, dOr = dev["O"].cpu().numpy() - pb["O"], Or - pb["O"]
    assert helpers.rel_l2(dO, dOr) < 1e-4 and helpers.rel_l2(dev["A"].cpu().numpy() - pb["A"], Ar - pb["A"]) < 1e-4
    assert abs(cost - costs[-1]) <= 1e-4 * abs(costs[-1])
    assert costs[-1] < costs[0]                                       # LM made progress
    ex = pb["M"] != 0
    assert np.array_equal(dev["O"].cpu().numpy()[ex], pb["O"][ex])
    print("LM %dx%d: %d steps, final radius %g, mismatching floats %d" % (W, H, steps, rad, int((dO != dOr).sum())))


def test_solver_reuse_across_frames_of_different_tile_counts(gpu_state):
    """One FrameSolver fed small frames (16 resident groups), then a large one (fewer, wider groups), then small
    again: the cached hipGraph must follow the group count.  Each result equals a fresh solver's."""
    from arap_flow_amd import synth
    W, H = 854, 480
    small = synth.make_frame(W, H, seed=31, K=3, fd=2)
    segs = synth.segment_masks(small)                       # ~35 k active vertices each
    big = synth.make_frame(W, H, seed=32, full_mask=True)   # 1680 tiles
    sched = (1, 2, 30)
    fs = opt.FrameSolver(gpu_state, W, H, batch=3)
    got = []
    for rnd in range(3):
        frames = [big] if rnd == 1 else segs
        for b, f in enumerate(frames):
            fs.set_frame(b, f["mask_red"], f["constraints"])
        fs.solve(len(frames), *sched)
        got.append([fs.results(b, want_rgb=False)["offset"] for b in range(len(frames))])
    fs.close()
    for rnd in range(3):
        frames = [big] if rnd == 1 else segs
        for b, f in enumerate(frames):
            one = opt.FrameSolver(gpu_state, W, H, batch=1)
            one.set_frame(0, f["mask_red"], f["constraints"])
            one.solve(1, *sched)
            assert np.array_equal(one.results(0, want_rgb=False)["offset"], got[rnd][b])
            one.close()


def test_solver_reuse_across_batches_of_the_same_shape(gpu_state):
    """Batch after batch of DIFFERENT frames whose deal has the same shape (launches, slot counts, list length rounded to 64
    workgroups) replays the captured graph of the first batch: only the tables behind it change.  Each result equals a fresh
    solver's (a stale table or a baked-in argument would show)."""
    from arap_flow_amd import synth
    W, H, nfr, sched = 854, 480, 4, (2, 3, 30)
    fs = opt.FrameSolver(gpu_state, W, H, batch=nfr)
    got, sets = [], []
    for rnd in range(3):
        frames = [synth.make_frame(W, H, seed=200 + 10 * rnd + s) for s in range(nfr)]
        sets.append(frames)
        for b, f in enumerate(frames):
            fs.set_frame(b, f["mask_red"], f["constraints"])
        fs.solve(nfr, *sched)
        got.append([fs.results(b, want_rgb=False) for b in range(nfr)])
    fs.close()
    for rnd in (0, 2):
        one = opt.FrameSolver(gpu_state, W, H, batch=nfr)
        for b, f in enumerate(sets[rnd]):
            one.set_frame(b, f["mask_red"], f["constraints"])
        one.solve(nfr, *sched)
        for b in range(nfr):
            r = one.results(b, want_rgb=False)
            assert np.array_equal(r["offset"], got[rnd][b]["offset"]) and np.array_equal(r["angle"], got[rnd][b]["angle"])
            assert r["cost"] == got[rnd][b]["cost"]
        one.close()


def test_resident_packing_of_many_uneven_solves(gpu_state):
    """The host deals every solve a group of the resident launch's 512 workgroups sized by its active-tile count and
    packs small solves onto one XCD (arapopt.hip: plan_resident_pack).  48 solves of very different sizes need two
    launches per step with more than 16 solves in flight; every one equals the two-kernel path bit for bit."""
    from arap_flow_amd import synth
    W, H = 256, 128                                             # 4 x 32 = 128 tiles of 64 x 4
    solves = []
    for s in range(36):                                         # whole-frame solves: 128 tiles = 15 workgroups each
        solves.append(synth.make_frame(W, H, seed=100 + s, full_mask=True))
    for s in range(4):                                          # and the three segments of four --multseg frames
        solves += synth.segment_masks(synth.make_frame(W, H, seed=200 + s, K=3, fd=2))
    n = len(solves)
    sched = (1, 2, 30)
    outs = []
    for resident in (True, False):
        gpu_state.set_resident(resident)
        fs = opt.FrameSolver(gpu_state, W, H, batch=n)
        for b, f in enumerate(solves):
            fs.set_frame(b, f["mask_red"], f["constraints"])
        fs.solve(n, *sched)
        outs.append([fs.results(b, want_rgb=False) for b in range(n)])
        st = fs.stats()
        if resident:
            assert st["resident_launches_per_step"] == 2 and st["resident_solves_in_flight"] > 16, st
        else:
            assert st["resident_launches_per_step"] == 0
        fs.close()
    gpu_state.set_resident(True)
    for a, b in zip(*outs):
        assert np.array_equal(a["offset"], b["offset"]) and np.array_equal(a["angle"], b["angle"])
        assert a["cost"] == b["cost"]


def test_resident_mixed_wide_and_narrow_solves(gpu_state):
    """One batch with solves of very different widths: a whole-frame solve (1680 tiles: four whole XCDs, two-level
    sums), a half-frame one (two XCDs) and DAVIS-shaped ones that share the remaining XCDs.  The host packs them into
    two launches; every result equals the two-kernel path bit for bit."""
    from arap_flow_amd import synth
    W, H = 854, 480
    full = synth.make_frame(W, H, seed=40, full_mask=True)
    half = synth.make_frame(W, H, seed=41, full_mask=True)
    half["mask_red"] = half["mask_red"].copy()
    half["mask_red"][:, W // 2:] = 255                           # ~840 tiles: two XCDs
    half["constraints"] = np.asarray([c for c in half["constraints"] if c[0] < W // 2 - 2 and c[2] < W // 2 - 2], np.int32)
    solves = [full, half] + [synth.make_frame(W, H, seed=42 + s) for s in range(5)]
    n = len(solves)
    sched = (1, 2, 40)
    outs = []
    for resident in (True, False):
        gpu_state.set_resident(resident)
        fs = opt.FrameSolver(gpu_state, W, H, batch=n)
        for b, f in enumerate(solves):
            fs.set_frame(b, f["mask_red"], f["constraints"])
        fs.solve(n, *sched)
        outs.append([fs.results(b, want_rgb=False) for b in range(n)])
        st = fs.stats()
        if resident:
            assert st["resident_launches_per_step"] == 2 and st["resident_solves_in_flight"] >= 4, st
        fs.close()
    gpu_state.set_resident(True)
    for a, b in zip(*outs):
        assert np.array_equal(a["offset"], b["offset"]) and np.array_equal(a["angle"], b["angle"])


def test_frame_edge_cases_empty_mask_and_no_constraints(gpu_state, oracle):
    """a frame whose mask excludes every vertex (nothing to solve, flow stays 0), a frame without any file
    constraint (only the border pins act) and a batch mixing them with a normal frame"""
    from arap_flow_amd import synth
    W, H = 100, 60
    normal = synth.make_frame(W, H, seed=8)
    none_active = np.full((H, W), 255, np.uint8)
    all_active = np.zeros((H, W), np.uint8)
    empty = np.zeros((0, 4), np.int32)
    fs = opt.FrameSolver(gpu_state, W, H, batch=3)
    fs.set_frame(0, none_active, normal["constraints"], rgb=normal["rgb"])
    fs.set_frame(1, all_active, empty, rgb=normal["rgb"])
    fs.set_frame(2, normal["mask_red"], normal["constraints"], rgb=normal["rgb"])
    fs.solve(3, 2, 2, 40)
    fs.warp(3)
    r = [fs.results(b) for b in range(3)]
    fs.close()
    assert np.all(r[0]["flow"] == 0) and np.all(r[0]["warped_mask"] == 0) and r[0]["cost"] == 0.0
    assert np.all(r[1]["flow"] == 0) and r[1]["cost"] == 0.0        # pinned border, rest state is the optimum
    assert np.array_equal(r[1]["warped_rgb"][:-1, :-1], normal["rgb"][:-1, :-1])
    O, A, costs = oracle.frame(normal["mask_red"], normal["constraints"], numIter=2, nIterations=2, lIterations=40,
                               dtype=np.float32, mode=1, trig=1)
    assert np.array_equal(r[2]["offset"], O)


@pytest.mark.parametrize("tile", [(0, 0), (16, 16), (32, 8), (64, 4), (32, 16), (64, 8), (-1, -1)])
def test_lds_tiled_phase_a_is_bit_identical(gpu_state, oracle, tile):
    """two-kernel path with an LDS-staged phase A (BASELINE config 5's tile shapes): same bits as the oracle, for a
    generic UrShape (Opt_* path) and for a batch of frames whose size is not a multiple of the tile"""
    from arap_flow_amd import synth
    gpu_state.set_resident(False)
    gpu_state.set_tile(*tile)
    try:
        W, H = 149, 83
        pb = helpers.random_problem(W, H, seed=5, generic_urshape=True, ncons=50)
        O, A, cost = _solve_opt(gpu_state, pb, 2, 30)
        Or, Ar, costs = oracle.solve(pb["O"], pb["A"], pb["U"], pb["C"], pb["M"], pb["wf"], pb["wr"], 2, 30,
                                     dtype=np.float32, mode=1, trig=1)
        assert np.array_equal(O, Or) and np.array_equal(A, Ar) and cost == costs[-1]
        f = synth.make_frame(W, H, seed=6, K=2)
        fs = opt.FrameSolver(gpu_state, W, H, batch=2)
        fs.set_frame(0, f["mask_red"], f["constraints"]); fs.set_frame(1, np.zeros((H, W), np.uint8), f["constraints"])
        fs.solve(2, 1, 2, 25)
        r0 = fs.results(0, want_rgb=False)
        fs.close()
        Of, Af, _ = oracle.frame(f["mask_red"], f["constraints"], numIter=1, nIterations=2, lIterations=25, dtype=np.float32,
                                 mode=1, trig=1)
        assert np.array_equal(r0["offset"], Of) and np.array_equal(r0["angle"], Af)
    finally:
        gpu_state.set_tile(-1, -1)
        gpu_state.set_resident(True)


def test_profiled_solve_records_cost_per_gn_iteration(gpu_state, oracle):
    """launchProfiledSolve (OptUtils.h:47-64): the step-by-step solve gives the same unknowns as Opt_ProblemSolve
    and one (cost, ms) record per Gauss-Newton iteration whose costs are the oracle's"""
    W, H = 120, 70
    pb = helpers.random_problem(W, H, seed=2, generic_urshape=False, ncons=40)
    outs = []
    for profiled in (False, True):
        dev = {k: torch.from_numpy(pb[k].copy()).cuda() for k in "OAUCM"}
        s = opt.OptSolver(gpu_state, (W, H))
        pp = opt.NamedParameters()
        for n, k in [("Offset", "O"), ("Angle", "A"), ("UrShape", "U"), ("Constraints", "C"), ("Mask", "M")]:
            pp.set(n, dev[k])
        pp.set("w_fitSqrt", 10.0); pp.set("w_regSqrt", 0.1)
        sp = opt.NamedParameters()
        sp.set("nIterations", 3); sp.set("lIterations", 25)
        recs = []
        s.solve(sp, pp, profiled=profiled, iters=recs)
        outs.append((dev["O"].cpu().numpy(), recs))
        s.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    _, _, costs = oracle.solve(pb["O"], pb["A"], pb["U"], pb["C"], pb["M"], 10.0, 0.1, 3, 25, dtype=np.float32, mode=1, trig=1)
    recs = outs[1][1]
    assert len(recs) == 4 and [c for c, _ in recs] == list(costs) and all(ms >= 0 for _, ms in recs)


def test_drop_in_step_sees_mask_and_urshape_changed_in_place(gpu_state, oracle):
    """Opt.h:58-66 lets the caller change the parameters between Opt_ProblemInit and the Opt_ProblemSteps and the
    reference re-reads them at every step (solverGPUGaussNewton.t:960,1026).  Changing the CONTENTS of Mask in place
    (same buffer: new tiles become active, others inactive) between two Steps must give what the oracle gives stepping
    with the new mask; so must turning UrShape into a generic shape (the resident kernel no longer applies)."""
    W, H = 300, 150
    pb = helpers.random_problem(W, H, seed=91, generic_urshape=False, mask_frac=0.0, ncons=80)
    M1 = np.full((H, W), 255.0, np.float32); M1[20:90, 30:170] = 0.0           # active block in the upper left
    M2 = np.full((H, W), 255.0, np.float32); M2[60:140, 120:290] = 0.0         # moves: new tiles active, old ones not
    dev = {k: torch.from_numpy(pb[k].copy()).cuda() for k in "OAUC"}
    dev["M"] = torch.from_numpy(M1.copy()).cuda()
    s = opt.OptSolver(gpu_state, (W, H))
    pp = opt.NamedParameters()
    for n, k in [("Offset", "O"), ("Angle", "A"), ("UrShape", "U"), ("Constraints", "C"), ("Mask", "M")]:
        pp.set(n, dev[k])
    pp.set("w_fitSqrt", 10.0); pp.set("w_regSqrt", 0.1)
    sp = opt.NamedParameters()
    sp.set("nIterations", 3); sp.set("lIterations", 30)
    s.set_solver_parameters(sp)
    s.init(pp)
    assert s.step(pp) == 1
    n_res = s.resident_launches()
    assert n_res > 0
    O1, A1, c1 = oracle.solve(pb["O"], pb["A"], pb["U"], pb["C"], M1, 10.0, 0.1, 1, 30, dtype=np.float32, mode=1, trig=1)
    assert np.array_equal(dev["O"].cpu().numpy(), O1) and np.array_equal(dev["A"].cpu().numpy(), A1)
    dev["M"].copy_(torch.from_numpy(M2))                                      # in place: same device pointer
    torch.cuda.synchronize()
    assert s.step(pp) == 1
    assert s.resident_launches() > n_res                                      # still the resident kernel, new tile list
    O2, A2, c2 = oracle.solve(O1, A1, pb["U"], pb["C"], M2, 10.0, 0.1, 1, 30, dtype=np.float32, mode=1, trig=1)
    assert np.array_equal(dev["O"].cpu().numpy(), O2) and np.array_equal(dev["A"].cpu().numpy(), A2)
    assert s.current_cost() == c2[-1]
    # UrShape becomes generic, in place: the step must leave the resident kernel and still match
    U3 = pb["U"] + np.random.default_rng(5).normal(size=pb["U"].shape).astype(np.float32) * 0.2
    dev["U"].copy_(torch.from_numpy(U3))
    torch.cuda.synchronize()
    n_res = s.resident_launches()
    assert s.step(pp) == 1 and s.resident_launches() == n_res
    O3, A3, c3 = oracle.solve(O2, A2, U3, pb["C"], M2, 10.0, 0.1, 1, 30, dtype=np.float32, mode=1, trig=1)
    assert np.array_equal(dev["O"].cpu().numpy(), O3) and np.array_equal(dev["A"].cpu().numpy(), A3)
    assert s.step(pp) == 0
    s.close()



def _stepwise_with_urshape_turning_generic(state, oracle, expect_resident):
    """Init + three Steps of a drop-in plan; before the third Step UrShape turns generic IN PLACE.  Returns nothing,
    asserts the oracle's bits after every Step."""
    W, H = 300, 150
    pb = helpers.random_problem(W, H, seed=92, generic_urshape=False, mask_frac=0.0, ncons=80)
    M1 = np.full((H, W), 255.0, np.float32); M1[20:90, 30:170] = 0.0
    M2 = np.full((H, W), 255.0, np.float32); M2[60:140, 120:290] = 0.0
    dev = {k: torch.from_numpy(pb[k].copy()).cuda() for k in "OAUC"}
    dev["M"] = torch.from_numpy(M1.copy()).cuda()
    s = opt.OptSolver(state, (W, H))
    pp = opt.NamedParameters()
    for n, k in [("Offset", "O"), ("Angle", "A"), ("UrShape", "U"), ("Constraints", "C"), ("Mask", "M")]:
        pp.set(n, dev[k])
    pp.set("w_fitSqrt", 10.0); pp.set("w_regSqrt", 0.1)
    sp = opt.NamedParameters()
    sp.set("nIterations", 3); sp.set("lIterations", 30)
    s.set_solver_parameters(sp)
    s.init(pp)
    n0 = s.resident_launches()
    assert s.step(pp) == 1
    assert (s.resident_launches() > n0) == expect_resident
    O1, A1, c1 = oracle.solve(pb["O"], pb["A"], pb["U"], pb["C"], M1, 10.0, 0.1, 1, 30, dtype=np.float32, mode=1, trig=1)
    assert np.array_equal(dev["O"].cpu().numpy(), O1) and np.array_equal(dev["A"].cpu().numpy(), A1)
    dev["M"].copy_(torch.from_numpy(M2)); torch.cuda.synchronize()            # new mask, same buffer
    assert s.step(pp) == 1
    O2, A2, c2 = oracle.solve(O1, A1, pb["U"], pb["C"], M2, 10.0, 0.1, 1, 30, dtype=np.float32, mode=1, trig=1)
    assert np.array_equal(dev["O"].cpu().numpy(), O2) and np.array_equal(dev["A"].cpu().numpy(), A2)
    # UrShape becomes generic in place.  The two-kernel path picks its phase-A kernel from the analysis of UrShape:
    # a stale "UrShape is the pixel grid" would make phase A (no UrShape loads) disagree with k_gn_init (real UrShape).
    U3 = pb["U"] + np.random.default_rng(6).normal(size=pb["U"].shape).astype(np.float32) * 0.2
    dev["U"].copy_(torch.from_numpy(U3)); torch.cuda.synchronize()
    n_res = s.resident_launches()
    assert s.step(pp) == 1 and s.resident_launches() == n_res
    O3, A3, c3 = oracle.solve(O2, A2, U3, pb["C"], M2, 10.0, 0.1, 1, 30, dtype=np.float32, mode=1, trig=1)
    assert np.array_equal(dev["O"].cpu().numpy(), O3) and np.array_equal(dev["A"].cpu().numpy(), A3)
    assert s.current_cost() == c3[-1]
    assert s.step(pp) == 0
    s.close()


def test_drop_in_step_sees_urshape_changed_in_place_with_the_resident_kernel_off(gpu_state, oracle):
    """The same protocol with ArapFlow_SetResident(state, 0): every Step runs on the two-kernel path, whose streaming
    phase A applies only while UrShape is the pixel grid -- the analysis must run before EVERY step there too."""
    gpu_state.set_resident(False)
    try:
        _stepwise_with_urshape_turning_generic(gpu_state, oracle, expect_resident=False)
    finally:
        gpu_state.set_resident(True)
