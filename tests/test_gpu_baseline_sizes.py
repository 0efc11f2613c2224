"""HIP vs the float32 CPU oracle, bit for bit, AT THE SIZES OF BASELINE.json's GPU configs (854x480 and 1920x1080),
with the resident kernel's fast paths on.  These are the cases that exercise the packed 8-frames-per-launch deal
of the bench workload, the 21-solve --multseg packing, the groups that span 2 and 4 XCDs (two-level sums
`group_sum_h`, XCD-spanning z publication) and the large-solve path -- on short schedules the oracle finishes in
seconds -- plus one full 19/8/400 run of the bench's frame seed 0.

Reference for the workloads: ARAP/deformation/src/main.cpp:215-221 (19/8/400), para_gen.py:518-540 (multseg split).
Bit equality is possible because both sides perform the same float32 operation list with float64-accumulated,
order-independent dot products (DESIGN.md "Bit-reproducible arithmetic")."""
import numpy as np
import pytest

from arap_flow_amd import opt, synth

pytestmark = pytest.mark.gpu


def _oracle_frame(oracle, f, sched, pins=True):
    return oracle.frame(f["mask_red"], f["constraints"], numIter=sched[0], nIterations=sched[1], lIterations=sched[2],
                        dtype=np.float32, mode=1, trig=1, border_pins=pins)


def _run_batch(gpu_state, W, H, solves, sched, pins=True):
    n = len(solves)
    fs = opt.FrameSolver(gpu_state, W, H, batch=n)
    for b, f in enumerate(solves):
        fs.set_frame(b, f["mask_red"], f["constraints"], border_pins=pins)
    fs.solve(n, *sched)
    out = [fs.results(b, want_rgb=False) for b in range(n)]
    st = fs.stats()
    fs.close()
    return out, st


def _assert_bits(oracle, solves, outs, sched, pins=True, what=""):
    for b, (f, r) in enumerate(zip(solves, outs)):
        O, A, costs = _oracle_frame(oracle, f, sched, pins)
        nO, nA = int((r["offset"] != O).sum()), int((r["angle"] != A).sum())
        assert nO == 0 and nA == 0, "%s solve %d: %d Offset / %d Angle floats differ from the oracle" % (what, b, nO, nA)
        assert r["cost"] == costs[-1], "%s solve %d: cost %r vs oracle %r" % (what, b, r["cost"], costs[-1])
        assert np.all(r["flow"][f["mask_red"] != 0] == 0)


def test_bench_workload_8_frames_in_one_launch_vs_oracle(gpu_state, oracle):
    """configs[1], exactly what bench.py times: 854x480, frame seeds 0..7 in one batch of 8 (one resident launch per
    Gauss-Newton step, one group per XCD), K=1, fd=1; short schedule (2,2,60)."""
    W, H, sched = 854, 480, (2, 2, 60)
    frames = [synth.make_frame(W, H, seed=s, K=1, fd=1) for s in range(8)]
    outs, st = _run_batch(gpu_state, W, H, frames, sched)
    assert st["resident_launches"] > 0 and st["resident_launches_per_step"] == 1 and st["resident_solves_in_flight"] == 8, st
    _assert_bits(oracle, frames, outs, sched, what="854x480 batch of 8")


def test_multseg_854x480_21_packed_solves_vs_oracle(gpu_state, oracle):
    """configs[2]: 854x480 --multseg K=3 fd=2, 7 frames = 21 segment solves packed into the resident launches."""
    W, H, sched = 854, 480, (2, 2, 60)
    solves = [sg for s in range(7) for sg in synth.segment_masks(synth.make_frame(W, H, seed=s, K=3, fd=2))]
    assert len(solves) == 21
    outs, st = _run_batch(gpu_state, W, H, solves, sched)
    assert st["resident_launches"] > 0 and st["resident_solves_in_flight"] >= 16, st
    _assert_bits(oracle, solves, outs, sched, what="854x480 multseg")


def test_full_mask_854x480_four_xcd_groups_vs_oracle(gpu_state, oracle):
    """roofline config: 854x480 with every vertex active (1680 tiles per frame -> groups of 256 workgroups on four
    XCDs each: two-level sums, z published across XCDs), two frames per launch."""
    W, H, sched = 854, 480, (2, 2, 60)
    frames = [synth.make_frame(W, H, seed=s, full_mask=True) for s in range(2)]
    outs, st = _run_batch(gpu_state, W, H, frames, sched)
    assert st["resident_launches"] > 0 and st["resident_launches_per_step"] == 1, st
    _assert_bits(oracle, frames, outs, sched, what="854x480 full mask")


def test_multseg_1920x1080_two_xcd_groups_vs_oracle(gpu_state, oracle):
    """configs[4]: 1920x1080 --multseg K=3 fd=5 (~700-900 tiles per segment: more than one XCD holds).  Every such solve
    takes a home XCD plus a short piece in a bin it shares with other pieces (arapopt.hip: resident_deal; one-hop sums
    with a short second run): 4 frames = 12 segment solves in at most 3 launches per Gauss-Newton step (whole pairs of XCDs
    needed 3)."""
    W, H, sched = 1920, 1080, (1, 2, 40)
    solves = [sg for s in range(4) for sg in synth.segment_masks(synth.make_frame(W, H, seed=s, K=3, fd=5))]
    outs, st = _run_batch(gpu_state, W, H, solves, sched)
    assert st["resident_launches"] > 0 and 1 <= st["resident_launches_per_step"] <= 3, st
    _assert_bits(oracle, solves, outs, sched, what="1920x1080 multseg")


def test_single_segment_1920x1080_vs_oracle(gpu_state, oracle):
    """1920x1080 K=1 (one ~2000-tile solve: four XCDs) and, in the same batch, a frame with every vertex active
    (8100 tiles: the large-solve path, more tiles than the register-resident kernel holds)."""
    W, H, sched = 1920, 1080, (1, 2, 30)
    solves = [synth.make_frame(W, H, seed=3, K=1, fd=5)]
    outs, st = _run_batch(gpu_state, W, H, solves, sched)
    assert st["resident_launches"] > 0, st
    _assert_bits(oracle, solves, outs, sched, what="1920x1080 K=1")
    full = [synth.make_frame(W, H, seed=4, full_mask=True)]
    outs, st = _run_batch(gpu_state, W, H, full, sched)
    _assert_bits(oracle, full, outs, sched, what="1920x1080 full mask")


def test_full_schedule_19_8_400_bench_frame_seed0_vs_oracle(gpu_state, oracle):
    """The benchmarked work itself: the full 19/8/400 schedule on bench frame seed 0 at 854x480, solved inside a
    batch of 8 (as the bench does), slot 0 compared with the oracle bit for bit (60 800 PCG iterations)."""
    W, H, sched = 854, 480, (19, 8, 400)
    frames = [synth.make_frame(W, H, seed=s, K=1, fd=1) for s in range(8)]
    outs, st = _run_batch(gpu_state, W, H, frames, sched)
    assert st["resident_launches"] > 0
    _assert_bits(oracle, frames[:1], outs[:1], sched, what="854x480 full schedule")


def _random_mask(rng, W, H):
    kind = rng.integers(0, 5)
    m = np.full((H, W), 255, np.uint8)
    if kind == 0:
        m[:] = 0
    elif kind == 1:
        for _ in range(rng.integers(1, 5)):
            x0, y0 = rng.integers(0, W), rng.integers(0, H)
            m[y0:y0 + rng.integers(1, H + 1), x0:x0 + rng.integers(1, W + 1)] = 0
    elif kind == 2:
        m[rng.random((H, W)) < rng.uniform(0.05, 0.9)] = 0
    elif kind == 3:
        ys, xs = np.mgrid[0:H, 0:W]
        m[((xs - W / 2) / (W * rng.uniform(0.1, 0.6))) ** 2 + ((ys - H / 2) / (H * rng.uniform(0.1, 0.6))) ** 2 < 1] = 0
    else:
        m[::rng.integers(2, 5)] = 0
    return m


def _random_constraints(rng, W, H):
    c = []
    for _ in range(int(rng.integers(0, 40))):
        x, y = int(rng.integers(0, W)), int(rng.integers(0, H))
        c.append((x, y, x + int(rng.integers(-6, 7)), y + int(rng.integers(-6, 7))))
    return np.asarray(c, np.int32).reshape(-1, 4)


@pytest.mark.parametrize("seed,long_", [(0, False), (1, False), (2, False), (3, True)])
def test_seeded_random_cases_vs_oracle(gpu_state, oracle, seed, long_):
    """The core of tools/fuzz_parity.py inside the suite: random sizes (1x1 .. 900x500), masks (full, rectangles,
    noise, ellipses, one-vertex stripes), constraints, border pins on/off and batch mixes; every solve equals the
    oracle bit for bit (non-finite results of a diverged solve must be non-finite on both sides)."""
    rng = np.random.default_rng(1000 + seed)
    ncases = 3 if long_ else 12
    for it in range(ncases):
        W, H = int(rng.integers(1, 330)), int(rng.integers(1, 200))
        if rng.random() < 0.2:
            W, H = int(rng.integers(600, 900)), int(rng.integers(300, 500))
        nb = int(rng.integers(1, 6))
        solves = [dict(mask_red=_random_mask(rng, W, H), constraints=_random_constraints(rng, W, H)) for _ in range(nb)]
        sched = (3, 3, 200) if long_ else (int(rng.integers(1, 3)), int(rng.integers(1, 3)), int(rng.integers(1, 25)))
        pins = True if long_ else bool(rng.integers(0, 2))
        outs, _ = _run_batch(gpu_state, W, H, solves, sched, pins)
        for b, (f, r) in enumerate(zip(solves, outs)):
            O, A, _ = _oracle_frame(oracle, f, sched, pins)
            fin = np.isfinite(O) & np.isfinite(r["offset"])
            assert np.array_equal(r["offset"][fin], O[fin]), (seed, it, b, W, H, sched, pins)
            if np.isfinite(O).all():
                assert np.array_equal(r["offset"], O) and np.array_equal(r["angle"], A), (seed, it, b, W, H, sched, pins)
