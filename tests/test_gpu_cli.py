"""GPU: the command-line twins end to end (child processes, as the reference's para_gen.py runs its binaries)."""
import os
import os.path as osp
import subprocess
import sys

import numpy as np
import pytest
from PIL import Image

import helpers
from arap_flow_amd import flo, opt, pipeline, synth

pytestmark = pytest.mark.gpu
ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))


def _run(args, cwd):
    env = dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", "0"))
    r = subprocess.run([sys.executable] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_warp_image_cli_reproduces_reference_fixture(tmp_path, golden_dir):
    d = osp.join(golden_dir, "cat512")
    out = _run([osp.join(ROOT, "warp_image.py"), osp.join(d, "cat512_iRGB.png"), osp.join(d, "cat512_iMsk.png"),
                osp.join(d, "cat512_iFlo.flo"), str(tmp_path / "o.png"), str(tmp_path / "om.png")], str(tmp_path))
    assert "Saved" in out
    cat = helpers.load_cat512(golden_dir)
    wm = np.array(Image.open(tmp_path / "om.png").convert("L"))
    assert np.array_equal(wm, cat["golden_wmsk"])
    diff = np.abs(np.array(Image.open(tmp_path / "o.png").convert("RGB")).astype(int) - cat["golden_wrgb"].astype(int))
    assert diff.max() <= 1
    bad = subprocess.run([sys.executable, osp.join(ROOT, "warp_image.py"), "a", "b"], capture_output=True, text=True)
    assert bad.returncode == 1 and "Invalid Input!" in bad.stdout


def test_arap_deform_cli_list_and_single(tmp_path, gpu_state):
    """list-file mode (two sizes in one list -> plan rebuilt) and 6-argument mode; output equals the library"""
    frames = [synth.make_frame(96, 64, seed=1), synth.make_frame(96, 64, seed=2), synth.make_frame(80, 48, seed=3)]
    lines = []
    for k, f in enumerate(frames):
        Image.fromarray(f["rgb"]).save(tmp_path / ("r%d.png" % k))
        Image.fromarray(np.stack([f["mask_red"]] * 3, -1)).save(tmp_path / ("m%d.png" % k))
        pipeline.write_constraints(str(tmp_path / ("c%d.txt" % k)), [tuple(c) for c in f["constraints"]])
        lines.append(" ".join(str(tmp_path / (n % k)) for n in ("r%d.png", "m%d.png", "c%d.txt", "f%d.flo", "w%d.png", "wm%d.png")))
    (tmp_path / "list.txt").write_text("\n".join(lines))
    out = _run([osp.join(ROOT, "arap_deform.py"), str(tmp_path / "list.txt")], str(tmp_path))
    assert out.count("Saved") == 3
    fs = opt.FrameSolver(gpu_state, 96, 64, batch=1)
    fs.set_frame(0, frames[1]["mask_red"], frames[1]["constraints"], rgb=frames[1]["rgb"])
    fs.solve(1, 19, 8, 400); fs.warp(1)
    r = fs.results(0); fs.close()
    assert np.array_equal(flo.flow_read(str(tmp_path / "f1.flo")), r["flow"])
    assert np.array_equal(np.array(Image.open(tmp_path / "w1.png")), r["warped_rgb"])
    assert np.array_equal(np.array(Image.open(tmp_path / "wm1.png")), r["warped_mask"] > 0)
    assert flo.flow_read(str(tmp_path / "f2.flo")).shape == (48, 80, 2)
    single = [str(tmp_path / n) for n in ("r0.png", "m0.png", "c0.txt", "s.flo", "s.png", "sm.png")]
    _run([osp.join(ROOT, "arap_deform.py")] + single, str(tmp_path))
    assert np.array_equal(flo.flow_read(single[3]), flo.flow_read(str(tmp_path / "f0.flo")))
    bad = subprocess.run([sys.executable, osp.join(ROOT, "arap_deform.py")], capture_output=True, text=True)
    assert bad.returncode == 1 and "Invalid Input!" in bad.stdout


@pytest.mark.parametrize("multseg", [False, True])
def test_para_gen_end_to_end(tmp_path, multseg):
    """tiny DAVIS-shaped tree -> para_gen.py --gpu 0 [--multseg]: outputs, all_files.list, flatten"""
    W, H = 96, 64
    inp, outp, mdir = tmp_path / "in", tmp_path / "out", tmp_path / "matches"
    for seq in ("a", "b"):
        os.makedirs(inp / "orgRGB" / seq); os.makedirs(inp / "orgMasks" / seq); os.makedirs(mdir / seq)
        fr = synth.make_frame(W, H, seed=len(seq) + ord(seq), K=2, fd=1)
        for n in range(3):
            Image.fromarray(fr["rgb"]).save(inp / "orgRGB" / seq / ("%05d.png" % n))
            Image.fromarray(fr["labels"].astype(np.uint8)).save(inp / "orgMasks" / seq / ("%05d.png" % n))
            (mdir / seq / ("%05d.txt" % n)).write_text("\n".join("%d %d %d %d 1.0 0" % tuple(c) for c in fr["constraints"]))
    args = [osp.join(ROOT, "para_gen.py"), "--input", str(inp), "--output", str(outp), "--gpu", "0", "--fd", "1",
            "--matches", str(mdir)] + (["--multseg"] if multseg else [])
    _run(args, str(tmp_path))
    lst = open(outp / "all_files.list").read().splitlines()
    assert len(lst) == 4                                              # 2 sequences x (3 - fd) pairs
    for ln in lst:
        rgb1, rgb2, fl = ln.split(" ")
        f = flo.flow_read(fl)
        assert f.shape == (H, W, 2) and np.abs(f).max() > 0.5
        assert np.array(Image.open(rgb2)).shape == (H, W, 3)
    wm = np.array(Image.open(str(outp / "wMasks" / "a" / "00000.png")))
    lab = np.array(Image.open(inp / "orgMasks" / "a" / "00000.png"))
    assert abs(int((wm != 0).sum()) - int((lab != 0).sum())) < 0.2 * (lab != 0).sum()
    assert not [f for f in os.listdir(outp / "Flow" / "a") if "_seg" in f]          # flattened and removed



def test_para_gen_with_the_builtin_matcher(tmp_path):
    """`para_gen.py --dm_bin builtin`: the pairs are matched by this repo's GPU matcher (libarapmatch.so through the
    matcher server; the reference calls the external DeepMatching binary here, para_gen.py:227-240), the matches are
    filtered into constraints and solved.  Frame 2 of every pair is frame 1 moved by a few pixels, label mask included:
    the flow written for the object must be that translation."""
    W, H, dx, dy = 192, 128, 5, -3
    inp, outp = tmp_path / "in", tmp_path / "out"
    os.makedirs(inp / "orgRGB" / "s"); os.makedirs(inp / "orgMasks" / "s")
    rng = np.random.default_rng(7)
    big = synth.make_rgb(W + 64, H + 64, 5).astype(np.int32) + rng.integers(-40, 41, (H + 64, W + 64, 1))
    big = np.clip(big, 0, 255).astype(np.uint8)
    lab_big = np.zeros((H + 64, W + 64), np.uint8)
    yy, xx = np.mgrid[0:H + 64, 0:W + 64]
    lab_big[((xx - 32 - W / 2) / (W * 0.28)) ** 2 + ((yy - 32 - H / 2) / (H * 0.30)) ** 2 < 1.0] = 1
    for n in range(3):                                                  # frame n = the content moved by n * (dx, dy)
        y0, x0 = 32 - n * dy, 32 - n * dx
        Image.fromarray(big[y0:y0 + H, x0:x0 + W]).save(inp / "orgRGB" / "s" / ("%05d.png" % n))
        Image.fromarray(lab_big[y0:y0 + H, x0:x0 + W]).save(inp / "orgMasks" / "s" / ("%05d.png" % n))
    _run([osp.join(ROOT, "para_gen.py"), "--input", str(inp), "--output", str(outp), "--gpu", "0", "--fd", "1",
          "--dm_bin", "builtin"], str(tmp_path))
    lst = open(outp / "all_files.list").read().splitlines()
    assert len(lst) == 2
    cst = open(outp / "tmpCnstr" / "s" / "00000.txt").read().split()
    assert int(cst[0]) > 50                                            # matches on the object that passed the filter
    f = flo.flow_read(lst[0].split(" ")[2])
    lab = np.array(Image.open(inp / "orgMasks" / "s" / "00000.png")) != 0
    core = lab & np.roll(lab, 8, 0) & np.roll(lab, -8, 0) & np.roll(lab, 8, 1) & np.roll(lab, -8, 1)
    med = np.median(f[core], axis=0)
    assert abs(med[0] - dx) <= 1.0 and abs(med[1] - dy) <= 1.0, med


def _bins():
    from arap_flow_amd import build
    return {osp.basename(o): o for o in build.build_host()}


def test_cpp_warp_image_reproduces_reference_fixture(tmp_path, golden_dir):
    b = _bins()
    d = osp.join(golden_dir, "cat512")
    r = subprocess.run([b["warp_image"], osp.join(d, "cat512_iRGB.png"), osp.join(d, "cat512_iMsk.png"),
                        osp.join(d, "cat512_iFlo.flo"), str(tmp_path / "o.png"), str(tmp_path / "om.png")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "Saved" in r.stdout, r.stdout + r.stderr
    cat = helpers.load_cat512(golden_dir)
    assert np.array_equal(np.array(Image.open(tmp_path / "om.png").convert("L")), cat["golden_wmsk"])
    diff = np.abs(np.array(Image.open(tmp_path / "o.png").convert("RGB")).astype(int) - cat["golden_wrgb"].astype(int))
    assert diff.max() <= 1 and (diff.max(-1) > 0).mean() < 0.005


def test_cpp_arap_deform_equals_python_twin_and_serves_para_gen(tmp_path, gpu_state):
    """the C++ driver (arap_flow_amd/host/arap_deform.cpp) writes the same .flo / PNG content as the library path,
    and para_gen.py accepts it as --arap_bin exactly like the reference's executable"""
    b = _bins()
    frames = [synth.make_frame(96, 64, seed=5), synth.make_frame(96, 64, seed=6), synth.make_frame(70, 50, seed=7)]
    lines = []
    for k, f in enumerate(frames):
        Image.fromarray(f["rgb"]).save(tmp_path / ("r%d.png" % k))
        Image.fromarray(f["mask_red"]).save(tmp_path / ("m%d.png" % k))                 # greyscale mask PNG
        pipeline.write_constraints(str(tmp_path / ("c%d.txt" % k)), [tuple(c) for c in f["constraints"]])
        lines.append(" ".join(str(tmp_path / (n % k)) for n in ("r%d.png", "m%d.png", "c%d.txt", "f%d.flo", "w%d.png", "wm%d.png")))
    (tmp_path / "list.txt").write_text("\n".join(lines) + "\n")
    r = subprocess.run([b["arap_deform"], str(tmp_path / "list.txt")], capture_output=True, text=True, timeout=600,
                       cwd=str(tmp_path))
    assert r.returncode == 0 and r.stdout.count("Saved") == 3, r.stdout + r.stderr
    assert "Starting to re-build plan" in r.stdout                                        # third frame has another size
    for k, f in enumerate(frames):
        H, W = f["mask_red"].shape
        fs = opt.FrameSolver(gpu_state, W, H, batch=1)
        fs.set_frame(0, f["mask_red"], f["constraints"], rgb=f["rgb"])
        fs.solve(1, 19, 8, 400); fs.warp(1)
        ref = fs.results(0); fs.close()
        assert np.array_equal(flo.flow_read(str(tmp_path / ("f%d.flo" % k))), ref["flow"])
        assert np.array_equal(np.array(Image.open(tmp_path / ("w%d.png" % k))), ref["warped_rgb"])
        wm = Image.open(tmp_path / ("wm%d.png" % k))
        assert wm.mode == "1" and np.array_equal(np.array(wm), ref["warped_mask"] > 0)
    # para_gen with the C++ executable as --arap_bin
    inp, outp, mdir = tmp_path / "in", tmp_path / "out", tmp_path / "matches"
    os.makedirs(inp / "orgRGB" / "s"); os.makedirs(inp / "orgMasks" / "s"); os.makedirs(mdir / "s")
    fr = synth.make_frame(96, 64, seed=11, K=2)
    for n in range(2):
        Image.fromarray(fr["rgb"]).save(inp / "orgRGB" / "s" / ("%05d.png" % n))
        Image.fromarray(fr["labels"].astype(np.uint8)).save(inp / "orgMasks" / "s" / ("%05d.png" % n))
        (mdir / "s" / ("%05d.txt" % n)).write_text("\n".join("%d %d %d %d 1.0 0" % tuple(c) for c in fr["constraints"]))
    _run([osp.join(ROOT, "para_gen.py"), "--input", str(inp), "--output", str(outp), "--gpu", "0", "--matches", str(mdir),
          "--multseg", "--arap_bin", b["arap_deform"]], str(tmp_path))
    lst = open(outp / "all_files.list").read().splitlines()
    assert len(lst) == 1 and np.abs(flo.flow_read(lst[0].split(" ")[2])).max() > 0.5


def test_bench_contract_json_line(tmp_path):
    """bench.py prints ONE JSON line with the contract's keys (tiny schedule and frames here)"""
    import json
    out = _run([osp.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2", "--size", "214", "120",
                "--schedule", "2", "2", "30"], str(tmp_path))
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "frames/s" and d["value"] > 0 and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "valu_issue", "lds", "group_wait") and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    for k in ("hbm_frac_by_counters", "valu_issue_frac", "wait_frac"):     # null unless a matching profiles/ record exists
        assert k in r, k
    assert d["parity_check"]["bit_equal"] is True                          # the timed work is the verified work
    assert d["per_rank_ms_per_step"] and d["world_size_seen"] == 1 and d["gpus_flag"] == 1
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1


def test_bench_two_ranks_under_torch_distributed_run(tmp_path):
    """The driver's N > 1 command line, rehearsed on one GPU: two ranks launched by torch.distributed.run share the
    card (ARAP_BENCH_BACKEND=gloo moves only the timing barrier and the max-over-ranks off RCCL, which wants one GPU
    per rank).  Rank 0 prints the one JSON line for the whole job; the CPU baseline is an N = 1 figure."""
    import json
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, ARAP_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), osp.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--batch", "2", "--size", "214", "120", "--schedule", "2", "2", "30"]
    r = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["frames_per_gpu_per_step"] == 2
    assert abs(d["value"] - 2 * 2 * 1 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]      # all ranks' frames / max time
    assert "roofline" in d and "cpu_baseline" not in d
    assert len(d["per_rank_ms_per_step"]) == 2 and d["world_size_seen"] == 2 and d["gpus_flag"] == 2


def test_bench_gpus_2_without_a_launcher_starts_two_ranks(tmp_path):
    """`python bench.py --gpus 2` as the driver types it for N = 1, with N = 2 and no torch.distributed.run around it:
    bench.py starts the two ranks itself (fresh child processes, before torch / HIP is touched).  Rehearsed on one GPU
    (gloo for the timing barrier; the ranks share the card)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ARAP_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, osp.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "2",
                        "--size", "214", "120", "--schedule", "2", "2", "30", "--no-kernel-timing"],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size_seen"] == 2 and d["gpus_flag"] == 2 and d["value"] > 0
    assert len(d["per_rank_ms_per_step"]) == 2


def test_resident_kernel_without_the_xcd_fast_paths(tmp_path):
    """ARAPOPT_NO_XCD_FAST=1 makes the resident kernel publish everything write-through, as it must when the workgroups
    of a group (or of an XCD run of a wide group) do not share an XCD.  Placement is the hardware's choice, so this
    flavour is exercised explicitly: a group on one XCD and a group that spans XCDs both give the oracle's bits."""
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from arap_flow_amd import opt, synth\n"
        "from oracle import oracle as orc\n"
        "st = opt.State()\n"
        "for (W, H, full) in ((160, 96, False), (640, 360, True), (854, 480, True)):\n"      # 1 XCD, 2 XCDs (one-hop sums), 4 XCDs (two-level sums)
        "    f = synth.make_frame(W, H, seed=5, full_mask=full)\n"
        "    fs = opt.FrameSolver(st, W, H, batch=1); fs.set_frame(0, f['mask_red'], f['constraints'])\n"
        "    fs.solve(1, 1, 2, 20); r = fs.results(0, want_rgb=False)\n"
        "    assert fs.stats()['resident_launches'] > 0\n"
        "    O, A, c = orc.frame(f['mask_red'], f['constraints'], numIter=1, nIterations=2, lIterations=20, dtype=np.float32, mode=1, trig=1)\n"
        "    assert np.array_equal(r['offset'], O) and np.array_equal(r['angle'], A), (W, H)\n"
        "    fs.close()\n"
        "print('write-through ok')\n" % ROOT)
    env = dict(os.environ, ARAPOPT_NO_XCD_FAST="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "write-through ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_two_solver_objects_do_not_wait_for_each_other(tmp_path):
    """The persistent worker alternates two solver objects on one state (own compute stream): while one solves, the other
    is filled, enqueued and, later, waited for.  Neither enqueueing B nor waiting for A may wait for the other object's
    solve (they did, through synchronisations of the shared stream: the worker then ran strictly serially, 298 instead of
    286 ms per batch).  Own process: ArapFlow_UseOwnStream changes the state."""
    code = (
        "import sys, time, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from arap_flow_amd import opt, synth\n"
        "st = opt.State(); st.use_own_stream()\n"
        "W, H = 854, 480\n"
        "lanes = [opt.FrameSolver(st, W, H, batch=8) for _ in range(2)]\n"
        "fr = [synth.make_frame(W, H, seed=s) for s in range(32)]\n"
        "worst = []\n"
        "for rnd in range(2):\n"
        "    for k in range(2):\n"
        "        for b in range(8):\n"
        "            f = fr[16 * rnd + 8 * k + b]; lanes[k].set_frame(b, f['mask_red'], f['constraints'], rgb=f['rgb'])\n"
        "    t0 = time.perf_counter(); lanes[0].solve_async(8, 4, 8, 400)\n"
        "    t1 = time.perf_counter(); lanes[1].solve_async(8, 4, 8, 400)\n"
        "    t2 = time.perf_counter(); lanes[0].wait()\n"
        "    t3 = time.perf_counter(); lanes[1].wait()\n"
        "    t4 = time.perf_counter()\n"
        "    total = t4 - t0\n"
        "    print('enqueue A %%.1f  enqueue B %%.1f  wait A %%.1f  wait B %%.1f ms' %% tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3)))\n"
        "    assert (t2 - t1) < 0.25 * total, 'enqueueing B waited for A'\n"
        "    assert (t3 - t2) < 0.70 * total and (t4 - t3) > 0.30 * total, 'waiting for A waited for B as well'\n"
        "    assert st.lib.ArapFlow_ResidentFailed(st.handle) == 0\n"
        "print('overlap ok')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "overlap ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_resident_failure_falls_back_to_two_kernel_path(tmp_path):
    """ARAPOPT_FORCE_RES_FAIL=1 makes every resident launch report a timed-out group wait (what happens when the GPU
    is shared and the 512 workgroups are not co-resident).  The frame solver and the drop-in path must notice, redo the
    work on the two-kernel path and still return the oracle's bits."""
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import helpers\n"
        "from arap_flow_amd import opt, synth\n"
        "from oracle import oracle as orc\n"
        "st = opt.State()\n"
        "f = synth.make_frame(160, 96, seed=3)\n"
        "fs = opt.FrameSolver(st, 160, 96, batch=1); fs.set_frame(0, f['mask_red'], f['constraints'])\n"
        "fs.solve(1, 2, 2, 30); r = fs.results(0, want_rgb=False)\n"
        "O, A, c = orc.frame(f['mask_red'], f['constraints'], numIter=2, nIterations=2, lIterations=30, dtype=np.float32, mode=1, trig=1)\n"
        "assert np.array_equal(r['offset'], O) and np.array_equal(r['angle'], A)\n"
        "assert st.lib.ArapFlow_ResidentFailed(st.handle) == 1 and fs.stats()['resident_launches'] > 0\n"
        "fs.close(); st.close()\n"
        "st = opt.State()\n"
        "pb = helpers.random_problem(90, 60, seed=4, generic_urshape=False, ncons=30)\n"
        "dev = {k: torch.from_numpy(pb[k].copy()).cuda() for k in 'OAUCM'}\n"
        "s = opt.OptSolver(st, (90, 60)); pp = opt.NamedParameters()\n"
        "[pp.set(n, dev[k]) for n, k in [('Offset','O'),('Angle','A'),('UrShape','U'),('Constraints','C'),('Mask','M')]]\n"
        "pp.set('w_fitSqrt', 10.0); pp.set('w_regSqrt', 0.1); sp = opt.NamedParameters(); sp.set('nIterations', 2); sp.set('lIterations', 20)\n"
        "cost = s.solve(sp, pp)\n"
        "Or, Ar, cs = orc.solve(pb['O'], pb['A'], pb['U'], pb['C'], pb['M'], 10.0, 0.1, 2, 20, dtype=np.float32, mode=1, trig=1)\n"
        "assert np.array_equal(dev['O'].cpu().numpy(), Or) and cost == cs[-1]\n"
        "assert st.lib.ArapFlow_ResidentFailed(st.handle) == 1\n"
        "print('fallback ok')\n" % (ROOT, osp.join(ROOT, "tests")))
    env = dict(os.environ, ARAPOPT_FORCE_RES_FAIL="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fallback ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    assert "Falling back to the two-kernel path" in r.stderr


def test_real_group_wait_timeout_and_recovery(tmp_path):
    """ARAPOPT_FORCE_RES_FAIL=2 leaves ONE workgroup of the first group out of the first resident launch: the rest of
    that group spins in a real group wait until the bounded spin gives up (arap_resident.h: group_sum) and sets the
    error word.  The solver redoes the schedule on the two-kernel path (oracle's bits), pauses the resident path for a
    few solve calls and then takes it again (the failure is not sticky)."""
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
        "from arap_flow_amd import opt, synth\n"
        "from oracle import oracle as orc\n"
        "st = opt.State()\n"
        "f = synth.make_frame(320, 200, seed=3)\n"
        "fs = opt.FrameSolver(st, 320, 200, batch=2)\n"
        "fs.set_frame(0, f['mask_red'], f['constraints']); fs.set_frame(1, f['mask_red'], f['constraints'])\n"
        "O, A, c = orc.frame(f['mask_red'], f['constraints'], numIter=2, nIterations=2, lIterations=30, dtype=np.float32, mode=1, trig=1)\n"
        "fs.solve(2, 2, 2, 30); r = fs.results(0, want_rgb=False)\n"
        "assert np.array_equal(r['offset'], O) and np.array_equal(r['angle'], A)\n"
        "assert st.lib.ArapFlow_ResidentFailed(st.handle) == 1\n"
        "n0 = fs.stats()['resident_launches']\n"
        "for k in range(8):\n"                      # the pause: two-kernel path, no resident launches
        "    fs.solve(2, 2, 2, 30)\n"
        "assert fs.stats()['resident_launches'] == n0, (n0, fs.stats())\n"
        "fs.solve(2, 2, 2, 30); r = fs.results(1, want_rgb=False)\n"     # pause over: resident again, same bits
        "assert fs.stats()['resident_launches'] > n0\n"
        "assert np.array_equal(r['offset'], O) and np.array_equal(r['angle'], A)\n"
        "print('timeout ok')\n" % ROOT)
    env = dict(os.environ, ARAPOPT_FORCE_RES_FAIL="2")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "timeout ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    assert r.stderr.count("gave up at a group wait") == 1



def test_drop_in_steps_during_the_pause_after_a_timeout_see_urshape_change(tmp_path):
    """ARAPOPT_FORCE_RES_FAIL=2: the first resident launch of a drop-in plan times out for real, the Step is redone on
    the two-kernel path and the resident path pauses.  During the pause every Step must still look at UrShape: it
    turns generic in place before the third Step, and the result must be the oracle's."""
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import test_gpu_solve as T\n"
        "from arap_flow_amd import opt\n"
        "from oracle import oracle as orc\n"
        "st = opt.State()\n"
        "T._stepwise_with_urshape_turning_generic(st, orc, expect_resident=True)\n"
        "assert st.lib.ArapFlow_ResidentFailed(st.handle) == 1\n"
        "print('pause ok')\n" % (ROOT, osp.join(ROOT, "tests")))
    env = dict(os.environ, ARAPOPT_FORCE_RES_FAIL="2")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "pause ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    assert r.stderr.count("gave up at a group wait") == 1


def test_para_gen_hands_the_gpu_full_batches(tmp_path):
    """The real CLI must feed the kernel what bench.py measures: through `para_gen.py --matches` with the persistent
    C++ worker, 854x480 DAVIS-shaped pairs reach the GPU in batches of 8 (one resident launch); the mean batch over
    the run must be >= 6 (the tail batch may be short).  The reference's own loop would hand out 1-2 frames per child
    process (para_gen.py:560-567)."""
    import json
    _bins()
    W, H, pairs = 854, 480, 28
    inp, outp, mdir = tmp_path / "in", tmp_path / "out", tmp_path / "matches"
    for n in range(pairs):
        seq = "s%03d" % n
        os.makedirs(inp / "orgRGB" / seq); os.makedirs(inp / "orgMasks" / seq); os.makedirs(mdir / seq)
        fr = synth.make_frame(W, H, seed=n, K=1, fd=1)
        for k in range(2):
            Image.fromarray(fr["rgb"]).save(inp / "orgRGB" / seq / ("%05d.png" % k))
            Image.fromarray(fr["labels"].astype(np.uint8)).save(inp / "orgMasks" / seq / ("%05d.png" % k))
        (mdir / seq / "00000.txt").write_text("\n".join("%d %d %d %d 1.0 0" % tuple(c) for c in fr["constraints"]))
    _run([osp.join(ROOT, "para_gen.py"), "--input", str(inp), "--output", str(outp), "--gpu", "0", "--matches", str(mdir)],
         str(tmp_path))
    st = json.load(open(outp / "arap_stats.json"))
    assert st["worker"] == "serve" and st["frames"] == pairs and sum(st["batches"]) == pairs
    assert st["mean_batch"] >= 6.0, st
    lst = open(outp / "all_files.list").read().splitlines()
    assert len(lst) == pairs
    # spot check one frame against the library path: same flow as a direct solve of the files para_gen wrote
    rgb1, rgb2, fl = lst[5].split(" ")
    f = flo.flow_read(fl)
    assert f.shape == (H, W, 2) and np.abs(f).max() > 0.5
