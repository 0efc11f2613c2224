#!/usr/bin/env python3
"""bench.py -- ARAP solve+warp throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N > 1: either the driver launches the N ranks (python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...)
or, when no launcher environment is present, bench.py starts them itself as fresh child processes before it touches
torch or HIP and relays rank 0's line; a WORLD_SIZE that differs from --gpus is refused (exit 2).

A "step" = one pass of the hot path over one batch of synthetic frames on every rank: the full
arap_deform schedule (19 ramp steps x 8 Gauss-Newton steps x 400 PCG iterations,
ARAP/deformation/src/main.cpp:215-221) for `--batch` independent 854x480 frames, then flow emission
and the forward rasterisation of RGB and mask.  Inputs (RGB, mask, constraints) are uploaded before
the timed region.  Workload = BASELINE.json configs[1]: single-segment DAVIS-shaped mask, fd=1.
Frames shard across ranks with no collective: weak scaling, value = all frames / max time.

One JSON line on rank 0: metric/value/... plus
  roofline     : dominant PCG kernel.  achieved / peak / frac = SURVEY 8(d)'s figure: algorithmic bytes per launch /
                 average launch duration measured with HIP events around every launch (a separate, un-graphed pass)
                 against the 8 TB/s HBM peak.  The resident kernel keeps the PCG state on chip, so that figure exceeds 1
                 and does NOT bind; `bound` names the largest MEASURED fraction, and the line carries them all:
                 hbm_frac_by_counters (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE per launch / launch time / peak),
                 valu_issue_frac (SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x launch cycles by GRBM_GUI_ACTIVE)) and
                 valu_frac_of_ceiling (against the ceiling the phase-A instruction mix reaches in
                 tools/microbench/issue_peak.hip), lds_frac (SQ_LDS_IDX_ACTIVE per CU-cycle against a saturated
                 ds_read_b64 stream), wait_frac (share of an iteration inside the two group waits, instrumented
                 build).  They come from
                 profiles/*_counters.json (tools/collect_profile.py) and are quoted only when that record was made for
                 this workload AND for the kernel sources as they are now (else null + the reason).
  cpu_baseline : the CPU oracle (kind "port") timed on this host on a bounded sample
  parity_check : frame 0 (its first segment with --multseg) of the timed steps, Offset and Angle, against the oracle's
                 result from the cpu_baseline leg (same schedule): bit_equal true / false (outside the timed region)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

import hashlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_A, BYTES_B = 64, 96        # algorithmic bytes per vertex per PCG iteration: 16 + 24 floats


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=0,
                    help="frames solved concurrently per GPU per step; 0 = 8, or with --multseg the count <= 16 that "
                         "fills the resident launches best (probed with the frames of seeds 0, 1, ...)")
    ap.add_argument("--size", type=int, nargs=2, default=[854, 480], metavar=("W", "H"))
    ap.add_argument("--schedule", type=int, nargs=3, default=[19, 8, 400], metavar=("NUMITER", "NITER", "LITER"))
    ap.add_argument("--workload", choices=["davis", "full"], default="davis",
                    help="davis: configs[1] single-segment mask (~25%% active); full: mask == 0 (roofline config)")
    ap.add_argument("--multseg", type=int, default=0, metavar="K",
                    help="configs[2]/[4]: K segments per frame, each its own ARAP solve (para_gen.py --multseg); "
                         "--batch then counts frames, K x batch solves run per step")
    ap.add_argument("--fd", type=int, default=1, help="frame distance of the synthetic matches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--stamps", action="store_true",
                    help="diagnostic (needs ARAPOPT_STAMPS=1): print the per-phase times of the resident kernel's last launch")
    return ap.parse_args()


def _field_hash(offset, angle):
    return hashlib.sha256(np.ascontiguousarray(offset, np.float32).tobytes() +
                          np.ascontiguousarray(angle, np.float32).tobytes()).hexdigest()


def _time_oracle(spec, threads, numIter, nIter, lIter, first_only=False):
    """wall time of the CPU oracle on the solves of ONE frame of the workload, in a FRESH process (OMP_NUM_THREADS
    must be set before the OpenMP runtime starts; this process already runs torch's).  Returns (seconds, sha256 of the
    first solve's Offset + Angle)."""
    import subprocess
    code = ("import sys,time,json,hashlib,numpy as np;sys.path.insert(0,%r);from oracle import oracle as orc;"
            "from arap_flow_amd import synth;f=synth.make_frame(%d,%d,seed=%d,K=%d,fd=%d,full_mask=%r);"
            "sv=synth.segment_masks(f) if %r else [f];sv=sv[:1] if %r else sv;t=time.time();h=None\n"
            "for g in sv:\n"
            "    O,A,c=orc.frame(g['mask_red'],g['constraints'],numIter=%d,nIterations=%d,lIterations=%d,dtype=np.float32,mode=1,trig=1)\n"
            "    h=h or hashlib.sha256(np.ascontiguousarray(O,np.float32).tobytes()+np.ascontiguousarray(A,np.float32).tobytes()).hexdigest()\n"
            "print(json.dumps([time.time()-t,h,len(sv)]))"
            % (ROOT, spec["W"], spec["H"], spec["seed"], spec["K"], spec["fd"], spec["full"], spec["multseg"], first_only,
               numIter, nIter, lIter))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_WAIT_POLICY="active"),
                       capture_output=True, text=True, timeout=1500)
    dt, h, n = json.loads(r.stdout.strip().splitlines()[-1])
    return float(dt), h, int(n)


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(spec, schedule):
    """The CPU oracle (same algorithm, float32, OpenMP over rows; kind "port") on a bounded sample of the same
    workload -- one frame, all its segment solves -- on this box's CPU share (a 1-GPU box grants 16 of the host's
    cores).  Only this leg of the bench touches oracle/.  Also returns the hash of the first solve's result for the
    parity check."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, int(os.environ.get("ARAP_CPU_BASELINE_THREADS", "16")))
    numIter, nIter, lIter = schedule
    # bound the sample to ~30 s: the full schedule of a 854x480 DAVIS-shaped frame is ~15 s on 16 cores; larger or
    # fuller frames get fewer ramp steps (the cost per ramp step is constant), scaled
    work = spec["active"] / 102480.0 * (numIter * nIter * lIter) / 60800.0
    ns = numIter if work <= 2.2 else max(1, int(numIter * 2.2 / work))
    dt, h, nsolves = _time_oracle(spec, cores, ns, nIter, lIter)
    fps = 1.0 / (dt * numIter / ns)
    out = {"value": fps, "unit": "frames/s", "cores": cores, "kind": "port", "cpu": _cpu_model(),
           "sample": "%d of %d ramp steps (x %d GN x %d PCG iterations) of one frame (%d solve%s) of the same workload, %.1f s"
                     % (ns, numIter, nIter, lIter, nsolves, "" if nsolves == 1 else "s", dt),
           "pcg_iters_per_s": ns * nIter * lIter * nsolves / dt}
    try:                                                      # the same code on ONE thread (SURVEY 8d asks for both)
        n1 = min(nIter, 8)
        scale = max(1.0, spec["active"] / 102480.0)
        n1 = max(1, int(n1 / scale))
        t1, _, _ = _time_oracle(spec, 1, 1, n1, lIter, first_only=True)
        out["single_thread"] = {"value": 1.0 / (t1 * nsolves * numIter * nIter / n1), "unit": "frames/s", "cores": 1,
                                "sample": "1 ramp step x %d GN x %d PCG iterations of one solve, %.1f s, scaled" % (n1, lIter, t1)}
    except Exception as e:                                    # never fail the bench over the extra figure
        out["single_thread"] = {"error": str(e)[:200]}
    return out, (h if ns == numIter else None)


def launch_ranks(a):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed.run environment: start the N ranks as FRESH child
    processes (this process has not touched HIP or torch yet and never will), let rank 0's JSON line through on the
    inherited stdout, and return the launcher's exit code.  One process per GPU, as para_gen.py --gpu 0 .. N-1 does
    (/root/reference/para_gen.py:441-445,560-567,190)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # RCCL on this pool: dmabuf IPC only
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def _stub_main(a, rank, world):
    """ARAP_BENCH_STUB=1 (tests, no GPU): the launcher, the rendezvous, the barrier-bracketed timing, the max over ranks
    and the one JSON line of rank 0 with a sleep in place of the solve.  Nothing here is a measurement."""
    import torch
    import torch.distributed as dist
    from arap_flow_amd import shard
    if world > 1:
        dist.init_process_group(os.environ.get("ARAP_BENCH_BACKEND", "gloo"))
    B = a.batch if a.batch > 0 else 8
    seeds = shard.shard_indices(world * B, rank, world)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(a.warmup):
        time.sleep(0.001)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        time.sleep(0.01 * (rank + 1))
    barrier()
    dt = shard.max_over_ranks(time.perf_counter() - t0, dist if world > 1 else None)
    if rank == 0:
        print(json.dumps({"metric": "ARAP solve+warp frames/sec at 854x480 mesh", "value": world * B * a.steps / dt,
                          "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "STUB (ARAP_BENCH_STUB=1: launcher test, no solve)",
                          "config": {"workload": "stub", "frames_per_gpu_per_step": B, "frames_of_rank0": len(seeds)},
                          "world_size_seen": dist.get_world_size() if world > 1 else 1, "gpus_flag": a.gpus}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse()
    # --gpus N is the contract's way to ask for N ranks.  Before torch or HIP is touched: without a
    # torch.distributed.run environment start the ranks ourselves; with one, refuse a mismatch (a line that says
    # n_gpus = WORLD_SIZE while the caller asked for another N would be a scaling point that measures something else).
    if a.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if a.gpus > 1:
            sys.exit(launch_ranks(a))
    elif int(os.environ["WORLD_SIZE"]) != a.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s: launch with --nproc-per-node %d (or drop the "
                         "launcher: bench.py starts its own ranks)\n" % (a.gpus, os.environ["WORLD_SIZE"], a.gpus))
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("ARAP_BENCH_STUB") == "1":
        return _stub_main(a, rank, world)
    import torch
    # one process per GPU.  (ARAP_BENCH_BACKEND=gloo is a rehearsal switch for boxes with fewer GPUs than ranks:
    # the ranks then share the visible GPUs and only the timing barrier / max-over-ranks change transport.)
    backend = os.environ.get("ARAP_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    from arap_flow_amd import opt, shard, synth
    from tools import profile_key

    W, H = a.size
    numIter, nIter, lIter = a.schedule
    st = opt.State()
    K = max(1, a.multseg)
    B = a.batch if a.batch > 0 else 8
    if a.batch <= 0 and a.multseg:
        # every solve gets a group of the resident launch's workgroups sized by its active tiles and a launch costs
        # the same however full it is: take the frame count with the most frames per launch (same probe on every rank)
        probe = opt.FrameSolver(st, W, H, batch=16 * K)
        best, n = (0.0, 1), 0
        for sd in range(16):
            for sg in synth.segment_masks(synth.make_frame(W, H, seed=sd, K=K, fd=a.fd)):
                probe.set_frame(n, sg["mask_red"], sg["constraints"])
                n += 1
            launches = probe.launches_for(n)
            if launches <= 0:
                break
            if (sd + 1) / launches > best[0] + 1e-9:
                best = ((sd + 1) / launches, sd + 1)             # most frames per launch, fewest frames on a tie
        probe.close()
        B = best[1]
    S = B * K                                    # solves per step on this rank
    fs = opt.FrameSolver(st, W, H, batch=S)
    # the job's frame list (world x B frames per step) is dealt round-robin to the ranks: no collective
    seeds = shard.shard_indices(world * B, rank, world)
    frames = [dict(synth.make_frame(W, H, seed=sd, K=K, fd=a.fd, full_mask=(a.workload == "full")), seed=sd)
              for sd in seeds]
    solves = [sg for f in frames for sg in synth.segment_masks(f)] if a.multseg else frames
    S = len(solves)
    for b, f in enumerate(solves):
        fs.set_frame(b, f["mask_red"], f["constraints"], rgb=f["rgb"])
    torch.cuda.synchronize()

    def step():
        fs.solve(S, numIter, nIter, lIter)
        fs.warp(S)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    st.timer_begin()
    for _ in range(a.steps):
        step()
    ev_ms = st.timer_end()
    t_local = time.perf_counter() - t0              # this rank's own time for the K steps (its frames only)
    barrier()
    dt = time.perf_counter() - t0
    dev = "cuda" if backend == "nccl" else "cpu"
    dt = shard.max_over_ranks(dt, dist, device=dev)
    per_rank_ms = [1e3 * t_local / a.steps]
    seen_world = 1
    if dist is not None:
        t = torch.tensor([1e3 * t_local / a.steps], dtype=torch.float64, device=dev)
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        per_rank_ms = [float(g.item()) for g in got]
        seen_world = dist.get_world_size()
    stats = fs.stats()
    total_frames = world * B * a.steps
    fps = total_frames / dt
    pcg_per_frame = stats["pcg_iterations_per_frame"]
    n_active = stats["active_vertices"] / B          # per frame (all its segments)
    n_grid = W * H
    # result of the timed steps, frame 0 (its first segment), for the parity check below: taken NOW, before the
    # kernel-timing pass re-solves with another schedule
    parity_hash = None
    if rank == 0 and not a.no_cpu_baseline and world == 1:
        r_timed = fs.results(0, want_rgb=False)
        parity_hash = _field_hash(r_timed["offset"], r_timed["angle"])
        del r_timed

    sig = profile_key.signature(a.workload, W, H, S, K if a.multseg else 1, a.fd, B)
    out = {
        "metric": "ARAP solve+warp frames/sec at 854x480 mesh",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("configs[1]: single %dx%d DAVIS-shaped frame, single-segment mask, fd=%d; "
                                "schedule %d/%d/%d" % (W, H, a.fd, numIter, nIter, lIter) if not a.multseg else
                                "%dx%d --multseg, %d per-segment ARAP solves per frame batched on one GPU, fd=%d; "
                                "schedule %d/%d/%d" % (W, H, K, a.fd, numIter, nIter, lIter))
                   if a.workload == "davis" else
                   "roofline config: %dx%d, mask == 0 (all vertices active), schedule %d/%d/%d"
                   % (W, H, numIter, nIter, lIter),
                   "frames_per_gpu_per_step": B, "segments_per_frame": K, "fd": a.fd,
                   "parallelism": "frames sharded, no collective",
                   "active_vertices_per_frame": n_active, "grid_vertices_per_frame": n_grid},
        "pcg_iters_per_s": fps * pcg_per_frame,
        "hip_event_ms_per_step_rank0": ev_ms / a.steps,
        # multi-GPU bookkeeping: every rank's own time per step (its frames only, before the closing barrier) and the
        # world size torch.distributed (RCCL) reported -- the scaling curve is computed by the driver from `value`
        "per_rank_ms_per_step": per_rank_ms, "world_size_seen": seen_world, "gpus_flag": a.gpus, "backend": backend if world > 1 else None,
        "profile_signature": sig,
    }

    if rank == 0:
        # ---- roofline of the dominant kernel: HIP events around every launch (un-graphed pass) ----------
        r0 = None
        if not a.no_kernel_timing:
            tl = min(lIter, 400)
            G = 8                                      # Gauss-Newton steps timed: 2 solves x 4 steps of tl PCG iterations
            st.set_kernel_timing(True)
            fs.solve(S, 1, 1, tl)                      # (one un-graphed step first: the records of a cold launch are dropped)
            torch.cuda.synchronize()
            st.set_kernel_timing(True)                 # clears the records
            for _ in range(2):
                fs.solve(S, 1, 4, tl)
            torch.cuda.synchronize()
            kt = {k: st.kernel_time(k) for k in ("PCGResident", "PCGStepA", "PCGStepB")}
            st.set_kernel_timing(False)
            n_act_total = stats["active_vertices"]     # all B frames
            prof = profile_key.find_counters(sig)
            if kt["PCGResident"] is not None:
                tot_ms, n = kt["PCGResident"]
                # one launch = all `tl` PCG iterations of one GN step for the frames in flight;
                # algorithmic bytes: 160 B per active vertex per PCG iteration
                bytes_total = 160.0 * n_act_total * tl * G
                ach = bytes_total / (tot_ms * 1e-3) / 1e9
                frames_per_launch = S * float(G) / n
                rl = {
                    "bound": "group_wait", "kernel": "k_pcg_resident", "achieved": ach, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None, "equivalent_GBs": ach,
                    "avg_launch_us": tot_ms / n * 1e3, "launches": n,
                    "pcg_iterations_per_launch": tl, "frames_per_launch": frames_per_launch,
                    "us_per_pcg_iteration": tot_ms / n * 1e3 / tl,
                    "achieved_vs_grid_vertices": ach * n_grid / n_active, "frac_vs_grid_vertices": ach * n_grid / n_active / HBM_PEAK_GBS,
                    "note": "achieved/frac = SURVEY 8(d): 160 B x active vertices x PCG iterations of the launch / launch "
                            "time, against the HBM peak.  The kernel keeps the PCG state in registers/LDS, so that "
                            "ceiling does not bind (frac > 1); what binds is VALU issue during the phases and the latency "
                            "of the two group-wide sums per iteration: hbm_frac_by_counters, valu_issue_frac, wait_frac"}
                kname = "k_pcg_resident"
            else:
                per = {}
                # algorithmic bytes per vertex of each phase: SURVEY 8(d)'s split (64 + 96); the lean streaming schedule
                # moves z's and one p's traffic away and delta into phase A: 85 + 41 (arap_stream.h)
                lean = bool(fs.stats().get("lean_stream"))
                bytes_ab = (85, 41) if lean else (BYTES_A, BYTES_B)
                for k, bytes_v in (("PCGStepA", bytes_ab[0]), ("PCGStepB", bytes_ab[1])):
                    tot_ms, n = kt[k]
                    avg_s = tot_ms / n * 1e-3
                    per[k] = {"avg_us": avg_s * 1e6, "launches": n, "GBs_active": bytes_v * n_act_total / avg_s / 1e9,
                              "GBs_grid": bytes_v * n_grid * B / avg_s / 1e9}
                dom = max(per, key=lambda k: per[k]["avg_us"])
                kname = {"PCGStepA": "k_pcg_a", "PCGStepB": "k_pcg_b"}[dom]
                t_it = (per["PCGStepA"]["avg_us"] + per["PCGStepB"]["avg_us"]) * 1e-6
                iteration = {"us": t_it * 1e6, "GBs_survey_160B": 160.0 * n_act_total / t_it / 1e9,
                             "frac_survey_160B": 160.0 * n_act_total / t_it / 1e9 / HBM_PEAK_GBS,
                             "schedule": "lean (k_pcg_a_march2 + k_pcg_b4_r)" if lean else "k_pcg_a_march + k_pcg_b4_lean"}
                rl = {"bound": "hbm", "kernel": kname, "iteration": iteration,
                      "achieved": per[dom]["GBs_active"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": per[dom]["GBs_active"] / HBM_PEAK_GBS, "traffic": None,
                      "avg_launch_us": per[dom]["avg_us"], "achieved_vs_grid_vertices": per[dom]["GBs_grid"],
                      "frac_vs_grid_vertices": per[dom]["GBs_grid"] / HBM_PEAK_GBS, "per_kernel": per,
                      "note": "algorithmic bytes = %d (A) / %d (B) per active vertex per launch; `iteration` = SURVEY 8(d)'s "
                              "160 B per vertex over the two launches of one PCG iteration" % bytes_ab}
            # counters of the same workload from rocprofv3 passes (collected separately: tools/collect_profile.py)
            rl["hbm_frac_by_counters"] = rl["valu_issue_frac"] = rl["wait_frac"] = None
            rl["valu_frac_of_ceiling"] = rl["lds_frac"] = None
            if prof is None:
                rl["counters_source"] = None
                rl["counters_note"] = ("no profiles/*_counters.json matches this workload and the current sources of %s "
                                       "(hash %s): traffic and utilisation figures withheld"
                                       % (kname, profile_key.source_hash(kname if kname == "k_pcg_resident" else kname[:7])))
            else:
                pf, rec = prof
                rl["counters_source"] = pf
                pm = rec.get("pmc", {}).get(kname, {})
                hb = pm.get("hbm_bytes_per_launch")
                pmc_ns = rec.get("kernel_stats", {}).get(kname, {}).get("avg_ns")
                launch_s = rl["avg_launch_us"] * 1e-6
                if hb:
                    rl["traffic"] = hb["total"]
                    rl["traffic_unit"] = "bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes)"
                    rl["traffic_detail"] = hb
                    rl["hbm_frac_by_counters"] = hb["total"] / launch_s / 1e9 / HBM_PEAK_GBS
                # shader cycles of a launch: GRBM_GUI_ACTIVE is summed over the 8 XCDs (the clock the chip really held
                # under this kernel); nominal 2.4 GHz where that pass is missing
                cyc = pm["GRBM_GUI_ACTIVE"]["avg_per_launch"] / 8.0 if "GRBM_GUI_ACTIVE" in pm else launch_s * 2.4e9
                rl["launch_cycles"] = cyc
                rl["clock_GHz"] = cyc / (pmc_ns * 1e-9) / 1e9 if pmc_ns else None
                ceil = profile_key.ceilings()
                cd = ceil[1] if ceil else {}
                rl["ceilings_source"] = ceil[0] if ceil else None
                if "SQ_ACTIVE_INST_VALU" in pm:
                    # SQ_ACTIVE_INST_VALU counts one quad-cycle per vector instruction (= SQ_INSTS_VALU): x 4 cycles over the
                    # cycles of the 1024 SIMDs.  Its CEILING depends on the instruction mix (tools/microbench/issue_peak.hip, two
                    # wavefronts per SIMD: plain v_fma_f32 1.55, v_pk_fma_f32 0.90, the phase-A mix 0.98): the kernel is
                    # quoted against the phase-A mix
                    rl["valu_issue_frac"] = pm["SQ_ACTIVE_INST_VALU"]["avg_per_launch"] * 4.0 / (1024.0 * cyc)
                    vc = cd.get("derived", {}).get("mix", {}).get("2_per_cu", {}).get("valu_active_x4_per_simd_cycle")
                    if vc:
                        rl["valu_ceiling"] = vc
                        rl["valu_frac_of_ceiling"] = rl["valu_issue_frac"] / vc
                if "SQ_LDS_IDX_ACTIVE" in pm:
                    # cycles the LDS arrays of the 256 CUs were busy; ceiling = a saturated ds_read_b64 stream (0.89)
                    lc = cd.get("lds_idx_active_ceiling")
                    rl["lds_idx_active_frac"] = pm["SQ_LDS_IDX_ACTIVE"]["avg_per_launch"] / (256.0 * cyc)
                    if lc:
                        rl["lds_ceiling"] = lc
                        rl["lds_frac"] = rl["lds_idx_active_frac"] / lc
                    if "SQ_LDS_BANK_CONFLICT" in pm and pm["SQ_LDS_IDX_ACTIVE"]["avg_per_launch"] > 0:
                        rl["lds_bank_conflict_share"] = pm["SQ_LDS_BANK_CONFLICT"]["avg_per_launch"] / pm["SQ_LDS_IDX_ACTIVE"]["avg_per_launch"]
                if rec.get("stamps"):
                    rl["wait_frac"] = rec["stamps"].get("wait_frac")
                    rl["stamps_us_per_iteration"] = rec["stamps"].get("us")
                if pmc_ns:
                    rl["rocprof_avg_launch_us"] = pmc_ns * 1e-3
                # what binds = the largest measured fraction; inside the phases (the share of the time that is not group
                # waits) the pipes are that much busier
                if kname == "k_pcg_resident":
                    fr = {"hbm": rl["hbm_frac_by_counters"], "valu_issue": rl["valu_frac_of_ceiling"], "lds": rl["lds_frac"],
                          "group_wait": rl["wait_frac"]}
                    fr = {k: v for k, v in fr.items() if v is not None}
                    if fr:
                        rl["bound"] = max(fr, key=fr.get)
                        rl["measured_fractions"] = fr
                    if rl["wait_frac"] is not None and rl["wait_frac"] < 1.0:
                        for k in ("valu_frac_of_ceiling", "lds_frac"):
                            if rl.get(k) is not None:
                                rl[k + "_inside_phases"] = rl[k] / (1.0 - rl["wait_frac"])
            out["roofline"] = rl
            out["resident_path"] = stats.get("resident_launches", 0) > 0
            # measured device copy bandwidth next to the nominal peak: 1 GiB float32 copy, read + write
            try:
                src = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()
                dst = torch.empty_like(src)
                dst.copy_(src); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    dst.copy_(src)
                e1.record(); torch.cuda.synchronize()
                out["roofline"]["measured_copy_GBs"] = 5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
                del src, dst
            except Exception as e:
                out["roofline"]["measured_copy_GBs"] = None
            # warp stage alone: fused flow emission + rasteriser on the GPU (its CPU counterpart: cpu_baseline leg)
            try:
                torch.cuda.synchronize()
                st.timer_begin()
                for _ in range(5):
                    fs.warp(S)
                out["warp_stage"] = {"gpu_ms_per_frame": st.timer_end() / 5 / S}
                r0 = fs.results(0)
            except Exception as e:
                out["warp_stage"] = {"error": str(e)[:200]}
        if a.stamps:
            arr = np.zeros((512, 16), np.uint64)
            us = None
            if st.lib.ArapFlow_SolverStamps(fs.h, arr.ctypes.data) == 0:
                o = arr.astype(np.float64)
                used = o[:, 0] > 0
                us = o[used, :5].mean(0) * 0.01 / lIter if used.any() else None  # 100 MHz ticks -> us per iteration, mean over workgroups
            if us is not None:
                print(json.dumps({"stamps": {"us": dict(zip(["phaseA", "wait1", "phaseB", "wait2", "update"], [float(v) for v in us])),
                                             "wait_frac": float((us[1] + us[3]) / us.sum()), "workgroups": int(used.sum()),
                                             "note": "instrumented build (ARAPOPT_STAMPS=1), last resident launch; a workgroup that "
                                                     "finishes a phase before its CU-mate counts the mate's remaining phase time as wait"}}))
        if not a.no_cpu_baseline and world == 1:               # the CPU side is timed at N = 1 only
            f0 = frames[0]
            spec = dict(W=W, H=H, seed=f0["seed"], K=K, fd=a.fd, full=(a.workload == "full"), multseg=bool(a.multseg),
                        active=float(n_active))
            out["cpu_baseline"], oracle_hash = cpu_baseline(spec, a.schedule)
            # the timed work is the verified work: frame 0 of the timed steps equals the oracle's result bit for bit
            if oracle_hash is None:
                out["parity_check"] = {"frame": 0, "bit_equal": None,
                                       "note": "the CPU sample was cut short of the full schedule (bounded CPU time): no comparison"}
            else:
                out["parity_check"] = {"frame": 0, "bit_equal": bool(oracle_hash == parity_hash),
                                       "what": "sha256 of Offset + Angle of frame 0%s after the timed steps vs the float32 CPU "
                                               "oracle on the same input and schedule" % (" (first segment)" if a.multseg else "")}
            if not a.no_kernel_timing and r0 is not None:
                try:                                            # the CPU rasteriser (oracle) on one frame of the same field
                    from oracle import oracle as orc
                    t0w = time.time()
                    orc.warp_offset(solves[0]["rgb"], solves[0]["mask_red"], r0["offset"])
                    out["cpu_baseline"]["warp_stage"] = {
                        "cpu_ms_per_frame": (time.time() - t0w) * 1e3,
                        "cpu": "oracle rasteriser, 1 thread (the reference's warp code is single threaded)"}
                except Exception as e:
                    out["cpu_baseline"]["warp_stage"] = {"error": str(e)[:200]}
        print(json.dumps(out))
    fs.close()
    st.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
