#!/usr/bin/env python3
"""bench.py -- ARAP solve+warp throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" = one pass of the hot path over one batch of synthetic frames on every rank: the full
arap_deform schedule (19 ramp steps x 8 Gauss-Newton steps x 400 PCG iterations,
ARAP/deformation/src/main.cpp:215-221) for `--batch` independent 854x480 frames, then flow emission
and the forward rasterisation of RGB and mask.  Inputs (RGB, mask, constraints) are uploaded before
the timed region.  Workload = BASELINE.json configs[1]: single-segment DAVIS-shaped mask, fd=1.
Frames shard across ranks with no collective: weak scaling, value = all frames / max time.

One JSON line on rank 0: metric/value/... plus
  roofline     : dominant PCG kernel; achieved = algorithmic bytes per launch / average launch
                 duration measured with HIP events around every launch (a separate, un-graphed pass)
  cpu_baseline : the CPU oracle (kind "port") timed on this host on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np

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
    return ap.parse_args()


def _time_oracle(frame, threads, numIter, nIter, lIter):
    """wall time of the CPU oracle in a FRESH process (OMP_NUM_THREADS must be set before the OpenMP runtime starts;
    this process already runs torch's)"""
    import subprocess
    H, W = frame["mask_red"].shape
    code = ("import sys,time,json,numpy as np;sys.path.insert(0,%r);from oracle import oracle as orc;"
            "from arap_flow_amd import synth;f=synth.make_frame(%d,%d,seed=%d,full_mask=%r);t=time.time();"
            "orc.frame(f['mask_red'],f['constraints'],numIter=%d,nIterations=%d,lIterations=%d,dtype=np.float32,mode=1,trig=1);"
            "print(json.dumps(time.time()-t))" % (ROOT, W, H, frame.get("seed", 0), bool((frame["mask_red"] == 0).all()),
                                                  numIter, nIter, lIter))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_WAIT_POLICY="active"),
                       capture_output=True, text=True, timeout=900)
    return float(r.stdout.strip().splitlines()[-1])


def cpu_baseline(frame, schedule):
    """The CPU oracle (same algorithm, float32, OpenMP over rows; kind "port") on a bounded sample of the same
    workload, on this box's CPU share (a 1-GPU box grants 16 of the host's cores).  Only this leg of the bench
    touches oracle/."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, int(os.environ.get("ARAP_CPU_BASELINE_THREADS", "16")))
    numIter, nIter, lIter = schedule
    ns = min(numIter, 19)
    dt = _time_oracle(frame, cores, ns, nIter, lIter)
    fps = 1.0 / (dt * numIter / ns)
    out = {"value": fps, "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": "%d of %d ramp steps (x %d GN x %d PCG iterations) of one frame of the same workload, %.1f s"
                     % (ns, numIter, nIter, lIter, dt),
           "pcg_iters_per_s": ns * nIter * lIter / dt}
    try:                                                      # the same code on ONE thread (SURVEY 8d asks for both)
        n1 = min(nIter, 8)
        t1 = _time_oracle(frame, 1, 1, n1, lIter)
        out["single_thread"] = {"value": 1.0 / (t1 * numIter * nIter / n1), "unit": "frames/s", "cores": 1,
                                "sample": "1 ramp step x %d GN x %d PCG iterations, %.1f s, scaled" % (n1, lIter, t1)}
    except Exception as e:                                    # never fail the bench over the extra figure
        out["single_thread"] = {"error": str(e)[:200]}
    return out


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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
    barrier()
    dt = time.perf_counter() - t0
    dt = shard.max_over_ranks(dt, dist, device="cuda" if backend == "nccl" else "cpu")
    stats = fs.stats()
    total_frames = world * B * a.steps
    fps = total_frames / dt
    pcg_per_frame = stats["pcg_iterations_per_frame"]
    n_active = stats["active_vertices"] / B          # per frame (all its segments)
    n_grid = W * H

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
    }

    if rank == 0:
        # ---- roofline of the dominant kernel: HIP events around every launch (un-graphed pass) ----------
        if not a.no_kernel_timing:
            tl = min(lIter, 400)
            st.set_kernel_timing(True)
            fs.solve(S, 1, 4, tl)                      # 4 Gauss-Newton steps of tl PCG iterations
            torch.cuda.synchronize()
            kt = {k: st.kernel_time(k) for k in ("PCGResident", "PCGStepA", "PCGStepB")}
            st.set_kernel_timing(False)
            n_act_total = stats["active_vertices"]     # all B frames
            if kt["PCGResident"] is not None:
                tot_ms, n = kt["PCGResident"]
                # one launch = all `tl` PCG iterations of one GN step for the frames in flight;
                # algorithmic bytes: 160 B per active vertex per PCG iteration
                bytes_total = 160.0 * n_act_total * tl * 4
                ach = bytes_total / (tot_ms * 1e-3) / 1e9
                frames_per_launch = S * 4.0 / n
                out["roofline"] = {
                    "bound": "hbm", "kernel": "k_pcg_resident", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": tot_ms / n * 1e3, "launches": n,
                    "pcg_iterations_per_launch": tl, "frames_per_launch": frames_per_launch,
                    "us_per_pcg_iteration": tot_ms / n * 1e3 / tl,
                    "achieved_vs_grid_vertices": ach * n_grid / n_active, "frac_vs_grid_vertices": ach * n_grid / n_active / HBM_PEAK_GBS,
                    "note": "algorithmic bytes = 160 B x active vertices x PCG iterations of the launch; the kernel keeps "
                            "the PCG state in registers/LDS, so its HBM traffic is far below the algorithmic bytes and "
                            "the fraction may exceed 1 (BASELINE.md section 3)"}
            else:
                per = {}
                for k, bytes_v in (("PCGStepA", BYTES_A), ("PCGStepB", BYTES_B)):
                    tot_ms, n = kt[k]
                    avg_s = tot_ms / n * 1e-3
                    per[k] = {"avg_us": avg_s * 1e6, "launches": n, "GBs_active": bytes_v * n_act_total / avg_s / 1e9,
                              "GBs_grid": bytes_v * n_grid * B / avg_s / 1e9}
                dom = max(per, key=lambda k: per[k]["avg_us"])
                out["roofline"] = {"bound": "hbm", "kernel": {"PCGStepA": "k_pcg_a", "PCGStepB": "k_pcg_b"}[dom],
                                   "achieved": per[dom]["GBs_active"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": per[dom]["GBs_active"] / HBM_PEAK_GBS, "traffic": None,
                                   "avg_launch_us": per[dom]["avg_us"], "achieved_vs_grid_vertices": per[dom]["GBs_grid"],
                                   "frac_vs_grid_vertices": per[dom]["GBs_grid"] / HBM_PEAK_GBS, "per_kernel": per,
                                   "note": "algorithmic bytes = %d (A) / %d (B) per active vertex per launch" % (BYTES_A, BYTES_B)}
            # HBM bytes per launch from rocprofv3 PMC passes (collected separately, profiles/)
            tf = os.path.join(ROOT, "profiles", "traffic_%s_b%d.json" % (a.workload, B))
            if os.path.exists(tf):
                det = json.load(open(tf)).get(out["roofline"]["kernel"])
                if det:
                    out["roofline"]["traffic"] = det.get("hbm_bytes_per_launch")     # bytes per launch
                    out["roofline"]["traffic_unit"] = "bytes per launch"
                    out["roofline"]["traffic_detail"] = det
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
            r0 = None
            try:
                torch.cuda.synchronize()
                st.timer_begin()
                for _ in range(5):
                    fs.warp(S)
                out["warp_stage"] = {"gpu_ms_per_frame": st.timer_end() / 5 / S}
                r0 = fs.results(0)
            except Exception as e:
                out["warp_stage"] = {"error": str(e)[:200]}
        if not a.no_cpu_baseline and world == 1:               # the CPU side is timed at N = 1 only
            out["cpu_baseline"] = cpu_baseline(frames[0], a.schedule)
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
