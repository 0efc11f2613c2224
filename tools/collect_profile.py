"""GPU box: everything the bench line of ONE configuration rests on, collected in one go and written under profiles/.

    python tools/collect_profile.py NAME [--round r02] [-- extra bench.py arguments]

NAME is one of CONFIGS below (the BASELINE.json configs that fit one GPU, plus the roofline configuration mask == 0).
For the configuration it runs, each as its own process with the program directly after `--` (no env/sh hop):
  1. python3 bench.py <args>                                   -> <round>_<NAME>_bench.json   (the JSON line)
  2. rocprofv3 --kernel-trace --stats -- python3 bench.py ...   -> <round>_<NAME>_kernel_stats.csv
  3. rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes, MI355X_MICROARCH.md "HBM"), and two SQ passes
     on a short schedule (1 ramp step x 2 GN steps x 400 PCG iterations)
  4. the instrumented build of the resident kernel (ARAPOPT_STAMPS=1): share of an iteration spent in the two group waits
  5. python3 bench.py <args> again with the counters of 2-4 in place: that line (traffic, hbm_frac_by_counters,
     valu_issue_frac, wait_frac filled in) is the one kept as <round>_<NAME>_bench.json
and condenses 2-4 into <round>_<NAME>_counters.json (all three under gpurun_out/profiles/, the directory that travels
back from the GPU box; copied into profiles/ by hand), which bench.py reads back from profiles/ for `roofline.traffic`,
`hbm_frac_by_counters`, `valu_issue_frac`, `wait_frac` -- keyed by the configuration's signature and the hash of the
kernel sources, so a record made for other code or another workload is never quoted.
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[1]: what the driver's default bench run measures
    "davis_854x480": [],
    # configs[2]: 854x480 --multseg (3 segments per frame), fd = 2
    "multseg3_fd2_854x480": ["--multseg", "3", "--fd", "2"],
    # configs[4]: 1920x1080 --multseg, fd = 5
    "multseg3_fd5_1920x1080": ["--size", "1920", "1080", "--multseg", "3", "--fd", "5"],
    # roofline configuration: mask == 0, every vertex active
    "full_854x480": ["--workload", "full"],
    "full_1920x1080": ["--size", "1920", "1080", "--workload", "full", "--batch", "1", "--steps", "1", "--warmup", "0"],
}
KERNELS = ("k_pcg_resident", "k_pcg_a", "k_pcg_b", "k_pcg_stream")


def run(cmd, log, env=None, timeout=1500):
    t = time.time()
    with open(log, "w") as f:
        r = subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, env=env, cwd=ROOT, timeout=timeout)
    print("[%6.1f s] rc %d  %s" % (time.time() - t, r.returncode, " ".join(cmd)[:160]), flush=True)
    return r.returncode


def kernel_key(name):
    for k in KERNELS:
        if k in name:
            # k_pcg_a_lds<..> and k_pcg_b4 are forms of phase A / B
            return k
    return None


def pmc_pass(counters, bench_args, tag, outdir):
    d = "/tmp/cp_%s" % tag
    shutil.rmtree(d, ignore_errors=True)
    cmd = ["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", d, "--", "python3", "bench.py"] + bench_args
    rc = run(cmd, os.path.join(outdir, "pmc_%s.log" % tag))
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    if rc == 0 and files:
        for r in csv.DictReader(open(files[0])):
            k = kernel_key(r["Kernel_Name"])
            if k:
                a = agg[k][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    shutil.rmtree(d, ignore_errors=True)
    return {k: {c: {"launches": n, "avg_per_launch": v / n} for c, (n, v) in cs.items()} for k, cs in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name", choices=sorted(CONFIGS))
    ap.add_argument("--round", default="r03")
    ap.add_argument("--skip", nargs="*", default=[], choices=["bench", "stats", "pmc", "stamps"])
    ap.add_argument("--bench-only", action="store_true",
                    help="re-run only the bench line (the counters record under profiles/ is kept and read back by it)")
    ap.add_argument("extra", nargs="*")
    a = ap.parse_args()
    os.chdir("/tmp")
    os.environ["TMPDIR"] = "/tmp"
    from tools import profile_key
    args = CONFIGS[a.name] + a.extra
    # written under gpurun_out/ (the only directory that travels back from the GPU box); copy into profiles/ afterwards
    prof = os.path.join(ROOT, "gpurun_out", "profiles")
    os.makedirs(prof, exist_ok=True)
    outdir = os.path.join(ROOT, "gpurun_out", "cp_" + a.name)
    os.makedirs(outdir, exist_ok=True)
    pre = os.path.join(prof, "%s_%s" % (a.round, a.name))
    rec = {"config": a.name, "bench_args": args, "source_hash": profile_key.source_hash(),
           "made_by": "tools/collect_profile.py " + a.name, "unix_time": int(time.time())}
    if a.bench_only:
        log = os.path.join(outdir, "bench_final.log")
        if run(["python3", "bench.py"] + args, log) == 0:
            lines = [ln for ln in open(log).read().splitlines() if ln.startswith("{")]
            if lines:
                open(pre + "_bench.json", "w").write(lines[-1] + "\n")
                print(lines[-1][:300])
        return

    # ---- 1. the bench line ---------------------------------------------------------------------------------------
    bench_line = None
    if "bench" not in a.skip:
        log = os.path.join(outdir, "bench.log")
        if run(["python3", "bench.py"] + args, log) == 0:
            lines = [ln for ln in open(log).read().splitlines() if ln.startswith("{")]
            if lines:
                bench_line = json.loads(lines[-1])
                open(pre + "_bench.json", "w").write(lines[-1] + "\n")
                rec["signature"] = bench_line.get("profile_signature")
    if rec.get("signature") is None and os.path.exists(pre + "_bench.json"):
        rec["signature"] = json.load(open(pre + "_bench.json")).get("profile_signature")    # (bench step skipped)

    # ---- 2. kernel trace of the same command ------------------------------------------------------------------------
    if "stats" not in a.skip:
        d = "/tmp/cp_stats"
        shutil.rmtree(d, ignore_errors=True)
        rc = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", "python3", "bench.py",
                  "--no-cpu-baseline"] + args, os.path.join(outdir, "stats.log"))
        files = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
        if rc == 0 and files:
            rows = list(csv.reader(open(files[0])))
            with open(pre + "_kernel_stats.csv", "w", newline="") as f:
                csv.writer(f, quoting=csv.QUOTE_NONNUMERIC).writerows(rows[:14])
            rec["kernel_stats"] = {}
            for r in rows[1:]:
                k = kernel_key(r[0])
                if k and k not in rec["kernel_stats"]:
                    rec["kernel_stats"][k] = {"name": r[0], "calls": int(r[1]), "avg_ns": float(r[3]), "pct": float(r[4])}
        shutil.rmtree(d, ignore_errors=True)

    # ---- 3. counters, short schedule, one group per pass --------------------------------------------------------------
    short = args + ["--no-cpu-baseline", "--no-kernel-timing", "--steps", "1", "--warmup", "0", "--schedule", "1", "2", "400"]
    if "pmc" not in a.skip:
        pm = {}
        for tag, ctrs in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]),
                          ("sq1", ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU"]),
                          ("sq2", ["SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]),
                          ("sq_lds", ["SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"]),
                          ("sq_lds2", ["SQ_WAIT_INST_LDS", "SQ_LDS_ADDR_CONFLICT", "SQ_INSTS_SALU", "SQ_INSTS_VMEM"]),
                          ("grbm", ["GRBM_GUI_ACTIVE"])):
            for k, cs in pmc_pass(ctrs, short, tag, outdir).items():
                pm.setdefault(k, {}).update(cs)
        rec["pmc"] = pm
        rec["pmc_command"] = "rocprofv3 --pmc <group> -- python3 bench.py " + " ".join(short)
        for k, cs in pm.items():
            if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide streaming
                # reads (MI355X_MICROARCH.md, HBM): doubled before it is compared with a byte count
                fb = cs["FETCH_SIZE"]["avg_per_launch"] * 1024.0 * 2.0
                wb = cs["WRITE_SIZE"]["avg_per_launch"] * 1024.0
                cs["hbm_bytes_per_launch"] = {"fetch_bytes_x2": fb, "write_bytes": wb, "total": fb + wb}

    # ---- 4. group-wait share from the instrumented resident kernel -----------------------------------------------------
    if "stamps" not in a.skip:
        env = dict(os.environ, ARAPOPT_STAMPS="1")
        log = os.path.join(outdir, "stamps.log")
        if run(["python3", "bench.py", "--stamps"] + args + ["--no-cpu-baseline", "--no-kernel-timing", "--steps", "1",
                                                              "--warmup", "0", "--schedule", "1", "3", "400"], log, env=env) == 0:
            for ln in open(log).read().splitlines():
                if ln.startswith("{\"stamps\""):
                    rec["stamps"] = json.loads(ln)["stamps"]
    # the record is tied to the sources of the kernel it is about (tools/profile_key.py)
    ks = rec.get("kernel_stats", {})
    rec["dominant_kernel"] = max(ks, key=lambda k: ks[k]["pct"]) if ks else None
    rec["source_hash"] = profile_key.source_hash(rec["dominant_kernel"])
    open(pre + "_counters.json", "w").write(json.dumps(rec, indent=1) + "\n")
    print("wrote", pre + "_counters.json")
    # ---- 5. the bench line once more, now that its counters exist: the committed line carries traffic / utilisations ----
    if "bench" not in a.skip:
        os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
        shutil.copy(pre + "_counters.json", os.path.join(ROOT, "profiles", os.path.basename(pre) + "_counters.json"))
        log = os.path.join(outdir, "bench_final.log")
        if run(["python3", "bench.py"] + args, log) == 0:
            lines = [ln for ln in open(log).read().splitlines() if ln.startswith("{")]
            if lines:
                bench_line = json.loads(lines[-1])
                open(pre + "_bench.json", "w").write(lines[-1] + "\n")
    if bench_line:
        print(json.dumps({k: bench_line[k] for k in ("value", "ms_per_step", "roofline") if k in bench_line})[:900])


if __name__ == "__main__":
    main()
