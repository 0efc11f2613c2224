#!/bin/bash
# Run on the GPU box (through gpurun): default bench line, rocprofv3 kernel stats of the same command, and the two
# PMC passes (separate runs, as MI355X_MICROARCH.md prescribes).  Outputs under gpurun_out/ ; copy into profiles/.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > gpurun_out/bench_default.log 2>&1
tail -1 gpurun_out/bench_default.log > gpurun_out/bench_default.json
rm -rf /tmp/prof1 /tmp/prof2 /tmp/prof3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_stats_bench.log 2>&1
cp "$(find /tmp/prof1 -name '*kernel_stats.csv' | head -1)" gpurun_out/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof2 -- python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 1 --warmup 0 --schedule 1 2 400 > gpurun_out/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof3 -- python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 1 --warmup 0 --schedule 1 2 400 > gpurun_out/prof_write.log 2>&1
python3 - <<PY > gpurun_out/pmc_summary.txt
import csv, glob, collections
for d, c in (("/tmp/prof2", "FETCH_SIZE"), ("/tmp/prof3", "WRITE_SIZE")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            a = agg[r["Kernel_Name"][:48]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, v) in sorted(agg.items()):
        print("%-11s %-50s launches %4d  avg KiB %12.1f" % (c, k, n, v / n))
PY
head -8 gpurun_out/kernel_stats.csv
cat gpurun_out/pmc_summary.txt | grep -i "arap::k_pcg"
cat gpurun_out/bench_default.json | cut -c1-400
