#!/bin/bash
# GPU box: HBM-side traffic of the two-kernel path at 1920x1080, mask == 0 (8100 tiles: more than the resident kernel
# holds), per phase-A variant, plus the calibration of FETCH_SIZE for 8-byte-per-lane accesses: the two forms of phase B
# (k_pcg_b: 8 B / 4 B per lane, k_pcg_b4: 16 B per lane) move exactly the same bytes.
# usage: tools/pmc_twokernel.sh [bench args ...]   (default: --size 1920 1080 --workload full --batch 1)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="${@:---size 1920 1080 --workload full --batch 1}"
COMMON="--no-cpu-baseline --no-kernel-timing --steps 1 --warmup 0 --schedule 1 1 50"
run() {   # $1 tag, $2 counters, env already exported
  rm -rf /tmp/pmc2k
  rocprofv3 --pmc $2 --output-format csv -d /tmp/pmc2k -- python3 bench.py $ARGS $COMMON > gpurun_out/pmc2k_$1.log 2>&1 || { echo "$1 failed"; tail -3 gpurun_out/pmc2k_$1.log; return; }
  python3 - "$1" <<PY
import csv, glob, collections, sys
f = glob.glob("/tmp/pmc2k/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if "k_pcg_" in r["Kernel_Name"]:
        a = agg[(r["Kernel_Name"].split("(")[0][-28:], r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(agg.items()):
    print("%-10s %-30s %-14s launches %4d  avg %14.1f" % (sys.argv[1], k, c, n, v / n))
PY
}
run "auto_fetch" FETCH_SIZE
run "auto_write" WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc2k_s0 -- python3 bench.py $ARGS $COMMON > gpurun_out/pmc2k_stats_auto.log 2>&1
head -4 "$(find /tmp/pmc2k_s0 -name '*kernel_stats.csv' | head -1)"
export ARAPOPT_STREAM_A=1
run "grid_fetch" FETCH_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc2k_s1 -- python3 bench.py $ARGS $COMMON > gpurun_out/pmc2k_stats_grid.log 2>&1
head -3 "$(find /tmp/pmc2k_s1 -name '*kernel_stats.csv' | head -1)"
unset ARAPOPT_STREAM_A
if [ -n "$PMC2K_SHORT" ]; then exit 0; fi
for tile in 64x8 0x0; do
  export ARAPOPT_TILE=$tile
  run "A${tile}_fetch" FETCH_SIZE
  run "A${tile}_write" WRITE_SIZE
  run "A${tile}_l2" "TCC_HIT_sum TCC_MISS_sum"
done
export ARAPOPT_B8=1
run "B8_fetch" FETCH_SIZE
run "B8_write" WRITE_SIZE
unset ARAPOPT_B8
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmc2k_s -- python3 bench.py $ARGS $COMMON > gpurun_out/pmc2k_stats.log 2>&1
head -6 "$(find /tmp/pmc2k_s -name '*kernel_stats.csv' | head -1)"
