#!/bin/bash
# Fixed cost of a resident launch: rocprofv3 kernel durations of the headline bench at 1, 2 and 400 PCG iterations per launch.
# usage (on the GPU box):  bash tools/launch_intercept.sh TAG     -> gpurun_out/TAG_intercept.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$1_intercept.txt
: > "$out"
for L in 1 2 400; do
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$1_L$L -o p -- python3 bench.py --steps 2 --warmup 1 --schedule 19 8 $L --no-cpu-baseline > gpurun_out/$1_L$L.log 2>&1 || exit 1
    python3 - $L gpurun_out/$1_L$L >> "$out" <<PY
import csv, glob, sys
f = glob.glob(sys.argv[2] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_pcg_res" in r["Name"] or r["Name"].startswith("arap::k_gn"):
        print("L=%s %-28s calls %s avg %.1f us min %.1f us" % (sys.argv[1], r["Name"].split("(")[0][-28:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
    grep '^{' gpurun_out/$1_L$L.log | python3 -c "import json,sys; r=json.loads(sys.stdin.read()); print('L=$L frames/s %.3f ms_per_step %.3f' % (r['value'], r['ms_per_step']))" >> "$out"
done
cat "$out"
