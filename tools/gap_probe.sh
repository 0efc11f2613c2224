#!/bin/bash
# GPU box: kernel trace of a short solve; prints the timeline of two Gauss-Newton steps (start offsets, durations, gaps)
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/gapprof
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/gapprof -- python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 1 --warmup 0 --schedule 2 4 400 > gpurun_out/gap_probe.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/gapprof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
res = [i for i, r in enumerate(rows) if "k_pcg_resident" in r["Kernel_Name"]]
i0 = res[2] - 1 if len(res) > 3 else 0
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = None
for r in rows[i0:i0 + 22]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%-40s start %9.1f us  dur %8.1f us  gap before %6.1f us" % (r["Kernel_Name"][:40], (s - t0) / 1e3, (e - s) / 1e3, gap))
    prev_end = e
PY
