#!/bin/bash
# GPU box: SQ counters of the resident kernel (one rocprofv3 --pmc pass per counter group), short schedule
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM"; do
  i=$((i+1)); rm -rf /tmp/pmcp$i
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmcp$i -- python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 1 --warmup 0 --schedule 1 2 400 > gpurun_out/pmc_probe_$i.log 2>&1 || { echo "group $i failed"; tail -3 gpurun_out/pmc_probe_$i.log; continue; }
  python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmcp$i/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if "k_pcg_resident" in r["Kernel_Name"]:
        a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print("%-24s launches %3d  avg per launch %16.0f" % (k, n, v / n))
PY
done
