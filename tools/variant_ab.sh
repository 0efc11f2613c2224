#!/bin/bash
# A/B of library variants on ONE box: bash tools/variant_ab.sh TAG "bench args" lib1.so lib2.so ...   ("-" = the built library)
# Every variant is run twice, interleaved; prints frames/s and the HIP-event launch time of each run.
tag=$1; args=$2; shift 2
for round in 1 2; do
  for lib in "$@"; do
    name=$(basename "$lib" .so)
    if [ "$lib" = "-" ]; then unset ARAPOPT_LIB; name=current; else export ARAPOPT_LIB="$GRAFT_REPO_ROOT/arap_flow_amd/lib/$lib"; fi
    timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline $args > gpurun_out/${tag}_${name}_$round.log 2>&1 || { echo "$name failed"; exit 1; }
    grep '^{' gpurun_out/${tag}_${name}_$round.log | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print('%-28s %.3f frames/s  %.2f ms/step  launch %.1f us' % ('$name', r['value'], r['ms_per_step'], r['roofline'].get('avg_launch_us') or 0))"
  done
done
