#!/bin/bash
# developer loop: parity subset, then the cfg3 b4096 bench and a kernel-stats pass (usage: tests/gpu_quick_cfg3.sh TAG)
set -o pipefail
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -q -x -k "eigh or cfg3 or deep" > gpurun_out/q_${TAG}.log 2>&1
rc=$?
tail -3 gpurun_out/q_${TAG}.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --workload cfg3 --batch 4096 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/q_${TAG}_bench.json 2> gpurun_out/q_${TAG}_bench.err || exit 1
cut -c1-200 gpurun_out/q_${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/q_${TAG}_prof -o s --output-format csv -- python3 $R/bench.py --workload cfg3 --batch 4096 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/q_${TAG}_prof/s_kernel_stats.csv")))
for r in rows[:9]:
    print("%-60s calls %4s avg %9.3f ms  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e6, r["Percentage"]))
PY
