#!/bin/bash
# Counter passes for one command on the GPU box (kernel trace only, one rocprofv3 run per counter group, as the MI355X guide asks):
#   bash tools/pmc.sh <kernel-name-substring> "<group1 counters>" "<group2 counters>" ... -- python tools/bench_kernels.py ...
# prints, per counter, the average per dispatch over the kernels whose name contains the substring.
REPO=$(pwd)
KSUB=$1; shift
GROUPS_=()
while [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
i=0
for g in "${GROUPS_[@]}"; do
    i=$((i+1))
    rm -rf "$REPO/gpurun_out/pmc_$i"
    timeout 600 rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$REPO/gpurun_out/pmc_$i" -o p -- "$@" > "$REPO/gpurun_out/pmc_$i.out" 2> "$REPO/gpurun_out/pmc_$i.err"
    f=$(find "$REPO/gpurun_out/pmc_$i" -name "*counter_collection.csv" | head -1)
    python - "$f" "$KSUB" <<'PY'
import collections
import csv
import sys
acc = collections.defaultdict(float)
disp = set()
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
        disp.add((r["Counter_Name"], r["Dispatch_Id"]))
n = collections.Counter(c for c, _ in disp)
for c, v in sorted(acc.items()):
    print(f"{c:28s} per dispatch {v / max(n[c], 1):18.1f}   ({n[c]} dispatches)")
PY
    find "$REPO/gpurun_out/pmc_$i" -name "*.csv" -size +20M -delete
done
