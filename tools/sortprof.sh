#!/bin/bash
# per-kernel times of one bench_kernels.py / bench_frames.py entry: bash tools/sortprof.sh <tool.py> <args...>  (on the GPU box)
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf "$REPO/gpurun_out/kprof"
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/kprof" -o k -- python "$REPO/$1" "${@:2}" > "$REPO/gpurun_out/kprof.log" 2>&1
f=$(find "$REPO/gpurun_out/kprof" -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f'{r["Name"][:90]:90s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:10.1f} total_ms={float(r["TotalDurationNs"])/1e6:9.2f} pct={r["Percentage"]}')
PY
find "$REPO/gpurun_out/kprof" -name "*kernel_trace.csv" -delete; find "$REPO/gpurun_out/kprof" -name "*.db" -delete
