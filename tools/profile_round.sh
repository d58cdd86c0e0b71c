#!/bin/bash
# Re-creates the evidence under profiles/ for one round, on the GPU box:
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r01'   then   python tools/collect_profiles.py r01
# Everything is written under gpurun_out/profile_<round>/ (scratch); collect_profiles.py copies the summaries.
# Counters are collected in their own passes (one --pmc counter per pass, kernel trace only), as the MI355X guide asks.
set -u
R=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/profile_$R
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python "$REPO/bench.py" > "$OUT/bench_1e9.json" 2> "$OUT/bench_1e9.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python "$REPO/bench.py" --cpu-sample 0 > "$OUT/bench_1e9_under_rocprof.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 1 --cpu-sample 0 > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o bench -- python "$REPO/bench.py" --steps 5 --warmup 1 --cpu-sample 0 > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err"
python "$REPO/tools/bench_kernels.py" --rows 1000000000 --steps 5 2> "$OUT/kernels.err" | grep kernel_ms > "$OUT/kernels_1e9_microbench.jsonl"
for w in c3 c4 q1; do python "$REPO/bench.py" --workload $w --steps 5 --warmup 2 2>> "$OUT/workloads.err" | tail -1 >> "$OUT/workloads.jsonl"; done
# keep the merged payload small: the raw traces stay on the box, the stats / counter CSVs travel
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
ls -la "$OUT" "$OUT"/*/ 2>/dev/null | head -40
