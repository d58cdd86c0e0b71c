#!/bin/bash
# Round 6: long batches, many of them — a block per batch (its own running offset) against tiles by ticket + scanner wave; same box, alternating.
set -u
OUT=gpurun_out/owned_ab
mkdir -p $OUT
: > $OUT/ab.jsonl
for rep in 1 2; do
  for cr in 16384 65536 200000; do
    for ow in 1 0; do
      python tools/bench_frames.py --only filter_frame --steps 5 --chunk-rows $cr --owned $ow 2>> $OUT/err.txt | grep kernel_ms | grep -v three_pass | sed "s/^{/{\"owned\": $ow, \"chunk_rows\": $cr, /" >> $OUT/ab.jsonl
    done
  done
done
python tools/bench_kernels.py --rows 1000000000 --steps 5 --only filter_1col_65536_row_chunks,filter_1col_65536_row_chunks_scanner_wave 2>> $OUT/err.txt | grep kernel_ms >> $OUT/ab.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/owned_ab/ab.jsonl'):
    d = json.loads(l)
    print(d.get('owned'), d.get('chunk_rows'), d['kernel'], round(d['kernel_ms'], 3), d.get('last_kernel'))
PY
