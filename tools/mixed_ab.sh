#!/bin/bash
# Round 6: DataFrame::filter over frames of 8- and 4-byte columns — the block kernel twice (filter_mixed 2) against the wave-tile kernel (0), by batch length.
set -u
OUT=gpurun_out/mixed_ab
mkdir -p $OUT
: > $OUT/ab.jsonl
for cr in 1000 1024 1500 3000 4096 65536 1000000000; do
  python tools/bench_frames.py --only filter_frame_mixed --steps 5 --chunk-rows $cr 2>> $OUT/err.txt | grep kernel_ms | sed "s/^{/{\"chunk_rows\": $cr, /" >> $OUT/ab.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/mixed_ab/ab.jsonl'):
    d = json.loads(l)
    print(d.get('chunk_rows'), d['kernel'], round(d['kernel_ms'], 3), d.get('last_kernel'))
PY
