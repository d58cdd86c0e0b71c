#!/bin/bash
# Round 6: DataFrame::filter on batch lengths that are not multiples of a wave's rows — the block kernel as the host now picks its form
# (short slots / long tiles / a block per batch, by how full they come out) against the wave-tile kernel (--block 0); one box.
set -u
OUT=gpurun_out/ragged_ab
mkdir -p $OUT
: > $OUT/ab.jsonl
for cr in 1000 1500 3000 5000 6000 7000 10000; do
  for b in 1 0; do
    python tools/bench_frames.py --only filter_frame --steps 5 --chunk-rows $cr --block $b 2>> $OUT/err.txt | grep kernel_ms | grep -v three_pass | sed "s/^{/{\"chunk_rows\": $cr, \"filter_block\": $b, /" >> $OUT/ab.jsonl
  done
done
python tools/bench_kernels.py --rows 1000000000 --steps 5 --only filter_1col_1024_row_chunks,filter_1col_4096_row_chunks,filter_1col_65536_row_chunks 2>> $OUT/err.txt | grep kernel_ms >> $OUT/ab.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/ragged_ab/ab.jsonl'):
    d = json.loads(l)
    print(d.get('chunk_rows'), d.get('filter_block'), d['kernel'], round(d['kernel_ms'], 3), d.get('last_kernel'))
PY
