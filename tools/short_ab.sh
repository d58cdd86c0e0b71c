#!/bin/bash
# Round 6: the block kernel's short-batch mode against the wave-tile kernels, same box, alternating runs.
#   gpurun --timeout 1500 -- 'bash tools/short_ab.sh'
set -u
OUT=gpurun_out/short_ab
mkdir -p $OUT
: > $OUT/ab.jsonl
for rep in 1 2; do
  for cr in 1024 2048 4096; do
    for sh in 1 0; do
      python tools/bench_frames.py --only filter_frame --steps 5 --chunk-rows $cr --short $sh 2>> $OUT/err.txt | grep kernel_ms | grep -v three_pass | sed "s/^{/{\"short\": $sh, \"chunk_rows\": $cr, /" >> $OUT/ab.jsonl
    done
  done
done
python tools/bench_kernels.py --rows 1000000000 --steps 5 --only filter_1col_1024_row_chunks,filter_1col_1024_row_chunks_wave_tiles,filter_1col_4096_row_chunks,filter_1col_4096_row_chunks_three_kernels,filter_1col_65536_row_chunks 2>> $OUT/err.txt | grep kernel_ms >> $OUT/ab.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/short_ab/ab.jsonl'):
    d = json.loads(l)
    print(d.get('short'), d.get('chunk_rows'), d['kernel'], d['kernel_ms'], d.get('last_kernel'))
PY
