#!/bin/bash
# Round 6: digit passes of the sort — K tiles per ticket (os_scatter4_kernel) against a tile per ticket (os_scatter_kernel); same box, alternating.
set -u
OUT=gpurun_out/sort_super_ab
mkdir -p $OUT
: > $OUT/ab.jsonl
E=sort_to_indices_i64_full_range,sort_to_indices_i64_full_range_byte_passes,sort_to_indices_i64_full_range_1e9,sort_to_indices_f64_uniform,sort_to_indices_f64_normal,sort_to_indices_f32_uniform,sort_to_indices_2keys_i64_desc_f64_asc,sort_to_indices_i64_dictionary_codes
for rep in 1 2; do
  for k in ${KS:-8 1 4 16}; do
    python tools/bench_kernels.py --rows 1000000000 --steps 3 --sort-super $k --only $E 2>> $OUT/err.txt | grep kernel_ms | sed "s/^{/{\"sort_super\": $k, /" >> $OUT/ab.jsonl
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/sort_super_ab/ab.jsonl'):
    d = json.loads(l)
    print(d.get('sort_super'), d['kernel'], d.get('rows'), round(d['kernel_ms'], 3))
PY
