#!/bin/bash
# Round 6: tile size of the sort's digit passes — 12 / 16 (default) / 24 items per thread (3072 / 4096 / 6144-pair tiles; 3 / 2 / 2 resident blocks per CU).
set -u
OUT=gpurun_out/sort_items_ab
mkdir -p $OUT
: > $OUT/ab.jsonl
E=sort_to_indices_i64_full_range,sort_to_indices_i64_full_range_byte_passes,sort_to_indices_i64_full_range_1e9,sort_to_indices_f64_uniform,sort_to_indices_f64_normal,sort_to_indices_2keys_i64_desc_f64_asc
for rep in 1 2; do
  for it in 16 12 24; do
    L=rust_dataframe_amd/librdf_alt_items$it.so; [ $it = 16 ] && L=rust_dataframe_amd/librdf_mi355x.so
    RDF_LIB_PATH=$PWD/$L python tools/bench_kernels.py --rows 1000000000 --steps 3 --only $E 2>> $OUT/err.txt | grep kernel_ms | sed "s/^{/{\"items\": $it, /" >> $OUT/ab.jsonl
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/sort_items_ab/ab.jsonl'):
    d = json.loads(l)
    print(d.get('items'), d['kernel'], d.get('rows'), round(d['kernel_ms'], 3))
PY
