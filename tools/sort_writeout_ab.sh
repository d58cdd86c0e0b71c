set -u
E=sort_to_indices_i64_full_range,sort_to_indices_i64_full_range_byte_passes,sort_to_indices_i64_full_range_1e9,sort_to_indices_f64_uniform,sort_to_indices_f64_normal,sort_to_indices_2keys_i64_desc_f64_asc
for rep in 1 2 3; do
  for L in librdf_mi355x.so ${ALT:-librdf_alt_sort_base.so}; do
    echo "== $L"
    RDF_LIB_PATH=$PWD/rust_dataframe_amd/$L python tools/bench_kernels.py --rows 1000000000 --steps 3 --only $E 2>/dev/null | grep kernel_ms | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['kernel'], d['rows'], round(d['kernel_ms'], 3))"
  done
done
