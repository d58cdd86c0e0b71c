#!/bin/bash
# Round 6: the hash GROUP BY's aggregate pass — threads per block x records per lane and batch (alt builds of rdf_groupby.hip), same box, alternating.
set -u
E=groupby_sum_1000000_groups,groupby_max_1000000_groups,groupby_count_1000000_groups,groupby_sum_1000000_groups_scattered_keys,groupby_sum_1000000_groups_zipf
for rep in 1 2; do
  for L in librdf_mi355x.so $(cd rust_dataframe_amd && ls librdf_alt_agg_*.so); do
    echo "== $L"
    RDF_LIB_PATH=$PWD/rust_dataframe_amd/$L python tools/bench_kernels.py --rows 1000000000 --steps 5 --only $E 2>/dev/null | grep kernel_ms | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['kernel'], d['rows'], round(d['kernel_ms'], 3))"
  done
done
