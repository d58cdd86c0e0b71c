#!/bin/bash
# Re-creates the evidence under profiles/ for one round, on the GPU box:
#   gpurun --timeout 2400 -- 'bash tools/profile_round.sh r02'   then   python tools/collect_profiles.py r02
# Everything is written under gpurun_out/profile_<round>/ (scratch); collect_profiles.py copies the summaries.
# Counters are collected in their own passes (one --pmc counter per pass, kernel trace only), as the MI355X guide asks.
set -u
R=${1:-r02}
PART=${2:-all}      # all | micro (only the per-entry counter passes of step 3b) | workloads (only the three bench lines of step 2) | r3 | new (the counter passes of the round-3 entries only)
REPO=$(pwd)
OUT=$REPO/gpurun_out/profile_$R
mkdir -p "$OUT"
# the kernel sources these measurements are taken on (bench.py reports a committed traffic figure only for the same sources)
( cd "$REPO" && python -c "import bench; print(bench.kernel_sources_sha())" > "$OUT/kernel_sources_sha.txt" 2>/dev/null )
cd /tmp && export TMPDIR=/tmp
pmc() {   # pmc <tag> <command...>: FETCH_SIZE and WRITE_SIZE of every kernel of the command, two passes
    local tag=$1; shift
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_${tag}_$c" -o p -- "$@" > "$OUT/pmc_${tag}_$c.out" 2> "$OUT/pmc_${tag}_$c.err"
    done
}
if [ "$PART" = "new" ]; then
    for e in sort_to_indices_i64_full_range groupby_count_1000000_groups join_inner_1e8_x_1e7; do
        pmc micro_$e python "$REPO/tools/bench_kernels.py" --rows 1000000000 --steps 3 --only $e
    done
    find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
    exit 0
fi
if [ "$PART" = "r6f" ]; then
    # round 6, after the block filter kernel got its short-batch and block-per-batch forms (and the sort's write-out its batched reads): the
    # files those touch, on one box — the driver's line, the micro-benchmarks, the frame operators on 1024-row batches, DataFrame::filter by
    # batch length (block kernel in all its forms against the wave-tile kernels), the HBM counters of filter_frame on 1024-row batches
    python "$REPO/bench.py" > "$OUT/bench_1e9.json" 2> "$OUT/bench_1e9.err"
    python "$REPO/tools/bench_kernels.py" --rows 1000000000 --steps 5 2> "$OUT/kernels.err" | grep kernel_ms > "$OUT/kernels_1e9_microbench.jsonl"
    python "$REPO/tools/bench_frames.py" 2> "$OUT/frames.err" | grep kernel_ms > "$OUT/frames_1e9.jsonl"
    rm -f "$OUT/filter_frame_long_batches.jsonl"
    for cr in 1024 2048 4096 8192 16384 65536 1048576 16777216 1000000000; do for b in 1 0; do
        python "$REPO/tools/bench_frames.py" --rows 1000000000 --chunk-rows $cr --steps 5 --block $b --only filter_frame_1col,filter_frame_2col,filter_frame_4col 2>> "$OUT/frames.err" | grep kernel_ms | sed "s/^{/{\"chunk_rows\": $cr, \"filter_block\": $b, /" >> "$OUT/filter_frame_long_batches.jsonl"
    done; done
    for e in filter_frame_1col filter_frame_4col; do
        pmc frames_$e python "$REPO/tools/bench_frames.py" --steps 2 --only $e
    done
    find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete
    exit 0
fi
if [ "$PART" = "workloads" ]; then
    rm -f "$OUT/workloads.jsonl"
    for w in c3 c4 q1; do python "$REPO/bench.py" --workload $w --steps 10 --warmup 3 2>> "$OUT/workloads.err" | tail -1 >> "$OUT/workloads.jsonl"; done
    exit 0
fi
if [ "$PART" = "all" ]; then
# 1. the driver's bench line, the same under the kernel trace, and on the reference's 1024-row batches
python "$REPO/bench.py" > "$OUT/bench_1e9.json" 2> "$OUT/bench_1e9.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python "$REPO/bench.py" --cpu-sample 0 > "$OUT/bench_1e9_under_rocprof.json" 2> "$OUT/trace.err"
python "$REPO/bench.py" --cpu-sample 0 --chunk-rows 1024 > "$OUT/bench_1e9_1024_row_batches.json" 2>> "$OUT/bench_1e9.err"
python "$REPO/bench.py" --cpu-sample 0 --null-fraction 0.1 > "$OUT/bench_1e9_validity.json" 2>> "$OUT/bench_1e9.err"
pmc headline python "$REPO/bench.py" --steps 5 --warmup 1 --cpu-sample 0
# 2. the other configs of BASELINE.json: bench lines + HBM counters of their kernels
for w in c3 c4 q1; do
    python "$REPO/bench.py" --workload $w --steps 10 --warmup 3 2>> "$OUT/workloads.err" | tail -1 >> "$OUT/workloads.jsonl"
    pmc $w python "$REPO/bench.py" --workload $w --steps 3 --warmup 1 --cpu-sample 0
done
# 3. per-kernel micro-benchmarks, shape kernels, compaction / take counters
python "$REPO/tools/bench_kernels.py" --rows 1000000000 --steps 5 2> "$OUT/kernels.err" | grep kernel_ms > "$OUT/kernels_1e9_microbench.jsonl"
python "$REPO/tools/bench_shapes.py" --interp 2> "$OUT/shapes.err" | grep kernel_ms > "$OUT/shapes_2p5e8.jsonl"
fi
# 3b. HBM counters of compaction / take / small-group GROUP BY, one micro-benchmark entry per pass (the entries share kernels)
for e in filter_1col filter_2col filter_1col_selectivity_1_16 take_random_u32 take_sequential_u32 groupby_sum_1000_groups \
         sort_to_indices_i64_full_range groupby_count_1000000_groups join_inner_1e8_x_1e7 \
         groupby_sum_1000000_groups_zipf groupby_sum_1000000_groups_scattered_keys list_rows_of_1000_distinct sort_to_indices_i64_full_range_1e9 \
         filter_1col_1024_row_chunks filter_1col_4096_row_chunks; do
    pmc micro_$e python "$REPO/tools/bench_kernels.py" --rows 1000000000 --steps 3 --only $e
done
# 3c. round 3: frame-level operators on 976 563 batches of 1024 rows (wall against kernel time), ingestion, Int8 / UInt8 kernels,
#     the multi-GPU code path on a 1-rank RCCL communicator (every collective on device tensors), PMC of the frame take paths
if [ "$PART" = "all" ] || [ "$PART" = "r3" ]; then
python "$REPO/tools/bench_frames.py" 2> "$OUT/frames.err" | grep kernel_ms > "$OUT/frames_1e9.jsonl"
python "$REPO/tools/bench_ingest.py" 2> "$OUT/ingest.err" | grep '"case"' > "$OUT/ingest.jsonl"
python "$REPO/tools/bench_bytes.py" 2> "$OUT/bytes.err" | grep program > "$OUT/bytes_2p5e8.jsonl"
python "$REPO/tools/bench_shapes.py" --beyond 2> "$OUT/beyond.err" | grep kernel_ms > "$OUT/beyond_catalogs_2p5e8.jsonl"   # shapes no catalog holds: compiled at run time vs interpreted
rm -f "$OUT/rccl_one_rank.jsonl"
for w in headline c4 q1; do python "$REPO/bench.py" --workload $w --rows 200000000 --steps 10 --warmup 3 --cpu-sample 0 --force-exchange --backend nccl 2>> "$OUT/rccl.err" | tail -1 >> "$OUT/rccl_one_rank.jsonl"; done
RDF_C4_SHUFFLE_ROWS=1 python "$REPO/bench.py" --workload c4 --rows 200000000 --steps 10 --warmup 3 --cpu-sample 0 --force-exchange --backend nccl 2>> "$OUT/rccl.err" | tail -1 >> "$OUT/rccl_one_rank.jsonl"
python "$REPO/bench.py" --workload c4 --total-rows 1000000000 --steps 10 --warmup 3 --cpu-sample 0 2>> "$OUT/rccl.err" | tail -1 > "$OUT/c4_total_rows_1e9.json"
for e in take_frame_random_1col take_frame_random_4col filter_frame_1col filter_frame_4col; do
    pmc frames_$e python "$REPO/tools/bench_frames.py" --steps 2 --only $e
done
# 3d. round 4: the same collectives through the torch.distributed harness (A/B against the library's own communicator, which the
#     lines above use by default), the N-GPU GROUP BY from plain C, the streamed batch loop over host-resident frames (16 GB; and
#     4 GB with 1.5 GB of HBM left), FETCH_SIZE on gather patterns, the bare streaming loop by grid size
rm -f "$OUT/rccl_one_rank_torch.jsonl"
for w in headline c4; do python "$REPO/bench.py" --workload $w --rows 200000000 --steps 10 --warmup 3 --cpu-sample 0 --force-exchange --backend nccl --comm torch 2>> "$OUT/rccl.err" | tail -1 >> "$OUT/rccl_one_rank_torch.jsonl"; done
gcc -std=c11 -pthread -I "$REPO/include" "$REPO/integration/example_dist.c" -L "$REPO/rust_dataframe_amd" -lrdf_mi355x -Wl,-rpath,"$REPO/rust_dataframe_amd" -o /tmp/example_dist && { /tmp/example_dist peer 4; /tmp/example_dist rccl; } > "$OUT/example_dist.txt" 2>&1
python "$REPO/tools/bench_stream.py" --gb 16 2> "$OUT/stream.err" | grep '"bench"' > "$OUT/stream_16GB.jsonl"
python "$REPO/tools/bench_stream.py" --gb 4 --hbm-left-gb 1.5 2>> "$OUT/stream.err" | grep '"bench"' > "$OUT/stream_4GB_hbm_left_1p5GB.jsonl"
if [ -x "$REPO/tools/ubench_stream.bin" ]; then timeout 300 "$REPO/tools/ubench_stream.bin" > "$OUT/ubench_stream.txt" 2>&1; fi
fi
# 3e. round 5: launch shapes of the specialised kernels (blocks per CU x tile walk) on data built once; the sinks that materialise,
#     host -> host, 16 GB and 4 GB with 1.5 GB of HBM left; vector-ALU instructions per row of the headline kernel with and without a
#     validity bitmap, this build against round 4's (rust_dataframe_amd/librdf_base_r04.so, when it travelled); the fused combine against
#     the pair of calls on the 1-rank communicator; C4 at the 8-GPU configuration's per-rank size
if [ "$PART" = "all" ] || [ "$PART" = "r5" ]; then
python "$REPO/tools/exp_tilewalk.py" --steps 20 2> "$OUT/tilewalk.err" > "$OUT/tilewalk.jsonl"
python "$REPO/tools/bench_stream_sinks.py" --gb 16 2> "$OUT/stream_sinks.err" | grep '"bench"' > "$OUT/stream_sinks_16GB.jsonl"
python "$REPO/tools/bench_stream_sinks.py" --gb 4 --hbm-left-gb 1.5 2>> "$OUT/stream_sinks.err" | grep '"bench"' > "$OUT/stream_sinks_4GB_hbm_left_1p5GB.jsonl"
for lib in librdf_mi355x.so librdf_base_r04.so; do
    [ -f "$REPO/rust_dataframe_amd/$lib" ] || continue
    for nf in 0 0.1; do
        RDF_LIB_PATH="$REPO/rust_dataframe_amd/$lib" timeout 600 rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d "$OUT/pmc_valu_${lib}_nf$nf" -o p -- \
            python "$REPO/bench.py" --steps 5 --warmup 1 --cpu-sample 0 --null-fraction $nf > "$OUT/pmc_valu_${lib}_nf$nf.out" 2> "$OUT/pmc_valu_${lib}_nf$nf.err"
    done
done
rm -f "$OUT/rccl_one_rank_fused_combine.jsonl"
for fc in "" 1; do RDF_BENCH_FUSED_COMBINE=$fc python "$REPO/bench.py" --rows 200000000 --steps 50 --warmup 5 --cpu-sample 0 --force-exchange --backend nccl 2>> "$OUT/rccl.err" | tail -1 >> "$OUT/rccl_one_rank_fused_combine.jsonl"; done
python "$REPO/tools/bench_kernels.py" --rows 125000000 --steps 5 --only groupby_sum_1000000_groups,groupby_count_1000000_groups,groupby_max_1000000_groups,groupby_sum_1000000_groups_zipf,groupby_sum_1000000_groups_scattered_keys,groupby_sum_1000000_groups_hot_key_30pct 2>> "$OUT/kernels.err" | grep kernel_ms > "$OUT/groupby_1p25e8_rows.jsonl"
RDF_LIB_PATH="$REPO/rust_dataframe_amd/librdf_base_r04.so" python "$REPO/tools/bench_kernels.py" --rows 125000000 --steps 5 --only groupby_sum_1000000_groups,groupby_count_1000000_groups 2>> "$OUT/kernels.err" | grep kernel_ms > "$OUT/groupby_1p25e8_rows_round4_build.jsonl"
python "$REPO/tools/exp_gb_window.py" --windows 8,32,64,128,256 --reps 2 2>> "$OUT/kernels.err" > "$OUT/gb_window.jsonl"
grep '"probe"' "$OUT/stream_sinks.err" | head -1 > "$OUT/link_probe.jsonl"       # each direction alone, both at once (bench_stream_sinks.py prints it first)
# (rdf_filter_frame on long batches: step 3f)
# equi-join: the scan-placed table (default) against the compare-and-swap table
python "$REPO/tools/bench_kernels.py" --rows 1000000000 --steps 3 --only join_inner_1e8_x_1e7,join_inner_1e8_x_1e7_cas_table,join_inner_1e8_x_1e8,join_inner_1e8_x_1e8_cas_table 2>> "$OUT/kernels.err" | grep kernel_ms > "$OUT/join_table_ab.jsonl"
if [ -x "$REPO/tools/ubench_streams.bin" ]; then timeout 300 "$REPO/tools/ubench_streams.bin" > "$OUT/ubench_streams.txt" 2>&1; fi
fi
# 3f. round 6: the one-pass filter on block tiles (rdf_bfilter.hip) against the wave-tile kernel by batch length, its HBM counters on ONE
#     batch of 1e9 rows, the mask-given form, the design probe (tools/ubench_compact.bin), the bare-stream probes the memory model is
#     fitted on (rdf_probe_stream), the reference's own benchmark shape, the take through page-sorted pairs, the two digit-pass kernels,
#     the interpreter's two kernels (tools/lean_ab.py)
if [ "$PART" = "all" ] || [ "$PART" = "r6" ]; then
rm -f "$OUT/filter_frame_long_batches.jsonl"
for cr in 1024 2048 4096 8192 16384 65536 1048576 16777216 1000000000; do for b in 1 0; do
    python "$REPO/tools/bench_frames.py" --rows 1000000000 --chunk-rows $cr --steps 5 --block $b --only filter_frame_1col,filter_frame_2col,filter_frame_4col 2>> "$OUT/frames.err" | grep kernel_ms | sed "s/^{/{\"chunk_rows\": $cr, \"filter_block\": $b, /" >> "$OUT/filter_frame_long_batches.jsonl"
done; done
pmc frames_long_filter_frame_1col python "$REPO/tools/bench_frames.py" --rows 1000000000 --chunk-rows 1000000000 --steps 2 --only filter_frame_1col
pmc frames_long_filter_frame_4col python "$REPO/tools/bench_frames.py" --rows 1000000000 --chunk-rows 1000000000 --steps 2 --only filter_frame_4col
python "$REPO/bench.py" --workload ref_bench 2>> "$OUT/bench_1e9.err" | tail -1 > "$OUT/ref_bench.json"
python - > "$OUT/probe_stream.jsonl" 2>> "$OUT/kernels.err" <<PY
import json, sys
sys.path.insert(0, "$REPO")
import torch
from rust_dataframe_amd import lib
lib.set_device(0)
n = 1_000_000_000
a = torch.empty(n, dtype=torch.float64, device="cuda"); b = torch.empty_like(a); c = torch.empty_like(a)
a.uniform_(); b.uniform_(); torch.cuda.synchronize()
for kind, name in ((0, "read"), (1, "copy"), (2, "two reads + one write")):
    g, shape = lib.probe_stream(kind, a.data_ptr(), b.data_ptr(), c.data_ptr(), n * 8, 7)
    print(json.dumps({"probe": name, "bytes_per_stream": n * 8, "GBps_of_bytes_moved": round(g, 1), "frac_of_8TBps": round(g / 8000.0, 3), "shape": shape}))
PY
mkdir -p "$OUT/model"; rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/model" -o m -- python "$REPO/tools/model_check.py" > "$OUT/memory_model_same_box.jsonl" 2>> "$OUT/kernels.err"
cp "$(find "$OUT/model" -name '*kernel_stats.csv' | head -1)" "$OUT/memory_model_same_box_kernel_stats.csv" 2>/dev/null
if [ -x "$REPO/tools/ubench_compact.bin" ]; then timeout 300 "$REPO/tools/ubench_compact.bin" 1e9 7 > "$OUT/ubench_compact.jsonl" 2>> "$OUT/kernels.err"; fi
if [ -x "$REPO/tools/ubench_take_binned.bin" ]; then timeout 300 "$REPO/tools/ubench_take_binned.bin" > "$OUT/ubench_take_binned.jsonl" 2>> "$OUT/kernels.err"; fi
# the interpreter: eval_lean_kernel against eval_kernel, 400 random programs compared field by field + both kernels timed on 1e9 rows
python "$REPO/tools/lean_ab.py" --programs 400 --seed 21 --time 2>> "$OUT/kernels.err" | tail -1 > "$OUT/interpreter_lean_ab.json"
fi
# 4. the scatter micro-benchmark behind the C4 bound
if [ -x "$REPO/tools/ubench_scatter.bin" ]; then timeout 300 "$REPO/tools/ubench_scatter.bin" > "$OUT/ubench_scatter.txt" 2>&1; fi
# keep the merged payload small: the raw traces stay on the box, the stats / counter CSVs travel
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*agent_info.csv" -delete
ls -la "$OUT" | head -60
