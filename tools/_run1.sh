cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_frame_ops_gpu.py tests/test_parity_gpu.py -m gpu -x -q -k "sort or join" 2>&1 | tail -3
timeout 300 bash tools/sortprof.sh tools/bench_kernels.py --rows 50000000 --steps 5 --only sort_to_indices_i64_full_range 2>&1 | head -4
timeout 300 python tools/bench_kernels.py --rows 1000000000 --steps 3 --only sort_to_indices_i64,sort_to_indices_i64_full_range,sort_to_indices_i64_full_range_1e9 2>/dev/null | cut -c1-200
