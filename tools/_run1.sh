set -x
export RDF_COMM_TIMEOUT_S=60
timeout 900 python -m pytest tests/test_comm_gpu.py -m gpu -x -q 2>&1 | tail -30
timeout 600 python -m pytest tests/test_abi.py tests/test_cpp_frame.py -m gpu -x -q 2>&1 | tail -30
mkdir -p gpurun_out/r04a
timeout 300 python bench.py --steps 5 --warmup 2 --rows 200000000 --force-exchange --cpu-sample 0 > gpurun_out/r04a/headline_native.json 2> gpurun_out/r04a/headline_native.err; echo rc=$?; tail -3 gpurun_out/r04a/headline_native.err
timeout 300 python bench.py --workload c4 --steps 3 --warmup 1 --rows 200000000 --force-exchange > gpurun_out/r04a/c4_native.json 2> gpurun_out/r04a/c4_native.err; echo rc=$?; tail -3 gpurun_out/r04a/c4_native.err
RDF_C4_SHUFFLE_ROWS=1 timeout 300 python bench.py --workload c4 --steps 3 --warmup 1 --rows 200000000 --force-exchange > gpurun_out/r04a/c4_native_rows.json 2> gpurun_out/r04a/c4_native_rows.err; echo rc=$?; tail -3 gpurun_out/r04a/c4_native_rows.err
timeout 300 python bench.py --workload c4 --steps 3 --warmup 1 --rows 200000000 --force-exchange --comm torch > gpurun_out/r04a/c4_torch.json 2> gpurun_out/r04a/c4_torch.err; echo rc=$?
timeout 300 python bench.py --workload q1 --steps 3 --warmup 1 --rows 100000000 --force-exchange > gpurun_out/r04a/q1_native.json 2> gpurun_out/r04a/q1_native.err; echo rc=$?
cat gpurun_out/r04a/*.json | cut -c1-1500
