set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_diffuse.py tests/test_gpu_tiled.py tests/test_gpu_edge_sizes.py -m gpu -x -q > gpurun_out/r02q_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02q_pytest.log; tail -4 gpurun_out/r02q_pytest.log
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"]); print(d["config"]["full_pipe"]["ms_per_step"], d["config"]["full_pipe"]["kernels_ms_per_step"])'
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-legs --no-verify > gpurun_out/r02q_bench.log 2>&1; tail -1 gpurun_out/r02q_bench.log | python -c "$P"
ANSEL_HIP_PDE_PER_ROW=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-legs --no-verify > gpurun_out/r02q_bench_per_row.log 2>&1; tail -1 gpurun_out/r02q_bench_per_row.log | python -c "$P"
