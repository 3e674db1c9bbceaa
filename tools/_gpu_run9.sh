set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_denoiseprofile.py tests/test_gpu_tiled.py tests/test_gpu_edge_cases.py -m gpu -x -q > gpurun_out/r02i_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02i_pytest.log; tail -6 gpurun_out/r02i_pytest.log
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["config"]["full_pipe"]["ms_per_step"], d["config"]["full_pipe"]["kernels_ms_per_step"])'
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-legs --no-verify > gpurun_out/r02i_bench.log 2>&1; tail -1 gpurun_out/r02i_bench.log | python -c "$P"
ANSEL_HIP_DN_PER_ROW=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-legs --no-verify > gpurun_out/r02i_bench_per_row.log 2>&1; tail -1 gpurun_out/r02i_bench_per_row.log | python -c "$P"
