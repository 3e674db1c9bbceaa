set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_demosaic.py tests/test_gpu_tiled.py tests/test_gpu_edge_sizes.py tests/test_gpu_pipe.py -m gpu -x -q > gpurun_out/r02k_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02k_pytest.log; tail -6 gpurun_out/r02k_pytest.log
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"].get("kernels_ms_per_step"))'
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-legs --no-verify --no-full-pipe > gpurun_out/r02k_bench.log 2>&1; tail -1 gpurun_out/r02k_bench.log | python -c "$P"
ANSEL_HIP_RCD_V1=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-legs --no-verify --no-full-pipe > gpurun_out/r02k_bench_v1.log 2>&1; tail -1 gpurun_out/r02k_bench_v1.log | python -c "$P"
