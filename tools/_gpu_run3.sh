set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_nlmeans.py tests/test_gpu_denoiseprofile.py tests/test_gpu_diffuse.py tests/test_gpu_pipe.py tests/test_gpu_tiled.py tests/test_gpu_edge_sizes.py tests/test_gpu_host_tiling.py tests/test_golden.py tests/test_gpu_parity_at_size.py -m gpu -x -q > gpurun_out/r02c_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02c_pytest.log; tail -8 gpurun_out/r02c_pytest.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-legs --no-verify > gpurun_out/r02c_bench.log 2>&1; tail -1 gpurun_out/r02c_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['full_pipe']['ms_per_step'], d['config']['full_pipe']['kernels_ms_per_step'])"
ANSEL_HIP_NLM_V1=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-legs --no-verify > gpurun_out/r02c_bench_nlmv1.log 2>&1; tail -1 gpurun_out/r02c_bench_nlmv1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['full_pipe']['ms_per_step'], d['config']['full_pipe']['kernels_ms_per_step'])"
