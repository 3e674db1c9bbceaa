set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_nlmeans.py tests/test_gpu_host_tiling.py tests/test_gpu_tiled.py tests/test_gpu_parity_at_size.py tests/test_gpu_pipe.py tests/test_gpu_denoiseprofile.py -m gpu -x -q > gpurun_out/r02g_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02g_pytest.log; tail -6 gpurun_out/r02g_pytest.log
timeout 300 python tools/nlm_variants.py > gpurun_out/r02g_nlm_variants.json 2> gpurun_out/r02g_nlm_variants.txt; head -5 gpurun_out/r02g_nlm_variants.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-legs --no-verify > gpurun_out/r02g_bench.log 2>&1; tail -1 gpurun_out/r02g_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['full_pipe']['ms_per_step'], d['config']['full_pipe']['kernels_ms_per_step'])"
