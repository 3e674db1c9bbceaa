set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tiled.py tests/test_boundary_compile.py tests/test_gpu_blend.py -m gpu -x -q > gpurun_out/r02h_pytest_new.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02h_pytest_new.log; tail -8 gpurun_out/r02h_pytest_new.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02h_pytest_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02h_pytest_all.log; tail -8 gpurun_out/r02h_pytest_all.log
timeout 600 python tools/cpu_baseline_sweep.py 24MP > gpurun_out/r02h_cpu_sweep.txt 2>&1; cat gpurun_out/r02h_cpu_sweep.txt
