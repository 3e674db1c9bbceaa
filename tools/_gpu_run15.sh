set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_blend.py -m gpu -x -q > gpurun_out/r02n_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02n_pytest.log; tail -12 gpurun_out/r02n_pytest.log
