set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_finalscale.py tests/test_gpu_host_tiling.py tests/test_gpu_pipe.py -m gpu -x -q 2>&1 | tail -6
