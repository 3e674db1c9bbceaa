set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/nlm_variants.py 6000x4000 > gpurun_out/r02e_nlm_variants.json 2> gpurun_out/r02e_nlm_variants.txt; cat gpurun_out/r02e_nlm_variants.txt
timeout 900 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_c_example.py tests/test_gpu_runtime.py tests/test_gpu_batch.py -m gpu -x -q > gpurun_out/r02e_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02e_pytest.log; tail -12 gpurun_out/r02e_pytest.log
