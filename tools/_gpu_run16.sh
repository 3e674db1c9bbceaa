set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_nlmeans.py -m gpu -x -q > gpurun_out/r02o_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02o_pytest.log; tail -4 gpurun_out/r02o_pytest.log
timeout 300 python tools/nlm_variants.py > gpurun_out/r02o_nlm_variants.json 2> gpurun_out/r02o_nlm_variants.txt; head -16 gpurun_out/r02o_nlm_variants.txt
