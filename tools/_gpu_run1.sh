set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
nproc; free -g | head -2
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r02a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a_pytest.log; tail -15 gpurun_out/r02a_pytest.log
timeout 120 tools/valu_microbench > gpurun_out/r02a_valu.json 2> gpurun_out/r02a_valu.err; echo "valu rc=$?"
timeout 900 tools/profile_round.sh r02a --steps 10 --warmup 3
