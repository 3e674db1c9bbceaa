set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/nlm_phase_clocks.py > gpurun_out/r02_nlm_phase_clocks_13waves.json 2> gpurun_out/r02_nlm_phase_clocks_13waves.txt
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"]); print(d["config"]["full_pipe"]["ms_per_step"], d["config"]["full_pipe"]["kernels_ms_per_step"])'
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-legs --no-verify > gpurun_out/r02s_bench.log 2>&1; tail -1 gpurun_out/r02s_bench.log | python -c "$P"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02s_pytest_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02s_pytest_all.log; tail -3 gpurun_out/r02s_pytest_all.log
