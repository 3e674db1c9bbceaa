set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r02u_pytest_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02u_pytest_all.log; tail -3 gpurun_out/r02u_pytest_all.log
timeout 1500 bash tools/profile_round.sh r02e --steps 5 --warmup 2 > gpurun_out/r02e_profile_round.log 2>&1
python tools/valu_model.py gpurun_out/prof_r02e/pmc_sq.json profiles/r02_valu_issue_cycles.json 101756928 gpurun_out/r02e_isa_mix.json | head -12
tail -1 gpurun_out/prof_r02e/bench.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["valu_frac"], d["cpu_baseline"]["value"]); print(d["config"]["full_pipe"]["ms_per_step"], d["config"]["full_pipe"]["hbm_frac"], d["config"]["full_pipe"]["kernels_ms_per_step"])'
