set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 bash tools/profile_round.sh r02d --steps 5 --warmup 2 > gpurun_out/r02d_profile_round.log 2>&1
tail -2 gpurun_out/r02d_profile_round.log
python tools/valu_model.py gpurun_out/prof_r02d/pmc_sq.json profiles/r02_valu_issue_cycles.json 101756928 gpurun_out/r02d_isa_mix.json
