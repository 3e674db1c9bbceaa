set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 bash tools/profile_round.sh r02c --steps 5 --warmup 2 > gpurun_out/r02c_profile_round.log 2>&1
tail -3 gpurun_out/r02c_profile_round.log
timeout 120 tools/valu_microbench > gpurun_out/r02c_valu.json 2> gpurun_out/r02c_valu.err; echo "valu rc=$?"
python tools/valu_model.py gpurun_out/prof_r02c/pmc_sq.json gpurun_out/r02c_valu.json 101756928 gpurun_out/r02c_isa_mix.json
