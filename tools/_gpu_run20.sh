set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/nlm_phase_clocks.py > gpurun_out/r02_nlm_phase_clocks.json 2> gpurun_out/r02_nlm_phase_clocks.txt; cat gpurun_out/r02_nlm_phase_clocks.txt | head -60

