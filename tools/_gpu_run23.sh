set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/nlm_variants.py > gpurun_out/r02t_nlm_variants.json 2> gpurun_out/r02t_nlm_variants.txt; sed -n 2,8p gpurun_out/r02t_nlm_variants.txt
