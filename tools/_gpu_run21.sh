set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_nlmeans.py tests/test_gpu_denoiseprofile.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/nlm_phase_clocks.py > gpurun_out/r02_nlm_phase_clocks_13waves.json 2> gpurun_out/r02_nlm_phase_clocks_13waves.txt; sed -n 2,19p gpurun_out/r02_nlm_phase_clocks_13waves.txt
timeout 300 python tools/nlm_variants.py > gpurun_out/r02r_nlm_variants.json 2> gpurun_out/r02r_nlm_variants.txt; sed -n 2,4p gpurun_out/r02r_nlm_variants.txt
