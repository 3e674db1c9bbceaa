set -u
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_devmath.py tests/test_gpu_nlmeans.py -q -m gpu > gpurun_out/r2/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r2/tests.log
tail -4 gpurun_out/r2/tests.log
timeout 600 python -m pytest tests/test_gpu_parity_at_size.py -q -m gpu -k "full_pipe_24MP" -s > gpurun_out/r2/atsize24.log 2>&1; echo "atsize rc=$?" | tee -a gpurun_out/r2/atsize24.log
tail -4 gpurun_out/r2/atsize24.log
timeout 120 tools/valu_clock_microbench > gpurun_out/r2/valu_issue_cycles.json 2> gpurun_out/r2/valu.err; echo "valu rc=$?"
timeout 300 python tools/bench_module.py nlmeans --size 24MP > gpurun_out/r2/mod_nlm24.json 2>&1; echo "nlm24 rc=$?"
ANSEL_HIP_LIB=ansel_amd/libansel_hip_measuring.so ANSEL_HIP_NLM_V2=1 timeout 300 python tools/bench_module.py nlmeans --size 24MP > gpurun_out/r2/mod_nlm24_v2.json 2>&1; echo "nlm24 v2 rc=$?"
timeout 300 python tools/bench_module.py nlmeans --size 7952x5304 > gpurun_out/r2/mod_nlm42.json 2>&1; echo "nlm42 rc=$?"
timeout 300 python tools/bench_module.py nlmeans --size 100MP > gpurun_out/r2/mod_nlm100.json 2>&1; echo "nlm100 rc=$?"
grep -h ms_total gpurun_out/r2/mod_nlm*.json
