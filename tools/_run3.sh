set -u
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_devmath.py tests/test_gpu_nlmeans.py tests/test_gpu_diffuse.py tests/test_gpu_denoiseprofile.py -q -m gpu > gpurun_out/r3/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r3/tests.log
tail -4 gpurun_out/r3/tests.log
timeout 120 tools/valu_clock_microbench > gpurun_out/r3/valu_issue_cycles.json 2> gpurun_out/r3/valu.err; echo "valu rc=$?"
timeout 300 python tools/bench_module.py nlmeans --size 24MP > gpurun_out/r3/mod_nlm24.json 2>&1; echo "nlm24 rc=$?"
timeout 300 python tools/bench_module.py nlmeans --size 7952x5304 > gpurun_out/r3/mod_nlm42.json 2>&1; echo "nlm42 rc=$?"
timeout 300 python tools/bench_module.py diffuse --size 100MP --iterations 2 > gpurun_out/r3/mod_diffuse.json 2>&1; echo "diffuse rc=$?"
grep -h 'ms_total\|"nlm\|"diffuse' gpurun_out/r3/mod_*.json
