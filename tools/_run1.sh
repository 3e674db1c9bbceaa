set -u
mkdir -p gpurun_out/r1
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_devmath.py tests/test_gpu_diffuse.py tests/test_gpu_denoiseprofile.py -x -q -m gpu > gpurun_out/r1/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r1/tests.log
tail -5 gpurun_out/r1/tests.log
timeout 120 tools/valu_clock_microbench > gpurun_out/r1/valu_issue_cycles.json 2> gpurun_out/r1/valu.err; echo "valu rc=$?"
timeout 300 python tools/bench_module.py diffuse --size 100MP --iterations 2 > gpurun_out/r1/mod_diffuse.json 2>&1; echo "diffuse rc=$?"
timeout 300 python tools/bench_module.py denoiseprofile --size 100MP > gpurun_out/r1/mod_dn.json 2>&1; echo "dn rc=$?"
ANSEL_HIP_LIB=ansel_amd/libansel_hip_measuring.so timeout 300 python tools/pde_div_ab.py --size 100MP > gpurun_out/r1/pde_div_ab.json 2>&1; echo "ab rc=$?"
timeout 400 python bench.py > gpurun_out/r1/bench.log 2>&1; echo "bench rc=$?"
tail -c 1500 gpurun_out/r1/bench.log
