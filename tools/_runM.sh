set -u
mkdir -p gpurun_out/rM
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_diffuse.py -q -x > gpurun_out/rM/diffuse_tests.log 2>&1; echo "diffuse rc=$?"; tail -2 gpurun_out/rM/diffuse_tests.log
for rep in 1 2 3; do
for off in 0 32; do
  ANSEL_HIP_LIB=ansel_amd/libansel_hip_measuring.so ANSEL_HIP_PDE_OFF=$off python tools/bench_module.py diffuse --size 100MP --preset lens_deblur_soft --iterations 2 > gpurun_out/rM/rot_$off.json 2>&1
  echo "off $off: $(grep -A1 '"diffuse_pde"' gpurun_out/rM/rot_$off.json | grep ms_total)"
done
done
