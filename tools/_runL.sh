set -u
mkdir -p gpurun_out/rL
export TMPDIR=/tmp
for rep in 1 2; do
for st in 32 64 128 48; do
  ANSEL_HIP_LIB=ansel_amd/libansel_hip_measuring.so ANSEL_HIP_PDE_STRIP=$st python tools/bench_module.py diffuse --size 100MP --preset lens_deblur_soft --iterations 2 > gpurun_out/rL/strip_$st.json 2>&1
  echo "strip $st: $(grep -A1 '"diffuse_pde"' gpurun_out/rL/strip_$st.json | grep ms_total)"
done
done
