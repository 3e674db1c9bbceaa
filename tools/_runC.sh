mkdir -p gpurun_out/rC
for rep in 1 2; do
ANSEL_HIP_LIB=ansel_amd/libansel_hip_measuring.so ANSEL_HIP_DIFFUSE_HF_PLANES=1 ANSEL_HIP_PDE_LATE=1 timeout 300 python tools/bench_module.py diffuse --size 100MP --iterations 2 --steps 5 > gpurun_out/rC/hf_late$rep.json 2>&1; echo "hf late$rep $(grep -A1 '"diffuse_pde"' gpurun_out/rC/hf_late$rep.json | tail -1)"
ANSEL_HIP_LIB=ansel_amd/libansel_hip_measuring.so ANSEL_HIP_DIFFUSE_HF_PLANES=1 timeout 300 python tools/bench_module.py diffuse --size 100MP --iterations 2 --steps 5 > gpurun_out/rC/hf_early$rep.json 2>&1; echo "hf early$rep $(grep -A1 '"diffuse_pde"' gpurun_out/rC/hf_early$rep.json | tail -1)"
done
