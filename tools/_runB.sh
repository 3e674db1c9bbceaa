mkdir -p gpurun_out/rB
ANSEL_HIP_LIB=ansel_amd/libansel_hip_measuring.so timeout 300 python tools/pde_off_ab.py --size 100MP > gpurun_out/rB/pde_off.json 2>&1; echo rc=$?
cat gpurun_out/rB/pde_off.json
