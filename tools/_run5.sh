set -u
mkdir -p gpurun_out/r5
export TMPDIR=/tmp
for r in 1 2 3; do for sz in 24MP 100MP; do timeout 300 python tools/bench_module.py nlmeans --size $sz --radius $r > gpurun_out/r5/nlm_${sz}_r$r.json 2>&1; echo "$sz r$r rc=$?"; done; done
ANSEL_HIP_LIB=ansel_amd/libansel_hip_measuring.so ANSEL_HIP_NLM_V2=1 timeout 300 python tools/bench_module.py nlmeans --size 100MP --radius 2 > gpurun_out/r5/nlm_100MP_r2_v2.json 2>&1
timeout 300 python tools/bench_module.py denoiseprofile_nlm --size 24MP > gpurun_out/r5/dn_nlm_24MP.json 2>&1; echo "dn nlm rc=$?"
grep -H 'ms_total' gpurun_out/r5/*.json
