mkdir -p gpurun_out/rA
timeout 600 python -m pytest tests/test_gpu_nlmeans.py -q -m gpu -x > gpurun_out/rA/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/rA/tests.log
for sz in 100MP 60MP 24MP; do timeout 300 python tools/bench_module.py nlmeans --size $sz --steps 4 > gpurun_out/rA/nlm_$sz.json 2>&1; echo "$sz $(grep -A1 '"nlm_' gpurun_out/rA/nlm_$sz.json | grep ms_total | tr -d ' \n')"; done
