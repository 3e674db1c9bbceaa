set -u
mkdir -p gpurun_out/r4
export TMPDIR=/tmp
bash tools/profile_round.sh r05_full > gpurun_out/r4/profile_round.log 2>&1; echo "profile rc=$?"
tail -3 gpurun_out/r4/profile_round.log
timeout 300 python bench.py --pipe light --size 24MP --no-cpu-baseline > gpurun_out/r4/bench_config2_24MP.log 2>&1; echo "c2 rc=$?"
timeout 400 python bench.py --pipe denoise --size 60MP --no-cpu-baseline > gpurun_out/r4/bench_config3_60MP.log 2>&1; echo "c3 rc=$?"
timeout 400 python bench.py --pipe full --size 45MP --no-cpu-baseline > gpurun_out/r4/bench_config5_frame_45MP.log 2>&1; echo "c5 rc=$?"
timeout 400 python bench.py --pipe denoise --size 24MP --no-cpu-baseline > gpurun_out/r4/bench_denoise_24MP.log 2>&1; echo "d24 rc=$?"
timeout 400 python bench.py --mode bands --bands 8 --gpus 1 --no-cpu-baseline > gpurun_out/r4/bench_bands8_n1.log 2>&1; echo "bands rc=$?"
timeout 400 python bench.py --mode tiled --gpus 1 --no-cpu-baseline > gpurun_out/r4/bench_tiled_n1.log 2>&1; echo "tiled rc=$?"
for f in gpurun_out/r4/bench_*.log; do echo "$f"; tail -1 "$f" | cut -c1-300; done
