set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/band_overhead.py --size 100MP --pipe denoise > gpurun_out/r02_band_overhead_100MP_denoise.jsonl 2> gpurun_out/r02_band_overhead_denoise.err; cat gpurun_out/r02_band_overhead_100MP_denoise.jsonl
timeout 600 python tools/band_overhead.py --size 100MP --pipe light > gpurun_out/r02_band_overhead_100MP_light.jsonl 2> gpurun_out/r02_band_overhead_light.err; cat gpurun_out/r02_band_overhead_100MP_light.jsonl
