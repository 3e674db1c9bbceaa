#!/bin/bash
# Run ON THE GPU BOX (gpurun): bench line + rocprofv3 kernel stats + HBM byte counters for one bench.py
# configuration.  Outputs under gpurun_out/prof_<tag>/; copy what is to be judged into profiles/.
#   tools/profile_round.sh <tag> [bench.py args...]
set -u
TAG="$1"; shift
OUT="gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py "$@" > "$OUT/bench.log" 2>&1
tail -1 "$OUT/bench.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py "$@" --no-cpu-baseline --no-host-legs > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python bench.py "$@" --no-cpu-baseline --no-host-legs > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python bench.py "$@" --no-cpu-baseline --no-host-legs > "$OUT/write.log" 2>&1
F=$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1)
W=$(find "$OUT/write" -name '*counter_collection.csv' | head -1)
S=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
[ -n "$S" ] && cp "$S" "$OUT/kernel_stats.csv"
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_hbm_json.py "$F" "$W" "$OUT/pmc_hbm_bytes.json" "bench.py $*;"
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write"
ls -la "$OUT"
