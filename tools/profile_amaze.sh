#!/bin/bash
# Run ON THE GPU BOX (gpurun): the AMaZE demosaic of a 24 MP frame on its own -- rocprofv3 kernel stats and the HBM byte
# counters (separate passes) -> gpurun_out/prof_amaze/{kernel_stats.csv,pmc_hbm_bytes.json,bench.log}
set -u
OUT="gpurun_out/prof_amaze"
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
CMD="python tools/bench_module.py amaze --size ${1:-24MP} --steps 5"
$CMD > "$OUT/bench.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1
F=$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1)
W=$(find "$OUT/write" -name '*counter_collection.csv' | head -1)
S=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
[ -n "$S" ] && cp "$S" "$OUT/kernel_stats.csv"
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_hbm_json.py "$F" "$W" "$OUT/pmc_hbm_bytes.json" "tools/bench_module.py amaze --size ${1:-24MP};"
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write"
ls -la "$OUT"
