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
# SQ counters of the same command (own passes, no trace): instruction counts per class, waits, LDS conflicts
if [ "${2:-}" = "sq" ]; then
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU \
    --output-format csv -d "$OUT/sqa" -- $CMD > "$OUT/sqa.log" 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 \
    --output-format csv -d "$OUT/sqb" -- $CMD > "$OUT/sqb.log" 2>&1
  A=$(find "$OUT/sqa" -name '*counter_collection.csv' | head -1)
  B=$(find "$OUT/sqb" -name '*counter_collection.csv' | head -1)
  [ -n "$A" ] && python tools/pmc_sq_json.py "$OUT/pmc_sq.json" "tools/bench_module.py amaze --size ${1:-24MP};" $A $B > "$OUT/pmc_sq.txt"
  rm -rf "$OUT/sqa" "$OUT/sqb"
fi
