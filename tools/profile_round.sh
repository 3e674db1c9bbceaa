#!/bin/bash
# Run ON THE GPU BOX (gpurun): bench line + rocprofv3 kernel stats + HBM byte counters + SQ counters for one
# bench.py configuration.  Outputs under gpurun_out/prof_<tag>/; copy what is to be judged into profiles/.
#   tools/profile_round.sh <tag> [bench.py args...]
# The counters are collected in their own runs, never together with a trace (gpurun refuses that combination).
set -u
TAG="$1"; shift
OUT="gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-host-legs --no-verify --no-light-pipe"
python bench.py "$@" > "$OUT/bench.log" 2>&1
tail -1 "$OUT/bench.log"
# the timed region alone (the pipe bench.py's `value` is quoted on, --pipe full unless the arguments say otherwise): kernel
# stats and the HBM byte counters bench.py's roofline.traffic / kernel_bounds cite
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python bench.py "$@" $Q > "$OUT/stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python bench.py "$@" $Q --steps 2 --warmup 1 > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python bench.py "$@" $Q --steps 2 --warmup 1 > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU \
  --output-format csv -d "$OUT/sqa" -- python bench.py "$@" $Q --steps 3 --warmup 1 > "$OUT/sqa.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 \
  --output-format csv -d "$OUT/sqb" -- python bench.py "$@" $Q --steps 3 --warmup 1 > "$OUT/sqb.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d "$OUT/sqc" -- python bench.py "$@" $Q --steps 3 --warmup 1 > "$OUT/sqc.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/grbm" -- python bench.py "$@" $Q --steps 3 --warmup 1 > "$OUT/grbm.log" 2>&1
G=$(find "$OUT/grbm" -name '*counter_collection.csv' | head -1)
[ -n "$G" ] && python tools/pmc_sclk_json.py "$G" "$OUT/sclk_per_kernel.json" "bench.py $* (the timed steps only);" > "$OUT/sclk_per_kernel.txt"
rm -rf "$OUT/grbm"
F=$(find "$OUT/fetch" -name '*counter_collection.csv' | head -1)
W=$(find "$OUT/write" -name '*counter_collection.csv' | head -1)
S=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
[ -n "$S" ] && cp "$S" "$OUT/kernel_stats.csv"
[ -n "$F" ] && [ -n "$W" ] && python tools/pmc_hbm_json.py "$F" "$W" "$OUT/pmc_hbm_bytes.json" "bench.py $*;"
A=$(find "$OUT/sqa" -name '*counter_collection.csv' | head -1)
B=$(find "$OUT/sqb" -name '*counter_collection.csv' | head -1)
C=$(find "$OUT/sqc" -name '*counter_collection.csv' | head -1)
[ -n "$A" ] && python tools/pmc_sq_json.py "$OUT/pmc_sq.json" "bench.py $* (the timed steps only);" $A $B $C > "$OUT/pmc_sq.txt"
# the kernels' VALU issue floors: the instruction mix of the SQ passes at the architectural issue rates (tools/valu_model.py ... arch)
PIX=$(python - "$@" <<'PYEOF'
import sys
sys.path.insert(0, ".")
from ansel_amd import synth
size = "100MP"
a = sys.argv[1:]
if "--size" in a:
    size = a[a.index("--size") + 1]
w, h = synth.SIZES[size] if size in synth.SIZES else map(int, size.lower().split("x"))
print(w * h)
PYEOF
)
[ -f "$OUT/pmc_sq.json" ] && python tools/valu_model.py "$OUT/pmc_sq.json" arch "$PIX" "$OUT/isa_mix.json" > "$OUT/isa_mix.txt" 2>&1
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write" "$OUT/sqa" "$OUT/sqb" "$OUT/sqc"
ls -la "$OUT"
