#!/bin/bash
# Run ON THE GPU BOX: the hardware's own utilisation figures for the kernels of the bench (rocprofv3 derived metrics VALUBusy,
# SALUBusy, MemUnitBusy, LDSBankConflict ... = counter formulas of the profiler's metric definitions), one pass each.
#   tools/valu_busy.sh > gpurun_out/valu_busy.txt
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-host-legs --no-verify --no-light-pipe --steps 2 --warmup 1"
for M in VALUBusy SALUBusy MemUnitBusy MemUnitStalled LDSBankConflict VALUUtilization; do
  rm -rf gpurun_out/vb_$M
  rocprofv3 --pmc $M --output-format csv -d gpurun_out/vb_$M -- python bench.py $Q > gpurun_out/vb_$M.log 2>&1
done
python - <<'PY'
import csv, glob, collections, re
def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("ansel::", "")
    m = re.match(r"(?:void\s+)?([A-Za-z_0-9]+(?:<[^>(]*>)?)", name)
    return m.group(1) if m else name[:48]
tab = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/vb_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            tab[k][c] = sum(v) / len(v)
cols = sorted({c for v in tab.values() for c in v})
print("%-40s" % "kernel" + "".join("%18s" % c for c in cols))
for k, v in sorted(tab.items(), key=lambda kv: -kv[1].get("VALUBusy", 0)):
    print("%-40s" % k[:40] + "".join("%18.2f" % v.get(c, float("nan")) for c in cols))
PY
rm -rf gpurun_out/vb_*/
