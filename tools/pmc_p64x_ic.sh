#!/bin/bash
# instruction-cache counters of the kernel variants inside tools/p64x_bench.  Run on the GPU box: bash tools/pmc_p64x_ic.sh <tag>
set -u
TAG=${1:-ic}
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RUN="$REPO/tools/p64x_bench pair"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_STALL SQ_IFETCH" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F SQ_INSTS_VALU_MUL_F SQ_INSTS_VALU_FMA_F SQ_INSTS_VALU_INT SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  timeout 300 rocprofv3 --output-format csv --pmc $set -d $OUT/p$i -o pmc -- $RUN > $OUT/p$i.log 2>&1
done
cd $REPO
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "LIB" if "regtile64p" in k else ("EXP" if "24, 0>" in k else ("BASE" if "false, 0, 0>" in k else None))
        if name: agg[r["Counter_Name"]][name].append(float(r["Counter_Value"]))
print(f"{'counter':40s} {'BASE (r02)':>16s} {'EXP early+late':>16s} {'LIB':>16s}   LIB/EXP  BASE/EXP")
for c in sorted(agg):
    v = {n: (sum(x[1:]) / max(1, len(x) - 1) if len(x) > 1 else (x[0] if x else float('nan'))) for n, x in agg[c].items()}
    b, e, l = v.get("BASE", float('nan')), v.get("EXP", float('nan')), v.get("LIB", float('nan'))
    print(f"{c:40s} {b:16.0f} {e:16.0f} {l:16.0f}   {l / e if e else float('nan'):.3f}   {b / e if e else float('nan'):.3f}")
PY
