#!/bin/bash
# L2 <-> memory request mix of the headline kernel (separate --pmc passes, counters only).  Run on the GPU box: bash tools/pmc_ea.sh <tag> [run_mix args]
set -u
TAG=${1:-ea}; shift || true
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RUN="python $REPO/tools/run_mix.py $*"
rocprofv3 --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $OUT/rd -o pmc -- $RUN > $OUT/rd.log 2>&1
rocprofv3 --output-format csv --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum -d $OUT/wr -o pmc -- $RUN > $OUT/wr.log 2>&1
rocprofv3 --output-format csv --pmc TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum -d $OUT/st -o pmc -- $RUN > $OUT/st.log 2>&1
cd $REPO
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spectre_mix" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg): print(f"{k:42s} {sum(agg[k]) / len(agg[k]):16.0f}   (per launch, n={len(agg[k])})")
PY
