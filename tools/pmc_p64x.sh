#!/bin/bash
# PMC comparison of kernel variants inside tools/p64x_bench (separate --pmc passes, counters only).  Run on the GPU box: bash tools/pmc_p64x.sh <tag>
set -u
TAG=${1:-p64x}
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RUN="$REPO/tools/p64x_bench pair"
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" \
           "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_sum"; do
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
        name = "LIB" if "regtile64p" in k else ("EXP" if "ELi24E" in k else ("BASE" if "p64x" in k else None))
        if name: agg[r["Counter_Name"]][name].append(float(r["Counter_Value"]))
print(f"{'counter':40s} {'BASE (r02)':>16s} {'EXP early+late':>16s} {'LIB':>16s}   LIB/EXP")
for c in sorted(agg):
    v = {n: (sum(x[1:]) / max(1, len(x) - 1) if len(x) > 1 else (x[0] if x else float('nan'))) for n, x in agg[c].items()}
    b, e, l = v.get("BASE", float('nan')), v.get("EXP", float('nan')), v.get("LIB", float('nan'))
    print(f"{c:40s} {b:16.0f} {e:16.0f} {l:16.0f}   {l / e if e else float('nan'):.3f}")
PY
