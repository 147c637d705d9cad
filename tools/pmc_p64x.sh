#!/bin/bash
# PMC comparison of kernel variants inside tools/p64x_bench (separate --pmc passes, counters only; every variant runs 6 x 8 launches back to
# back so that the counters describe the steady state).  Run on the GPU box: bash tools/pmc_p64x.sh <tag>
set -u
TAG=${1:-p64x}
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RUN="$REPO/tools/p64x_bench pair"
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" \
           "TCC_NORMAL_WRITEBACK_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --output-format csv --pmc $set -d $OUT/p$i -o pmc -- $RUN > $OUT/p$i.log 2>&1
done
cd $REPO
python - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
def short(k):
    m = re.search(r"spectre_mix_(p64x|regtile64p)<([^>]*)>", k)
    if not m: return None
    a = [x.strip() for x in m.group(2).split(",")]
    return ("exp" if m.group(1) == "p64x" else "LIB") + f"({a[0]},{a[1]})" + ("" if m.group(1) != "p64x" else ("+early" if a[-2] != "0" else " r02"))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = short(r["Kernel_Name"])
        if n: agg[r["Counter_Name"]][n].append(float(r["Counter_Value"]))
names = sorted({n for c in agg.values() for n in c})
print(f"{'counter':36s} " + " ".join(f"{n:>16s}" for n in names))
for c in sorted(agg):
    print(f"{c:36s} " + " ".join(f"{(sum(agg[c][n][8:]) / max(1, len(agg[c][n]) - 8)) if len(agg[c].get(n, [])) > 8 else float('nan'):16.0f}" for n in names))
PY
