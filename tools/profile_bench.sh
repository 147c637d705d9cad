#!/bin/bash
# Round-3 evidence in one go (run on the GPU box:  bash tools/profile_bench.sh <tag>):
#   1) rocprofv3 --kernel-trace --stats of `python bench.py` (the command the driver runs)   -> summary.txt (per-kernel averages)
#   2) separate --pmc passes (counters only) of the same command: FETCH_SIZE / WRITE_SIZE     -> profiles/pmc_latest.json via collect_pmc.py
#   3) L2 <-> memory request mix of the headline kernel (TCC_EA0_*, write-backs)
set -u
TAG=${1:-r05}
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT/summary
cd /tmp && export TMPDIR=/tmp
RUN="python $REPO/bench.py --no-cpu-baseline"
timeout 600 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o b -- $RUN > $OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --output-format csv --pmc $c -d $OUT/pmc_$c -o pmc -- $RUN > $OUT/pmc_$c.log 2>&1
done
timeout 600 rocprofv3 --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $OUT/pmc_ea -o pmc -- $RUN > $OUT/pmc_ea.log 2>&1
timeout 600 rocprofv3 --output-format csv --pmc TCC_NORMAL_WRITEBACK_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE -d $OUT/pmc_wb -o pmc -- $RUN > $OUT/pmc_wb.log 2>&1
timeout 600 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $OUT/pmc_sq -o pmc -- $RUN > $OUT/pmc_sq.log 2>&1
cd $REPO
python tools/summarize_prof.py $OUT > $OUT/summary/summary.txt 2>&1
python tools/collect_pmc.py $OUT f32 256,4096,768 "spectre_mix_regtile64p<3, 3, false, false, false, true, true, 1>" > $OUT/summary/pmc_latest.log 2>&1
tail -3 $OUT/summary/pmc_latest.log
head -40 $OUT/summary/summary.txt
