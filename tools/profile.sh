#!/bin/bash
# rocprofv3 passes for the headline kernel; run on the GPU box:  bash tools/profile.sh <tag> [extra run_mix args]
# Writes raw output under gpurun_out/prof_<tag>/ and compact summaries under gpurun_out/prof_<tag>/summary/.
set -u
TAG=${1:-r01}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT/summary
cd /tmp && export TMPDIR=/tmp
RUN="python $REPO/tools/run_mix.py $*"
# 1) kernel trace + stats (timing)
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/trace -o trace -- $RUN > $OUT/trace.log 2>&1
# 2) PMC passes (counters only, never with trace domains)
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $RUN > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $RUN > $OUT/pmc_write.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY -d $OUT/pmc_sq1 -o pmc -- $RUN > $OUT/pmc_sq1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_SALU -d $OUT/pmc_sq2 -o pmc -- $RUN > $OUT/pmc_sq2.log 2>&1
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_grbm -o pmc -- $RUN > $OUT/pmc_grbm.log 2>&1
cd $REPO
python tools/summarize_prof.py $OUT > $OUT/summary/summary.txt 2>&1
cat $OUT/summary/summary.txt
