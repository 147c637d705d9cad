#!/bin/bash
# round 6: rocprofv3 kernel trace + PMC passes of the bench command (tools/profile_bench.sh), summaries into gpurun_out/prof_r06/summary
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash tools/profile_bench.sh r06 2>&1 | tail -60
ls gpurun_out/prof_r06/summary
rm -rf gpurun_out/prof_r06/trace gpurun_out/prof_r06/pmc_*    # (the raw CSVs are large; the summaries are what is kept)
