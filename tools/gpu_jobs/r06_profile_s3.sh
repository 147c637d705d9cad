#!/bin/bash
# round 6, third session: rocprofv3 kernel trace + PMC passes of the bench command on the final build (tools/profile_bench.sh), summaries only
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash tools/profile_bench.sh r06s3 2>&1 | tail -60
ls gpurun_out/prof_r06s3/summary
rm -rf gpurun_out/prof_r06s3/trace gpurun_out/prof_r06s3/pmc_*
