#!/bin/bash
# round 5, GPU call 2: pair tickets with one chip-wide counter (uncached memory) in the harness; PMC evidence (copy forms + product forms)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 tools/p64v_bench 5 > gpurun_out/r05_p64v_ab_31b_one_counter.log 2>&1
echo "p64v_bench rc $?"
tail -22 gpurun_out/r05_p64v_ab_31b_one_counter.log
timeout 1300 python tools/pmc_probe.py --out gpurun_out/r05_pmc --tag forms --passes 0,5,6,2,9,7,8,10,3,16,17 -- bash tools/gpu_jobs/pmc_target.sh > gpurun_out/r05_pmc_forms.stdout 2>&1
echo "pmc rc $?"
tail -5 gpurun_out/r05_pmc/forms_passes.log
