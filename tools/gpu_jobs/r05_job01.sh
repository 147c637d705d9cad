#!/bin/bash
# round 5, GPU call 1: copy-form tile maps (timing + PMC), pair tickets in the product kernel's harness, PMC of the product
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
rocm-smi --showmeminfo vram > gpurun_out/r05_job01_box.txt 2>&1
timeout 300 tools/window_lab 10 > gpurun_out/r05_window_lab.log 2>&1
echo "window_lab rc $?"
timeout 600 tools/p64v_bench 5 > gpurun_out/r05_p64v_ab_31_pair_tickets.log 2>&1
echo "p64v_bench rc $?"
timeout 900 python tools/pmc_probe.py --out gpurun_out/r05_pmc --tag window_lab -- tools/window_lab 2 copy > gpurun_out/r05_pmc_window_lab.stdout 2>&1
echo "pmc window_lab rc $?"
timeout 900 python tools/pmc_probe.py --out gpurun_out/r05_pmc --tag product --kernel-filter spectre_mix --max-passes 18 -- python tools/run_mix.py --iters 3 > gpurun_out/r05_pmc_product.stdout 2>&1
echo "pmc product rc $?"
tail -30 gpurun_out/r05_p64v_ab_31_pair_tickets.log
