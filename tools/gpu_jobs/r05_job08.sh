#!/bin/bash
# round 5, GPU call 8: PMC evidence for the copy forms (flat / static far / tickets) and for the gate gradient
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tools/window_lab 2 'flat copy|seg 128&static far&copy|seg 128&dynamic per workgroup&copy|seg  64&static far&copy|seg  64&dynamic per PAIR &copy' > gpurun_out/r05_window_lab_filtered.log 2>&1
cat gpurun_out/r05_window_lab_filtered.log
timeout 900 python tools/pmc_probe.py --out gpurun_out/r05_pmc --tag copies --passes 0,5,6,7,8,2,9 -- tools/window_lab 2 'flat copy|seg 128&static far&copy|seg 128&dynamic per workgroup&copy|seg  64&static far&copy|seg  64&dynamic per PAIR &copy' > gpurun_out/r05_pmc_copies.stdout 2>&1
echo "pmc copies rc $?"; cat gpurun_out/r05_pmc/copies_passes.log | cut -c1-200 | tail -8
timeout 900 python tools/pmc_probe.py --out gpurun_out/r05_pmc --tag dgate --kernel-filter gate_grad --passes 0,5,6,7,8,2,9 -- python tools/run_bwd.py 256,4096,768 dgate 3 > gpurun_out/r05_pmc_dgate.stdout 2>&1
echo "pmc dgate rc $?"; tail -3 gpurun_out/r05_pmc/dgate_passes.log | cut -c1-200
