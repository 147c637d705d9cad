#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python tools/order_ab.py 7 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_order_ab.log
timeout 1200 python -m pytest tests/test_fullsize_oracle_gpu.py tests/test_tickets_gpu.py tests/test_perf_sanity_gpu.py tests/test_parity_gpu.py -x -q -m gpu 2>&1 | tail -4
