#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 tools/q32_lab ${1:-5} "${2:-}" > gpurun_out/r06_q32_lab.log 2>&1
cat gpurun_out/r06_q32_lab.log | tail -40
