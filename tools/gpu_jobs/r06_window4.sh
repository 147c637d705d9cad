#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 tools/window_lab 10 "flat copy|bf16->f32" > gpurun_out/r06_window_lab_bf16_overfetch.log 2>&1
cat gpurun_out/r06_window_lab_bf16_overfetch.log
