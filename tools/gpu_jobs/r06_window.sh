#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 tools/window_lab 10 "flat|two-CU|seg 128 dynamic per workgroup|seg  64 dynamic per PAIR  |seg  64 static far|seg 128 static far" > gpurun_out/r06_window_lab_two_cu.log 2>&1
cat gpurun_out/r06_window_lab_two_cu.log
