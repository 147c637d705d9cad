#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 tools/window_lab 10 "flat copy|ticket per|seg  64 dynamic per PAIR  |per PAIR, 4 in flight" > gpurun_out/r06_window_lab_gang.log 2>&1
cat gpurun_out/r06_window_lab_gang.log
