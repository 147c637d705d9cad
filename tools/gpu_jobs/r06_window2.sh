#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 tools/window_lab 10 "flat copy|in flight|seg  64 dynamic per PAIR  |seg  64 static far  " > gpurun_out/r06_window_lab_in_flight.log 2>&1
cat gpurun_out/r06_window_lab_in_flight.log
