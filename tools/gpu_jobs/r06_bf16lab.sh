#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 tools/bf16_lab 5 f32out > gpurun_out/r06_bf16_lab_f32out.log 2>&1; tail -12 gpurun_out/r06_bf16_lab_f32out.log
timeout 600 tools/bf16_lab 5 bf16out > gpurun_out/r06_bf16_lab_bf16out.log 2>&1; tail -12 gpurun_out/r06_bf16_lab_bf16out.log
