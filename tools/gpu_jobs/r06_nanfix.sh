#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
tools/nan_cvt_probe
for i in 1 2; do
for b in fix nofix; do for m in bf16out f32out; do echo "== $b $m"; timeout 300 tools/bf16_lab_$b 5 $m 2>&1 | tail -3; done; done
done
