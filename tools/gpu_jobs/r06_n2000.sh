#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
echo "== shipped (no NaN patch)"; python tools/n2000_time.py 2>&1 | grep n_fft
echo "== regtile_n2000 with -DSPECTRE_BF16_CANONICAL_NAN"; SPECTRE_HIP_LIB=$PWD/fft_amd/lib/libspectre_hip_canonnan.so python tools/n2000_time.py 2>&1 | grep n_fft
done
