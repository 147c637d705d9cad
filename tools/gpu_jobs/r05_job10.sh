#!/bin/bash
# round 5, job 10: gate gradient, prefetch form: share of the prefetched rows requested at the top of the tile (compile-time, three libraries)
mkdir -p gpurun_out; cd /root/repo
export TMPDIR=/tmp
: > gpurun_out/r05_dgate_top.log
for lib in libspectre_hip.so libspectre_hip_fine.so libspectre_hip_top10.so libspectre_hip_top2.so; do
  echo "== $lib" >> gpurun_out/r05_dgate_top.log
  SPECTRE_HIP_LIB=$PWD/fft_amd/lib/$lib timeout 600 python tools/dgate_pn.py 4 0,120,124 >> gpurun_out/r05_dgate_top.log 2>&1
done
grep "==\|median\|NOT DET" gpurun_out/r05_dgate_top.log
