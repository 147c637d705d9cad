#!/bin/bash
# round 5, job 9: gate gradient with prefetch registers (PN = 16 / 24 / 32) against the shipped form: same-process A/B + parity under each PN
mkdir -p gpurun_out; cd /root/repo
export TMPDIR=/tmp
timeout 900 python tools/dgate_pn.py 4 0,16,116,120,124,120:99999,120:128,120:512,4216 > gpurun_out/r05_dgate_pn.log 2>&1
tail -20 gpurun_out/r05_dgate_pn.log
for pn in 120 124; do
  SPECTRE_TUNING=1 SPECTRE_DGATE_PREFETCH=$pn timeout 900 python -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "4096 or regtile or dgate or gate_grad" 2>&1 | tail -3 | tee -a gpurun_out/r05_dgate_pn.log
done
