#!/bin/bash
# round 5, GPU call 5: ticket kernels (4096 + 3000) robustness + timing, harness with the library's ticket kernel on one box, GPU tier, bench
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 tools/tickets_lab 5 > gpurun_out/r05_tickets_lab.log 2>&1
echo "tickets_lab rc $?"; grep -v "differing dwords: 0$" gpurun_out/r05_tickets_lab.log | tail -14
timeout 600 tools/p64v_bench 5 'LIBRARY|ONE counter, uncached|shipped (3,3)' > gpurun_out/r05_p64v_ab_32_library_tickets.log 2>&1
echo "p64v rc $?"; grep -A12 "^variant" gpurun_out/r05_p64v_ab_32_library_tickets.log | cut -c1-110
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_pytest_gpu_1.log 2>&1
echo "pytest rc $?"; tail -8 gpurun_out/r05_pytest_gpu_1.log
timeout 600 python bench.py > gpurun_out/r05_bench_1.json 2> gpurun_out/r05_bench_1.err
echo "bench rc $?"; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05_bench_1.json') if l.startswith('{')][-1])
r=d['roofline']; print('kernel', d['config']['kernel']); print('ms_per_step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'cold', d.get('cold_start',{}).get('kernel_ms'))
print({k:(round(v['kernel_ms'],4), round(v['roofline_frac'],3)) for k,v in d['variants'].items()})
PY
