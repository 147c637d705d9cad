#!/bin/bash
# round 5, GPU call 11: after the persistent gate-gradient form: ticket lab, GPU tier, bench line, rocprofv3 evidence of `python bench.py`
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 tools/tickets_lab 5 > gpurun_out/r05_tickets_lab_d.log 2>&1
echo "tickets_lab rc $?"; grep -v "differing dwords: 0$" gpurun_out/r05_tickets_lab_d.log | tail -14
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_pytest_gpu_3.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/r05_pytest_gpu_3.log
timeout 600 python bench.py > gpurun_out/r05_bench_3.json 2> gpurun_out/r05_bench_3.err
echo "bench rc $?"; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05_bench_3.json') if l.startswith('{')][-1])
r=d['roofline']; print('kernel', d['config']['kernel']); print('ms_per_step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'cold', d.get('cold_start',{}).get('kernel_ms'))
print({k:(round(v['kernel_ms'],4), round(v['roofline_frac'],3)) for k,v in d['variants'].items()})
PY
timeout 1500 bash tools/profile_bench.sh r05 > gpurun_out/r05_profile_bench.log 2>&1; tail -3 gpurun_out/r05_profile_bench.log | cut -c1-300

