#!/bin/bash
# round 6 job 1: wave priority A/B under the ticket order (harness) + a baseline bench line of the round-5 library on this box
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 tools/p64v_bench 7 "shipped (3,3) spreads|pair tickets, ONE counter, uncached|prio" > gpurun_out/r06_p64v_ab_35_prio.log 2>&1
tail -12 gpurun_out/r06_p64v_ab_35_prio.log
timeout 600 python bench.py > gpurun_out/r06_bench_baseline.json 2> gpurun_out/r06_bench_baseline.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_bench_baseline.json') if l.startswith('{')][-1])
r=d['roofline']; print(d['config']['kernel'].split('order=')[1], '| kernel_ms %.4f frac %.4f cold %.4f' % (r['kernel_ms'], r['frac'], d['cold_start']['kernel_ms']))
print({k:(round(v['kernel_ms'],4), round(v['roofline_frac'],3)) for k,v in d['variants'].items() if 'kernel_ms' in v})
PY
