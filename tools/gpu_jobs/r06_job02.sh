#!/bin/bash
# round 6 job 2: tile-order API + slice ownership tests, ticket tests, bench line
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_tile_order_gpu.py tests/test_tickets_gpu.py -x -q 2>&1 | tail -15
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r06_bench_job02.json 2> gpurun_out/r06_bench_job02.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_bench_job02.json') if l.startswith('{')][-1])
r=d['roofline']; print(d['config']['kernel'].split('order=')[1], '| kernel_ms %.4f frac %.4f cold %.4f' % (r['kernel_ms'], r['frac'], d['cold_start']['kernel_ms']))
print({k:(round(v['kernel_ms'],4), round(v['roofline_frac'],3)) for k,v in d['variants'].items() if 'kernel_ms' in v})
PY
