#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_tickets_gpu.py tests/test_harness_gpu.py -x -q 2>&1 | tail -3
for i in 1 2 3; do
  timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r05_bench_auto_$i.json 2> gpurun_out/r05_bench_auto_$i.err
  python - $i <<'PY'
import json,sys
d=json.loads([l for l in open(f'gpurun_out/r05_bench_auto_{sys.argv[1]}.json') if l.startswith('{')][-1])
r=d['roofline']; print('run', sys.argv[1], d['config']['kernel'].split('order=')[1], '| kernel_ms %.4f frac %.4f cold %.4f' % (r['kernel_ms'], r['frac'], d['cold_start']['kernel_ms']))
print('   ', {k:(round(v['kernel_ms'],4), round(v['roofline_frac'],3)) for k,v in d['variants'].items()})
PY
done
