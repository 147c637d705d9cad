#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r06_bench_extra.json 2> gpurun_out/r06_bench_extra.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_bench_extra.json') if l.startswith('{')][-1])
r=d['roofline']; print(d['config']['kernel'].split('order=')[1], '| kernel_ms %.4f frac %.4f cold %.4f (%.4f) traffic x%.3f' % (r['kernel_ms'], r['frac'], d['cold_start']['kernel_ms'], d.get('cold_start_frac', 0), r.get('traffic_over_algorithmic') or 0))
print({k:(round(v['kernel_ms'],4), round(v['roofline_frac'],3)) for k,v in d['variants'].items() if 'kernel_ms' in v})
PY
