#!/bin/bash
# round 6 job 4: whole GPU tier, smoke, bench line
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -7
timeout 600 python bench.py > gpurun_out/r06_bench_job04.json 2> gpurun_out/r06_bench_job04.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_bench_job04.json') if l.startswith('{')][-1])
r=d['roofline']; print(d['config']['kernel'].split('order=')[1], '| kernel_ms %.4f frac %.4f cold %.4f (%.4f) traffic x%.3f' % (r['kernel_ms'], r['frac'], d['cold_start']['kernel_ms'], d.get('cold_start_frac', 0), r.get('traffic_over_algorithmic') or 0))
print({k:(round(v['kernel_ms'],4), round(v['roofline_frac'],3), v.get('frac_of_pattern_copy')) for k,v in d['variants'].items() if 'kernel_ms' in v})
print(d['cpu_baseline'])
PY
