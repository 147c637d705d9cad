#!/bin/bash
# round 6, third session: the rebuilt tree on a fresh box — full GPU tier, smoke(), the default bench line
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 ) 2>&1 | tee gpurun_out/r06_s3_gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r06_s3_smoke.log
timeout 600 python bench.py > gpurun_out/r06_s3_bench.json 2> gpurun_out/r06_s3_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_s3_bench.json') if l.startswith('{')][-1])
r=d['roofline']; print(d['config']['kernel'].split('order=')[1], '| value %.4g kernel_ms %.4f frac %.4f cold %.4f (%.4f) traffic x%.3f' % (d['value'], r['kernel_ms'], r['frac'], d['cold_start']['kernel_ms'], d.get('cold_start_frac', 0), r.get('traffic_over_algorithmic') or 0))
print({k:(round(v['kernel_ms'],4), round(v['roofline_frac'],3)) for k,v in d['variants'].items() if 'kernel_ms' in v})
print('cpu_baseline', d.get('cpu_baseline'))
PY
