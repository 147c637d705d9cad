#!/bin/bash
# round 6 job 3: the whole GPU tier + a bench line + the rocprofv3 kernel trace of the bench command
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8
timeout 600 python bench.py > gpurun_out/r06_bench_job03.json 2> gpurun_out/r06_bench_job03.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06_bench_job03.json') if l.startswith('{')][-1])
r=d['roofline']; print(d['config']['kernel'].split('order=')[1], '| kernel_ms %.4f frac %.4f cold %.4f (%.4f)' % (r['kernel_ms'], r['frac'], d['cold_start']['kernel_ms'], d.get('cold_start_frac', 0)))
print({k:(round(v['kernel_ms'],4), round(v['roofline_frac'],3), v.get('frac_of_pattern_copy')) for k,v in d['variants'].items() if 'kernel_ms' in v})
PY
