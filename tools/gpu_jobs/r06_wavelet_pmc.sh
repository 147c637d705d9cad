#!/bin/bash
# round 6, third session: HBM traffic of the wavelet refinement launch (separate --pmc passes, counters only)
set -u
REPO="$GRAFT_REPO_ROOT"; mkdir -p $REPO/gpurun_out/wv_pmc
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --output-format csv --pmc $c -d $REPO/gpurun_out/wv_pmc/$c -o pmc -- python $REPO/tools/wavelet_time.py > $REPO/gpurun_out/wv_pmc/$c.log 2>&1
done
cd $REPO
python - <<'PY'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/wv_pmc/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "wavelet" in r["Kernel_Name"]:
                rows[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in rows.items():
        # tools/wavelet_time.py launches 13 times per rate (3 warm-up + 10 timed), rates 0.0 / 0.1 / 0.5 / 1.0 in this order
        per = [sum(v[i * 13:(i + 1) * 13]) / 13 for i in range(len(v) // 13)]
        print(c, k, "dispatches", len(v), "mean KB per launch by rate:", [round(x) for x in per])
PY
rm -rf gpurun_out/wv_pmc/FETCH_SIZE gpurun_out/wv_pmc/WRITE_SIZE
