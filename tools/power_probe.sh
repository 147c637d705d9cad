#!/bin/bash
# Socket power and shader clock while ONE kernel variant runs back to back (tools/p64x_bench loop ...).  Run on the GPU box: bash tools/power_probe.sh
cd "$(dirname "$0")/.."
run() {  # variant name
  python tools/power_sampler.py 4.5 "$1" &
  S=$!
  timeout 60 tools/p64x_bench loop "$1" 5 | grep "^loop"
  wait $S
}
python tools/power_sampler.py 2 "idle"
run "LIBRARY <4,2>"
run "LIBRARY, no traffic (rows_in = rows_out = 0)"
run "LIBRARY, stores dropped (rows_out = 0)"
run "LIBRARY, loads answered with 0 (rows_in = 0)"
run "no traffic: exchanges without LDS ops (VALU + barriers)"
run "no traffic: no butterflies (LDS + barriers)"
run "baseline (3,3)"
run "LIBRARY bf16 -> bf16 <3,3>"
