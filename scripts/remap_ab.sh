#!/bin/bash
# horizontal_remap at the bench size: fused flux + update kernel (default) vs the three kernels through HBM, one box
cd "$(dirname "$0")/.."
for rep in 1 2; do
  python3 scripts/remap_bench.py --cpu-grid 0 --reps 4 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('fused    ms/call', j['ms_per_call'])"
  EVPK_REMAP_FUSED=0 python3 scripts/remap_bench.py --cpu-grid 0 --reps 4 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('unfused  ms/call', j['ms_per_call'])"
done
