#!/bin/bash
# does the tripole band's share of the resident workgroup slots (32 of 512 on the bench grid) want another strip height?
# bench default (3600x2700 tripole) with the tuner counting the band workgroups or not, fixed heights around its choice, open N-S beside it
cd "$(dirname "$0")/.."
run() { python3 bench.py --steps 6 --warmup 3 --cpu-subcycles 0 --no-variants "$@" 2>/dev/null | python3 -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=o['roofline']
print('%-22s ms/evp=%.3f loop=%.3f kern=%.4f ms R=%d strips=%d'%('$TAG', o['ms_per_step'], r['loop_ms_per_step'], r['avg_launch_ms'], o['config']['strip_rows_rank0'], o['config']['strips_per_launch_rank0']))"; }
for rep in 1 2; do
  TAG="open" run --ns open
  TAG="tripole tune_band=0" EVPK_TUNE_BAND=0 run --ns tripole
  TAG="tripole tune_band=1" EVPK_TUNE_BAND=1 run --ns tripole
  for R in 22 24 26 28; do TAG="tripole R=$R" EVPK_STRIP_ROWS=$R run --ns tripole; done
done
