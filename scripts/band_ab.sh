#!/bin/bash
# tripole top band on one rank: fused into the pair's launch (default) vs band launches on the second stream
# (EVPK_BAND_FUSED=0) vs the open grid, alternated on ONE box.  usage: scripts/band_ab.sh ["<bench args>"]
cd "$(dirname "$0")/.."
args="${1:---no-variants}"
line() { python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$1 ms/step=%.3f loop=%.3f k=%.4f frac=%.3f'%(j['ms_per_step'], r['loop_ms_per_step'], r['avg_launch_ms'], r['frac']))"; }
for rep in 1 2 3; do
  python3 bench.py --steps 8 --warmup 2 --cpu-subcycles 0 $args 2>/dev/null | tail -1 | line "fused   rep$rep"
  EVPK_BAND_FUSED=0 python3 bench.py --steps 8 --warmup 2 --cpu-subcycles 0 $args 2>/dev/null | tail -1 | line "stream2 rep$rep"
  python3 bench.py --steps 8 --warmup 2 --cpu-subcycles 0 --ns open $args 2>/dev/null | tail -1 | line "open    rep$rep"
done
