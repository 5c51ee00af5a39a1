#!/bin/bash
# A/B of library builds on ONE box (boxes differ by several %): scripts/ab.sh "<bench args>" build/ab/libevpk_A.so build/ab/libevpk_B.so ...
# Alternates the variants three times; restores nothing (use on the scratch copy of a gpurun box).
cd "$(dirname "$0")/.."
args="$1"; shift
for rep in 1 2 3; do for lib in "$@"; do
  cp "$lib" cice5_amd/libevpk.so
  EVPK_BALANCE=${EVPK_BALANCE:-0} python3 bench.py --steps 8 --warmup 2 --cpu-subcycles 0 $args 2>/dev/null | tail -1 | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('$(basename $lib) rep$rep ms/step=%.3f loop=%.3f k=%.4f R=%d'%(j['ms_per_step'], r['loop_ms_per_step'], r['avg_launch_ms'], j['config']['strip_rows_rank0']))"
done; done
