#!/bin/bash
# Round 5: the rolling tile kernel (k_subcycle2r, EVPK_TILE=2) against the five-row tiles (k_subcycle2t, EVPK_TILE=1) and the tuner's own
# choice on the small-slab grids, one GPU; alternating so that box drift shows.  usage: scripts/roll_ab.sh [reps]
cd "$(dirname "$0")/.." || exit 1
reps=${1:-2}
run() { python3 bench.py --steps 6 --warmup 2 --cpu-subcycles 0 "$@" 2>/dev/null | python3 -c "
import json,sys
o=json.loads([l for l in sys.stdin.readlines() if l.startswith('{')][-1]); r=o['roofline']
print('%-40s R=%-2d strips=%-5d ms/evp=%.3f loop=%.3f kern=%.4f ms frac=%.3f'%(o['config']['workload'][:40],o['config']['strip_rows_rank0'],o['config']['strips_per_launch_rank0'],o['ms_per_step'],r['loop_ms_per_step'],r['avg_launch_ms'],r['frac']))"; }
grids=("--grid 450x2700 --xblocks 1 --yblocks 10 --ns open" "--grid 1440x1080 --xblocks 8 --yblocks 4 --dt 1800 --ns open" "--grid 360x300 --xblocks 24 --yblocks 1 --dt 3600 --ns open" "--grid 320x384 --xblocks 1 --yblocks 1 --dt 3600 --land rows --ns open" "--grid 1440x1080 --xblocks 8 --yblocks 4 --dt 1800 --ns tripole")
for rep in $(seq 1 $reps); do
for g in "${grids[@]}"; do
  echo "== $g (rep $rep)"
  unset EVPK_STRIP_ROWS
  EVPK_TILE=1 run $g | sed 's/^/tile H=5 (2t)   /'
  for R in 5 11 17 23 35 47; do
    EVPK_TILE=2 EVPK_STRIP_ROWS=$R run $g | sed "s/^/roll R=$R (2r)   /"
  done
  (unset EVPK_TILE; run $g | sed 's/^/tuner           /')
done
done
