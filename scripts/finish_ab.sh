#!/bin/bash
# evp_finish + u2tgrid_vector in one launch (k_finish_tgrid, one rank) against k_finish, halo, k_to_tgrid2: what is outside the loop
cd "$(dirname "$0")/.."
run() { python3 bench.py --steps 10 --warmup 3 --cpu-subcycles 0 --no-variants "$@" 2>/dev/null | python3 -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=o['roofline']
print('%-10s %-34s ms/evp=%.3f loop=%.3f prep+finish=%.3f ms'%('$TAG', o['config']['workload'][:34], o['ms_per_step'], r['loop_ms_per_step'], o['ms_per_step']-r['loop_ms_per_step']))"; }
for rep in 1 2; do for v in 1 0; do
  export EVPK_FINISH_FUSED=$v; TAG="fused=$v"
  run --ns tripole
  run --ns open
  run --grid 450x2700 --xblocks 1 --yblocks 10 --ns open
  run --grid 360x300 --xblocks 24 --yblocks 1 --dt 3600 --ns tripole
done; done
