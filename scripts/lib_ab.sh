#!/bin/bash
# A/B of two builds of libevpk.so on one box, alternating: scripts/lib_ab.sh <other.so> [bench flags]
other=$1; shift
run() {
  local label=$1; shift
  env "$@" python3 bench.py --steps 5 --warmup 2 --cpu-subcycles 0 --no-variants $FLAGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-12s ms_per_step %.3f loop %.3f x%d avg_launch_ms %.4f frac %.3f'%('$label',d['ms_per_step'],r['loop_ms_per_step'],r['subcycles_per_launch'],r['avg_launch_ms'],r['frac']))"
}
FLAGS="$@"
for rep in 1 2 3; do run default X=1; run other EVPK_LIB=$other; done
