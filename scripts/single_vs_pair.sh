#!/bin/bash
# small / medium slabs: the tuned pair kernels (tile or march) against the one-subcycle kernel with its strip height tuned (tune_R1)
run() { env "$@" python3 bench.py --steps 5 --warmup 3 --cpu-subcycles 0 --no-variants $G 2>/dev/null | python3 -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=o['roofline']
print('    ms/evp=%.3f loop=%.3f kern(%d sub)=%.4f ms R=%d strips=%d'%(o['ms_per_step'],r['loop_ms_per_step'],r['subcycles_per_launch'],r['avg_launch_ms'],o['config']['strip_rows_rank0'],o['config']['strips_per_launch_rank0']))"; }
for G in "--grid 450x2700 --xblocks 1 --yblocks 10 --ns open" "--grid 1440x1080 --xblocks 8 --yblocks 4 --dt 1800 --ns open" "--grid 3600x2700 --ns open"; do
  echo "== $G"
  echo "  pairs (default)"; run EVPK_DOUBLE=1
  echo "  one subcycle per launch, tuned R"; run EVPK_DOUBLE=0
  for R in ${ROWS:-2 3 4 6}; do echo "  one subcycle per launch, R=$R"; run EVPK_DOUBLE=0 EVPK_STRIP_ROWS=$R; done
done
