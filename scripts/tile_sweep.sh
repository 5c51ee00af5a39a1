#!/bin/bash
# tile heights of k_subcycle2t (EVPK_TILE=1, EVPK_STRIP_ROWS=H: H + 3 waves per workgroup) and strip heights of the marching pair
# kernel on the medium / small slabs, against the tuner's own choice
run() { env "$@" python3 bench.py --steps 5 --warmup 3 --cpu-subcycles 0 --no-variants $G 2>/dev/null | python3 -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=o['roofline']
print('    ms/evp=%.3f loop=%.3f kern(%d sub)=%.4f ms R=%d strips=%d'%(o['ms_per_step'],r['loop_ms_per_step'],r['subcycles_per_launch'],r['avg_launch_ms'],o['config']['strip_rows_rank0'],o['config']['strips_per_launch_rank0']))"; }
for G in "--grid 450x2700 --xblocks 1 --yblocks 10 --ns open" "--grid 1440x1080 --xblocks 8 --yblocks 4 --dt 1800 --ns open"; do
  echo "== $G"
  echo "  tuner"; run X=1
  for H in 3 4 5 6 7 8 9 10 13; do echo "  tile H=$H"; run EVPK_TILE=1 EVPK_STRIP_ROWS=$H; done
  for R in 3 4 5 6 8; do echo "  march R=$R"; run EVPK_TILE=0 EVPK_STRIP_ROWS=$R; done
done
