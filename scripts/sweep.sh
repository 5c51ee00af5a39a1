#!/bin/bash
# usage: tools_sweep.sh "<R list>" "<ice list>"
for ice in $2; do for R in $1; do
  EVPK_STRIP_ROWS=$R python3 bench.py --steps 2 --warmup 1 --cpu-subcycles 0 --ice $ice 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.read()); r=o['roofline']
print('ice=$ice R=$R value=%.3e ms/step=%.2f loop=%.2f k(%d sub)=%.4f ms frac=%.3f strips=%d R_used=%d'%(o['value'],o['ms_per_step'],r['loop_ms_per_step'],r['subcycles_per_launch'],r['avg_launch_ms'],r['frac'],o['config']['strips_per_launch_rank0'],o['config']['strip_rows_rank0']))"
done; done
