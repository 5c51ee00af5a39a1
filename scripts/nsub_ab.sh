#!/bin/bash
# k_subcycleNt (NS subcycles per launch, small one-rank slabs) against the pair kernel: EVPK_NSUB = 0 / 4 / 6 / auto
# (the record of an experiment: EVPK_NSUB exists in commit 395dac5 only -- profiles/r02_v3/nsub_rejected.txt)
cd "$(dirname "$0")/.." || exit 1
run() { python3 bench.py --steps 5 --warmup 2 --cpu-subcycles 0 "$@" 2>/dev/null | python3 -c "
import json,sys
o=json.loads([l for l in sys.stdin.readlines() if l.startswith('{')][-1]); r=o['roofline']
print('%-42s %-22s sub/launch=%d R=%-2d strips=%-5d ms/evp=%.3f loop=%.3f kern=%.4f ms frac=%.3f value=%.3e'%(o['config']['workload'][:42],r['kernel'][:22],r['subcycles_per_launch'],o['config']['strip_rows_rank0'],o['config']['strips_per_launch_rank0'],o['ms_per_step'],r['loop_ms_per_step'],r['avg_launch_ms'],r['frac'],o['value']))"; }
for t in 0 4 6 auto; do
  if [ $t = auto ]; then unset EVPK_NSUB; else export EVPK_NSUB=$t; fi
  echo "== EVPK_NSUB=$t"
  run --grid 320x384 --xblocks 1 --yblocks 1 --dt 3600 --land rows --ns open
  run --grid 320x384 --xblocks 1 --yblocks 1 --dt 3600 --land rows --ice full --ns open
  run --grid 360x300 --xblocks 24 --yblocks 1 --dt 3600 --ns open
  run --grid 450x2700 --xblocks 1 --yblocks 10 --ns open
  run --grid 1440x1080 --xblocks 8 --yblocks 4 --dt 1800 --ns open
done
