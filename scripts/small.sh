#!/bin/bash
run() { python3 bench.py --steps 10 --warmup 3 --cpu-subcycles 0 "$@" 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.read()); r=o['roofline']
print('%-50s value=%.3e ms/evp=%.3f loop=%.3f kern=%.4f'%(o['config']['workload'][:50],o['value'],o['ms_per_step'],r['loop_ms_per_step'],r['avg_launch_ms']))"; }
for tk in 1 0; do echo "EVPK_TIME_KERNELS=$tk"; export EVPK_TIME_KERNELS=$tk
run --grid 320x384 --xblocks 1 --yblocks 1 --dt 3600 --land rows
run --grid 360x300 --xblocks 24 --yblocks 1 --dt 3600
run --grid 1440x1080 --xblocks 8 --yblocks 4 --dt 1800
run
done
