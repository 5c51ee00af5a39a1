#!/bin/bash
# BASELINE.json configs 2, 3 (+ tripole variant of the headline grid) on one GPU: python bench.py lines, compact
run() { python3 bench.py --steps 5 --warmup 2 --cpu-subcycles 0 "$@" 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.read()); r=o['roofline']
print('%-70s value=%.3e ms/evp=%.3f loop=%.3f kern(%d sub)=%.4f ms frac=%.3f R=%d'%(o['config']['workload'][:70],o['value'],o['ms_per_step'],r['loop_ms_per_step'],r['subcycles_per_launch'],r['avg_launch_ms'],r['frac'],o['config']['strip_rows_rank0']))"; }
run --grid 320x384 --xblocks 1 --yblocks 1 --dt 3600 --land rows --ns open
run --grid 320x384 --xblocks 1 --yblocks 1 --dt 3600 --land rows --ice full --ns open
run --grid 360x300 --xblocks 24 --yblocks 1 --dt 3600 --ns open
run --grid 360x300 --xblocks 24 --yblocks 1 --dt 3600 --ns tripole
run --grid 1440x1080 --xblocks 8 --yblocks 4 --dt 1800 --ns open
run --grid 1440x1080 --xblocks 8 --yblocks 4 --dt 1800 --ns tripole
run --grid 450x2700 --xblocks 1 --yblocks 10 --ns open
run --grid 3600x2700 --ns open
run --grid 3600x2700 --ns tripole
run --grid 3600x2700 --ns tripole --ndte 240
