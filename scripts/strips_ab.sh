#!/bin/bash
# strip list of the pair kernels compacted on the device (default: evpk_prep never waits for the GPU) vs on the host, one box
cd "$(dirname "$0")/.."
run() { python3 bench.py --steps 8 --warmup 3 --cpu-subcycles 0 --no-variants "$@" 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.read()); r=o['roofline']
print('%-6s %-44s ms/evp=%.3f loop=%.3f'%('$TAG', o['config']['workload'][:44], o['ms_per_step'], r['loop_ms_per_step']))"; }
for rep in 1 2; do for v in 1 0; do
  export EVPK_DEVICE_STRIPS=$v; TAG="dev=$v"
  run --grid 320x384 --xblocks 1 --yblocks 1 --dt 3600 --land rows --ns open
  run --grid 360x300 --xblocks 24 --yblocks 1 --dt 3600 --ns open
  run --grid 360x300 --xblocks 24 --yblocks 1 --dt 3600 --ns tripole
  run --grid 450x2700 --xblocks 1 --yblocks 10 --ns open
  run --grid 3600x2700 --ns tripole
done; done
