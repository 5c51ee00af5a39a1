#!/bin/bash
# k_subcycle3w on small slabs (one rank, open N-S) against the tile / marching pair kernels: is the pipeline a better use of a chip
# that has more SIMDs than strips?
export EVPK_LIB=${EVPK_LIB:-cice5_amd/libevpk_exp.so}      # k_subcycle3w lives in the experimental build only (make -C cice5_amd/csrc exp)
run() { env "$@" python3 bench.py --steps 5 --warmup 2 --cpu-subcycles 0 --no-variants $G 2>/dev/null | python3 -c "
import json,sys
o=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=o['roofline']
print('    ms/evp=%.3f loop=%.3f kern(%d sub)=%.4f ms launches=%d others=%s'%(o['ms_per_step'],r['loop_ms_per_step'],r['subcycles_per_launch'],r['avg_launch_ms'],r['launches'],[(q['subcycles_per_launch'],q['launches'],round(q['avg_launch_ms'],4)) for q in r['other_kernels']]))"; }
for G in "--grid 450x2700 --xblocks 1 --yblocks 10 --ns open" "--grid 1440x1080 --xblocks 8 --yblocks 4 --dt 1800 --ns open" "--grid 360x300 --xblocks 24 --yblocks 1 --dt 3600 --ns open" "--grid 320x384 --xblocks 1 --yblocks 1 --dt 3600 --land rows --ns open"; do
  echo "== $G"
  echo "  default (tuner)"; run EVPK_TRIPLE=0
  for R in ${K3_ROWS:-3 4 6 8 12}; do echo "  triple R3=$R"; run EVPK_TRIPLE=1 EVPK_TILE=0 EVPK_STRIP_ROWS3=$R; done
done
