#!/bin/bash
# A/B of library builds for horizontal_remap on one box: scripts/remap_libs.sh build/ab/libevpk_A.so build/ab/libevpk_B.so ...
cd "$(dirname "$0")/.."
for rep in 1 2; do for lib in "$@"; do
  cp "$lib" cice5_amd/libevpk.so
  python3 scripts/remap_bench.py --cpu-grid 0 --reps 4 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$(basename $lib) rep$rep ms/call', j['ms_per_call'])"
done; done
