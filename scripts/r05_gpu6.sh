#!/bin/bash
# Round 5, sixth GPU pass: band workgroups last (A/B), stepa inside k_eap_sub (A/B), their parity tests
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05_v6}; out=gpurun_out/$tag; mkdir -p "$out"
timeout 1800 python3 -m pytest tests/test_parity_gpu.py tests/test_multirank_gpu.py tests/test_edge_gpu.py -m gpu -x -q -k "tripole or eap or cfg5 or fold or band or x_slabs" > "$out/tests.txt" 2>&1
tail -5 "$out/tests.txt"
run() { python3 bench.py --steps 6 --warmup 2 --cpu-subcycles 0 --no-variants "$@" 2>/dev/null | python3 -c "
import json,sys
o=json.loads([l for l in sys.stdin.readlines() if l.startswith('{')][-1]); r=o['roofline']
print('%-40s R=%-2d ms/evp=%.3f loop=%.3f kern=%.4f ms frac=%.3f'%(o['config']['workload'][:40],o['config']['strip_rows_rank0'],o['ms_per_step'],r['loop_ms_per_step'],r['avg_launch_ms'],r['frac']))"; }
{
echo "== tripole band workgroups first (EVPK_BAND_LAST=0, rounds 3-4) against last (default), alternating"
for rep in 1 2 3; do
  EVPK_BAND_LAST=0 run | sed 's/^/band first  /'
  EVPK_BAND_LAST=1 run | sed 's/^/band last   /'
done
for g in "--grid 1440x1080 --xblocks 8 --yblocks 4 --dt 1800" "--grid 360x300 --xblocks 24 --yblocks 1 --dt 3600"; do
  for rep in 1 2; do
    EVPK_BAND_LAST=0 run $g | sed 's/^/band first  /'
    EVPK_BAND_LAST=1 run $g | sed 's/^/band last   /'
  done
done
} > "$out/band_last_ab.txt" 2>&1
cat "$out/band_last_ab.txt"
{
echo "== eap(dt), 3600x2700 tripole, ndte = 120: stepa as a launch of its own (EVPK_EAP_STEPA_FUSED=0) against inside k_eap_sub (default)"
for rep in 1 2; do
  EVPK_EAP_STEPA_FUSED=0 timeout 300 python3 scripts/eap_bench.py --cpu-grid 0 2>/dev/null | tail -1 | cut -c1-400 | sed 's/^/own launch  /'
  timeout 300 python3 scripts/eap_bench.py --cpu-grid 0 2>/dev/null | tail -1 | cut -c1-400 | sed 's/^/in k_eap_sub /'
done
} > "$out/eap_ab.txt" 2>&1
cat "$out/eap_ab.txt"
