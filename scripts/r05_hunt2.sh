#!/bin/bash
# Round 5: is the FIRST evp of a fresh context deterministic?  The two failing draws' shapes, thousands of fresh contexts each, with the
# library that spills in k_subcycle2t<.., LAST2> (in-tree at the time of this pass) and with the scratch-free build (build/libevpk_new.so)
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05_hunt2}; out=gpurun_out/$tag; mkdir -p "$out"; T=${2:-400}
{
for lib in cice5_amd/libevpk.so build/libevpk_new.so; do
  echo "=== EVPK_LIB=$lib"
  EVPK_LIB=$lib timeout $((T + 200)) python3 scripts/first_evp_stress.py --seconds $T --grid 733x12 --blocks 245x6 --ndte 20 --revised --pin --mode EVPK_FORCE_EXCHANGE=2 EVPK_ZONE_M=1 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
  EVPK_LIB=$lib timeout $((T + 200)) python3 scripts/first_evp_stress.py --seconds $T --grid 200x40 --blocks 20x20 --ndte 20 --pin --mode EVPK_COMPACT_METRICS=0 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
done
} > "$out/first_evp.txt" 2>&1
cat "$out/first_evp.txt"
