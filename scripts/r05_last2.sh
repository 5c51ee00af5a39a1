#!/bin/bash
# Round 5, the last GPU minutes: more seeded draws on the final library in the product's default configuration (no verification switch)
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05_last2}; out=gpurun_out/$tag; mkdir -p "$out"
EVPK_FUZZ_N=${2:-3000} EVPK_FUZZ_BASE=${3:-810000} EVPK_FUZZ_R_N=0 EVPK_FUZZ_E_N=0 timeout 720 python3 -m pytest tests/test_fuzz_gpu.py -q -k random_configuration 2>&1 \
  | tee "$out/fuzz_full.txt" | grep -E "passed|failed|FAILED" | tail -4 > "$out/fuzz.txt"
cat "$out/fuzz.txt"
