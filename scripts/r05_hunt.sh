#!/bin/bash
# Round 5: hunting the lost in-place delivery under pytest (where it showed twice): every draw through page-locked arrays, every in-place plane
# verified against a staged delivery and REPAIRED (EVPK_VERIFY_DELIVERY=2) so that the run goes on; the events land in verify.log
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05_hunt}; out=gpurun_out/$tag; mkdir -p "$out"
grep -E "^thp_|^numa_|^pgmigrate|^compact_migrate" /proc/vmstat > "$out/vmstat_before.txt"
for base in ${2:-410000 420000 430000}; do
  EVPK_FUZZ_PIN=1 EVPK_VERIFY_DELIVERY=2 EVPK_VERIFY_LOG=$out/verify.log EVPK_FUZZ_N=3000 EVPK_FUZZ_BASE=$base EVPK_FUZZ_R_N=0 EVPK_FUZZ_E_N=0 \
    timeout 1500 python3 -m pytest tests/test_fuzz_gpu.py -q -k random_configuration 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -3 | sed "s/^/base $base: /"
done > "$out/hunt.txt" 2>&1
grep -E "^thp_|^numa_|^pgmigrate|^compact_migrate" /proc/vmstat > "$out/vmstat_after.txt"
cat "$out/hunt.txt"
echo "events logged:"; grep -c "delivery check" "$out/verify.log" 2>/dev/null
head -150 "$out/verify.log" 2>/dev/null
