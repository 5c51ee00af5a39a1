#!/bin/bash
# Round 5, third GPU pass: the rolling tile kernel with its register budget fixed, the headline kernel's instruction diet (A/B against the
# previous build, build/libevpk_head.so), the tests the second pass did not reach, the fuzz replays with their summary lines
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05_v3}; out=gpurun_out/$tag; mkdir -p "$out"
{
timeout 1800 python3 -m pytest tests/test_parity_gpu.py tests/test_multirank_gpu.py -q -x -k "rolling or tile_kernel or upwind_state_on or without_a_block or all_eliminated or bench_line or two_subcycle_kernel or cfg5 or tripole_fold" 2>&1 | tail -25
} > "$out/tests.txt" 2>&1
tail -30 "$out/tests.txt"
{
echo "== headline: this build against build/libevpk_head.so (before the instruction diet), alternating"
bash scripts/lib_ab.sh build/libevpk_head.so
echo "== 3600x2700 open"
bash scripts/lib_ab.sh build/libevpk_head.so --ns open
} > "$out/diet_ab.txt" 2>&1
cat "$out/diet_ab.txt"
bash scripts/roll_ab.sh 1 > "$out/roll_ab.txt" 2>&1
cat "$out/roll_ab.txt"
{
timeout 200 python3 scripts/delivery_stress.py --variant stale --iters 60 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -8
echo "== the 1 029 draws that preceded round 4's difference, every in-place plane checked; then 1 200 new draws, all through page-locked arrays"
EVPK_FUZZ_BASE=110000 EVPK_VERIFY_DELIVERY=1 EVPK_VERIFY_LOG=$out/verify.log timeout 900 python3 scripts/fuzz_one.py 0-1028 2>&1 | grep "draws\|bad\|Error" | tail -4
EVPK_FUZZ_BASE=150000 EVPK_FUZZ_PIN=1 EVPK_VERIFY_DELIVERY=1 EVPK_VERIFY_LOG=$out/verify.log timeout 900 python3 scripts/fuzz_one.py 0-1199 2>&1 | grep "draws\|bad\|Error" | tail -4
cat $out/verify.log 2>/dev/null
} > "$out/delivery.txt" 2>&1
cat "$out/delivery.txt"
