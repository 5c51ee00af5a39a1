#!/bin/bash
# Round 5, second GPU pass: new tests, rolling-kernel A/B, delivery checks with the (fixed) verification switch
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05_v2}; out=gpurun_out/$tag; mkdir -p "$out"
{
echo "== pytest: delivery, rolling tile kernel, evp after upwind, idle ranks, self-verifying multi-rank bench line"
timeout 1500 python3 -m pytest tests/test_delivery_gpu.py tests/test_parity_gpu.py tests/test_multirank_gpu.py -q -x -k "delivery or registered or registry or rolling or tile_kernel or upwind_state_on or page_locked or device_memory or without_a_block or all_eliminated or bench_line" 2>&1 | tail -25
} > "$out/tests.txt" 2>&1
tail -30 "$out/tests.txt"
bash scripts/roll_ab.sh 1 > "$out/roll_ab.txt" 2>&1
tail -50 "$out/roll_ab.txt"
{
echo "== delivery_stress with the library's own check on as well (EVPK_VERIFY_DELIVERY=1)"
EVPK_VERIFY_DELIVERY=1 timeout 400 python3 scripts/delivery_stress.py --variant heap_raw --seconds 200 --churn --oracle 2>&1 | tail -6
EVPK_VERIFY_DELIVERY=1 timeout 300 python3 scripts/delivery_stress.py --variant heap --seconds 100 --churn 2>&1 | tail -6
timeout 200 python3 scripts/delivery_stress.py --variant stale --iters 60 2>&1 | tail -8
echo "== the 1 029 draws that preceded round 4's difference, every in-place plane checked; then 1 200 new draws, all through page-locked arrays"
EVPK_FUZZ_BASE=110000 EVPK_VERIFY_DELIVERY=1 EVPK_VERIFY_LOG=$out/verify.log timeout 900 python3 scripts/fuzz_one.py 0-1028 2>&1 | tail -4
EVPK_FUZZ_BASE=150000 EVPK_FUZZ_PIN=1 EVPK_VERIFY_DELIVERY=1 EVPK_VERIFY_LOG=$out/verify.log timeout 900 python3 scripts/fuzz_one.py 0-1199 2>&1 | tail -4
cat $out/verify.log 2>/dev/null
} > "$out/delivery.txt" 2>&1
tail -40 "$out/delivery.txt"
