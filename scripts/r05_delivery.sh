#!/bin/bash
# Round 5, item 1: the in-place delivery into page-locked caller arrays under test (one GPU box).  usage: scripts/r05_delivery.sh <tag> [seconds per variant]
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05_delivery}; T=${2:-120}
out=gpurun_out/$tag; mkdir -p "$out"
{
echo "== host settings"; uname -r; cat /sys/kernel/mm/transparent_hugepage/enabled /proc/sys/kernel/numa_balancing 2>/dev/null; nproc; numactl -H 2>/dev/null | head -3
echo "== pytest: delivery fences + the evp-after-upwind regression + the page-locked parity cases"
timeout 900 python3 -m pytest tests/test_delivery_gpu.py tests/test_parity_gpu.py -q -x -k "delivery or registered or registry or upwind or page_locked or device_memory or sparse" 2>&1 | tail -15
echo "== delivery_stress (in-place vs staged delivery of one device state, no oracle in between)"
for v in heap_raw heap aligned alloc; do
  timeout $((T + 120)) python3 scripts/delivery_stress.py --variant $v --seconds $T --churn 2>&1 | tail -12
done
timeout $((T + 120)) python3 scripts/delivery_stress.py --variant heap_raw --seconds $T --churn --oracle 2>&1 | tail -12
timeout 300 python3 scripts/delivery_stress.py --variant stale --iters 150 2>&1 | tail -8
echo "== the 1 029 draws that preceded round 4's difference, in one process, every in-place plane checked (EVPK_VERIFY_DELIVERY=1)"
EVPK_FUZZ_BASE=110000 EVPK_VERIFY_DELIVERY=1 EVPK_VERIFY_LOG=$out/verify.log timeout 1500 python3 scripts/fuzz_one.py 0-1028 2>&1 | tail -12
cat $out/verify.log 2>/dev/null
} > "$out/delivery.txt" 2>&1
tail -60 "$out/delivery.txt"
