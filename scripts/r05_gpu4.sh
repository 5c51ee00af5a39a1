#!/bin/bash
# Round 5, fourth GPU pass: the whole GPU suite on the current library, then the PCIe-inclusive stage times with and without the fused transfers
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05_v4}; out=gpurun_out/$tag; mkdir -p "$out"
timeout 2400 python3 -m pytest tests -m gpu -x -q > "$out/gpu_tests.txt" 2>&1
tail -15 "$out/gpu_tests.txt"
{
echo "== PCIe-inclusive evp, resident state + every-step outputs, 3600x2700 tripole: stage times (ms); one launch per array"
EVPK_XFER_FUSED=0 timeout 600 python3 scripts/pcie_stages.py 2>&1 | grep "sparse_io"
echo "== ... up to twelve arrays per launch (default)"
timeout 600 python3 scripts/pcie_stages.py 2>&1 | grep "sparse_io"
} > "$out/pcie.txt" 2>&1
cat "$out/pcie.txt"
