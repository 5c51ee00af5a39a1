#!/bin/bash
# Round 5, fifth GPU pass: the whole GPU suite on the final library, the rejected kernels' tests against libevpk_exp.so, then seeded fuzz
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05_v5}; out=gpurun_out/$tag; mkdir -p "$out"
timeout 2400 python3 -m pytest tests -m gpu -x -q > "$out/gpu_tests.txt" 2>&1
tail -6 "$out/gpu_tests.txt"
EVPK_LIB=cice5_amd/libevpk_exp.so timeout 900 python3 -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "three_subcycle or two_subcycle" > "$out/exp_tests.txt" 2>&1
tail -3 "$out/exp_tests.txt"
{
echo "== seeded fuzz on the final library of round 5 (one MI355X)"
EVPK_FUZZ_N=3000 EVPK_FUZZ_BASE=210000 EVPK_FUZZ_R_N=300 EVPK_FUZZ_E_N=300 timeout 2400 python3 -m pytest tests/test_fuzz_gpu.py -x -q 2>&1 | tail -4
EVPK_FUZZ_BIG=1 EVPK_FUZZ_N=120 EVPK_FUZZ_BASE=220000 EVPK_FUZZ_R_N=0 EVPK_FUZZ_E_N=0 timeout 1200 python3 -m pytest tests/test_fuzz_gpu.py -x -q -k random_configuration 2>&1 | tail -3
EVPK_FUZZ_MR_N=150 EVPK_FUZZ_BASE=230000 timeout 2400 python3 -m pytest tests/test_multirank_gpu.py -x -q -k "random" 2>&1 | tail -3
EVPK_FUZZ_F_N=40 EVPK_FUZZ_BASE=240000 timeout 900 python3 -m pytest tests/test_fortran_host.py -x -q -k random 2>&1 | tail -3
} > "$out/fuzz.txt" 2>&1
cat "$out/fuzz.txt"
