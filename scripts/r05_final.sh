#!/bin/bash
# Round 5, final GPU pass on the final library: the whole GPU suite, seeded fuzz (with summary lines), then the profile pass
cd "$(dirname "$0")/.." || exit 1
tag=${1:-r05_final2}; out=gpurun_out/$tag; mkdir -p "$out"
timeout 2400 python3 -m pytest tests -m gpu -x -q > "$out/gpu_tests.txt" 2>&1
grep -E "passed|failed" "$out/gpu_tests.txt" | tail -2
EVPK_LIB=cice5_amd/libevpk_exp.so timeout 900 python3 -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "three_subcycle" 2>&1 | grep -E "passed|failed" > "$out/exp_tests.txt"
{
echo "Seeded fuzz on the final library of round 5 (one MI355X), pytest summary lines:"
echo "tests/test_fuzz_gpu.py EVPK_FUZZ_N=1500 EVPK_FUZZ_BASE=510000 EVPK_FUZZ_R_N=150 EVPK_FUZZ_E_N=150 (evp / remap / eap draws; modes incl. the rolling tile kernel):"
EVPK_FUZZ_N=1500 EVPK_FUZZ_BASE=510000 EVPK_FUZZ_R_N=150 EVPK_FUZZ_E_N=150 timeout 2400 python3 -m pytest tests/test_fuzz_gpu.py -x -q 2>&1 | tee -a "$out/fuzz_full.txt" | grep -E "passed|failed|FAILED|Error" | tail -4
echo "EVPK_FUZZ_BIG=1 EVPK_FUZZ_N=60 EVPK_FUZZ_BASE=520000 (large grids):"
EVPK_FUZZ_BIG=1 EVPK_FUZZ_N=60 EVPK_FUZZ_BASE=520000 EVPK_FUZZ_R_N=0 EVPK_FUZZ_E_N=0 timeout 1200 python3 -m pytest tests/test_fuzz_gpu.py -x -q -k random_configuration 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -3
echo "tests/test_multirank_gpu.py -k random, EVPK_FUZZ_MR_N=40 EVPK_FUZZ_BASE=530000 (2-5 rank processes on one GPU, both transports, idle ranks among the draws):"
EVPK_FUZZ_MR_N=40 EVPK_FUZZ_BASE=530000 timeout 2400 python3 -m pytest tests/test_multirank_gpu.py -x -q -k "random" 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -3
echo "tests/test_fortran_host.py -k random, EVPK_FUZZ_F_N=20 EVPK_FUZZ_BASE=540000:"
EVPK_FUZZ_F_N=20 EVPK_FUZZ_BASE=540000 timeout 900 python3 -m pytest tests/test_fortran_host.py -x -q -k random 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -3
} > "$out/fuzz.txt" 2>&1
cat "$out/fuzz.txt"
# the self-verifying multi-rank bench line at FULL size: 2 and 4 rank processes sharing the one GPU over the peer-mapped transport (a functional
# check, not a scaling number): every rank's blocks against the one-rank device run of the whole grid, bit for bit
for n in 2 4; do
  EVPK_FORCE_DEVICE=0 timeout 900 python3 bench.py --gpus $n --steps 2 --warmup 1 --transport ipc 2>/dev/null | grep '^{' | tail -1 > "$out/bench_n${n}_shared_gpu_ipc.json"
  python3 -c "
import json,sys
o=json.load(open('$out/bench_n${n}_shared_gpu_ipc.json')); v=o.get('verify',{})
print('bench --gpus $n (shared GPU, ipc): ms/evp %.2f  verify.bit_identical=%s against=%s values=%s blocks=%s/%s checks=%s' % (o['ms_per_step'], v.get('bit_identical'), v.get('against'), v.get('values_compared'), v.get('blocks_compared'), v.get('blocks_total'), o.get('checks')))" 2>&1 | tee -a "$out/bench_multirank_verify.txt"
done
bash scripts/profile_round.sh "$tag" > "$out/profile_round.log" 2>&1
tail -5 "$out/profile_round.log"
cat "$out/traffic_updates.txt" 2>/dev/null
