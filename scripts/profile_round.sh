#!/bin/bash
# One GPU box, everything the profiles/ directory of a round is made of.  usage: scripts/profile_round.sh <tag>
# Output: gpurun_out/<tag>/  (copy the summaries into profiles/<tag>/).  rocprofv3 gets the program itself after `--`.
cd "$(dirname "$0")/.." || exit 1
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="bench.py --steps 4 --warmup 1 --cpu-subcycles 0"
# 1. the bench line as the driver runs it (CPU baseline included)
python3 bench.py > "$out/bench_n1.json" 2> "$out/bench_n1.err"
# 2. kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- python3 $B > "$out/trace.log" 2>&1
cp "$(ls "$out"/trace/*kernel_stats.csv | head -1)" "$out/kernel_stats.csv" 2>/dev/null
# 3. HBM traffic: separate FETCH_SIZE / WRITE_SIZE passes with calibration copies
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -o f -- python3 $B --calib 3 > "$out/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -o w -- python3 $B --calib 3 > "$out/pmc_write.log" 2>&1
python3 scripts/pmc_traffic.py "$(ls "$out"/pmc_fetch/*counter_collection.csv | head -1)" "$(ls "$out"/pmc_write/*counter_collection.csv | head -1)" 3600 2700 "$out/pmc_traffic.json" > /dev/null
# 4. SQ counters of the hot kernel (two passes of <= 8 SQ counters)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d "$out/pmc_sq" -o sq -- python3 $B > "$out/pmc_sq.log" 2>&1
python3 scripts/pmc_sq.py "$out/pmc_sq" k_subcycle2p > "$out/sq_counters.txt"
# 5. other configurations, PCIe-inclusive run, x-slab machinery on one GPU
bash scripts/configs.sh > "$out/configs.txt" 2>&1
python3 scripts/pcie_run.py > "$out/pcie.txt" 2>&1
bash scripts/zone_sweep.sh > "$out/zone_sweep.txt" 2>&1
EVPK_FORCE_DEVICE=0 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 4 --steps 2 --warmup 1 --cpu-subcycles 0 --transport shm 2> "$out/shm4.err" | tail -1 > "$out/bench_shm_4ranks_one_gpu.json"
rm -rf "$out"/trace "$out"/pmc_fetch "$out"/pmc_write "$out"/pmc_sq
ls -la "$out"
