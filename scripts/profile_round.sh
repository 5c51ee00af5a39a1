#!/bin/bash
# One GPU box, everything the profiles/ directory of a round is made of.  The pass labels its numbers with the hash of the csrc/ files
# of the snapshot and measures the libevpk.so of the snapshot: build first, and leave csrc/ alone until gpurun has sent the tree (a
# call can queue for minutes before it takes the snapshot -- round 4 lost a pass to an edit made in that window).  usage: scripts/profile_round.sh <tag> [quick]
# Output: gpurun_out/<tag>/  (copy the summaries into profiles/<tag>/).  rocprofv3 gets the program itself after `--`;
# the counter passes carry --kernel-trace only (no other trace domain beside --pmc).
cd "$(dirname "$0")/.." || exit 1
tag=${1:-prof}
quick=${2:-}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
COMMON="--steps 4 --warmup 1 --cpu-subcycles 0"
# 1. the bench line as the driver runs it (CPU baseline and extras included)
python3 bench.py > "$out/bench_n1.json" 2> "$out/bench_n1.err"
# 2. per workload: kernel trace, then FETCH_SIZE / WRITE_SIZE passes (they do not fit in one) with calibration copies
profile() {   # name, bench flags
  local name=$1; shift
  local w=$out/$name; mkdir -p "$w"
  python3 bench.py $COMMON "$@" > "$w/bench.json" 2> "$w/bench.err"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$w/trace" -o t -- python3 bench.py $COMMON "$@" > "$w/trace.log" 2>&1
  cp "$(ls "$w"/trace/*/*kernel_stats.csv "$w"/trace/*kernel_stats.csv 2>/dev/null | head -1)" "$w/kernel_stats.csv" 2>/dev/null
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$w/pmc_fetch" -o f -- python3 bench.py $COMMON "$@" --calib 3 > "$w/pmc_fetch.log" 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$w/pmc_write" -o w -- python3 bench.py $COMMON "$@" --calib 3 > "$w/pmc_write.log" 2>&1
  python3 scripts/pmc_traffic.py "$(ls "$w"/pmc_fetch/*/*counter_collection.csv "$w"/pmc_fetch/*counter_collection.csv 2>/dev/null | head -1)" \
      "$(ls "$w"/pmc_write/*/*counter_collection.csv "$w"/pmc_write/*counter_collection.csv 2>/dev/null | head -1)" "$w/pmc_traffic.json" "$name" > /dev/null \
    && python3 scripts/update_traffic.py "$w/pmc_traffic.json" "$w/bench.json" "profiles/$tag/$name/pmc_traffic.json" "$w/kernel_stats.csv" >> "$out/traffic_updates.txt"
  rm -rf "$w"/trace "$w"/pmc_fetch "$w"/pmc_write
}
profile cfg5_3600x2700_tripole                                   # the bench default = BASELINE config 5's grid and boundary, ndte = 120
profile cfg5_3600x2700_open --ns open
if [ -z "$quick" ]; then
profile cfg5_3600x2700_tripole_ndte240 --ndte 240
profile cfg2_gx1_320x384 --grid 320x384 --xblocks 1 --yblocks 1 --dt 3600 --land rows --ns open
profile cfg3_360x300_24blocks --grid 360x300 --xblocks 24 --yblocks 1 --dt 3600 --ns open
profile cfg4_1440x1080 --grid 1440x1080 --xblocks 8 --yblocks 4 --dt 1800 --ns open
# 3. SQ counters of the hot kernel (one pass of <= 8 SQ counters)
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d "$out/pmc_sq" -o sq -- python3 bench.py $COMMON > "$out/pmc_sq.log" 2>&1
python3 scripts/pmc_sq.py "$out/pmc_sq" k_subcycle2p > "$out/sq_counters.txt"
rm -rf "$out"/pmc_sq
# 3b. timeline of one launch of the hot kernel (per-strip start / end clocks written by the kernel itself), prep / finish kernels one by one
EVPK_DEBUG_CLOCKS=$out/tl.txt python3 bench.py --steps 2 --warmup 2 --cpu-subcycles 0 --no-variants > /dev/null 2>&1
python3 scripts/k_timeline.py "$out/tl.txt" > "$out/timeline_pair.txt" 2>&1; rm -f "$out/tl.txt"
bash scripts/prep_prof.sh "$tag/prep" > /dev/null 2>&1; rm -f "$out/prep/kernel_trace.csv" "$out/prep/trace.log"
# 4. other configurations in one table, PCIe-inclusive run, x-slab machinery on one GPU
bash scripts/configs.sh > "$out/configs.txt" 2>&1
python3 scripts/pcie_run.py > "$out/pcie.txt" 2>&1
bash scripts/zone_sweep.sh > "$out/zone_sweep.txt" 2>&1
bash scripts/xp_compare.sh > "$out/xp_compare.txt" 2>&1
bash scripts/tile_ab.sh > "$out/tile_ab.txt" 2>&1
# 5. rows f-3 / f-4: transport_remap and eap at the bench size (timing line, kernel trace)
python3 scripts/remap_bench.py > "$out/remap_bench.json" 2> "$out/remap_bench.err"
bash scripts/prof_remap.sh "$tag/remap" > "$out/remap_kernels.txt" 2>&1
python3 scripts/eap_bench.py > "$out/eap_bench.json" 2> "$out/eap_bench.err"
bash scripts/prof_eap.sh "$tag/eap" > "$out/eap_kernels.txt" 2>&1
fi
cp profiles/traffic.json "$out/traffic.json" 2>/dev/null
ls -la "$out"
