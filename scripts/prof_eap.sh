#!/bin/bash
# kernel trace of eap(dt) at the bench size.  usage: scripts/prof_eap.sh [out-dir under gpurun_out]
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/${1:-eap_prof}
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o rp -- python3 scripts/eap_bench.py --cpu-grid 0 --steps 2 > "$out/bench.log" 2>&1
f=$(find "$out/trace" -name "*kernel_stats.csv" | tail -1)
cp "$f" "$out/kernel_stats.csv"
rm -rf "$out/trace"
python3 - "$out/kernel_stats.csv" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print(r["Name"][:70].ljust(70), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
grep '"what"' "$out/bench.log" | cut -c1-300
# HBM traffic per kernel (separate FETCH_SIZE / WRITE_SIZE passes, ONE eap call + calibration copies) and the SQ counters
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -o f -- python3 scripts/eap_bench.py --cpu-grid 0 --steps 1 --ndte 20 --calib 2 > "$out/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -o w -- python3 scripts/eap_bench.py --cpu-grid 0 --steps 1 --ndte 20 --calib 2 > "$out/pmc_write.log" 2>&1
python3 scripts/pmc_kernels.py "$(find "$out/pmc_fetch" -name "*counter_collection.csv" | tail -1)" "$(find "$out/pmc_write" -name "*counter_collection.csv" | tail -1)" \
    "k_eap_" "$out/pmc_traffic.json" 16
rm -rf "$out/pmc_fetch" "$out/pmc_write"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d "$out/pmc_sq" -o q -- python3 scripts/eap_bench.py --cpu-grid 0 --steps 1 --ndte 20 > "$out/pmc_sq.log" 2>&1
python3 scripts/pmc_sq.py "$out/pmc_sq" k_eap_ > "$out/sq_counters.txt" 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d "$out/pmc_tc" -o q -- python3 scripts/eap_bench.py --cpu-grid 0 --steps 1 --ndte 20 > "$out/pmc_tc.log" 2>&1
python3 scripts/pmc_sq.py "$out/pmc_tc" k_eap_ >> "$out/sq_counters.txt" 2>&1
rm -rf "$out/pmc_sq" "$out/pmc_tc"
cat "$out/sq_counters.txt"
