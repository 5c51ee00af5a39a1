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
