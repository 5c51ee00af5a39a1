#!/bin/bash
# kernel trace of evpk_transport_remap at the bench size.  usage: scripts/prof_remap.sh [out-dir under gpurun_out]
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
out=gpurun_out/${1:-remap_prof}
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o rp -- python3 scripts/remap_bench.py --cpu-grid 0 --reps 3 > "$out/bench.log" 2>&1
f=$(find "$out/trace" -name "*kernel_stats.csv" | tail -1)
cp "$f" "$out/kernel_stats.csv"
rm -rf "$out/trace"
python3 - "$out/kernel_stats.csv" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(r["Name"][:70].ljust(70), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
tail -1 "$out/bench.log"
# HBM traffic per kernel: separate FETCH_SIZE / WRITE_SIZE passes (ONE call of the remap + calibration copies)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -o f -- python3 scripts/remap_bench.py --cpu-grid 0 --reps 1 --calib 2 > "$out/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -o w -- python3 scripts/remap_bench.py --cpu-grid 0 --reps 1 --calib 2 > "$out/pmc_write.log" 2>&1
python3 scripts/pmc_kernels.py "$(find "$out/pmc_fetch" -name "*counter_collection.csv" | tail -1)" "$(find "$out/pmc_write" -name "*counter_collection.csv" | tail -1)" \
    "k_remap|k_gather_planes|k_scatter_planes|k_state_|k_planes_" "$out/pmc_traffic.json" 8
rm -rf "$out/pmc_fetch" "$out/pmc_write"

# the whole call's bytes into profiles/traffic.json (bench.py: next_rows.transport_remap.roofline.traffic_pmc), keyed by the kernel sources
python3 - "$out/pmc_traffic.json" "profiles/${1:-remap_prof}" <<'PY'
import hashlib, json, os, sys
pmc = json.load(open(sys.argv[1]))
h = hashlib.sha256()
for n in ("evpk_kernels.hip", "evpk_api.hip", "evpk_internal.h", "evpk_remap.hip", "evpk_eap.hip", "evpk_fmath.h"):
    h.update(open(os.path.join("cice5_amd", "csrc", n), "rb").read())
calls = 2.0      # --reps 1 = one warm-up + one timed call
tot = sum(v["hbm_bytes_all_launches"] for k, v in pmc.items() if isinstance(v, dict) and "hbm_bytes_all_launches" in v) / calls
path = "profiles/traffic.json"
db = json.load(open(path)) if os.path.exists(path) else {"entries": []}
e = {"source_sha": h.hexdigest()[:16], "cells": 3600 * 2700, "fields": 66, "hbm_bytes_per_call": tot, "profile": sys.argv[2],
     "factors": pmc.get("factors_used")}
db["remap"] = [x for x in db.get("remap", []) if x.get("source_sha") != e["source_sha"]] + [e]
json.dump(db, open(path, "w"), indent=1)
print("remap: %.2f GB per call, sha %s" % (tot / 1e9, e["source_sha"]))
PY
