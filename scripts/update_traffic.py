#!/usr/bin/env python3
"""Upserts one entry of profiles/traffic.json (what bench.py reports as roofline.traffic) from a PMC pass.

usage: update_traffic.py <pmc_traffic.json> <bench.json of the same command> <path the summary is kept under> [kernel_stats.csv]
(kernel_stats.csv: the rocprofv3 --kernel-trace --stats summary of the same command; its average for the dominant kernel is kept as
kernel_avg_ms, which bench.py prints beside its own event-timed average)

The entry is keyed by the workload string AND by the hash of the kernel sources (bench.py: source_sha), so a number
measured on another build of the kernels is never reported against this one."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    pmc = json.load(open(sys.argv[1]))
    line = [l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1]
    bench = json.loads(line)
    kerns = {k: v for k, v in pmc.items() if k.startswith("evpk::k_subcycle")}
    if not kerns:
        raise SystemExit("no subcycle kernel in " + sys.argv[1])
    name = max(kerns, key=lambda k: kerns[k]["launches"] * kerns[k]["hbm_bytes"])          # the dominant one: most bytes in all
    e = {"workload": bench["config"]["workload"], "source_sha": bench["roofline"]["source_sha"], "kernel": name,
         "launches_profiled": kerns[name]["launches"], "hbm_bytes_per_launch": kerns[name]["hbm_bytes"],
         "read_bytes": kerns[name]["read_bytes"], "write_bytes": kerns[name]["write_bytes"],
         "correction": pmc.get("correction"), "calibration": pmc.get("calibration"), "profile": sys.argv[3]}
    if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):
        import csv
        for r in csv.DictReader(open(sys.argv[4])):
            if r["Name"].split("(")[0].replace("void ", "") == name:
                e["kernel_avg_ms"] = float(r["AverageNs"]) / 1e6
                e["kernel_calls"] = int(r["Calls"])
    path = os.path.join(ROOT, "profiles", "traffic.json")
    db = json.load(open(path)) if os.path.exists(path) else {"entries": []}
    db["entries"] = [x for x in db["entries"] if not (x["workload"] == e["workload"] and x["source_sha"] == e["source_sha"])] + [e]
    json.dump(db, open(path, "w"), indent=1)
    print(f"{name}: {e['hbm_bytes_per_launch'] / 1e9:.4f} GB per launch, sha {e['source_sha']}, workload {e['workload'][:60]}")


if __name__ == "__main__":
    main()
