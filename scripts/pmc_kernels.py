#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate passes, --kernel-trace only), for
kernels whose name matches a pattern.  The factors that turn the raw counters into bytes are CALIBRATED IN THE SAME RUN on
copies of a known byte count (1 GiB each way between two buffers four times the Infinity Cache): k_calib_copy_big (16 B per
lane; the guide's gfx950 correction FETCH x 2, WRITE x 1) and k_calib_copy_big8 (8 B per lane, the access shape of the plain
planes of horizontal_remap / eap) -- MI355X_MICROARCH.md, HBM: "other access widths are uncalibrated: calibrate on a known
byte count in your own access pattern".

usage: pmc_kernels.py <fetch counter_collection.csv> <write counter_collection.csv> <pattern> <out.json> [8|16]
"""
import collections
import csv
import json
import re
import sys

BIG = float(1 << 30)


def means(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
    pat, width = re.compile(sys.argv[3]), (sys.argv[5] if len(sys.argv) > 5 else "8")
    cal = {}
    for k in ("evpk::k_calib_copy_big", "evpk::k_calib_copy_big8"):
        if k in fetch and k in write:
            f, w = fetch[k][0] * 1024.0, write[k][0] * 1024.0
            cal[k] = {"known_bytes_each_way": BIG, "fetch_raw_bytes": f, "write_raw_bytes": w, "fetch_factor": BIG / f, "write_factor": BIG / w}
    key = "evpk::k_calib_copy_big8" if width == "8" else "evpk::k_calib_copy_big"
    ff = cal.get(key, {}).get("fetch_factor", 2.0)
    wf = cal.get(key, {}).get("write_factor", 1.0)
    out = {"unit": "bytes per launch", "calibration": cal, "factors_used": {"fetch": ff, "write": wf, "from": key}}
    tot = 0.0
    for k in sorted(fetch):
        if pat.search(k):
            f, n = fetch[k]
            w = write.get(k, (0.0, 0))[0]
            rb, wb = f * 1024.0 * ff, w * 1024.0 * wf
            out[k] = {"launches": n, "read_bytes": rb, "write_bytes": wb, "hbm_bytes": rb + wb, "hbm_bytes_all_launches": (rb + wb) * n}
            tot += (rb + wb) * n
    out["total_bytes_all_launches"] = tot
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    for k, v in out.items():
        if isinstance(v, dict) and "hbm_bytes" in v:
            print(f"{k[:60]:60s} n={v['launches']:4d} read {v['read_bytes']/1e9:8.3f} GB  write {v['write_bytes']/1e9:8.3f} GB per launch")
    print("calibration:", {k: (round(v["fetch_factor"], 4), round(v["write_factor"], 4)) for k, v in cal.items()}, "total GB", round(tot / 1e9, 2))


if __name__ == "__main__":
    main()
