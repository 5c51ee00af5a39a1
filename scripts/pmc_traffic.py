#!/usr/bin/env python3
"""HBM traffic of k_subcycle from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), calibrated on
the k_calib_copy_pair launches of the same run (known byte count, same 16 B/lane access shape), as
MI355X_MICROARCH.md prescribes ("calibrate on a known byte count in your own access pattern").

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <nxl> <nyl> [out.json]
"""
import collections
import csv
import json
import sys


def means(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
    nxl, nyl = int(sys.argv[3]), int(sys.argv[4])
    known = (nxl + 2) * (nyl + 2) * 16.0                     # bytes each way per calibration copy
    cf = known / (fetch["evpk::k_calib_copy_pair"][0] * 1024.0)
    cw = known / (write["evpk::k_calib_copy_pair"][0] * 1024.0)
    out = {"unit": "bytes per launch", "calibration": {"known_bytes_each_way": known, "fetch_factor": cf, "write_factor": cw,
                                                       "fetch_KiB": fetch["evpk::k_calib_copy_pair"][0],
                                                       "write_KiB": write["evpk::k_calib_copy_pair"][0]}}
    for k in sorted(fetch):
        if "k_subcycle" in k:
            f, n = fetch[k]
            w = write.get(k, (0.0, 0))[0]
            out[k] = {"launches": n, "fetch_KiB_raw": f, "write_KiB_raw": w,
                      "read_bytes": f * 1024.0 * cf, "write_bytes": w * 1024.0 * cw,
                      "hbm_bytes": f * 1024.0 * cf + w * 1024.0 * cw,
                      "hbm_bytes_guide_correction": f * 1024.0 * 2.0 + w * 1024.0}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 5:
        json.dump(out, open(sys.argv[5], "w"), indent=1)


if __name__ == "__main__":
    main()
