#!/usr/bin/env python3
"""HBM traffic of the subcycle kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; the two do not fit in one
pass: MI355X_MICROARCH.md, rocprofv3 PMC slots), corrected as that guide's HBM section prescribes for gfx950:

    bytes read    = FETCH_SIZE [KiB] x 1024 x 2      (FETCH_SIZE tallies 128-B requests at 64 B for 16 B-per-lane reads)
    bytes written = WRITE_SIZE [KiB] x 1024 x 1

and checked in the same run against copies of a known byte count with the hot kernel's access shape (16 B per lane):
k_calib_copy_big moves 1 GiB each way between two buffers four times the Infinity Cache (the factor the correction must
reproduce: 2.0 / 1.0), k_calib_copy_pair one 155 MB pair plane in place (cache-resident: its reads are partly served
on the die, which is why round 1's own factor came out at 1.77).

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [label]
"""
import collections
import csv
import json
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
    out = {"unit": "bytes per launch", "correction": {"fetch": 2.0, "write": 1.0, "source": "MI355X_MICROARCH.md, HBM"},
           "label": sys.argv[4] if len(sys.argv) > 4 else ""}
    cal = {}
    for k, known in (("evpk::k_calib_copy_big", BIG), ("evpk::k_calib_copy_pair", None)):
        if k in fetch and k in write:
            f, w = fetch[k][0] * 1024.0, write[k][0] * 1024.0
            cal[k] = {"fetch_raw_bytes": f, "write_raw_bytes": w, "launches": fetch[k][1]}
            if known:
                cal[k].update({"known_bytes_each_way": known, "fetch_factor": known / f, "write_factor": known / w})
    out["calibration"] = cal
    for k in sorted(fetch):
        if "k_subcycle" in k:
            f, n = fetch[k]
            w = write.get(k, (0.0, 0))[0]
            out[k] = {"launches": n, "fetch_KiB_raw": f, "write_KiB_raw": w,
                      "read_bytes": f * 1024.0 * 2.0, "write_bytes": w * 1024.0,
                      "hbm_bytes": f * 1024.0 * 2.0 + w * 1024.0}
    print(json.dumps(out, indent=1))
    json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
