#!/usr/bin/env python3
"""Timeline of a few pairs of subcycles from a rocprofv3 --kernel-trace CSV: kernel, stream/queue, start offset, duration.
usage: trace_pair.py <kernel_trace.csv> [first_row_of_window] [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k0 = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
t0 = int(rows[k0]["Start_Timestamp"])
for r in rows[k0:k0 + n]:
    print("%9.2f us  +%7.2f us  q%-3s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                          r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]))
