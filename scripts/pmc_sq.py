#!/usr/bin/env python3
"""Per-kernel means of the counters of a `rocprofv3 --pmc ... -d DIR` run (reads the *_results.db files under DIR).

usage: pmc_sq.py DIR [kernel-name-substring]
SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over all waves; GRBM_GUI_ACTIVE is summed
over the 8 XCDs (MI355X_MICROARCH.md, "rocprofv3 PMC slots")."""
import collections
import glob
import sqlite3
import sys

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(d + "/**/*.db", recursive=True):
    cur = sqlite3.connect(fn).cursor()
    for name, counter, value in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        acc[name.split("(")[0].replace("void ", "")][counter].append(float(value))
for k, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    if pat not in k:
        continue
    print(f"{k[:90]}  launches={max(len(v) for v in cs.values())}")
    for cname, v in sorted(cs.items()):
        print(f"    {cname:24s} {sum(v) / len(v):16.1f}")
