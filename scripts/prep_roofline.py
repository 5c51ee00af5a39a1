#!/usr/bin/env python3
"""Bytes / time rows of the once-per-evp kernels from a rocprofv3 kernel trace of bench.py (steady-state evps: the median launch).
usage: prep_roofline.py kernel_trace.csv [bench flags: --grid NXxNY]"""
import csv, sys, collections
import numpy as np
nx, ny = 3600, 2700
for k, a in enumerate(sys.argv):
    if a == "--grid": nx, ny = (int(v) for v in sys.argv[k + 1].split("x"))
cells = (nx + 2) * (ny + 2)
rows = list(csv.DictReader(open(sys.argv[1])))
dur = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("evpk::", "")
    dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
# compulsory bytes per CELL the kernel touches when every tile is active (fp64 planes 8 B, int masks 4 B, byte masks 1 B), by hand from the
# kernels' loads / stores; `act` = the kernel skips tiles outside act_any / act_ice, so its bytes scale with the active share
model = {
    "k_prep1a":       (6 * 8 + 4 + 8 + 1 + 16 + 40, "reads 3 pairs + tmask, writes tmass, tmphm, the wind pair, zeroes 5 diagnostics where a T cell was active"),
    "k_prep1b":       (9 * 1 + 4 + 8, "3x3 of tmphm, tmask, writes icetmask"),
    "k_to_ugrid4":    (4 * 8 + 8 + 4 * 8 + 8, "4 T fields + tarea in (each cell read once, 4-fold reuse in cache), 4 U fields out, uarea"),
    "k_prep2":        (8 + 1 + 16 + 4 + 4 + 16 + 16 + 16 + 8 + 8 + 16 + 10 * 8, "icetm, cmask, aiu/umass, masks, uocn/vocn, u/v, fcor, strair, Cw in; 10 planes + masks out where active"),
    "k_strip_flags2": (1, "cmask bytes under the strips"),
    "k_finish":       (16 + 16 + 8 + 8 + 8 + 4 + 16 + 16, "u/v, uocn/vocn, aiu, fm, Cw, mask in; strocnx/y, work pair out"),
    "k_to_tgrid2":    (16 + 8 + 8 + 16, "the work pair + uarea in (4-fold reuse), tarea, strocnxT/yT out"),
}
print(f"{'kernel':18s} {'launches':>8s} {'median us':>10s} {'min us':>8s} {'max us':>8s}   bytes/cell  GB(all cells)  GB/s at the median (all cells; active share ~0.35 for the skipping kernels)")
tot = 0.0
for name, v in sorted(dur.items(), key=lambda kv: -np.median(kv[1]) * len(kv[1])):
    if name.startswith("k_subcycle") or "rocclr" in name or name.startswith("k_calib") or name.startswith("k_gather") or name.startswith("k_scatter"):
        continue
    med = float(np.median(v))
    per_evp = med * len(v)
    b = model.get(name)
    extra = ""
    if b:
        gb = b[0] * cells / 1e9
        extra = f"   {b[0]:6d}     {gb:8.3f}       {gb / (med * 1e-6):8.0f}    {b[1]}"
    print(f"{name[:18]:18s} {len(v):8d} {med:10.1f} {min(v):8.1f} {max(v):8.1f}{extra}")
