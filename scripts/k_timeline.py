#!/usr/bin/env python3
"""Timeline of one subcycle-kernel launch from EVPK_DEBUG_CLOCKS=<file> (per strip: start / end of its wave in 100 MHz ticks, XCC, HW_ID).
usage: k_timeline.py FILE"""
import sys
import numpy as np
rows = []
t0 = None
for ln in open(sys.argv[1]):
    if ln.startswith("#"):
        t0 = int(ln.split()[2]); continue
    k, a, b, hw = ln.split()
    rows.append((int(k), int(a), int(b), int(hw, 16)))
r = np.array([(k, a, b) for k, a, b, _ in rows], dtype=np.int64)
hw = np.array([h for *_, h in rows], dtype=np.uint64)
start, end = (r[:, 1] - t0) / 100.0, (r[:, 2] - t0) / 100.0          # microseconds after the marker kernel in front of the launch
life = end - start
print(f"strips {len(r)}  start: min {start.min():.1f} p50 {np.median(start):.1f} p90 {np.percentile(start, 90):.1f} max {start.max():.1f} us")
print(f"           end:   min {end.min():.1f} p10 {np.percentile(end, 10):.1f} p50 {np.median(end):.1f} p90 {np.percentile(end, 90):.1f} max {end.max():.1f} us")
print(f"           life:  min {life.min():.1f} p50 {np.median(life):.1f} mean {life.mean():.1f} max {life.max():.1f} us")
span = end.max()
print(f"slot-time filled: {life.sum() / (span * 2048):.3f} of 2048 wave slots x {span:.1f} us")
# resident waves over time
ts = np.linspace(0, span, 21)
print("resident waves at t:", " ".join(f"{int(((start <= t) & (end > t)).sum())}" for t in ts))
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
cu = (hw.astype(np.int64) >> 8) & 0xf
se = (hw.astype(np.int64) >> 13) & 0x7
simd = (hw.astype(np.int64) >> 4) & 0x3
key = xcc * 1000 + se * 100 + cu
u, cnt = np.unique(key, return_counts=True)
print(f"distinct (xcc, se, cu): {len(u)}; strips per CU: min {cnt.min()} max {cnt.max()} hist {dict(zip(*np.unique(cnt, return_counts=True)))}")
ks = key * 10 + simd
u2, c2 = np.unique(ks, return_counts=True)
print(f"distinct SIMDs used: {len(u2)}; waves per SIMD hist {dict(zip(*np.unique(c2, return_counts=True)))}")
print("per XCC strips:", dict(zip(*np.unique(xcc, return_counts=True))))
