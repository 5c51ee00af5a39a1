#!/usr/bin/env python3
"""Time evpk_transport_remap (horizontal_remap) on one MI355X with the ice state resident in HBM (the
caller's mm / tm are device arrays), and the CPU restatement on a bounded sample beside it.

    python scripts/remap_bench.py --grid 3600x2700 --ns tripole --ncat 5 --trcr 0,1,1,1,1,2,1,1,1,1

--trcr: trcr_depend of the tracers beyond hice, hsno (0 area, 1 ice volume, 2 snow volume): the default is Tsfc, 4 x qice,
qsno, 4 x sice -- the tracer set of a COSIMA run (nilyr = 4, nslyr = 1), ntrace = 12.
Prints one JSON line.
"""
import argparse
import ctypes as ct
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def tables(trcr_depend):
    """init_transport (ice_transport_driver.F90:88-125)"""
    ntrace = 2 + len(trcr_depend)
    depend, ttype = np.zeros(ntrace, np.int32), np.ones(ntrace, np.int32)
    for nt, dep in enumerate(trcr_depend):
        depend[2 + nt] = dep
        ttype[2 + nt] = 1 if dep == 0 else (3 if dep > 2 and trcr_depend[dep - 3] > 0 else 2)
    has = np.zeros(ntrace, np.int32)
    for nt in range(ntrace):
        if depend[nt] > 0:
            has[depend[nt] - 1] = 1
    return ttype, depend, has


def state(d, f, ncat, ntrace, I, J, nx, ny):
    mm = np.zeros((d.nblocks, ncat + 1, d.ny_block, d.nx_block))
    tm = np.zeros((d.nblocks, ncat, ntrace, d.ny_block, d.nx_block))
    for b in range(d.nblocks):
        x = (2 * np.pi * ((I[b] - 1) % nx + 1) / nx)[None, :]
        y = (np.pi * J[b] / ny)[:, None]
        ocean = f["tmask"][b] > 0
        ice = ocean & (np.sin(3 * x + 0.5) * np.cos(2 * y) > -0.3)
        tot = np.zeros((d.ny_block, d.nx_block))
        for n in range(1, ncat + 1):
            a = np.where(ice, 0.25 * (1 + 0.8 * np.sin(n * x + y)) / ncat * 2.0, 0.0)
            mm[b, n] = a
            tot += a
            for k in range(ntrace):
                tm[b, n - 1, k] = np.where(a > 0, (k + 1.0) * (0.5 + 0.3 * np.cos(2 * x - y + k)), 0.0)
        mm[b, 0] = np.where(ocean, 1.0 - tot, 0.0)
    return mm, tm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="3600x2700")
    ap.add_argument("--ns", default="tripole")
    ap.add_argument("--xblocks", type=int, default=8)
    ap.add_argument("--ncat", type=int, default=5)
    ap.add_argument("--trcr", default="0,1,1,1,1,2,1,1,1,1")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu-grid", default="360x300", help="grid of the CPU sample (same state per cell, 0 = skip)")
    ap.add_argument("--calib", type=int, default=0, help="untimed calibration copies (1 GiB each way, 16 and 8 B per lane) for rocprofv3 --pmc runs")
    a = ap.parse_args()
    import torch
    from cice5_amd import blocks, constants as C, dyn, synth
    nx, ny = (int(v) for v in a.grid.split("x"))
    trcr = [int(v) for v in a.trcr.split(",")] if a.trcr else []
    ttype, depend, has = tables(trcr)
    ntrace = len(ttype)

    def case_of(nx, ny, xb):
        case = synth.SynthCase(nx=nx, ny=ny, ns_boundary=C.BND_NAMES[a.ns], land="continents")
        d = blocks.create_distrb_cart(nx, ny, nx // xb, ny, ns_boundary_type=a.ns)
        f = synth.make_block_fields(case, d)
        synth.add_remap_grid(case, d, f)
        I, J = blocks.block_index_windows(d)
        for b in range(d.nblocks):
            x = (2 * np.pi * ((I[b] - 1) % nx + 1) / nx)[None, :]
            y = (np.pi * J[b] / ny)[:, None]
            f["uvel"][b] = 0.3 * np.sin(2 * x) * np.cos(y) * f["umask"][b]
            f["vvel"][b] = 0.2 * np.cos(3 * x + 1.0) * np.sin(2 * y) * f["umask"][b]
        mm, tm = state(d, f, a.ncat, ntrace, I, J, nx, ny)
        return case, d, f, mm, tm

    case, d, f, mm, tm = case_of(nx, ny, a.xblocks)
    dt = 0.3 * synth.global_min_dx(case) / 0.3                   # departure points up to 0.3 cells away
    s = dyn.EvpDynamics(d, f, ndte=120, xmin=synth.global_min_dx(case))
    s.set_evp_parameters(3600.0)
    s.ctx.upload(f)                                               # (ghost cells of uvel, vvel: refreshed by the library's own halo where it needs them)
    s.ctx.remap_init(f["dxu"], f["dyu"], f["hm"])
    L, ctx = s.ctx._L, s.ctx._ctx
    dmm, dtm = torch.from_numpy(mm).cuda(), torch.from_numpy(tm).cuda()
    m0, t0 = dmm.clone(), dtm.clone()
    p64, p32 = ct.POINTER(ct.c_double), ct.POINTER(ct.c_int32)
    args = lambda: (ctx, float(dt), a.ncat, ntrace, ct.cast(dmm.data_ptr(), p64), ct.cast(dtm.data_ptr(), p64),
                    ttype.ctypes.data_as(p32), depend.ctypes.data_as(p32), has.ctypes.data_as(p32), 3, 1, 0)
    times = []
    for r in range(a.reps + 1):
        dmm.copy_(m0); dtm.copy_(t0)
        torch.cuda.synchronize()
        t = time.perf_counter()
        rc = L.evpk_transport_remap(*args())
        times.append(time.perf_counter() - t)
        assert rc == 0, (rc, L.evpk_last_error(ctx))
    moved = float((dmm - m0).abs().max())
    if a.calib:
        s.ctx.calibrate(a.calib)
    s.close()
    ms = 1e3 * min(times[1:])
    cells = nx * ny
    out = {"what": "evpk_transport_remap, state resident in HBM", "grid": a.grid, "ns": a.ns, "ncat": a.ncat, "ntrace": ntrace,
           "ms_per_call": round(ms, 3), "first_call_ms": round(1e3 * times[0], 1), "cells": cells,
           "field_cell_updates_per_s": cells * (a.ncat + 1 + a.ncat * ntrace) / (ms * 1e-3), "max_area_change": moved,
           # each plane of mm, tm read and written once: the least any implementation moves
           "compulsory_GB": 2 * 8 * cells * (a.ncat + 1 + a.ncat * ntrace) / 1e9}
    out["GBps_of_compulsory"] = out["compulsory_GB"] / (ms * 1e-3)
    if a.cpu_grid != "0":
        from oracle import orc
        cx, cy = (int(v) for v in a.cpu_grid.split("x"))
        case2, d2, f2, mm2, tm2 = case_of(cx, cy, 1)
        dt2 = 0.3 * synth.global_min_dx(case2) / 0.3
        t = time.perf_counter()
        rc = orc.horizontal_remap(d2, dt2, f2, mm2, tm2, ttype, depend, has)
        sec = time.perf_counter() - t
        assert rc == 0
        out["cpu_port"] = {"grid": a.cpu_grid, "seconds": round(sec, 3), "cores": 1,
                           "field_cell_updates_per_s": cx * cy * (a.ncat + 1 + a.ncat * ntrace) / sec}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
