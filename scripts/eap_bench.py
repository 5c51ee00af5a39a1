#!/usr/bin/env python3
"""Time eap(dt) (kdyn = 2) on one MI355X with the state resident in HBM, the same workload as bench.py's
default (3600x2700 tripole, ndte = 120), and the CPU restatement on a bounded sample.  Prints one JSON line.

    python scripts/eap_bench.py [--grid 3600x2700] [--ndte 120] [--steps 3]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="3600x2700")
    ap.add_argument("--ns", default="tripole")
    ap.add_argument("--ndte", type=int, default=120)
    ap.add_argument("--dt", type=float, default=450.0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--xblocks", type=int, default=8)
    ap.add_argument("--yblocks", type=int, default=10)
    ap.add_argument("--cpu-grid", default="360x300", help="CPU sample (0 = skip)")
    ap.add_argument("--calib", type=int, default=0, help="untimed calibration copies (1 GiB each way) for rocprofv3 --pmc runs")
    a = ap.parse_args()
    import torch
    from cice5_amd import blocks, constants as C, dyn, synth
    from cice5_amd.eap_tables import eap_tables
    nx, ny = (int(v) for v in a.grid.split("x"))
    case = synth.SynthCase(nx=nx, ny=ny, ns_boundary=C.BND_NAMES[a.ns], land="continents", ice="polar", dt=a.dt, ndte=a.ndte)
    d = blocks.create_distrb_cart(nx, ny, nx // a.xblocks, ny // a.yblocks, ns_boundary_type=a.ns)
    f = synth.make_block_fields(case, d)
    synth.add_eap_state(f)
    s = dyn.EvpDynamics(d, f, ndte=a.ndte, xmin=synth.global_min_dx(case))
    s.init_eap(a.dt, eap_tables())
    ctx = s.ctx
    ctx.upload(f)
    times = []
    for n in range(a.steps + 1):
        torch.cuda.synchronize()
        t = time.perf_counter()
        ctx.prep(); ctx.subcycle(a.ndte); ctx.finish(); ctx.sync()
        times.append(time.perf_counter() - t)
    st = ctx.stats()
    ms = 1e3 * min(times[1:])
    n_active = 0.5 * (st.icellt + st.icellu)
    out = {"what": "eap(dt), state resident in HBM", "grid": a.grid, "ns": a.ns, "ndte": a.ndte, "ms_per_eap": round(ms, 3),
           "loop_ms": round(float(st.loop_ms), 3), "icellt": int(st.icellt), "icellu": int(st.icellu),
           "cell_updates_per_s": n_active * a.ndte / (ms * 1e-3)}
    if a.calib:
        ctx.calibrate(a.calib)
    s.close()
    if a.cpu_grid != "0":
        from oracle import orc
        cx, cy = (int(v) for v in a.cpu_grid.split("x"))
        case2 = synth.SynthCase(nx=cx, ny=cy, ns_boundary=C.BND_NAMES[a.ns], land="continents", ice="polar", dt=a.dt, ndte=a.ndte)
        d2 = blocks.create_distrb_cart(cx, cy, cx, cy, ns_boundary_type=a.ns)
        f2 = synth.make_block_fields(case2, d2)
        synth.add_eap_state(f2)
        orc.set_num_threads(1) if hasattr(orc, "set_num_threads") else None
        p = orc.make_params(a.dt, a.ndte, synth.global_min_dx(case2))
        nt, nu, secs = orc.eap(d2, p, f2, eap_tables())
        out["cpu_port"] = {"grid": a.cpu_grid, "loop_seconds": round(secs, 3), "cores": 1, "cell_updates_per_s": 0.5 * (nt + nu) * a.ndte / secs}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
