#!/usr/bin/env python3
"""PCIe-inclusive time of a whole evpk_run (upload + prep + ndte subcycles + finish + download through the C ABI) on
the bench workload, host arrays pageable (staged copies) and page-locked with evpk_pin_host (moved in place).
Not the metric; noted in DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cice5_amd import blocks, dyn, synth

nx, ny = 3600, 2700
case = synth.SynthCase(nx=nx, ny=ny, land="continents", dt=450.0)
d = blocks.create_distrb_cart(nx, ny, 450, 270)
f = synth.make_block_fields(case, d)
for pin in (False, True):
    s = dyn.EvpDynamics(d, f, ndte=120, pin_host=pin)
    s.init_evp(450.0)
    s.evp(450.0)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); s.evp(450.0); ts.append(time.perf_counter() - t)
    st = s.ctx.stats()
    n = 0.5 * (st.icellt + st.icellu) * 120
    print(f"evpk_run incl. PCIe, host arrays {'page-locked' if pin else 'pageable'}: {min(ts)*1e3:.1f} ms per evp -> "
          f"{n/min(ts):.3e} cell-updates/s (loop alone {st.loop_ms:.2f} ms)")
    s.close()
