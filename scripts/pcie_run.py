#!/usr/bin/env python3
"""PCIe-inclusive time of a whole evpk_run (upload + prep + ndte subcycles + finish + download through the C ABI) on
the bench workload, host arrays pageable (staged copies) and page-locked with evpk_pin_host (moved in place).
Not the metric; noted in DESIGN.md."""
import os, sys, time
try:
    import torch          # before libevpk: the process must end up with ONE HIP runtime (torch bundles its own)
    torch.cuda.is_available()
except ImportError:
    torch = None
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

# caller arrays already in device memory (torch tensors): no PCIe at all
try:
    import ctypes as ct
    import numpy as np
    from cice5_amd import evpk
    if torch is None:
        raise ImportError
    s = dyn.EvpDynamics(d, f, ndte=120)
    s.init_evp(450.0)
    dev = {n: torch.from_numpy(np.ascontiguousarray(a)).cuda() for n, a in f.items() if isinstance(a, np.ndarray)}
    ptr = lambda n, T: ct.cast(dev[n].data_ptr(), T) if n in dev else None
    si, st = evpk.StepIn(), evpk.State()
    for n in evpk.STEP_IN_F64:
        setattr(si, n, ptr(n, evpk.c_f64p))
    si.aicen = si.vicen = si.aice0 = None
    st.uvel, st.vvel = ptr("uvel", evpk.c_f64p), ptr("vvel", evpk.c_f64p)
    for k in ("stressp", "stressm", "stress12"):
        setattr(st, k, (evpk.c_f64p * 4)(*[ptr(f"{k}_{c}", evpk.c_f64p) for c in (1, 2, 3, 4)]))
    st.iceumask = ptr("iceumask", evpk.c_i32p)
    for n in evpk.STATE_OUT_F64:
        setattr(st, n, ptr(n, evpk.c_f64p))
    st.icetmask, st.strength = None, None
    torch.cuda.synchronize()
    ts = []
    for k in range(4):
        t = time.perf_counter()
        assert s.ctx._L.evpk_run(s.ctx._ctx, ct.byref(si), ct.byref(st)) == 0
        ts.append(time.perf_counter() - t)
    st_ = s.ctx.stats()
    print(f"evpk_run, caller arrays in device memory: {min(ts[1:])*1e3:.1f} ms per evp (loop alone {st_.loop_ms:.2f} ms)")
    s.close()
except ImportError:
    pass
