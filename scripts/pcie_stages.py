#!/usr/bin/env python3
"""Where the PCIe-inclusive evp of a CPU-resident host goes: stage times of the resident-state + sparse-transfer mode
(state on the device, inputs up, the every-step outputs down) on the bench workload.  Not the metric."""
import os, sys, time
import torch
torch.cuda.is_available()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cice5_amd import blocks, dyn, synth, constants as C

nx, ny = 3600, 2700
case = synth.SynthCase(nx=nx, ny=ny, land="continents", dt=450.0, ns_boundary=C.BND_NAMES["tripole"])
d = blocks.create_distrb_cart(nx, ny, 450, 270, ns_boundary_type="tripole")
f = synth.make_block_fields(case, d)
for sparse in (0, 1, 2):
    s = dyn.EvpDynamics(d, f, ndte=120, pin_host=True, resident=True, outputs=dyn.EVERY_STEP_OUTPUTS, sparse_io=sparse)
    s.init_evp(450.0)
    s.evp(450.0); s.evp(450.0)
    ctx = s.ctx
    out = {n: f[n] for n in dyn.EVERY_STEP_OUTPUTS}
    acc = {}
    for rep in range(3):
        t0 = time.perf_counter(); ctx.upload_inputs(f); ctx.sync(); t1 = time.perf_counter()
        ctx.prep(); ctx.sync(); t2 = time.perf_counter()
        ctx.subcycle(120); ctx.sync(); t3 = time.perf_counter()
        ctx.finish(); ctx.sync(); t4 = time.perf_counter()
        ctx.download(out); ctx.sync(); t5 = time.perf_counter()
        for k, v in (("upload_inputs", t1 - t0), ("prep", t2 - t1), ("subcycle", t3 - t2), ("finish", t4 - t3), ("download_outputs", t5 - t4), ("total", t5 - t0)):
            acc[k] = min(acc.get(k, 1e9), v)
    print(("sparse_io=%d " % sparse) + "  ".join(f"{k} {1e3 * v:.2f} ms" for k, v in acc.items()))
    s.close()
