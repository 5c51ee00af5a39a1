"""where does k_subcycle3w differ from k_subcycle?  (debugging aid)"""
import os, sys
os.environ.setdefault("EVPK_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cice5_amd", "libevpk_exp.so"))      # k_subcycle3w: experimental build only
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cice5_amd import blocks, constants as C, dyn, synth
from tests import util

nx, ny = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "130x96").split("x"))
ndte = int(sys.argv[2]) if len(sys.argv) > 2 else 31
nsub = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ice = sys.argv[4] if len(sys.argv) > 4 else "polar"
os.environ["EVPK_TILE"] = "0"
case = synth.SynthCase(nx=nx, ny=ny, land="continents", ice=ice)
d = blocks.create_distrb_cart(nx, ny, nx, ny)
f = synth.make_block_fields(case, d)
xmin = synth.global_min_dx(case)
outs = []
for mode in ("single", "triple"):
    os.environ["EVPK_DOUBLE"] = "0" if mode == "single" else "1"
    os.environ["EVPK_TRIPLE"] = "0" if mode == "single" else "1"
    g = util.clone(f)
    s = dyn.EvpDynamics(d, g, ndte=ndte, xmin=xmin)
    s.init_evp(3600.0)
    s.ctx.upload(g); s.ctx.prep(); s.ctx.subcycle(nsub)
    st = s.ctx.stats()
    print(mode, "k3", st.kernel3_launches, "k2", st.kernel2_launches, "k1", st.kernel_launches, "R3", st.strip_rows3, "nstrips3", st.nstrips3)
    s.ctx.finish(); s.ctx.download(g); s.close()
    outs.append(g)
for name in ("uvel", "stressp_1"):
    a, b = outs[0][name][0], outs[1][name][0]
    bad = np.argwhere(a != b)
    print(name, "mismatches", len(bad), "of", int((a != 0).sum()), "nonzero")
    if len(bad):
        js, is_ = bad[:, 0], bad[:, 1]
        print("  rows (j index in block array) histogram:", dict(zip(*np.unique(js, return_counts=True))))
        print("  cols histogram:", dict(zip(*np.unique(is_, return_counts=True))))
        for j, i in bad[:10]:
            print("   ", j, i, a[j, i], b[j, i])
