"""Ad-hoc GPU check (not a pytest file): python scripts/gpu_debug.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
from oracle import orc
from cice5_amd import dyn, synth

def run(nx, ny, bsx, bsy, ndte=120, nsub=None, **kw):
    case, d, f = util.make_case(nx, ny, bsx, bsy, **kw)
    xmin = synth.global_min_dx(case)
    fo, fg = util.clone(f), util.clone(f)
    p = orc.make_params(3600.0, ndte, xmin)
    t = time.time(); nt, nu, secs = orc.evp(d, p, fo); t_or = time.time() - t
    solver = dyn.EvpDynamics(d, fg, ndte=ndte, xmin=xmin)
    solver.init_evp(3600.0)
    t = time.time(); solver.evp(3600.0); t_g = time.time() - t
    st = solver.ctx.stats()
    bad = util.compare(d, fg, fo)
    print(f"{nx}x{ny} blocks {bsx}x{bsy} kw={kw}: oracle icellt={nt} icellu={nu} ({t_or:.2f}s) gpu icellt={st.icellt} icellu={st.icellu} "
          f"strips {st.nstrips}/{st.nstrips_total} loop {st.loop_ms:.2f} ms total {t_g:.2f}s max|u|={np.abs(fo['uvel']).max():.4f}")
    print("   MISMATCH:" if bad else "   bit-exact", bad[:8])
    solver.close()
    return bad

if __name__ == "__main__":
    run(100, 116, 100, 116, land="continents")
    run(100, 116, 25, 29, land="continents")
    run(320, 384, 320, 384)
    run(360, 300, 15, 300, land="continents")
