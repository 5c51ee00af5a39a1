#!/usr/bin/env python3
"""Regenerates tests/golden/evp_*.npz:  python tests/golden/make_golden.py

What these vectors are -- and are not.  The reference cannot be built in this container (ice_grid.F90 needs netCDF,
which the image lacks), so there are NO reference-generated vectors.  The files hold the outputs of THIS
repository's CPU restatement (oracle/, gcc -O2 -ffp-contract=off) on deterministic synthetic inputs
(cice5_amd/synth.py: hash-based fields, reproducible bit for bit from the parameters stored in each file).  They pin
the restatement against drifting from round to round and give the GPU tests a committed expected output; they say
nothing about the reference beyond what the oracle's own checks do (oracle/evp_oracle.h: PARITY UNPINNED).

Per case: the SynthCase / decomposition parameters, the EVP parameters, and after `ncalls` calls of evp(dt) the cells
the reference defines of every output array (tests/util.py: ALL_CELLS / NE_CELLS / PHYS_CELLS), packed in global order.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cice5_amd import blocks, constants as C, synth          # noqa: E402
from oracle import orc                                       # noqa: E402
from tests import util                                       # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (nx, ny, bsx, bsy, ns, ndte, dt, revised_evp, ncalls, synth kwargs)
    "evp_open_1block": (24, 20, 24, 20, "open", 12, 3600.0, False, 1, dict(land="continents")),
    "evp_open_12blocks_2calls": (48, 40, 12, 10, "open", 20, 3600.0, False, 2, dict(land="continents")),
    "evp_tripole_8blocks": (48, 32, 12, 16, "tripole", 16, 1800.0, False, 1, dict(land="continents")),
    "evp_revised_full_ice": (40, 24, 20, 12, "open", 10, 3600.0, True, 1, dict(ice="full")),
}


def run_case(name):
    nx, ny, bsx, bsy, ns, ndte, dt, revised, ncalls, kw = CASES[name]
    case, d, f = util.make_case(nx, ny, bsx, bsy, ns=ns, **kw)
    xmin = synth.global_min_dx(case)
    p = orc.make_params(dt, ndte, xmin, revised_evp=revised)
    for call in range(ncalls):
        if call:
            f["aice"] *= 0.97
            f["vice"] *= 0.97
            f["strairxT"], f["strairyT"] = f["strairyT"].copy(), -f["strairxT"]
        orc.evp(d, p, f)
    return d, f


def defined_cells(d, f):
    """{name: 1-D array} of the cells the reference defines, local blocks in order."""
    out = {}
    for grp, kind in ((util.ALL_CELLS, "all"), (util.NE_CELLS, "ne"), (util.PHYS_CELLS, "phys")):
        m = util.cell_mask(d, kind)
        for n in grp:
            out[n] = np.ascontiguousarray(f[n][m])
    return out


def main():
    for name in CASES:
        d, f = run_case(name)
        cells = defined_cells(d, f)
        assert np.abs(cells["uvel"]).max() > 1e-4, name
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **cells)
        print(name, {k: v.shape for k, v in list(cells.items())[:2]}, "max|u| = %.6e" % np.abs(cells["uvel"]).max())


if __name__ == "__main__":
    main()
