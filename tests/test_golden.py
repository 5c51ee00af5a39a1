"""Committed vectors of tests/golden/ (made by tests/golden/make_golden.py from the CPU restatement -- see its header
for what they do and do not pin).  CPU: the oracle still reproduces them bit for bit.  GPU: the HIP path reproduces
them through the C ABI without the oracle in the loop."""
import os

import numpy as np
import pytest

from cice5_amd import dyn, synth
from tests import util
from tests.golden import make_golden as mg


def _mismatch(cells, gold):
    bad = []
    for n in gold.files:
        a, b = cells[n], gold[n]
        assert a.shape == b.shape, n
        neq = ~((a == b) | (np.isnan(a) & np.isnan(b))) if a.dtype.kind == "f" else (a != b)
        if neq.any():
            bad.append((n, int(neq.sum())))
    return bad


@pytest.mark.parametrize("name", list(mg.CASES))
def test_oracle_reproduces_golden(name):
    gold = np.load(os.path.join(mg.HERE, name + ".npz"))
    d, f = mg.run_case(name)
    assert set(gold.files) == set(util.ALL_CELLS + util.NE_CELLS + util.PHYS_CELLS)
    assert not _mismatch(mg.defined_cells(d, f), gold)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(mg.CASES))
def test_hip_path_reproduces_golden(name):
    gold = np.load(os.path.join(mg.HERE, name + ".npz"))
    nx, ny, bsx, bsy, ns, ndte, dt, revised, ncalls, kw = mg.CASES[name]
    case, d, f = util.make_case(nx, ny, bsx, bsy, ns=ns, **kw)
    s = dyn.EvpDynamics(d, f, ndte=ndte, revised_evp=revised, xmin=synth.global_min_dx(case))
    s.init_evp(dt)
    for call in range(ncalls):
        if call:
            f["aice"] *= 0.97
            f["vice"] *= 0.97
            f["strairxT"], f["strairyT"] = f["strairyT"].copy(), -f["strairxT"]
        s.evp(dt)
    s.close()
    assert not _mismatch(mg.defined_cells(d, f), gold)
