"""The C oracle against a second, independently written restatement of evp(dt) (tests/npref.py: whole-array numpy, another
reading of the Fortran) -- bit for bit, on one-block domains.  CPU only.  What it buys: oracle/evp_oracle.c is no longer the
only reading of ice_dyn_evp.F90 / ice_dyn_shared.F90 / ice_grid.F90 the HIP kernels are compared with."""
import numpy as np
import pytest

from cice5_amd import synth
from oracle import orc
from tests import npref, util


def _run_both(nx, ny, ndte, revised=False, cosw=1.0, sinw=0.0, ncalls=1, nsub=None, **kw):
    case, d, f = util.make_case(nx, ny, nx, ny, **kw)
    xmin = synth.global_min_dx(case)
    fo, fn = util.clone(f), util.clone(f)
    p = orc.make_params(3600.0, ndte, xmin, revised_evp=revised, cosw=cosw, sinw=sinw)
    for call in range(ncalls):
        if call:
            for ff in (fo, fn):
                ff["aice"] *= 0.9
                ff["vice"] *= 0.97
                ff["strairxT"], ff["strairyT"] = ff["strairyT"].copy(), -ff["strairxT"]
        nt, nu, _ = orc.evp(d, p, fo, nsub=nsub or 0)
        g = {k: v[0] for k, v in fn.items() if isinstance(v, np.ndarray) and v.ndim == 3}      # the one block, views into fn
        mt, mu = npref.evp(g, 3600.0, ndte, xmin, revised_evp=revised, cosw=cosw, sinw=sinw, nsub=nsub)
        assert (nt, nu) == (mt, mu)
        bad = util.compare(d, fn, fo)
        assert not bad, (call, bad[:6])
    assert nu > 0 and np.abs(fo["uvel"]).max() > 1e-3
    # the parameters of set_evp_parameters too (ice_dyn_shared.F90:185-259)
    P = npref.set_evp_parameters(3600.0, ndte, revised, xmin)
    for n in ("ecci", "revp", "arlx1i", "brlx", "denom1"):
        assert P[n] == getattr(p, n), n


def test_numpy_restatement_equals_c_oracle_classic():
    _run_both(48, 40, 30, land="continents", ncalls=3)


def test_numpy_restatement_equals_c_oracle_gx3_shape():
    _run_both(100, 116, 120, land="continents")


def test_numpy_restatement_equals_c_oracle_revised_evp_and_turning_angle():
    _run_both(40, 36, 24, revised=True, land="continents", ncalls=2)
    _run_both(37, 29, 17, cosw=np.cos(0.4), sinw=np.sin(0.4), ice="full", ncalls=2)


def test_numpy_restatement_equals_c_oracle_partial_loop():
    _run_both(48, 40, 30, nsub=7, land="rows")


def test_numpy_restatement_equals_c_oracle_on_a_tripole_domain():
    """the headline boundary: the physics read a second time, the halo update and the stress fold taken from the routines the
    reference's own output pins (tests/golden halo fixtures) -- so the SEQUENCE of updates, their field locations / types and
    the stress fold's array pairing are checked against the C oracle's too"""
    from cice5_amd import constants as C
    nx, ny, ndte = 48, 40, 24
    case, d, f = util.make_case(nx, ny, nx, ny, ns="tripole", land="continents")
    xmin = synth.global_min_dx(case)
    fo, fn = util.clone(f), util.clone(f)
    p = orc.make_params(3600.0, ndte, xmin)
    calls = []

    def halo_update(a, loc, kind):
        t = np.ascontiguousarray(a[None])
        orc.halo_r8(d, t, loc, kind, 0.0)
        a[...] = t[0]
        calls.append((loc, kind))

    def stress_fold(a1, a2):
        t1, t2 = np.ascontiguousarray(a1[None]), np.ascontiguousarray(a2[None])
        orc.halo_stress(d, t1, t2)
        a1[...] = t1[0]
        calls.append("fold")

    assert (npref.LOC_CENTER, npref.LOC_NECORNER, npref.KIND_SCALAR, npref.KIND_VECTOR) == \
        (C.LOC_CENTER, C.LOC_NECORNER, C.KIND_SCALAR, C.KIND_VECTOR)
    for call in range(2):
        if call:
            for ff in (fo, fn):
                ff["aice"] *= 0.9
                ff["vice"] *= 0.97
        nt, nu, _ = orc.evp(d, p, fo)
        g = {k: v[0] for k, v in fn.items() if isinstance(v, np.ndarray) and v.ndim == 3}
        mt, mu = npref.evp(g, 3600.0, ndte, xmin, halo_update=halo_update, stress_fold=stress_fold)
        assert (nt, nu) == (mt, mu)
        bad = util.compare(d, fn, fo)
        assert not bad, (call, bad[:6])
    assert calls.count("fold") == 24 and np.abs(fo["uvel"][0, -3:]).max() > 1e-3          # ice at the fold


def test_numpy_upwind_equals_c_oracle():
    """row f-3, first step: the edge velocities and upwind_field of transport_upwind, after a real evp"""
    from cice5_amd import constants as C
    nx, ny = 48, 40
    case, d, f = util.make_case(nx, ny, nx, ny, land="continents")
    xmin = synth.global_min_dx(case)
    orc.evp(d, orc.make_params(3600.0, 20, xmin), f)
    synth.add_thickness_distribution(f)
    planes = [f["aice0"]] + [a for n in range(f["aicen"].shape[1]) for a in (f["aicen"][:, n], f["vicen"][:, n])]
    works = np.ascontiguousarray(np.stack(planes, axis=1))
    for k in range(works.shape[1]):
        w = np.ascontiguousarray(works[:, k]); orc.halo_r8(d, w, C.LOC_CENTER, C.KIND_SCALAR, 0.0); works[:, k] = w
    wo, wn = works.copy(), works.copy()
    orc.transport_upwind(d, 3600.0, f, wo)
    npref.transport_upwind({k: v[0] for k, v in f.items() if isinstance(v, np.ndarray) and v.ndim == 3}, wn[0], 3600.0)
    assert np.abs(wo - works).max() > 1e-6
    assert np.array_equal(wn, wo)


def test_numpy_eap_equals_libm_build_of_the_c_oracle():
    """row f-4: eap(dt) -- stress_eap, update_stress_rdg, stepa, calc_ffrac and the driver around them -- read a second time
    (tests/npref.py) against the C oracle built on the host's libm (the build whose sin / cos / atan2 numpy's math.* shares)"""
    from cice5_amd.eap_tables import eap_tables
    T = eap_tables()
    for (nx, ny, ndte, ncalls) in ((40, 36, 22, 2), (48, 40, 31, 1)):
        case, d, f = util.make_case(nx, ny, nx, ny, land="continents")
        synth.add_eap_state(f)
        xmin = synth.global_min_dx(case)
        fo, fn = util.clone(f), util.clone(f)
        p = orc.make_params(3600.0, ndte, xmin)
        for call in range(ncalls):
            if call:
                for ff in (fo, fn):
                    ff["aice"] *= 0.9
                    ff["vice"] *= 0.97
            orc.eap(d, p, fo, T, libm=True)
            g = {k: v[0] for k, v in fn.items() if isinstance(v, np.ndarray) and v.ndim == 3}
            npref.evp(g, 3600.0, ndte, xmin, eap_tables=T)
            bad = util.compare(d, fn, fo)
            assert not bad, (call, bad[:6])
            ne = util.cell_mask(d, "ne")
            for n in synth.EAP_STATE + synth.EAP_HISTORY:
                assert np.array_equal(fn[n][ne], fo[n][ne]), (call, n, int((fn[n][ne] != fo[n][ne]).sum()))
        assert np.abs(fo["uvel"]).max() > 1e-3 and np.abs(fo["a11_1"] - 0.5).max() > 1e-6
