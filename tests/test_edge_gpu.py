"""GPU edge cases through the C ABI: no ice, tiny / narrow grids, eliminated land blocks, odd shapes."""
import numpy as np
import pytest

from cice5_amd import blocks, constants as C, dyn, evpk, synth
from oracle import orc
from tests import util

pytestmark = pytest.mark.gpu


def _check(case, d, f, ndte=20, ncalls=1, dt=3600.0):
    xmin = synth.global_min_dx(case)
    fo, fg = util.clone(f), util.clone(f)
    p = orc.make_params(dt, ndte, xmin)
    s = dyn.EvpDynamics(d, fg, ndte=ndte, xmin=xmin)
    s.init_evp(dt)
    for _ in range(ncalls):
        nt, nu, _ = orc.evp(d, p, fo)
        s.evp(dt)
        st = s.ctx.stats()
        assert (st.icellt, st.icellu) == (nt, nu)
        bad = util.compare(d, fg, fo)
        assert not bad, bad[:5]
    s.close()
    return fo, (nt, nu)


def test_no_ice_at_all():
    case, d, f = util.make_case(100, 116, 50, 58, land="continents")
    for n in ("aice", "vice", "vsno", "aice_init", "strength", "strairxT", "strairyT"):
        f[n][...] = 0.0
    fo, (nt, nu) = _check(case, d, f)
    assert (nt, nu) == (0, 0) and not fo["uvel"].any()


def test_ice_melts_away_between_calls():
    case, d, f = util.make_case(100, 116, 50, 58, land="continents")
    xmin = synth.global_min_dx(case)
    fo, fg = util.clone(f), util.clone(f)
    p = orc.make_params(3600.0, 20, xmin)
    s = dyn.EvpDynamics(d, fg, ndte=20, xmin=xmin)
    s.init_evp(3600.0)
    for call in range(3):
        if call == 1:
            for ff in (fo, fg):
                for n in ("aice", "vice", "vsno", "aice_init"):
                    ff[n][...] = 0.0
        if call == 2:
            for ff in (fo, fg):
                for n in ("aice", "vice", "vsno", "aice_init"):
                    ff[n][...] = f[n]
        nt, nu, _ = orc.evp(d, p, fo)
        s.evp(3600.0)
        assert (s.ctx.stats().icellu == nu) and (call != 1 or nu == 0)
        assert not util.compare(d, fg, fo), call
    s.close()


@pytest.mark.parametrize("nx,ny,bsx,bsy", [(8, 8, 8, 8), (20, 12, 10, 6), (61, 9, 61, 9), (62, 10, 31, 5), (63, 17, 63, 17),
                                           (64, 16, 16, 16), (122, 33, 61, 11), (125, 40, 25, 8), (127, 7, 127, 7)])
def test_small_and_odd_grids(nx, ny, bsx, bsy):
    """grids narrower than one strip, widths around the 61/63-column strip sizes, very few rows"""
    case, d, f = util.make_case(nx, ny, bsx, bsy, ice="full")
    _check(case, d, f, ndte=13, ncalls=2)


@pytest.mark.parametrize("nx,ny,bsx,bsy", [(16, 8, 8, 4), (62, 9, 31, 9), (126, 12, 63, 6)])
def test_small_tripole_grids(nx, ny, bsx, bsy):
    case, d, f = util.make_case(nx, ny, bsx, bsy, ice="full", ns="tripole")
    _check(case, d, f, ndte=12, ncalls=2)


def test_eliminated_land_blocks():
    """land-block elimination (ice_domain.F90:387-441): blocks without ocean are not handed to the library;
    their cells read as land.  The oracle runs on the same reduced block list."""
    nx, ny, bsx, bsy = 120, 96, 6, 4
    case = synth.SynthCase(nx=nx, ny=ny, land="continents")
    full = blocks.create_distrb_cart(nx, ny, bsx, bsy)
    ff = synth.make_block_fields(case, full)
    work = [int(ff["tmask"][n, b.jlo - 1:b.jhi, b.ilo - 1:b.ihi].any()) for n, b in enumerate(full.local_blocks)]
    # make sure the corner blocks stay so that the bounding rectangle is still the whole grid
    if sum(work) == len(work):
        pytest.skip("no all-land block in this mask")
    for k in (0, len(work) - 1):
        work[k] = 1
    d = blocks.create_distrb_cart(nx, ny, bsx, bsy, work_per_block=work)
    assert d.nblocks < full.nblocks
    f = synth.make_block_fields(case, d)
    fo, (nt, nu) = _check(case, d, f, ndte=24, ncalls=2)
    # same physical answer as with all blocks present
    xmin = synth.global_min_dx(case)
    orc.evp(full, orc.make_params(3600.0, 24, xmin), ff)
    orc.evp(full, orc.make_params(3600.0, 24, xmin), ff)
    for name in ("uvel", "vvel", "stressp_1", "divu"):
        G = blocks.gather_global(full, ff[name])
        L = blocks.gather_global(d, fo[name])
        cov = blocks.gather_global(d, np.ones_like(fo[name])) > 0
        assert np.array_equal(G[cov], L[cov]), name


def test_closed_boundaries():
    case = synth.SynthCase(nx=70, ny=40, ew_boundary=C.BND_CLOSED, ns_boundary=C.BND_CLOSED, ice="full")
    d = blocks.create_distrb_cart(70, 40, 35, 20, ew_boundary_type="closed", ns_boundary_type="closed")
    f = synth.make_block_fields(case, d)
    _check(case, d, f, ndte=15, ncalls=2)


def test_tripole_top_block_of_one_row_is_rejected():
    """With a one-row top block the reference's halo leaves the ghost copy of the (fold-symmetrised) top row stale in the
    block below, so its own results depend on the decomposition (the oracle reproduces that: 1 block != 20x11 blocks on a
    122x23 grid).  The library refuses the layout instead of silently answering for a different decomposition."""
    case, d, f = util.make_case(122, 23, 20, 11, ns="tripole", land="rows", ice="full")
    with pytest.raises(evpk.EvpkError, match="at least two physical rows"):
        dyn.EvpDynamics(d, f, ndte=2)
    case, d, f = util.make_case(122, 23, 20, 12, ns="tripole", land="rows", ice="full")
    dyn.EvpDynamics(d, f, ndte=2).close()
