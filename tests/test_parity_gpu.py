"""GPU parity tests: the HIP path (libevpk through its C ABI) against the CPU oracle on the
same bytes.  fp64, bit-exact: both sides evaluate the reference's operation order without
FMA contraction, and gfx950's fp64 sqrt/divide are correctly rounded, so equality is demanded
(signed zeros compare equal).  The oracle itself is PARITY UNPINNED (oracle/evp_oracle.h)."""
import os

import numpy as np
import pytest

from cice5_amd import blocks, constants as C, dyn, evpk, synth
from oracle import orc
from tests import util

pytestmark = pytest.mark.gpu

# k_subcycle2 (EVPK_PREFETCH=0) and k_subcycle3w (EVPK_TRIPLE=1) were measured and not adopted: the product library does not contain them
# (csrc/evpk_experimental.hip).  Their tests run against the other build: EVPK_LIB=cice5_amd/libevpk_exp.so python -m pytest tests -m gpu -k three_subcycle
needs_experimental = pytest.mark.skipif(not evpk.experimental(), reason="kernels of evpk_experimental.hip: EVPK_LIB=cice5_amd/libevpk_exp.so")


def _both(nx, ny, bsx, bsy, *, ndte=120, dt=3600.0, ncalls=1, revised_evp=False, cosw=1.0, sinw=0.0,
          tilt_from_slope=False, wind_on_ugrid=False, ns="open", nsub=None, pin_host=False, **kw):
    case, d, f = util.make_case(nx, ny, bsx, bsy, ns=ns, **kw)
    xmin = synth.global_min_dx(case)
    fo, fg = util.clone(f), util.clone(f)
    p = orc.make_params(dt, ndte, xmin, revised_evp=revised_evp, cosw=cosw, sinw=sinw,
                        tilt_from_slope=tilt_from_slope, wind_on_ugrid=wind_on_ugrid)
    solver = dyn.EvpDynamics(d, fg, ndte=ndte, revised_evp=revised_evp, xmin=xmin, cosw=cosw, sinw=sinw,
                             tilt_from_slope=tilt_from_slope, wind_on_ugrid=wind_on_ugrid, pin_host=pin_host)
    assert not pin_host or len(solver._pinned) > 40
    solver.init_evp(dt)
    # host mirror of set_evp_parameters must agree with the oracle's to the bit
    for n in ("revp", "ecci", "denom1", "arlx1i", "brlx"):
        assert getattr(solver.params, n) == getattr(p, n), n
    for call in range(ncalls):
        if call:   # next step: thermodynamics changed the ice a little, the wind turned
            for ff in (fo, fg):
                ff["aice"] *= 0.97
                ff["vice"] *= 0.97
                ff["strairxT"], ff["strairyT"] = ff["strairyT"].copy(), -ff["strairxT"]
        nt, nu, _ = orc.evp(d, p, fo)
        solver.evp(dt)
        st = solver.ctx.stats()
        assert (st.icellt, st.icellu) == (nt, nu)
        bad = util.compare(d, fg, fo)
        assert not bad, f"call {call}: {bad[:6]}"
        if not wind_on_ugrid:      # t2ugrid_vector leaves zeros in every ghost cell (to_ugrid: work2(:,:,:) = c0, ice_grid.F90:1852)
            assert np.array_equal(fg["strairx"], fo["strairx"]) and np.array_equal(fg["strairy"], fo["strairy"])
    assert nu > 0 and np.abs(fo["uvel"]).max() > 1e-3
    solver.close()
    return fo, fg


def test_cfg1_gx3_one_block():
    """BASELINE config 1 shape: gx3 100x116, one 102x118 block, ndte=120."""
    _both(100, 116, 100, 116, land="continents")


def test_cfg1_gx3_sixteen_blocks():
    """same grid cut into 16 blocks of 25x29 (the reference's serial multi-block mode)."""
    _both(100, 116, 25, 29, land="continents")


def test_cfg2_gx1_one_block():
    """BASELINE config 2: gx1 320x384, whole grid in one device block, ndte=120."""
    _both(320, 384, 320, 384)


def test_cfg2_gx1_full_ice():
    _both(320, 384, 320, 384, ice="full", ndte=40)


def test_cfg3_auscom_360x300_slender_blocks():
    """BASELINE config 3: 360x300 in 24 blocks of 15x300 (bld/config.nci.auscom.360x300), land mask."""
    _both(360, 300, 15, 300, land="continents")


def test_cfg3_auscom_360x300_six_blocks():
    """config.ubuntu.auscom.360x300: 6 blocks of 60x300."""
    _both(360, 300, 60, 300, land="continents", ndte=60)


def test_cfg3_auscom_360x300_tripole():
    """the 1-degree ACCESS-OM2 grid is tripolar in production: the same 24 blocks with ns_boundary_type = 'tripole', whole evp,
    ndte = 120 (one-subcycle tile launches + the in-place fold of a small one-rank slab)"""
    _both(360, 300, 15, 300, ns="tripole", land="continents")


def test_padded_blocks():
    """block size that does not divide the grid (ice_blocks.F90:148-150 padding)."""
    _both(100, 116, 32, 40, land="continents", ndte=30)


def test_two_calls_warm_start():
    """second evp() call starts from non-zero u, sigma and iceumask (new-ice / lost-ice branches of evp_prep2)."""
    _both(100, 116, 25, 29, land="continents", ndte=40, ncalls=3)


def test_revised_evp():
    _both(100, 116, 50, 58, land="continents", revised_evp=True, ndte=60, ncalls=2)


def test_turning_angle_and_slope_tilt():
    """AusCOM namelist knobs: cosw/sinw (ice_dyn_shared.F90:66-72) and use_ocnslope (:604-608)."""
    th = np.deg2rad(25.0)
    _both(100, 116, 50, 58, land="continents", cosw=float(np.cos(th)), sinw=float(np.sin(th)),
          tilt_from_slope=True, ndte=40)


def test_wind_on_ugrid():
    """ACCESS: strairx/y := strax/stray, no t2ugrid_vector (ice_dyn_evp.F90:226-228)."""
    _both(100, 116, 50, 58, land="continents", wind_on_ugrid=True, ndte=40)


def test_open_ew_boundary_not_cyclic():
    case = synth.SynthCase(nx=64, ny=48, ew_boundary=C.BND_OPEN)
    d = blocks.create_distrb_cart(64, 48, 16, 16, ew_boundary_type="open")
    f = synth.make_block_fields(case, d)
    xmin = synth.global_min_dx(case)
    fo, fg = util.clone(f), util.clone(f)
    p = orc.make_params(3600.0, 30, xmin)
    orc.evp(d, p, fo)
    s = dyn.EvpDynamics(d, fg, ndte=30, xmin=xmin)
    s.init_evp(3600.0)
    s.evp(3600.0)
    assert not util.compare(d, fg, fo)
    s.close()


@pytest.mark.parametrize("bs", [(48, 40), (12, 10), (24, 20)])
def test_tripole_fold(bs):
    """ns_boundary_type = 'tripole' (ACCESS-OM2): per-subcycle velocity fold with top-row symmetrisation
    and the post-loop ice_HaloUpdate_stress pairs (ice_dyn_evp.F90:416-481)."""
    _both(48, 40, *bs, ns="tripole", land="continents", ndte=60, ncalls=2)


def test_strip_rows_do_not_matter():
    """the wave-strip height is a pure scheduling choice: results are bit-identical for any R."""
    case, d, f = util.make_case(320, 384, 320, 384)
    xmin = synth.global_min_dx(case)
    out = []
    for R in ("2", "7", "32"):
        os.environ["EVPK_STRIP_ROWS"] = R
        g = util.clone(f)
        s = dyn.EvpDynamics(d, g, ndte=24, xmin=xmin)
        s.init_evp(3600.0)
        s.evp(3600.0)
        s.close()
        out.append(g)
    del os.environ["EVPK_STRIP_ROWS"]
    for g in out[1:]:
        assert not util.compare(d, g, out[0])


@pytest.mark.parametrize("sw", [dict(kstrength=1, krdg_partic=1, krdg_redist=1), dict(kstrength=1, krdg_partic=0, krdg_redist=0),
                                dict(kstrength=1, krdg_partic=0, krdg_redist=1), dict(kstrength=1, krdg_partic=1, krdg_redist=0),
                                dict(kstrength=0)])
def test_ice_strength_on_the_device(sw):
    """evpk_step_in.strength == NULL: ice_strength (ice_mechred.F90:2111-2269, Rothrock with both participation and both
    redistribution functions, or Hibler) runs on the device where the reference calls it; bit-identical to the oracle doing
    the same (both use the same fixed exp algorithm), strength included, on all cells (it is halo-updated, :311)."""
    for (nx, ny, bsx, bsy, ns) in ((100, 116, 25, 29, "open"), (96, 64, 96, 64, "tripole")):
        case, d, f = util.make_case(nx, ny, bsx, bsy, ns=ns, land="continents")
        synth.add_thickness_distribution(f)
        f["strength"][...] = -7.0                      # not an input any more
        xmin = synth.global_min_dx(case)
        fo, fg = util.clone(f), util.clone(f)
        p = orc.make_params(3600.0, 24, xmin, strength_mode=1, **sw)
        solver = dyn.EvpDynamics(d, fg, ndte=24, xmin=xmin, device_strength=dict(sw))
        solver.init_evp(3600.0)
        for call in range(2):
            if call:
                for ff in (fo, fg):
                    ff["aice"] *= 0.9
                    ff["vice"] *= 0.9
                    ff["aicen"] *= 0.9
                    ff["vicen"] *= 0.9
                    ff["aice0"][...] = np.maximum(1.0 - ff["aice"], 0.0)
            orc.evp(d, p, fo)
            solver.evp(3600.0)
            bad = util.compare(d, fg, fo)
            m = util.cell_mask(d, "all")
            assert np.array_equal(fg["strength"][m], fo["strength"][m]), (sw, ns, call)
            assert not bad, (sw, ns, call, bad[:6])
        assert fo["strength"].max() > 1e3 and np.abs(fo["uvel"]).max() > 1e-3
        solver.close()


@pytest.mark.parametrize("ns", ["open", "tripole"])
def test_page_locked_host_arrays(ns):
    """evpk_pin_host: the gather / scatter kernels read and write the caller's arrays in place over PCIe; a download
    must leave every cell it does not deliver untouched (the oracle comparison covers ghost cells too)."""
    _both(100, 116, 25, 29, ns=ns, land="continents", ndte=30, ncalls=2, pin_host=True)
    _both(130, 60, 130, 60, ns=ns, ice="full", ndte=12, pin_host=True)


@pytest.mark.parametrize("ns,m", [("open", 4), ("open", 1), ("tripole", 1)])
def test_self_exchange_through_rccl(ns, m, monkeypatch):
    """EVPK_FORCE_EXCHANGE=2: the single rank's self-exchange goes through a ONE-rank RCCL communicator created by the
    library -- ncclSend / ncclRecv to itself in a group (ghost zones, one-column halos, compacted row lists) and
    ncclAllGather (tripole fold) on the production buffers, counts and stream.  RCCL refuses two ranks on one GPU, so this
    is how far the RCCL call path can be exercised on a one-GPU box."""
    monkeypatch.setenv("EVPK_FORCE_EXCHANGE", "2")
    monkeypatch.setenv("EVPK_ZONE_M", str(m))
    _both(200, 96, 50, 48, ns=ns, land="continents", ndte=31, ncalls=2)
    _both(130, 60, 130, 60, ns=ns, ice="full", ndte=18, revised_evp=True)


@pytest.mark.parametrize("m", [1, 2, 4])
def test_forced_exchange_zone_depth(m, monkeypatch):
    """Self-exchange on a cyclic ring of one rank with ghost zones of 2*m columns (EVPK_ZONE_M)."""
    monkeypatch.setenv("EVPK_FORCE_EXCHANGE", "1")
    monkeypatch.setenv("EVPK_ZONE_M", str(m))
    _both(200, 96, 50, 48, land="continents", ndte=31, ncalls=2)
    _both(130, 60, 130, 60, ice="full", ndte=18, revised_evp=True)


@pytest.mark.parametrize("ns", ["open", "tripole"])
def test_forced_exchange_path(ns, monkeypatch):
    """EVPK_FORCE_EXCHANGE=1 makes a single rank take the multi-rank code path (pack edge columns ->
    exchange -> unpack, fold through the all-gather re-pack); only the RCCL calls themselves are
    replaced by device copies.  Must still match the oracle bit for bit."""
    monkeypatch.setenv("EVPK_FORCE_EXCHANGE", "1")
    _both(48, 40, 12, 10, ns=ns, land="continents", ndte=40, ncalls=2)
    _both(100, 116, 25, 29, ns=ns, land="continents", ndte=30)
    # the two-subcycle kernel with two-column ghost zones filled by the exchange, several strips wide
    _both(200, 96, 50, 48, ns=ns, ice="full", ndte=31, ncalls=2)
    _both(200, 96, 50, 48, ns=ns, land="continents", ndte=24, revised_evp=True)


def test_two_subcycle_kernel_equals_single(monkeypatch):
    """k_subcycle2 (two subcycles per launch) vs k_subcycle only: bit-identical, including odd ndte,
    subcycles issued in odd pieces, revised EVP, and a non-cyclic E-W boundary.  (EVPK_TILE=0: the marching kernels; on a
    one-rank tripole grid this small the tuner would otherwise drop the pairs for one-row-per-wave single launches.)"""
    monkeypatch.setenv("EVPK_TILE", "0")
    for kw, ndte, pieces in [(dict(land="continents"), 31, [31]), (dict(land="continents"), 40, [7, 12, 21]),
                             (dict(ice="full"), 24, [24]), (dict(land="continents", ns="tripole"), 33, [33]),
                             (dict(ice="full", ns="tripole"), 26, [9, 17])]:
        case, d, f = util.make_case(130, 96, 130, 96, **kw)
        xmin = synth.global_min_dx(case)
        outs = []
        for dbl in ("0", "1"):
            monkeypatch.setenv("EVPK_DOUBLE", dbl)
            g = util.clone(f)
            s = dyn.EvpDynamics(d, g, ndte=ndte, xmin=xmin, revised_evp=(ndte == 24))
            s.init_evp(3600.0)
            s.ctx.upload(g); s.ctx.prep()
            for n in pieces:
                s.ctx.subcycle(n)
            st = s.ctx.stats()
            assert (st.kernel2_launches > 0) == (dbl == "1")
            s.ctx.finish(); s.ctx.download(g)
            s.close()
            outs.append(g)
        assert not util.compare(d, outs[1], outs[0])


@needs_experimental
@pytest.mark.parametrize("R3", ["", "3", "7", "40"])
def test_three_subcycle_kernel_equals_single(R3, monkeypatch):
    """k_subcycle3w (three subcycles per launch, one wave per subcycle stage, sigma and (u, v) handed from stage to stage through
    the LDS) vs k_subcycle only: bit-identical -- ndte with every remainder mod 3, subcycles issued in odd pieces, revised EVP, a
    non-cyclic E-W boundary, grids narrower / wider than one 58-column strip, strip heights 3, 7, 24 (default) and 40."""
    monkeypatch.setenv("EVPK_TILE", "0")
    if R3:
        monkeypatch.setenv("EVPK_STRIP_ROWS3", R3)
    for (nx, ny), kw, ndte, pieces in [((130, 96), dict(land="continents"), 31, [31]), ((130, 96), dict(land="continents"), 41, [7, 12, 22]),
                                       ((130, 96), dict(ice="full"), 24, [24]), ((200, 75), dict(land="continents"), 36, [36]),
                                       ((57, 40), dict(ice="full"), 17, [4, 13]), ((117, 64), dict(land="continents", ew="open"), 30, [30])]:
        ew = kw.pop("ew", "cyclic")
        case = synth.SynthCase(nx=nx, ny=ny, ew_boundary=C.BND_NAMES[ew], **kw)
        d = blocks.create_distrb_cart(nx, ny, nx, ny, ew_boundary_type=ew)
        f = synth.make_block_fields(case, d)
        xmin = synth.global_min_dx(case)
        outs = []
        for mode in ("single", "triple"):
            monkeypatch.setenv("EVPK_DOUBLE", "0" if mode == "single" else "1")
            monkeypatch.setenv("EVPK_TRIPLE", "0" if mode == "single" else "1")
            g = util.clone(f)
            s = dyn.EvpDynamics(d, g, ndte=ndte, xmin=xmin, revised_evp=(ndte == 24))
            s.init_evp(3600.0)
            s.ctx.upload(g); s.ctx.prep()
            for n in pieces:
                s.ctx.subcycle(n)
            st = s.ctx.stats()
            assert (st.kernel3_launches > 0) == (mode == "triple"), (mode, st.kernel3_launches)
            s.ctx.finish(); s.ctx.download(g)
            s.close()
            outs.append(g)
        bad = util.compare(d, outs[1], outs[0])
        assert not bad, ((nx, ny), kw, ndte, bad[:4])
        assert np.abs(outs[0]["uvel"]).max() > 1e-4


@needs_experimental
def test_three_subcycle_kernel_equals_oracle(monkeypatch):
    """the same kernel against the oracle: BASELINE config 3's grid in 24 blocks and a 3-call warm start (new / lost ice)"""
    monkeypatch.setenv("EVPK_TILE", "0")
    monkeypatch.setenv("EVPK_TRIPLE", "1")
    seen = []
    real = dyn.EvpDynamics.close

    def close(self):
        seen.append(int(self.ctx.stats().kernel3_launches))
        real(self)

    monkeypatch.setattr(dyn.EvpDynamics, "close", close)
    _both(360, 300, 15, 300, land="continents", ndte=24, ncalls=2)
    _both(320, 384, 80, 96, ndte=31, ncalls=3)
    assert all(n > 0 for n in seen), seen


@pytest.mark.parametrize("H", ["", "2", "5", "13"])
def test_tile_kernel_equals_oracle(H, monkeypatch):
    """k_subcycle2t (EVPK_TILE=1: one row per wave, three workgroup barriers instead of the north march -- the small-slab
    variant of the two-subcycle kernel) against the oracle: BASELINE configs 2 and 3, padded blocks, revised EVP, an open
    E-W boundary, the tripole band launches beside it, ghost zones (forced exchange) at two depths, odd ndte and subcycles
    issued in pieces.  H = tile height (EVPK_STRIP_ROWS; '' = tuned)."""
    monkeypatch.setenv("EVPK_TILE", "1")
    if H:
        monkeypatch.setenv("EVPK_STRIP_ROWS", H)
    seen = []
    real = dyn.EvpDynamics.close

    def close(self):
        seen.append(int(self.ctx.stats().tile_kernel))
        real(self)

    monkeypatch.setattr(dyn.EvpDynamics, "close", close)
    _both(320, 384, 320, 384, ndte=30)                                         # config 2
    _both(360, 300, 15, 300, land="continents", ndte=24, ncalls=2)             # config 3
    _both(100, 116, 32, 40, land="continents", ndte=31)                        # padded blocks, odd ndte
    _both(130, 60, 130, 60, ice="full", ndte=18, revised_evp=True)
    _both(48, 40, 12, 10, ns="tripole", land="continents", ndte=40, ncalls=2)
    _both(260, 140, 65, 35, ns="tripole", ice="full", ndte=14)
    assert seen and all(seen)
    if H in ("", "5"):
        for m in ("1", "4"):
            monkeypatch.setenv("EVPK_FORCE_EXCHANGE", "1")
            monkeypatch.setenv("EVPK_ZONE_M", m)
            _both(200, 96, 50, 48, land="continents", ndte=31, ncalls=2)
            _both(200, 96, 50, 48, ns="tripole", ice="full", ndte=12)
        monkeypatch.delenv("EVPK_FORCE_EXCHANGE")
        monkeypatch.delenv("EVPK_ZONE_M")
        # open E-W boundary, subcycles in pieces
        case = synth.SynthCase(nx=64, ny=48, ew_boundary=C.BND_OPEN)
        d = blocks.create_distrb_cart(64, 48, 16, 16, ew_boundary_type="open")
        f = synth.make_block_fields(case, d)
        xmin = synth.global_min_dx(case)
        fo, fg = util.clone(f), util.clone(f)
        orc.evp(d, orc.make_params(3600.0, 30, xmin), fo)
        s = dyn.EvpDynamics(d, fg, ndte=30, xmin=xmin)
        s.init_evp(3600.0)
        s.ctx.upload(fg); s.ctx.prep(); s.ctx.subcycle(7); s.ctx.subcycle(12); s.ctx.subcycle(11); s.ctx.finish(); s.ctx.download(fg)
        assert s.ctx.stats().tile_kernel == 1
        assert not util.compare(d, fg, fo)
        s.close()


@pytest.mark.parametrize("R", ["", "5", "6", "11", "12", "23", "40"])
def test_rolling_tile_kernel_equals_oracle(R, monkeypatch):
    """k_subcycle2r (EVPK_TILE=2; round 5): the one-row-per-wave tile that rolls north through a strip of R rows in passes of six, the
    rows that cannot finish in a pass kept by their waves -- no redundant rows per tile.  Against the oracle on the cases of the tile
    kernel's test: R = one pass (5), one row into the second pass (6), whole passes (11, 23), ragged last passes (12, 40), tuned ('')."""
    monkeypatch.setenv("EVPK_TILE", "2")
    if R:
        monkeypatch.setenv("EVPK_STRIP_ROWS", R)
    seen = []
    real = dyn.EvpDynamics.close

    def close(self):
        seen.append(int(self.ctx.stats().tile_kernel))
        real(self)

    monkeypatch.setattr(dyn.EvpDynamics, "close", close)
    _both(320, 384, 320, 384, ndte=30)                                         # config 2
    _both(360, 300, 15, 300, land="continents", ndte=24, ncalls=2)             # config 3
    _both(100, 116, 32, 40, land="continents", ndte=31)                        # padded blocks, odd ndte
    _both(130, 60, 130, 60, ice="full", ndte=18, revised_evp=True)
    _both(48, 40, 12, 10, ns="tripole", land="continents", ndte=40, ncalls=2)
    _both(260, 140, 65, 35, ns="tripole", ice="full", ndte=14)
    assert seen and all(v == 2 for v in seen), seen
    if R in ("", "11", "40"):
        for m in ("1", "4"):
            monkeypatch.setenv("EVPK_FORCE_EXCHANGE", "1")
            monkeypatch.setenv("EVPK_ZONE_M", m)
            _both(200, 96, 50, 48, land="continents", ndte=31, ncalls=2)
            _both(200, 96, 50, 48, ns="tripole", ice="full", ndte=12)
        monkeypatch.delenv("EVPK_FORCE_EXCHANGE")
        monkeypatch.delenv("EVPK_ZONE_M")
        case = synth.SynthCase(nx=64, ny=48, ew_boundary=C.BND_OPEN)               # open E-W boundary, subcycles in pieces
        d = blocks.create_distrb_cart(64, 48, 16, 16, ew_boundary_type="open")
        f = synth.make_block_fields(case, d)
        xmin = synth.global_min_dx(case)
        fo, fg = util.clone(f), util.clone(f)
        orc.evp(d, orc.make_params(3600.0, 30, xmin), fo)
        s = dyn.EvpDynamics(d, fg, ndte=30, xmin=xmin)
        s.init_evp(3600.0)
        s.ctx.upload(fg); s.ctx.prep(); s.ctx.subcycle(7); s.ctx.subcycle(12); s.ctx.subcycle(11); s.ctx.finish(); s.ctx.download(fg)
        assert s.ctx.stats().tile_kernel == 2
        assert not util.compare(d, fg, fo)
        s.close()


def test_tile_kernel_is_chosen_for_small_slabs_only():
    """the tuner takes the tile variant where strips are scarce (gx1, 360x300) and the marching one on the headline grid"""
    for (nx, ny, bsx, bsy, want) in ((320, 384, 320, 384, 1), (360, 300, 60, 300, 1)):
        case, d, f = util.make_case(nx, ny, bsx, bsy, land="continents")
        s = dyn.EvpDynamics(d, f, ndte=4, xmin=synth.global_min_dx(case))
        s.init_evp(3600.0)
        s.evp(3600.0)
        assert (s.ctx.stats().tile_kernel > 0) == bool(want), (nx, ny)
        s.close()


def test_staged_api_equals_run():
    """upload/prep/subcycle/finish/download == evpk_run; subcycles may be issued in pieces."""
    case, d, f = util.make_case(100, 116, 25, 29, land="continents")
    xmin = synth.global_min_dx(case)
    a, b = util.clone(f), util.clone(f)
    s = dyn.EvpDynamics(d, a, ndte=50, xmin=xmin)
    s.init_evp(3600.0)
    s.evp(3600.0)
    s.close()
    s = dyn.EvpDynamics(d, b, ndte=50, xmin=xmin)
    s.init_evp(3600.0)
    s.ctx.upload(b); s.ctx.prep(); s.ctx.subcycle(17); s.ctx.subcycle(0); s.ctx.subcycle(33); s.ctx.finish(); s.ctx.download(b)
    assert s.ctx.stats().subcycles_done == 50
    s.close()
    assert not util.compare(d, b, a)


@pytest.mark.parametrize("mode", ["plain", "exchange", "exchange-serial", "single-kernel"])
def test_device_resident_steps_inputs_only_upload(mode, monkeypatch):
    """The state (u, v, sigma, iceumask) stays on the device between evp() calls; only the inputs are
    re-uploaded (evpk_upload with state == NULL).  Ice retreats and advances between the calls, which
    exercises the skip-if-still-inactive logic of the prep kernels.  Oracle: the same calls back to back."""
    if mode.startswith("exchange"):        # x-slab code path on one rank: ghost zones, two streams
        monkeypatch.setenv("EVPK_FORCE_EXCHANGE", "1")
    if mode == "exchange-serial":
        monkeypatch.setenv("EVPK_OVERLAP", "0")
    if mode == "single-kernel":
        monkeypatch.setenv("EVPK_DOUBLE", "0")
    case, d, f = util.make_case(200, 116, 50, 29, land="continents")
    xmin = synth.global_min_dx(case)
    fo, fg = util.clone(f), util.clone(f)
    p = orc.make_params(3600.0, 30, xmin)
    s = dyn.EvpDynamics(d, fg, ndte=30, xmin=xmin)
    s.init_evp(3600.0)
    ctx = s.ctx
    for call in range(4):
        for ff in (fo, fg):
            if call == 1:      # retreat: a band of ice disappears
                cut = (ff["aice"] > 0) & (ff["aice"] < 0.8)
                for n in ("aice", "vice", "vsno", "aice_init"):
                    ff[n] = np.where(cut, 0.0, ff[n])
            if call == 2:      # advance again, and new wind
                for n in ("aice", "vice", "vsno", "aice_init", "strairxT", "strairyT", "strength"):
                    ff[n] = f[n].copy()
                ff["strairxT"] = -ff["strairxT"]
            if call == 3:
                ff["uocn"] = ff["uocn"] * 0.5
        nt, nu, _ = orc.evp(d, p, fo)
        if call == 0:
            ctx.upload(fg)
        else:
            ctx.upload_inputs(fg)
        ctx.prep(); ctx.subcycle(30); ctx.finish()
        st = ctx.stats()
        assert (st.icellt, st.icellu) == (nt, nu), call
        out = util.clone(fg)
        ctx.download(out)
        bad = util.compare(d, out, fo)
        assert not bad, f"call {call}: {bad[:6]}"
    s.close()


@pytest.mark.parametrize("mode", ["plain", "exchange", "tripole", "device-strength"])
def test_soak_twelve_steps_with_wandering_ice(mode, monkeypatch):
    """Twelve consecutive evp() calls with the state resident on the device while the ice cover wanders: patches melt
    away completely, reappear elsewhere, thin out below the a_min / m_min thresholds and thicken again.  Every prep takes
    a different mix of skipped / re-zeroed / newly active tiles, strip lists and (exchange mode) ghost-zone row lists;
    after every call the download must equal the oracle run with the same inputs, bit for bit."""
    if mode == "exchange":
        monkeypatch.setenv("EVPK_FORCE_EXCHANGE", "1")
    ns = "tripole" if mode == "tripole" else "open"
    case, d, f = util.make_case(260, 140, 65, 35, ns=ns, land="continents")
    dev_strength = dict(kstrength=1, krdg_partic=1, krdg_redist=1) if mode == "device-strength" else None
    xmin = synth.global_min_dx(case)
    base = util.clone(f)
    fo, fg = util.clone(f), util.clone(f)
    p = orc.make_params(3600.0, 16, xmin, **(dict(strength_mode=1, **dev_strength) if dev_strength else {}))
    s = dyn.EvpDynamics(d, fg, ndte=16, xmin=xmin, device_strength=dev_strength)
    s.init_evp(3600.0)
    ctx = s.ctx
    I, J = blocks.block_index_windows(d)
    # ghost cells must hold what a halo update would put there (ice_dyn_evp.F90:157-169): wrap E-W, and on the tripole grid
    # the north ghost row is the mirror image of the top row, T cell (i, ny+1) = (nx-i+1, ny)  (serial/ice_boundary.F90:801-888)
    nxg, nyg = d.nx_global, d.ny_global
    I = (I - 1) % nxg + 1
    if ns == "tripole":
        top = J > nyg
        I = [np.broadcast_to(I[n][None, :], (d.ny_block, d.nx_block)).copy() for n in range(d.nblocks)]
        for n in range(d.nblocks):
            I[n][top[n], :] = nxg - I[n][top[n], :] + 1
        J = np.minimum(J, nyg)
    else:
        I = [np.broadcast_to(I[n][None, :], (d.ny_block, d.nx_block)) for n in range(d.nblocks)]
    rng = np.random.default_rng(2026)
    for call in range(12):
        # a smooth, moving mask in global coordinates (identical in ghost cells): ice survives where it is positive
        kx, ky, ph = rng.uniform(0.02, 0.09), rng.uniform(0.03, 0.12), rng.uniform(0, 6.28)
        thin = rng.uniform(0.0, 1.0)
        for ff in (fo, fg):
            for n in range(d.nblocks):
                w = np.sin(kx * I[n] + ph) * np.cos(ky * J[n][:, None] - 0.5 * ph) + 0.35 * np.sin(0.7 * call)
                keep = np.where(w > 0.0, 1.0, 0.0) * np.where(w > 0.6, 1.0, thin * 0.01 + 0.001)   # some cells just above / below a_min
                for name in ("aice", "vice", "vsno", "aice_init"):
                    ff[name][n] = base[name][n] * keep
                ff["strength"][n] = base["strength"][n] * keep
                ff["strairxT"][n] = base["strairxT"][n] * np.cos(0.5 * call) - base["strairyT"][n] * np.sin(0.5 * call)
                ff["strairyT"][n] = base["strairyT"][n] * np.cos(0.5 * call) + base["strairxT"][n] * np.sin(0.5 * call)
            if dev_strength:
                synth.add_thickness_distribution(ff)
        nt, nu, _ = orc.evp(d, p, fo)
        if call == 0:
            ctx.upload(fg)
        else:
            ctx.upload_inputs(fg)
        ctx.prep(); ctx.subcycle(16); ctx.finish()
        st = ctx.stats()
        assert (st.icellt, st.icellu) == (nt, nu), call
        out = util.clone(fg)
        ctx.download(out)
        bad = util.compare(d, out, fo)
        assert not bad, f"call {call}: {bad[:6]}"
    s.close()


@pytest.mark.parametrize("env", [{}, {"EVPK_DOUBLE": "0"}, {"EVPK_FORCE_EXCHANGE": "1"}])
def test_tripole_fold_rewrites_inactive_top_row_cells(env, monkeypatch):
    """The tripole halo update rewrites the WHOLE top row of u, v after every subcycle (serial/ice_boundary.F90:801-888),
    also a U cell without ice whose mirror image across the fold has ice; the reference keeps one velocity array, so
    that cell carries the value into the next update.  Here the ice mask is deliberately NOT mirror-symmetric across the
    fold (ghost cells inconsistent with their images), which makes such cells common: the double-buffered kernels must
    still reproduce the single-array behaviour bit for bit, for 1, 2, 3 and 16 subcycles."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    case, d, f = util.make_case(260, 140, 65, 35, ns="tripole", land="continents")
    xmin = synth.global_min_dx(case)
    I, J = blocks.block_index_windows(d)
    for n in range(d.nblocks):
        w = np.sin(0.05 * I[n][None, :] + 1.0) * np.cos(0.07 * J[n][:, None] - 0.5)
        keep = np.where(w > 0.0, 1.0, 0.0) * np.where(w > 0.6, 1.0, 0.004)
        for name in ("aice", "vice", "vsno", "aice_init", "strength"):
            f[name][n] = f[name][n] * keep
    for ndte in (1, 2, 3, 16):
        fo, fg = util.clone(f), util.clone(f)
        p = orc.make_params(3600.0, ndte, xmin)
        s = dyn.EvpDynamics(d, fg, ndte=ndte, xmin=xmin)
        s.init_evp(3600.0)
        orc.evp(d, p, fo)
        s.evp(3600.0)
        s.close()
        bad = util.compare(d, fg, fo)
        assert not bad, (ndte, bad[:4])


@pytest.mark.parametrize("mode", ["given", "absent", "inconsistent", "disabled"])
def test_compact_metrics_from_htn_hte(mode, monkeypatch):
    """HTN/HTE are optional: when they reproduce the eight metric planes bit for bit the kernel reads them instead
    (same results); when they are absent or do NOT reproduce the planes, the planes are used."""
    case, d, f = util.make_case(130, 96, 65, 48, land="continents")
    if mode == "absent":
        del f["HTN"], f["HTE"]
    if mode == "inconsistent":
        f["HTN"] = f["HTN"] * 1.0000001          # a grid whose metric planes were not built by the reference formulas
    if mode == "disabled":
        monkeypatch.setenv("EVPK_COMPACT_METRICS", "0")
    xmin = synth.global_min_dx(case)
    fo, fg = util.clone(f), util.clone(f)
    p = orc.make_params(3600.0, 24, xmin)
    s = dyn.EvpDynamics(d, fg, ndte=24, xmin=xmin)
    s.init_evp(3600.0)
    for _ in range(2):
        orc.evp(d, p, fo)
        s.evp(3600.0)
        assert not util.compare(d, fg, fo)
    s.close()


def test_principal_stress():
    """ice_dyn_shared.F90:853-893 on the device-resident state vs the oracle on the downloaded arrays."""
    import ctypes as ct
    case, d, f = util.make_case(100, 116, 50, 58, land="continents")
    xmin = synth.global_min_dx(case)
    s = dyn.EvpDynamics(d, f, ndte=40, xmin=xmin)
    s.init_evp(3600.0)
    s.evp(3600.0)
    g1, g2 = s.principal_stress()
    s.close()
    o1, o2 = np.zeros_like(g1), np.zeros_like(g2)
    for n in range(d.nblocks):
        orc.lib().orc_principal_stress(d.nx_block, d.ny_block, *[x[n].ctypes.data_as(orc.c_f64p) for x in
                                       (f["stressp_1"], f["stressm_1"], f["stress12_1"], f["prs_sig"])],
                                       o1[n].ctypes.data_as(orc.c_f64p), o2[n].ctypes.data_as(orc.c_f64p))
    m = util.cell_mask(d, "phys")
    assert np.array_equal(g1[m], o1[m]) and np.array_equal(g2[m], o2[m])
    assert (g1[m] < 1e29).sum() > 100


def test_context_create_destroy_many_times():
    case, d, f = util.make_case(100, 116, 50, 58, land="continents")
    xmin = synth.global_min_dx(case)
    ref = None
    for k in range(12):
        g = util.clone(f)
        s = dyn.EvpDynamics(d, g, ndte=6, xmin=xmin)
        s.init_evp(3600.0)
        s.evp(3600.0)
        s.close()
        if ref is None:
            ref = g
        else:
            assert not util.compare(d, g, ref)


def test_errors_are_reported_not_fatal():
    case, d, f = util.make_case(100, 116, 100, 116)
    ctx = evpk.Context(d, f)
    with pytest.raises(evpk.EvpkError, match="set_params"):
        ctx.upload(f)
    p = dyn.set_evp_parameters(3600.0, 120, False, 1.0)
    ctx.set_params(p)
    with pytest.raises(evpk.EvpkError, match="upload"):
        ctx.prep()
    g = dict(f); g["aice"] = None
    with pytest.raises(evpk.EvpkError, match="NULL"):
        ctx.upload(g)
    ctx.close()
    # blocks outside the rank's share of the domain (create_distrb_cart's x ranges) are refused; missing blocks are not:
    # they are eliminated land blocks (ice_domain.F90:387-441)
    d2 = blocks.create_distrb_cart(100, 116, 50, 116, nprocs=2, rank=1)
    d2.rank = 0
    f2 = synth.make_block_fields(case, d2)
    with pytest.raises(evpk.EvpkError, match="slenderX1"):
        evpk.Context(d2, f2)


def test_full_size_3600x2700_properties():
    """BASELINE headline grid on one GPU.  The oracle checks the first subcycles on the whole grid
    bit for bit; then size-independent properties after more subcycles: finite fields, inactive cells
    untouched, speeds bounded, and one more evp() on the result reproduces itself from a restart
    (uvel/vvel/sigma/iceumask are the complete cross-step state, ice_restart_driver.F90:122-176)."""
    nx, ny = 3600, 2700
    case, d, f = util.make_case(nx, ny, 450, 2700, land="continents", dt=450.0)   # 8 blocks = the 8-GPU slabs
    xmin = synth.global_min_dx(case)
    fo = util.clone(f)
    p = orc.make_params(450.0, 120, xmin)
    nt, nu, _ = orc.evp(d, p, fo, nsub=3)
    s = dyn.EvpDynamics(d, f, ndte=120, xmin=xmin)
    s.init_evp(450.0)
    g = util.clone(f)
    s.fields = g
    s.ctx.upload(g); s.ctx.prep(); s.ctx.subcycle(3); s.ctx.finish(); s.ctx.download(g)
    st = s.ctx.stats()
    assert (st.icellt, st.icellu) == (nt, nu)
    bad = util.compare(d, g, fo, names=["uvel", "vvel", "strocnxT", "strocnyT", "aiu", "umass", "fm", "iceumask"] + util.SIGMA)
    assert not bad, bad[:6]
    del fo
    # full evp
    g = util.clone(f)
    s.fields = g
    s.evp(450.0)
    phys = util.cell_mask(d, "phys")
    for n in ["uvel", "vvel", "divu", "shear", "strintx", "strocnxT"] + util.SIGMA:
        assert np.isfinite(g[n][phys]).all(), n
    speed = np.hypot(g["uvel"], g["vvel"])[phys]
    assert 0.01 < speed.max() < 3.0
    off = phys & (g["iceumask"] == 0)
    assert not g["uvel"][off].any() and not g["vvel"][off].any()
    offT = phys & (g["icetmask"] == 0)
    assert not g["stressp_1"][offT].any() and not g["divu"][offT].any()
    # restart exactness: same inputs + the state just produced, through a NEW context, twice
    h1, h2 = util.clone(g), util.clone(g)
    s.fields = h1
    s.evp(450.0)
    s.close()
    s2 = dyn.EvpDynamics(d, h2, ndte=120, xmin=xmin)
    s2.set_evp_parameters(450.0)
    s2.evp(450.0)
    s2.close()
    assert not util.compare(d, h1, h2)


def test_full_size_3600x2700_tripole_config5():
    """BASELINE config 5: 3600x2700 with ns_boundary_type = 'tripole', on one GPU, against the oracle on the whole grid.
    (a) the first 4 subcycles of an ndte = 240 evp: two inside pairs of the two-subcycle kernel, each with its two band
        launches and folds next to the main launch (ice_dyn_evp.F90:392-400, serial/ice_boundary.F90:801-888);
    (b) a whole evp with ndte = 4: one inside pair, then the LAST2 launch + LAST band launch that end the evp, the twelve
        ice_HaloUpdate_stress folds (ice_dyn_evp.F90:416-481), evp_finish and the U->T averages -- every output compared;
    (c) ndte = 240 (the configuration's subcycle count): finite, bounded, inactive cells untouched, the fold symmetry of
        the top row, and restart exactness through a new context."""
    nx, ny = 3600, 2700
    case, d, f = util.make_case(nx, ny, 450, 2700, ns="tripole", land="continents", dt=450.0)
    xmin = synth.global_min_dx(case)
    # (a)
    fo = util.clone(f)
    nt, nu, _ = orc.evp(d, orc.make_params(450.0, 240, xmin), fo, nsub=4)
    s = dyn.EvpDynamics(d, f, ndte=240, xmin=xmin)
    s.init_evp(450.0)
    g = util.clone(f)
    s.fields = g
    s.ctx.upload(g); s.ctx.prep(); s.ctx.subcycle(4); s.ctx.finish(); s.ctx.download(g)
    st = s.ctx.stats()
    assert (st.icellt, st.icellu) == (nt, nu)
    assert st.kernel2_launches == 2            # the two-subcycle kernel ran, with its band launches
    bad = util.compare(d, g, fo, names=["uvel", "vvel", "strocnx", "strocny", "strocnxT", "strocnyT", "aiu", "umass", "fm",
                                        "iceumask"] + util.SIGMA)
    assert not bad, bad[:6]
    s.close()
    # (b)
    fo = util.clone(f)
    orc.evp(d, orc.make_params(450.0, 4, xmin), fo)
    g = util.clone(f)
    s = dyn.EvpDynamics(d, g, ndte=4, xmin=xmin)
    s.init_evp(450.0)
    s.evp(450.0)
    assert s.ctx.stats().kernel2_launches == 2
    bad = util.compare(d, g, fo)
    assert not bad, bad[:6]
    s.close()
    del fo
    # (c)
    g = util.clone(f)
    s = dyn.EvpDynamics(d, g, ndte=240, xmin=xmin)
    s.init_evp(450.0)
    s.evp(450.0)
    assert s.ctx.stats().kernel2_launches == 120
    phys = util.cell_mask(d, "phys")
    for n in ["uvel", "vvel", "divu", "shear", "strintx", "strocnxT"] + util.SIGMA:
        assert np.isfinite(g[n][phys]).all(), n
    speed = np.hypot(g["uvel"], g["vvel"])[phys]
    assert 0.01 < speed.max() < 3.0
    off = phys & (g["iceumask"] == 0)
    assert not g["uvel"][off].any() and not g["vvel"][off].any()
    offT = phys & (g["icetmask"] == 0)
    assert not g["stressp_1"][offT].any() and not g["divu"][offT].any()
    # the fold leaves the top row antisymmetric about the poles: u(i, ny) = -u(nx - i, ny)  (serial/ice_boundary.F90:818-824)
    U = blocks.gather_global(d, g["uvel"]); V = blocks.gather_global(d, g["vvel"])
    i = np.arange(1, nx)
    assert np.array_equal(U[ny - 1, i - 1], -U[ny - 1, nx - i - 1]) and np.array_equal(V[ny - 1, i - 1], -V[ny - 1, nx - i - 1])
    assert np.abs(U[ny - 1]).max() > 1e-4
    del U, V
    h1, h2 = util.clone(g), util.clone(g)
    s.fields = h1
    s.evp(450.0)
    s.close()
    s2 = dyn.EvpDynamics(d, h2, ndte=240, xmin=xmin)
    s2.set_evp_parameters(450.0)
    s2.evp(450.0)
    s2.close()
    assert not util.compare(d, h1, h2)


@pytest.mark.parametrize("ns", ["tripole", "open"])
def test_restart_records_from_the_device_state(ns, tmp_path):
    """evpk_restart_write / evpk_restart_read: the dynamics records of the binary restart (ice_restart_driver.F90:122-176,
    :295-412) straight from / into the device state.  'run 2 steps' == 'run 1, write, NEW context, read, run 1', bit for
    bit, on a tripole grid (ghost cells refilled as restartfile does, incl. the ice_HaloUpdate_stress pairings, :370-395);
    and the file is byte-identical to what the host-side writer makes of the downloaded arrays."""
    import io
    from cice5_amd import restart
    case, d, f = util.make_case(96, 64, 24, 32, ns=ns, land="continents")
    xmin = synth.global_min_dx(case)

    def step(ctx, ff, first):
        if first:
            ctx.upload(ff)
        else:
            ctx.upload_inputs(ff)
        ctx.prep(); ctx.subcycle(20); ctx.finish()

    def second_inputs(ff):
        ff["aice"] *= 0.97
        ff["vice"] *= 0.97
        ff["strairxT"], ff["strairyT"] = ff["strairyT"].copy(), -ff["strairxT"]

    # A: two steps in one context
    fa = util.clone(f)
    sa = dyn.EvpDynamics(d, fa, ndte=20, xmin=xmin)
    sa.init_evp(3600.0)
    step(sa.ctx, fa, True)
    second_inputs(fa)
    step(sa.ctx, fa, False)
    sa.ctx.download(fa)
    sa.close()
    # B: one step, write
    fb = util.clone(f)
    sb = dyn.EvpDynamics(d, fb, ndte=20, xmin=xmin)
    sb.init_evp(3600.0)
    step(sb.ctx, fb, True)
    path = str(tmp_path / "iced.dyn")
    sb.ctx.restart_write(path)
    sb.ctx.download(fb)
    sb.close()
    buf = io.BytesIO()
    restart.write_dynamics_records(buf, d, fb)
    assert open(path, "rb").read() == buf.getvalue()
    # C: new context, read, second step
    fc = util.clone(f)
    second_inputs(fc)
    sc = dyn.EvpDynamics(d, fc, ndte=20, xmin=xmin)
    sc.set_evp_parameters(3600.0)
    sc.ctx.restart_read(path)
    step(sc.ctx, fc, False)
    sc.ctx.download(fc)
    sc.close()
    assert np.abs(fa["uvel"]).max() > 1e-3
    bad = util.compare(d, fc, fa)
    assert not bad, bad[:6]


@pytest.mark.parametrize("shape", [(360, 300, 15, 300, 120), (130, 96, 130, 96, 31), (1440, 1080, 360, 1080, 8), (62, 40, 31, 20, 13)])
@pytest.mark.parametrize("tile", ["", "0", "1"])
def test_tripole_band_launches_on_the_second_stream(shape, tile, monkeypatch):
    """EVPK_BAND_FUSED=0: the one-rank tripole top band as it ran until round 3 -- two band launches and two folds per pair
    on the second stream (still the path of x-slab ranks) -- stays bit-identical; and the fused band (default) under both the
    marching and the one-row-per-wave pair kernels (EVPK_TILE fixes the tuner's choice)."""
    nx, ny, bsx, bsy, ndte = shape
    if tile:
        monkeypatch.setenv("EVPK_TILE", tile)
    _both(nx, ny, bsx, bsy, ns="tripole", land="continents", ndte=ndte, ncalls=2)
    monkeypatch.setenv("EVPK_BAND_FUSED", "0")
    _both(nx, ny, bsx, bsy, ns="tripole", land="continents", ndte=ndte)


@pytest.mark.parametrize("ns", ["open", "tripole"])
def test_strip_list_compacted_on_the_host(ns, monkeypatch):
    """EVPK_DEVICE_STRIPS=0: the strip list of the pair kernels built on the host from the flags (the path of x-slab ranks and of
    rounds 1-2) against the default, where flags, ordered compaction and counts stay on the device and evpk_prep never waits;
    three calls with changing ice, the counts of active cells included"""
    monkeypatch.setenv("EVPK_DEVICE_STRIPS", "0")
    _both(100, 116, 25, 29, ns=ns, land="continents", ndte=30, ncalls=3)
    monkeypatch.delenv("EVPK_DEVICE_STRIPS")
    _both(100, 116, 25, 29, ns=ns, land="continents", ndte=30, ncalls=3)
    _both(320, 384, 320, 384, ns=ns, ndte=12, ncalls=2)


def test_bound_time_is_reported():
    """evpk_stats.bound_ms: the halo / fold updates of the subcycle loop (timer_bound in the reference) are timed with
    sampled HIP events: none on a cyclic one-rank open grid (the kernel wraps in place) nor on a one-rank tripole grid whose
    pairs fold inside the launch (band_pair; ndte even: no one-subcycle tail), some when the fold is a launch of its own
    (odd ndte: the tail subcycle; EVPK_BAND_FUSED=0: the band launches on the second stream)."""
    for ns, ndte, env, want in (("open", 30, None, False), ("tripole", 30, None, False), ("tripole", 31, None, True),
                                ("tripole", 30, "0", True)):
        if env is not None:
            os.environ["EVPK_BAND_FUSED"] = env
        case, d, f = util.make_case(130, 96, 130, 96, ns=ns, land="continents")
        s = dyn.EvpDynamics(d, f, ndte=ndte, xmin=synth.global_min_dx(case))
        os.environ.pop("EVPK_BAND_FUSED", None)
        s.init_evp(3600.0)
        s.evp(3600.0)
        st = s.ctx.stats()
        assert (st.bound_updates > 0) == want, (ns, ndte, env, st.bound_updates, st.bound_ms)
        if not want or st.bound_updates >= 5:          # (every fifth update is timed)
            assert (st.bound_ms > 0.0) == want, (ns, ndte, env, st.bound_updates, st.bound_ms)
        s.close()


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("ns", ["open", "tripole"])
def test_resident_state_sparse_transfers(ns, mode, fused, monkeypatch):
    """The drop-in as a host model drives it step after step: page-locked arrays, the state resident on the device, inputs
    up, only the every-step outputs down (dyn.EVERY_STEP_OUTPUTS), and evpk_params.sparse_io -- only the tiles with ice
    move.  Ice wanders, melts away and comes back over eight calls; what is delivered equals the oracle after every call,
    and on the last call every output is requested (arrays that were not delivered every step arrive whole).
    mode = evpk_params.sparse_io: 1 aice, vice, vsno whole, 2 aice alone whole (vice = vsno = 0 where aice = 0: _wander scales all
    three by one mask); fused: up to twelve arrays per gather / scatter launch (round 5) or one launch per array."""
    monkeypatch.setenv("EVPK_XFER_FUSED", fused)
    from tests.test_multirank_gpu import _wander
    case, d, f = util.make_case(520, 200, 130, 100, ns=ns, land="continents")
    xmin = synth.global_min_dx(case)
    base = util.clone(f)
    fo, fg = util.clone(f), util.clone(f)
    p = orc.make_params(3600.0, 12, xmin)
    every = list(dyn.EVERY_STEP_OUTPUTS)
    s = dyn.EvpDynamics(d, fg, ndte=12, xmin=xmin, pin_host=True, resident=True, outputs=every, sparse_io=mode)
    s.init_evp(3600.0)
    ncalls = 8
    for call in range(ncalls):
        for ff in (fo, fg):
            _wander(d, base, ff, call, ns)
            for n in ("strairxT", "strairyT"):          # the precondition of sparse_io: no T-grid forcing where there is no ice
                ff[n][...] = np.where(ff["aice"] > 0.0, ff[n], 0.0)
        if call == ncalls - 1:
            s._outputs = None
        nt, nu, _ = orc.evp(d, p, fo)
        s.evp(3600.0)
        st = s.ctx.stats()
        assert (st.icellt, st.icellu) == (nt, nu), call
        bad = util.compare(d, fg, fo, names=None if call == ncalls - 1 else every)
        assert not bad, f"call {call}: {bad[:6]}"
    assert np.abs(fo["uvel"]).max() > 1e-3
    s.close()


@pytest.mark.parametrize("ns,bs", [("open", (100, 116)), ("open", (25, 29)), ("tripole", (24, 16)), ("tripole", (48, 32))])
def test_transport_upwind_on_the_resident_velocities(ns, bs):
    """SURVEY S8 row f-3, first step: transport_upwind (ice_transport_driver.F90:634-772) -- edge velocities from the
    uvel / vvel the evp left on the device, their E-face / N-face halo updates (tripole: the fold rules of those field
    locations, serial/ice_boundary.F90:826-846) and upwind_field on every array of the work array -- against the oracle's
    restatement, bit for bit, on BASELINE config 1's shape and on tripole grids, one and many blocks."""
    nx, ny = (100, 116) if ns == "open" else (96, 64)
    case, d, f = util.make_case(nx, ny, *bs, ns=ns, land="continents")
    synth.add_thickness_distribution(f)
    xmin = synth.global_min_dx(case)
    fo, fg = util.clone(f), util.clone(f)
    orc.evp(d, orc.make_params(3600.0, 30, xmin), fo)
    s = dyn.EvpDynamics(d, fg, ndte=30, xmin=xmin)
    s.init_evp(3600.0)
    s.evp(3600.0)
    assert not util.compare(d, fg, fo, names=["uvel", "vvel"])
    # the work array of state_to_work (:1432-1448): aice0, then per category aicen, vicen, vsnon (+ one aicen-weighted tracer)
    ncat = f["aicen"].shape[1]
    planes = [f["aice0"]]
    for n in range(ncat):
        a, v = f["aicen"][:, n], f["vicen"][:, n]
        planes += [a, v, 0.2 * v, a * (0.3 + 0.1 * n)]
    works = np.ascontiguousarray(np.stack(planes, axis=1))              # (nblocks, narr, ny_block, nx_block)
    for k in range(works.shape[1]):                                      # ghost cells current, as after bound_state
        w = np.ascontiguousarray(works[:, k])
        orc.halo_r8(d, w, C.LOC_CENTER, C.KIND_SCALAR, 0.0)
        works[:, k] = w
    wo, wg = works.copy(), works.copy()
    orc.transport_upwind(d, 3600.0, fo, wo)
    s.ctx.transport_upwind(3600.0, wg)
    s.close()
    phys = util.cell_mask(d, "phys")
    assert np.abs(wo - works)[:, :, phys[0]].max() > 1e-6 if d.nblocks == 1 else np.abs(wo - works).max() > 1e-6
    assert np.array_equal(wg, wo)                                       # physical cells advected alike, ghost cells untouched by both


def _upwind_state(d, f, ntrcr_dim, tag):
    """aice0, aicen, vicen, vsnon, trcrn for transport_upwind with every tracer rule of state_to_work / compute_tracers:
    Tsfc, qice (ice volume), qsno (snow volume), alvl, vlvl, apnd (on alvl), hpnd (on apnd), fbri, a brine tracer; ghost cells current"""
    from tests.golden import refvec as rv
    dep, n_tsfc, n_alvl, n_apnd, n_fbri, pond = rv.TRACER_CASES[tag]
    ntrcr = len(dep)
    synth.add_thickness_distribution(f)
    aicen = np.ascontiguousarray(f["aicen"]); vicen = np.ascontiguousarray(f["vicen"])
    ncat = aicen.shape[1]
    vsnon = 0.2 * vicen
    vsnon[:, 1] = 0.0                                                    # a category without snow
    trcrn = np.zeros((d.nblocks, ncat, ntrcr_dim) + aicen.shape[2:])
    for n in range(ncat):
        for it in range(ntrcr):
            base = (-5.0 - n - 0.3 * it) if it < 3 else (0.2 + 0.1 * it + 0.05 * n)        # fractions for alvl / apnd / fbri, their dependents
            trcrn[:, n, it] = np.where(aicen[:, n] > 0, base * (1.0 + 0.05 * np.sin(aicen[:, n] * 40.0 + it)), 0.0)
        trcrn[:, n, ntrcr:] = 777.0
    aice0 = np.where(f["tmask"] > 0, 1.0 - aicen.sum(axis=1), 0.0)
    for arr in (aicen, vicen, vsnon, trcrn.reshape(d.nblocks, -1, *aicen.shape[2:])):
        for k in range(arr.shape[1]):
            w = np.ascontiguousarray(arr[:, k]); orc.halo_r8(d, w, C.LOC_CENTER, C.KIND_SCALAR, 0.0); arr[:, k] = w
    orc.halo_r8(d, aice0, C.LOC_CENTER, C.KIND_SCALAR, 0.0)
    return [aice0, aicen, vicen, vsnon, trcrn], dict(ntrcr=ntrcr, trcr_depend=dep, nt_Tsfc=n_tsfc, nt_alvl=n_alvl, nt_apnd=n_apnd, nt_fbri=n_fbri, ponds=pond)


@pytest.mark.parametrize("ns,bs,tag", [("open", (100, 116), "lvl_ponds"), ("open", (25, 29), "cesm_ponds"), ("tripole", (24, 32), "lvl_ponds"),
                                       ("tripole", (96, 64), "plain"), ("open", (32, 40), "plain")])
def test_transport_upwind_with_the_state_transforms(ns, bs, tag):
    """transport_upwind WHOLE (ice_transport_driver.F90:634-772): state_to_work inside the gather, upwind_field, work_to_state
    with compute_tracers and bound_state inside the scatter -- the caller's aice0, aicen, vicen, vsnon, trcrn in, the same arrays
    out, every cell of every block against the oracle (whose compute_tracers is pinned by the reference's own,
    tests/test_ref_pins.py), after a real evp; all tracer rules; padded blocks; tracer slots beyond ntrcr untouched."""
    nx, ny = (100, 116) if ns == "open" else (96, 64)
    case, d, f = util.make_case(nx, ny, *bs, ns=ns, land="continents")
    xmin = synth.global_min_dx(case)
    state, kw = _upwind_state(d, f, 11, tag)
    fo, fg = util.clone(f), util.clone(f)
    orc.evp(d, orc.make_params(3600.0, 30, xmin), fo)
    s = dyn.EvpDynamics(d, fg, ndte=30, xmin=xmin)
    s.init_evp(3600.0)
    s.evp(3600.0)
    so, sg = [a.copy() for a in state], [a.copy() for a in state]
    orc.transport_upwind_state(d, 3600.0, fo, *so, **kw)
    s.ctx.transport_upwind_state(3600.0, *sg, **kw)
    s.close()
    every = util.cell_mask(d, "all")
    for name, a, b_, a0 in zip(("aice0", "aicen", "vicen", "vsnon", "trcrn"), sg, so, state):
        m = every if a.ndim == 3 else (every[:, None] if a.ndim == 4 else every[:, None, None])
        m = np.broadcast_to(m, a.shape)
        assert np.array_equal(a[m], b_[m]), (name, int((a[m] != b_[m]).sum()), np.argwhere((a != b_) & m)[:4])
        assert np.abs(b_ - a0).max() > 0
    assert np.array_equal(sg[4][:, :, kw["ntrcr"]:], state[4][:, :, kw["ntrcr"]:])


@pytest.mark.parametrize("ns", ["open", "tripole"])
def test_transport_upwind_state_next_to_eliminated_land_blocks(ns):
    """bound_state next to an eliminated land block (ice_domain.F90:387-441): the halo update writes its fill 0 into the ghost
    cells of aicen / vicen / vsnon / trcrn there (mpi/ice_boundary.F90, srcBlock == 0) -- not compute_tracers of an empty cell,
    which would leave Tsfc = Tocnfrz and fbri = 1 (round 3's advisor finding)."""
    nx, ny, bsx, bsy = 120, 96, 6, 4
    case = synth.SynthCase(nx=nx, ny=ny, land="continents", ns_boundary=C.BND_NAMES[ns])
    full = blocks.create_distrb_cart(nx, ny, bsx, bsy, ns_boundary_type=ns)
    ff = synth.make_block_fields(case, full)
    work = [int(ff["tmask"][n, b.jlo - 1:b.jhi, b.ilo - 1:b.ihi].any()) for n, b in enumerate(full.local_blocks)]
    assert sum(work) < len(work)
    for k in (0, len(work) - 1):
        work[k] = 1
    d = blocks.create_distrb_cart(nx, ny, bsx, bsy, work_per_block=work, ns_boundary_type=ns)
    assert d.nblocks < full.nblocks
    f = synth.make_block_fields(case, d)
    xmin = synth.global_min_dx(case)
    state, kw = _upwind_state(d, f, 11, "lvl_ponds")
    fo, fg = util.clone(f), util.clone(f)
    orc.evp(d, orc.make_params(3600.0, 12, xmin), fo)
    s = dyn.EvpDynamics(d, fg, ndte=12, xmin=xmin)
    s.init_evp(3600.0)
    s.evp(3600.0)
    so, sg = [a.copy() for a in state], [a.copy() for a in state]
    orc.transport_upwind_state(d, 3600.0, fo, *so, **kw)
    s.ctx.transport_upwind_state(3600.0, *sg, **kw)
    s.close()
    every = util.cell_mask(d, "all")
    for name, a, b_ in zip(("aice0", "aicen", "vicen", "vsnon", "trcrn"), sg, so):
        m = every if a.ndim == 3 else (every[:, None] if a.ndim == 4 else every[:, None, None])
        m = np.broadcast_to(m, a.shape)
        assert np.array_equal(a[m], b_[m]), (name, int((a[m] != b_[m]).sum()), np.argwhere((a != b_) & m)[:4])
    # the case bites: some ghost cell next to a land block holds the fill where compute_tracers would have left Tocnfrz
    ghost = ~util.cell_mask(d, "phys")
    assert (so[4][:, :, kw["nt_Tsfc"] - 1][np.broadcast_to(ghost[:, None], so[1].shape)] == 0.0).any()


@pytest.mark.parametrize("keep_cover", [False, True])
def test_evp_after_transport_upwind_state_on_a_resident_context(keep_cover, monkeypatch):
    """Round 4's advisor finding: evpk_transport_upwind_state builds its land-block coverage in evp's T-grid wind plane; a later evp on
    the same resident context whose NEW ice lies two tiles away from the old ice skips the ice-free tile beside it (k_prep1a) and
    averages that tile's wind into strairx (k_to_ugrid4) -- the plane must hold zeros there again.  evp, transport_upwind_state, inputs
    with a new ice patch whose edge lies on a tile border (tiles are 64 x 4 cells), evp: every output against the oracle.
    keep_cover: the same with the restoring fill switched off (EVPK_DEBUG_KEEP_COVER) must DIFFER -- the case bites."""
    if keep_cover:
        monkeypatch.setenv("EVPK_DEBUG_KEEP_COVER", "1")
    nx, ny, bsx, bsy = 320, 96, 8, 4                      # (five tile columns: tiles on the slab's rim are never skipped)
    case = synth.SynthCase(nx=nx, ny=ny, land="continents", ice="full")
    full = blocks.create_distrb_cart(nx, ny, bsx, bsy)
    ff = synth.make_block_fields(case, full)
    work = [int(ff["tmask"][n, b.jlo - 1:b.jhi, b.ilo - 1:b.ihi].any()) for n, b in enumerate(full.local_blocks)]
    assert sum(work) < len(work)
    d = blocks.create_distrb_cart(nx, ny, bsx, bsy, work_per_block=work)
    assert d.nblocks < full.nblocks                       # eliminated land blocks: the coverage plane is built
    f = synth.make_block_fields(case, d)
    base = util.clone(f)
    xmin = synth.global_min_dx(case)
    state, kw = _upwind_state(d, util.clone(f), 11, "lvl_ponds")
    _, J = blocks.block_index_windows(d)
    Jg = np.stack([np.broadcast_to(J[n][:, None], (d.ny_block, d.nx_block)) for n in range(d.nblocks)])
    keep0 = (Jg <= 23)                                    # ice in tile rows 0 .. 5
    keep1 = keep0 | ((Jg >= 48) & (Jg <= 51))             # + a band that fills tile row 12: tile rows 11 and 13 hold no data and were
    fo, fg = util.clone(f), util.clone(f)                 #   not active at the first evp
    p = orc.make_params(3600.0, 12, xmin)
    s = dyn.EvpDynamics(d, fg, ndte=12, xmin=xmin, resident=True)
    s.init_evp(3600.0)
    # (the first evp of a context treats every tile as active -- its state was just uploaded --, so the tiles a later evp may skip are
    #  only known from the second one on: two evps on the first ice cover, the transport, then the evp with the new band)
    for call, keep in enumerate((keep0, keep0, keep1)):
        for x in (fo, fg):
            for name in ("aice", "vice", "vsno", "aice_init", "strength", "strairxT", "strairyT"):
                x[name][...] = base[name] * keep
        nt, nu, _ = orc.evp(d, p, fo)
        s.evp(3600.0)
        bad = util.compare(d, fg, fo)
        if call < 2:
            assert not bad, bad[:6]
        if call == 1:
            so, sg = [a.copy() for a in state], [a.copy() for a in state]
            orc.transport_upwind_state(d, 3600.0, fo, *so, **kw)
            s.ctx.transport_upwind_state(3600.0, *sg, **kw)
            assert all(np.array_equal(a, b_) for a, b_ in zip(sg[1:], so[1:]))
    s.close()
    assert nu > 0
    if keep_cover:
        assert bad and any(n in ("strairx", "uvel") for n, _, _ in bad), "the case does not exercise the stale coverage plane"
    else:
        assert not bad, bad[:6]


def _remap_on_device(d, f, mm, tm, tables, dt, order, midpt, env=None, monkeypatch=None):
    """horizontal_remap through the C ABI on synthetic velocities uploaded as the resident state"""
    ttype, depend, has = tables
    if env:
        for k, v in env.items():
            monkeypatch.setenv(k, v)
    s = dyn.EvpDynamics(d, f, ndte=10, xmin=1.0e4)
    s.set_evp_parameters(3600.0)
    s.ctx.upload(f)                                                      # uvel, vvel resident, as after an evp
    s.ctx.remap_init(f["dxu"], f["dyu"], f["hm"])
    rc = s.ctx.transport_remap(dt, mm, tm, ttype, depend, has, integral_order=order, l_dp_midpt=midpt)
    s.close()
    return rc


@pytest.mark.parametrize("ns,bs,order,midpt", [("open", (48, 40), 3, True), ("open", (12, 10), 2, False), ("open", (24, 20), 1, True),
                                              ("tripole", (48, 40), 3, True), ("tripole", (12, 10), 3, False),
                                              ("tripole", (24, 20), 2, True), ("tripole", (16, 8), 1, False)])
def test_transport_remap_matches_the_oracle(ns, bs, order, midpt):
    """SURVEY S8 row f-3, second step: horizontal_remap (ice_transport_remap.F90:309-850) -- masks, limited gradients,
    departure points (both rules), the triangles of every edge and their integrals (all three quadrature orders), the
    flux-form update of areas and of tracers of all three types -- against the oracle's restatement, bit for bit, open and
    tripole grids, one and many blocks."""
    case, d, f, mm, tm, tables = util.remap_case(48, 40, *bs, ns=ns, trcr_depend=(0, 1, 2 + 1, 2 + 2))     # types 1, 1, 1, 2, 2, 3
    assert list(tables[0]) == [1, 1, 1, 2, 2, 3]
    mo, to, mg, tg = mm.copy(), tm.copy(), mm.copy(), tm.copy()
    assert orc.horizontal_remap(d, 3600.0, f, mo, to, *tables, integral_order=order, l_dp_midpt=midpt) == 0
    assert _remap_on_device(d, f, mg, tg, tables, 3600.0, order, midpt) == 0
    assert np.abs(mo - mm).max() > 1e-3 and np.abs(to - tm).max() > 1e-3
    assert np.array_equal(mg, mo)            # physical cells advanced alike, ghost cells left alone by both
    assert np.array_equal(tg, to)


@pytest.mark.parametrize("ns", ["open", "tripole"])
def test_transport_remap_through_the_general_halo_path(ns, monkeypatch):
    """the ghost-cell updates of mm, tm, mx, my, tc, tx, ty through the pack / exchange / unpack path several ranks take
    (EVPK_FORCE_EXCHANGE=1), fourteen planes at a time through the scratch state planes; and ghost cells of the caller's
    arrays that are not current do not matter (the device refreshes them first)"""
    case, d, f, mm, tm, tables = util.remap_case(48, 40, 24, 20, ns=ns, ncat=5)
    mo, to, mg, tg = mm.copy(), tm.copy(), mm.copy(), tm.copy()
    assert orc.horizontal_remap(d, 3600.0, f, mo, to, *tables) == 0
    if d.nblocks == 1:
        ghost = ~util.cell_mask(d, "phys")[0]
        mg[0][:, ghost] = 7.0; tg[0][:, :, ghost] = -3.0
    assert _remap_on_device(d, f, mg, tg, tables, 3600.0, 3, True, {"EVPK_FORCE_EXCHANGE": "1"}, monkeypatch) == 0
    if d.nblocks == 1:
        mg[0][:, ghost] = mm[0][:, ghost]; tg[0][:, :, ghost] = tm[0][:, :, ghost]
    assert np.array_equal(mg, mo) and np.array_equal(tg, to)


def test_transport_remap_stale_ghost_cells_and_no_tracers():
    case, d, f, mm, tm, tables = util.remap_case(48, 40, 48, 40, ns="tripole")
    ghost = ~util.cell_mask(d, "phys")[0]
    mo, to, mg, tg = mm.copy(), tm.copy(), mm.copy(), tm.copy()
    assert orc.horizontal_remap(d, 3600.0, f, mo, to, *tables) == 0
    mg[0][:, ghost] = 7.0; tg[0][:, :, ghost] = -3.0
    assert _remap_on_device(d, f, mg, tg, tables, 3600.0, 3, True) == 0
    assert np.array_equal(mg[0][:, ~ghost], mo[0][:, ~ghost]) and np.array_equal(tg[0][:, :, ~ghost], to[0][:, :, ~ghost])
    assert (mg[0][:, ghost] == 7.0).all() and (tg[0][:, :, ghost] == -3.0).all()
    # areas alone (ntrace = 0)
    m1 = mm.copy()
    assert _remap_on_device(d, f, m1, None, ([], [], []), 3600.0, 3, True) == 0
    assert np.array_equal(m1, mo)


def test_transport_remap_through_planes_and_a_scatter_pass(monkeypatch):
    """EVPK_REMAP_DIRECT=0 (the update writes planes, a scatter pass delivers them) and EVPK_REMAP_FUSED=0 (transports and update as
    three kernels through HBM) stay bit-identical with the default, where the fused tile kernel writes the caller's arrays itself"""
    case, d, f, mm, tm, tables = util.remap_case(100, 116, 25, 29, ns="tripole", ncat=3)
    mo, to = mm.copy(), tm.copy()
    dt = 3600.0
    assert orc.horizontal_remap(d, dt, f, mo, to, *tables) == 0
    for env in ({}, {"EVPK_REMAP_DIRECT": "0"}, {"EVPK_REMAP_FUSED": "0"}):
        mg, tg = mm.copy(), tm.copy()
        assert _remap_on_device(d, f, mg, tg, tables, dt, 3, True, env=env, monkeypatch=monkeypatch) == 0
        for k in env:
            monkeypatch.delenv(k)
        phys = util.cell_mask(d, "phys")
        assert np.array_equal(mg[:, :, phys[0]] if d.nblocks == 1 else mg[np.broadcast_to(phys[:, None], mg.shape)],
                              mo[:, :, phys[0]] if d.nblocks == 1 else mo[np.broadcast_to(phys[:, None], mo.shape)]), env
        assert np.array_equal(tg[np.broadcast_to(phys[:, None, None], tg.shape)], to[np.broadcast_to(phys[:, None, None], to.shape)]), env
        ghost = ~phys
        assert np.array_equal(mg[np.broadcast_to(ghost[:, None], mg.shape)], mm[np.broadcast_to(ghost[:, None], mm.shape)]), env     # ghost cells left alone


def test_transport_remap_reports_the_two_abort_cases():
    """departure points outside the neighbouring cells (:1583-1607) and a negative area after the update (:3622-3640):
    the reference aborts; the library returns the case and leaves the arrays alone"""
    case, d, f, mm, tm, tables = util.remap_case(48, 40, 24, 20)
    mo, to, mg, tg = mm.copy(), tm.copy(), mm.copy(), tm.copy()
    dt = 3600.0 * 400
    assert orc.horizontal_remap(d, dt, f, mo, to, *tables) == 1
    assert _remap_on_device(d, f, mg, tg, tables, dt, 3, True) == evpk.REMAP_BAD_DEPARTURE
    assert np.array_equal(mg, mm) and np.array_equal(tg, tm)
    with pytest.raises(evpk.EvpkError):
        _remap_on_device(d, f, mg, tg, tables, 3600.0, 4, True)
    with pytest.raises(evpk.EvpkError):
        _remap_on_device(d, f, mg, tg, (tables[0], [0] * len(tables[0]), tables[2]), 3600.0, 3, True)


def test_transport_remap_after_an_evp():
    """the production order: evp leaves uvel, vvel on the device, transport_remap advects the ice state with them
    (ice_step_mod.F90:step_dynamics); BASELINE config 1's shape in 25 x 29 blocks, five categories"""
    case, d, f = util.make_case(100, 116, 25, 29, ns="open", land="continents")
    synth.add_thickness_distribution(f)
    synth.add_remap_grid(case, d, f)
    xmin = synth.global_min_dx(case)
    fo, fg = util.clone(f), util.clone(f)
    orc.evp(d, orc.make_params(3600.0, 30, xmin), fo)
    s = dyn.EvpDynamics(d, fg, ndte=30, xmin=xmin)
    s.init_evp(3600.0)
    s.evp(3600.0)
    assert not util.compare(d, fg, fo, names=["uvel", "vvel"])
    ncat = f["aicen"].shape[1]
    tables = orc.remap_tables([0, 1, 2, 2 + 1, 2 + 2])                   # hice, hsno; Tsfc (area), qice (hice), qsno (hsno), one on Tsfc, one on qice (type 3)
    ntrace = len(tables[0])
    mm = np.zeros((d.nblocks, ncat + 1) + f["aice0"].shape[1:])
    tm = np.zeros((d.nblocks, ncat, ntrace) + f["aice0"].shape[1:])
    mm[:, 0] = f["aice0"]
    for n in range(ncat):
        a, v = f["aicen"][:, n], f["vicen"][:, n]
        mm[:, n + 1] = a
        h = np.where(a > 1e-11, v / np.where(a > 1e-11, a, 1.0), 0.0)
        tm[:, n, 0], tm[:, n, 1] = h, 0.2 * h
        for k in range(2, ntrace):
            tm[:, n, k] = np.where(a > 1e-11, (-5.0 - n) * 10.0 ** (k - 2) * (1.0 + 0.2 * np.sin(0.1 * k * np.arange(a.shape[-1]))), 0.0)
    for arr in (mm.reshape(d.nblocks, -1, *mm.shape[2:]), tm.reshape(d.nblocks, -1, *mm.shape[2:])):
        for k in range(arr.shape[1]):
            w = np.ascontiguousarray(arr[:, k]); orc.halo_r8(d, w, C.LOC_CENTER, C.KIND_SCALAR, 0.0); arr[:, k] = w
    umax = max(np.abs(fo["uvel"]).max(), np.abs(fo["vvel"]).max())
    dt = 0.4 * xmin / umax                                                # departure points up to 0.4 cells away
    mo, to, mg, tg = mm.copy(), tm.copy(), mm.copy(), tm.copy()
    s.ctx.remap_init(f["dxu"], f["dyu"], f["hm"])
    assert orc.horizontal_remap(d, dt, fo, mo, to, *tables) == 0
    assert s.ctx.transport_remap(dt, mg, tg, *tables) == 0
    s.close()
    assert np.abs(mo - mm).max() > 1e-3
    assert np.array_equal(mg, mo) and np.array_equal(tg, to)


EAP_NE = [f"a11_{c}" for c in (1, 2, 3, 4)] + [f"a12_{c}" for c in (1, 2, 3, 4)] + list(evpk.EAP_HISTORY)


def _eap_both(ns, bs, ndte, calls=1, nx=100, ny=116, revised=False):
    from cice5_amd.eap_tables import eap_tables
    T = eap_tables()
    case, d, f = util.make_case(nx, ny, *bs, ns=ns, land="continents")
    synth.add_eap_state(f)
    xmin = synth.global_min_dx(case)
    fo, fg = util.clone(f), util.clone(f)
    p = orc.make_params(3600.0, ndte, xmin, revised_evp=revised) if revised else orc.make_params(3600.0, ndte, xmin)
    s = dyn.EvpDynamics(d, fg, ndte=ndte, xmin=xmin, revised_evp=revised)
    s.init_eap(3600.0, T)
    ne = util.cell_mask(d, "ne")
    bad = []
    for call in range(calls):
        if call:
            for ff in (fo, fg):
                ff["aice"] *= 0.9; ff["vice"] *= 0.9
                ff["strairxT"], ff["strairyT"] = ff["strairyT"].copy(), -ff["strairxT"]
                ff["aice"][:, :, : ff["aice"].shape[2] // 3] = 0.0          # ice disappears from a third of the domain
                ff["vice"][:, :, : ff["aice"].shape[2] // 3] = 0.0
                for n in ("aice", "vice"):
                    orc.halo_r8(d, ff[n], C.LOC_CENTER, C.KIND_SCALAR, 0.0)
        orc.eap(d, p, fo, T)
        s.eap(3600.0)
        bad += [(call,) + x for x in util.compare(d, fg, fo)]
        for n in EAP_NE:
            if not np.array_equal(fg[n][ne], fo[n][ne]):
                bad.append((call, n, int((fg[n][ne] != fo[n][ne]).sum()), float(np.abs(fg[n][ne] - fo[n][ne]).max())))
    s.close()
    return fo, fg, bad


@pytest.mark.parametrize("ns,bs,ndte", [("open", (100, 116), 120), ("open", (25, 29), 31), ("tripole", (50, 58), 40), ("tripole", (20, 29), 25)])
def test_eap_matches_the_oracle(ns, bs, ndte):
    """eap(dt) (ice_dyn_eap.F90:66-486) -- stress_eap with its table lookups, stepu, stepa every tenth
    subcycle, the velocity halo, no stress fold -- against the oracle's restatement, bit for bit: velocities, the twelve
    stresses, the structure tensor, the history fields and everything evp_finish leaves; BASELINE config 1's shape"""
    fo, fg, bad = _eap_both(ns, bs, ndte)
    assert not bad, bad[:8]
    assert np.abs(fo["a11_1"] - 0.5).max() > 0.05 and np.abs(fo["uvel"]).max() > 1e-3        # the anisotropy did evolve
    assert np.abs(fo["yieldstress12"]).max() > 0 and np.abs(fo["rdg_conv"]).max() > 0


def test_eap_two_launch_form_and_partial_subcycle_calls(monkeypatch):
    """EVPK_EAP_FUSED=0 (k_eap_stress -> k_eap_stepu with str(8) through memory, in place) gives the same bits as the fused
    launch; and the history planes, which only a call's last subcycle stores, are current after evpk_subcycle calls that
    stop in the middle of the loop"""
    monkeypatch.setenv("EVPK_EAP_FUSED", "0")
    fo, fg, bad = _eap_both("tripole", (25, 29), 22)
    assert not bad, bad[:8]
    monkeypatch.delenv("EVPK_EAP_FUSED")
    from cice5_amd.eap_tables import eap_tables
    T = eap_tables()
    case, d, f = util.make_case(100, 116, 50, 58, ns="tripole", land="continents")
    synth.add_eap_state(f)
    xmin = synth.global_min_dx(case)
    fo, fg = util.clone(f), util.clone(f)
    orc.eap(d, orc.make_params(3600.0, 23, xmin), fo, T)
    s = dyn.EvpDynamics(d, fg, ndte=23, xmin=xmin)
    s.init_eap(3600.0, T)
    s.ctx.upload(fg)
    s.ctx.prep()
    for n in (1, 9, 2, 11):                       # 23 subcycles in four calls: odd and even counts, stepa at ksub = 1, 11, 21
        s.ctx.subcycle(n)
    s.ctx.finish()
    s.ctx.download(fg)
    s.ctx.eap_download(fg)
    s.close()
    ne = util.cell_mask(d, "ne")
    assert not util.compare(d, fg, fo)
    for n in EAP_NE:
        assert np.array_equal(fg[n][ne], fo[n][ne]), n


def test_eap_three_calls_with_ice_that_disappears():
    """the structure tensor stays on the device between calls and is reset to isotropic where icetmask = 0 (:284-298)"""
    fo, fg, bad = _eap_both("tripole", (25, 29), 22, calls=3)
    assert not bad, bad[:8]
    assert (fo["a11_1"] == 0.5).any() and (fo["a11_1"] != 0.5).any()


def test_eap_structure_tensor_from_a_restart():
    """evpk_eap_upload: a11_1..4, a12_1..4 from the host (read_restart_eap, :1908-2010) before the first call"""
    from cice5_amd.eap_tables import eap_tables
    T = eap_tables()
    case, d, f = util.make_case(100, 116, 50, 58, ns="open", land="continents")
    synth.add_eap_state(f)
    rng = np.random.default_rng(3)
    for c in (1, 2, 3, 4):
        f[f"a11_{c}"] = rng.uniform(0.3, 0.7, f["uvel"].shape)
        f[f"a12_{c}"] = rng.uniform(-0.2, 0.2, f["uvel"].shape)
        for n in (f"a11_{c}", f"a12_{c}"):
            orc.halo_r8(d, f[n], C.LOC_CENTER, C.KIND_SCALAR, 0.0)
    xmin = synth.global_min_dx(case)
    fo, fg = util.clone(f), util.clone(f)
    orc.eap(d, orc.make_params(3600.0, 30, xmin), fo, T)
    s = dyn.EvpDynamics(d, fg, ndte=30, xmin=xmin)
    s.set_evp_parameters(3600.0)
    s.ctx.eap_init(T)
    s._eap = True
    s.ctx.eap_upload(fg)
    s.eap(3600.0)
    s.close()
    ne = util.cell_mask(d, "ne")
    assert not util.compare(d, fg, fo)
    for n in EAP_NE:
        assert np.array_equal(fg[n][ne], fo[n][ne]), n


def test_caller_arrays_in_device_memory():
    """A host model whose fields already live on the GPU passes device pointers in place of host arrays (same block
    layout): the library reads and writes them in place.  Here the arrays are torch tensors on the device."""
    import ctypes as ct
    torch = pytest.importorskip("torch")
    case, d, f = util.make_case(100, 116, 25, 29, ns="tripole", land="continents")
    xmin = synth.global_min_dx(case)
    fo = util.clone(f)
    p = orc.make_params(3600.0, 20, xmin)
    s = dyn.EvpDynamics(d, f, ndte=20, xmin=xmin)       # geometry from host arrays; the per-step arrays from the device
    s.init_evp(3600.0)
    dev = {n: torch.from_numpy(np.ascontiguousarray(a)).cuda() for n, a in f.items() if isinstance(a, np.ndarray)}
    ptr = lambda n, T: ct.cast(dev[n].data_ptr(), T) if n in dev else None
    for call in range(2):
        if call:
            for n in ("aice", "vice"):
                fo[n] *= 0.95
                dev[n] *= 0.95
        si, st = evpk.StepIn(), evpk.State()
        for n in evpk.STEP_IN_F64:
            setattr(si, n, ptr(n, evpk.c_f64p))
        st.uvel, st.vvel = ptr("uvel", evpk.c_f64p), ptr("vvel", evpk.c_f64p)
        for k in ("stressp", "stressm", "stress12"):
            setattr(st, k, (evpk.c_f64p * 4)(*[ptr(f"{k}_{c}", evpk.c_f64p) for c in (1, 2, 3, 4)]))
        st.iceumask = ptr("iceumask", evpk.c_i32p)
        for n in evpk.STATE_OUT_F64:
            setattr(st, n, ptr(n, evpk.c_f64p))
        st.icetmask, st.strength = None, None
        torch.cuda.synchronize()
        assert s.ctx._L.evpk_run(s.ctx._ctx, ct.byref(si), ct.byref(st)) == 0
        orc.evp(d, p, fo)
        got = {n: t.cpu().numpy() for n, t in dev.items()}
        got["icetmask"] = fo["icetmask"]
        bad = util.compare(d, got, fo)
        assert not bad, (call, bad[:4])
    s.close()


def _ice_state(d, f, ntrcr, ntrcr_dim, nt_qsno, nslyr):
    """aice0, aicen, vicen, vsnon, trcrn of a synthetic thickness distribution, ghost cells current"""
    synth.add_thickness_distribution(f)
    aicen = np.ascontiguousarray(f["aicen"]); vicen = np.ascontiguousarray(f["vicen"])
    ncat = aicen.shape[1]
    # holes: a category without ice here and there, and cells whose ice is below puny
    I, J = blocks.block_index_windows(d)
    for b in range(d.nblocks):
        x = (I[b] % d.nx_global)[None, :] * 0.37; y = J[b][:, None] * 0.23
        for n in range(ncat):
            hole = np.sin(x * (n + 1) + y) > 0.6
            aicen[b, n][hole] = 0.0; vicen[b, n][hole] = 0.0
            tiny = np.cos(x - y * (n + 2)) > 0.93
            aicen[b, n][tiny] *= 1e-12; vicen[b, n][tiny] *= 1e-12
    vsnon = 0.2 * vicen
    aice0 = np.where(f["tmask"] > 0, 1.0 - aicen.sum(axis=1), 0.0)
    trcrn = np.zeros((d.nblocks, ncat, ntrcr_dim) + aicen.shape[2:])
    for n in range(ncat):
        for it in range(ntrcr):
            base = -5.0 - n - 0.3 * it if not (nt_qsno <= it + 1 < nt_qsno + nslyr) else -1.2e8 * (1 + 0.1 * n)
            trcrn[:, n, it] = np.where(aicen[:, n] > 0, base * (1.0 + 0.05 * np.sin(aicen[:, n] * 40.0)), 0.0)
        trcrn[:, n, ntrcr:] = 777.0                                     # tracers not in use are not touched
    for arr in (aicen, vicen, vsnon, trcrn.reshape(d.nblocks, -1, *aicen.shape[2:])):
        for k in range(arr.shape[1]):
            w = np.ascontiguousarray(arr[:, k]); orc.halo_r8(d, w, C.LOC_CENTER, C.KIND_SCALAR, 0.0); arr[:, k] = w
    orc.halo_r8(d, aice0, C.LOC_CENTER, C.KIND_SCALAR, 0.0)
    return aice0, aicen, vicen, vsnon, trcrn


@pytest.mark.parametrize("ns,bs", [("open", (100, 116)), ("open", (25, 29)), ("tripole", (50, 58)), ("tripole", (20, 29))])
def test_transport_remap_with_the_state_transforms(ns, bs):
    """transport_remap as a whole (ice_transport_driver.F90:198-627, its optional checks off): state_to_tracers, horizontal_remap,
    tracers_to_state, bound_state -- the caller's aice0, aicen, vicen, vsnon, trcrn in, the same arrays out, every cell of every
    block (ghost cells included) against the oracle, after a real evp; snow enthalpy shift, unused tracer slots, categories
    without ice, cells whose new area is zero keep their values"""
    nx, ny = (100, 116)
    case, d, f = util.make_case(nx, ny, *bs, ns=ns, land="continents")
    synth.add_remap_grid(case, d, f)
    xmin = synth.global_min_dx(case)
    ntrcr, ntrcr_dim, nt_qsno, nslyr = 6, 8, 4, 2                      # Tsfc, 2 x qice, 2 x qsno, one on the area; two unused slots
    tables = orc.remap_tables([0, 1, 1, 2, 2, 0])
    state = _ice_state(d, f, ntrcr, ntrcr_dim, nt_qsno, nslyr)
    fo, fg = util.clone(f), util.clone(f)
    orc.evp(d, orc.make_params(3600.0, 30, xmin), fo)
    s = dyn.EvpDynamics(d, fg, ndte=30, xmin=xmin)
    s.init_evp(3600.0)
    s.evp(3600.0)
    so = [a.copy() for a in state]; sg = [a.copy() for a in state]
    dt = 0.4 * xmin / max(np.abs(fo["uvel"]).max(), np.abs(fo["vvel"]).max())
    shift = 330.0 * 3.34e5
    assert orc.transport_remap_state(d, dt, fo, *so, ntrcr, nt_qsno, nslyr, shift, *tables) == 0
    s.ctx.remap_init(f["dxu"], f["dyu"], f["hm"])
    assert s.ctx.transport_remap_state(dt, *sg, ntrcr, nt_qsno, nslyr, shift, *tables) == 0
    s.close()
    every = util.cell_mask(d, "all")
    for name, a, b_, a0 in zip(("aice0", "aicen", "vicen", "vsnon", "trcrn"), sg, so, state):
        m = every if a.ndim == 3 else (every[:, None] if a.ndim == 4 else every[:, None, None])
        m = np.broadcast_to(m, a.shape)
        assert np.array_equal(a[m], b_[m]), (name, int((a[m] != b_[m]).sum()))
        assert np.abs(b_ - a0).max() > 0
    assert np.array_equal(sg[4][:, :, ntrcr:], state[4][:, :, ntrcr:]) and (state[4][:, :, ntrcr:] == 777.0).any()


def test_full_size_3600x2700_tripole_rows_f3_f4():
    """BASELINE config 5's grid and boundary for the rows that follow the EVP path, whole grid against the oracle:
    (a) eap(dt), ndte = 12 (stepa after subcycles 1 and 11, the LAST launch): velocities, stresses, structure tensor, history;
    (b) horizontal_remap with the velocities that eap left on the device (2 categories, tracer types 1 and 2), bit for bit, and
        the size-independent properties at this size: area and area x tracer integrals conserved, tracers within their old range."""
    from cice5_amd.eap_tables import eap_tables
    nx, ny = 3600, 2700
    case, d, f = util.make_case(nx, ny, 450, 2700, ns="tripole", land="continents", dt=450.0)
    synth.add_eap_state(f)
    synth.add_remap_grid(case, d, f)
    xmin = synth.global_min_dx(case)
    T = eap_tables()
    fo, fg = util.clone(f), util.clone(f)
    orc.eap(d, orc.make_params(450.0, 12, xmin), fo, T)
    s = dyn.EvpDynamics(d, fg, ndte=12, xmin=xmin)
    s.init_eap(450.0, T)
    s.eap(450.0)
    bad = util.compare(d, fg, fo)
    ne = util.cell_mask(d, "ne")
    for n in EAP_NE:
        if not np.array_equal(fg[n][ne], fo[n][ne]):
            bad.append((n, int((fg[n][ne] != fo[n][ne]).sum())))
    assert not bad, bad[:8]
    assert np.abs(fo["a11_1"] - 0.5).max() > 1e-3 and np.abs(fo["uvel"]).max() > 1e-3
    # (b)
    ncat = 2
    tables = orc.remap_tables([0, 1])                               # hice, hsno, an area tracer, one on the ice volume
    ntrace = len(tables[0])
    aice = f["aice"]
    mm = np.zeros((d.nblocks, ncat + 1) + aice.shape[1:])
    tm = np.zeros((d.nblocks, ncat, ntrace) + aice.shape[1:])
    for n in range(ncat):
        a = aice * (0.4 + 0.2 * n)
        mm[:, n + 1] = a
        for k in range(ntrace):
            tm[:, n, k] = np.where(a > 0, (k + 1.0) * (0.5 + 0.3 * aice) * (1.0 + 0.1 * n), 0.0)
    mm[:, 0] = (1.0 - aice) * (f["tmask"] > 0)
    umax = max(np.abs(fo["uvel"]).max(), np.abs(fo["vvel"]).max())
    dt = 0.3 * xmin / umax
    mo, to, mg, tg = mm.copy(), tm.copy(), mm.copy(), tm.copy()
    assert orc.horizontal_remap(d, dt, fo, mo, to, *tables) == 0
    s.ctx.remap_init(f["dxu"], f["dyu"], f["hm"])
    assert s.ctx.transport_remap(dt, mg, tg, *tables) == 0
    s.close()
    assert np.array_equal(mg, mo) and np.array_equal(tg, to)
    phys = util.cell_mask(d, "phys")
    ta = f["tarea"]
    assert np.abs(mg - mm).max() > 1e-4
    for n in range(ncat + 1):
        a0, a1 = (mm[:, n] * ta)[phys].sum(), (mg[:, n] * ta)[phys].sum()
        assert abs(a1 - a0) <= 1e-11 * abs(a0), n
    for n in range(ncat):
        q0, q1 = (mm[:, n + 1] * tm[:, n, 0] * ta)[phys].sum(), (mg[:, n + 1] * tg[:, n, 0] * ta)[phys].sum()       # ice volume
        assert abs(q1 - q0) <= 1e-10 * abs(q0), n
        ice = phys & (mg[:, n + 1] > 1e-9)
        for k in (0, 2):                                                                                        # type-1 tracers stay in range
            lo, hi = tm[:, n, k][tm[:, n, k] > 0].min(), tm[:, n, k].max()
            assert tg[:, n, k][ice].min() >= lo * (1 - 1e-9) and tg[:, n, k][ice].max() <= hi * (1 + 1e-9), (n, k)
