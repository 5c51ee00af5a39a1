"""Pins from the REFERENCE ITSELF: tests/golden/ref_*.npz hold what the reference's own routines -- compiled unmodified from
/root/reference by oracle/ref/Makefile, driven by oracle/ref/ref_harness.F90 -- return on the inputs of tests/golden/refvec.py
(generator: tests/golden/make_ref_golden.py).  Here the host mirror (cice5_amd/blocks.py) and the C restatement (oracle/) are
held against them, bit for bit; tests/test_ref_pins_gpu.py does the same for the HIP kernels.

Covered (the ghost-cell part of f-3):
  create_blocks (ice_blocks.F90:111), create_distribution cartesian (ice_distribution.F90:535) incl. land-block elimination
  (ice_domain.F90:387-441), ice_HaloUpdate 2DR8 / 3DR8 / 2DI4 for every field location x field type on cyclic / open /
  closed / tripole grids in 1 and 16 (padded) blocks, ice_HaloUpdate_stress, bound_state (ice_state.F90:173),
  ice_strength (ice_mechred.F90:2111), global_minval.
"""
from __future__ import annotations

import os

import numpy as np
import pytest

from cice5_amd import blocks, constants as C
from oracle import orc
from tests.golden import refvec as rv

HERE = os.path.dirname(os.path.abspath(__file__))


def load(cfg):
    return np.load(os.path.join(HERE, "golden", f"ref_{cfg}.npz"))


def cases(cfg, z):
    nbt = ((rv.CONFIGS[cfg][0] - 1) // rv.CONFIGS[cfg][2] + 1) * ((rv.CONFIGS[cfg][1] - 1) // rv.CONFIGS[cfg][3] + 1)
    for ew, ns, land in rv.BOUNDARIES:
        if land == "landblock" and nbt == 1:
            continue
        case = rv.case_name(ew, ns, land)
        assert f"{case}/aborted" not in z.files, f"the reference aborted on {cfg} {case}: regenerate with a valid case"
        yield ew, ns, land, case


def decomp(cfg, z, ew, ns, case, nprocs=1, rank=0, shape="slenderX1"):
    nx, ny, bx, by, _ = rv.CONFIGS[cfg]
    loc = z[f"{case}/blocks/blockLocation"]
    return blocks.create_distrb_cart(nx, ny, bx, by, nprocs=nprocs, rank=rank, ew_boundary_type=ew, ns_boundary_type=ns,
                                     processor_shape=shape, work_per_block=(loc != 0).astype(int))


@pytest.mark.parametrize("cfg", list(rv.CONFIGS))
def test_create_blocks_equals_reference(cfg):
    """every member of the reference's `block` type (ice_blocks.F90:22-35) for every block, all boundary types"""
    z = load(cfg)
    nx, ny, bx, by, _ = rv.CONFIGS[cfg]
    n = 0
    for ew, ns, land, case in cases(cfg, z):
        bl = blocks.create_blocks(nx, ny, bx, by, ew, ns)
        desc, ig, jg = z[f"{case}/blocks/desc"], z[f"{case}/blocks/i_glob"], z[f"{case}/blocks/j_glob"]
        assert len(bl) == len(desc)
        for b, d, i_glob, j_glob in zip(bl, desc, ig, jg):
            assert (b.block_id, b.iblock, b.jblock, b.ilo, b.ihi, b.jlo, b.jhi, int(b.tripole)) == tuple(d), (case, d)
            assert np.array_equal(b.i_glob, i_glob), (case, d, b.i_glob, i_glob)
            assert np.array_equal(b.j_glob, j_glob), (case, d, b.j_glob, j_glob)
            n += 1
    assert n >= 7


@pytest.mark.parametrize("cfg", [c for c in rv.CONFIGS if rv.CONFIGS[c][4] > 1])
def test_cartesian_distribution_equals_reference(cfg):
    """blockLocation / blockLocalID of create_distrb_cart for 1, 2, 3, 4, 8 ranks, slenderX1 / slenderX2, with and without
    eliminated land blocks; and the local block list (create_local_block_ids)"""
    z = load(cfg)
    nx, ny, bx, by, _ = rv.CONFIGS[cfg]
    for ew, ns, land, case in cases(cfg, z):
        d = decomp(cfg, z, ew, ns, case)
        assert np.array_equal(d.block_location, z[f"{case}/blocks/blockLocation"])
        assert [b.block_id for b in d.local_blocks] == list(z[f"{case}/blocks/blocks_ice"])
        lid = np.zeros(len(d.all_blocks), dtype=np.int32)
        for b in d.local_blocks:
            lid[b.block_id - 1] = b.local_id
        assert np.array_equal(lid, z[f"{case}/blocks/blockLocalID"])
        for nprocs, shape in rv.DISTRIBUTIONS:
            want_loc, want_lid = z[f"{case}/distrb/{nprocs}_{shape}/blockLocation"], z[f"{case}/distrb/{nprocs}_{shape}/blockLocalID"]
            got_lid = np.zeros_like(want_lid)
            for r in range(nprocs):
                dr = decomp(cfg, z, ew, ns, case, nprocs=nprocs, rank=r, shape=shape)
                assert np.array_equal(dr.block_location, want_loc), (case, nprocs, shape)
                for b in dr.local_blocks:
                    got_lid[b.block_id - 1] = b.local_id
            assert np.array_equal(got_lid, want_lid), (case, nprocs, shape)


def mpi_semantics(serial_out, inp, fill):
    """The fixtures come from the reference's SERIAL backend (serial/ice_boundary.F90), the only one that builds here; the
    production backend (mpi/ice_boundary.F90), which oracle and kernels follow, differs in ONE documented way: before the
    copies it overwrites the outermost nghost rows / columns of every block array with the fill value (mpi/ice_boundary.F90:
    1409-1416 "fill out halo region ... for halo grid cells that are not updated"), where the serial code leaves a ghost
    cell that no message writes (open / closed boundary) as it was.  Every input cell carries its own random value, so a
    cell the serial update left alone is one whose output equals its input."""
    e = serial_out.copy()
    ring = np.zeros(serial_out.shape, dtype=bool)
    ring[..., 0, :] = ring[..., -1, :] = ring[..., :, 0] = ring[..., :, -1] = True
    e[ring & (serial_out == inp)] = fill
    return e


def _ghost_report(d, got, want, inp):
    bad = np.argwhere(~((got == want) | (np.isnan(got) & np.isnan(want))))
    return [(tuple(int(x) for x in k), float(got[tuple(k)]), float(want[tuple(k)]), float(inp[tuple(k)])) for k in bad[:6]], len(bad)


@pytest.mark.parametrize("cfg", list(rv.CONFIGS))
def test_halo_update_r8_equals_reference(cfg):
    """orc_halo_r8 == ice_HaloUpdate2DR8 / 3DR8 (serial/ice_boundary.F90:630, :1440) on every cell of every block"""
    z = load(cfg)
    checked = 0
    for ew, ns, land, case in cases(cfg, z):
        d = decomp(cfg, z, ew, ns, case)
        for key, nz, loc, typ, fill in rv.HALO_R8:
            inp = rv.halo_r8_input(cfg, case, key, d.nblocks, d.ny_block, d.nx_block, nz)
            want = mpi_semantics(z[f"{case}/halo_r8/{key}"], inp, 0.0 if fill is None else fill)
            got = inp.copy()
            kind = C.KIND_VECTOR if typ in (rv.TYPE["vector"], rv.TYPE["angle"]) else C.KIND_SCALAR
            if nz:
                for k in range(nz):
                    w = np.ascontiguousarray(got[:, k]); orc.halo_r8(d, w, loc, kind, 0.0 if fill is None else fill); got[:, k] = w
            else:
                orc.halo_r8(d, got, loc, kind, 0.0 if fill is None else fill)
            rep, nbad = _ghost_report(d, got, want, inp)
            assert nbad == 0, (cfg, case, key, nbad, rep)
            checked += 1
    assert checked >= 7 * len(rv.HALO_R8)


@pytest.mark.parametrize("cfg", list(rv.CONFIGS))
def test_halo_update_i4_and_stress_equal_reference(cfg):
    """orc_halo_i4 == ice_HaloUpdate2DI4 (:1170), orc_halo_stress == ice_HaloUpdate_stress (:3269)"""
    z = load(cfg)
    for ew, ns, land, case in cases(cfg, z):
        d = decomp(cfg, z, ew, ns, case)
        for key, loc, typ, fill in rv.HALO_I4:
            inp = rv.halo_i4_input(cfg, case, key, d.nblocks, d.ny_block, d.nx_block)
            got = inp.copy()
            orc.halo_i4(d, got, 0 if fill is None else fill)
            want = mpi_semantics(z[f"{case}/halo_i4/{key}"], inp, 0 if fill is None else fill)
            assert np.array_equal(got, want), (cfg, case, key, np.argwhere(got != want)[:5])
        a1 = rv.halo_r8_input(cfg, case, "stress1", d.nblocks, d.ny_block, d.nx_block, 0)
        a2 = rv.halo_r8_input(cfg, case, "stress2", d.nblocks, d.ny_block, d.nx_block, 0)
        got = a1.copy()
        orc.halo_stress(d, got, a2)
        want = z[f"{case}/halo_stress/center_scalar"]
        rep, nbad = _ghost_report(d, got, want, a1)
        assert nbad == 0, (cfg, case, nbad, rep)
        if ns != "tripole" and land not in ("landblock",) and not (cfg == "g26x18_b8x5" and ew == "closed"):
            assert np.array_equal(want, a1)          # without eliminated land blocks the stress update touches tripole grids only


@pytest.mark.parametrize("cfg", list(rv.CONFIGS))
def test_bound_state_equals_reference(cfg):
    """bound_state (ice_state.F90:173-238) == a centre / scalar halo update of every category and tracer plane"""
    z = load(cfg)
    n = 0
    for ew, ns, land, case in cases(cfg, z):
        if f"{case}/bound/aicen" not in z.files:
            continue
        d = decomp(cfg, z, ew, ns, case)
        mxb = rv.CONFIGS[cfg][4]
        aicen, vicen, vsnon, trcrn = rv.state_input(cfg, case, mxb, d.ny_block, d.nx_block, 3)
        assert bool(z[f"{case}/bound/trcrn_beyond_ntrcr_untouched"])
        for name, arr in (("aicen", aicen), ("vicen", vicen), ("vsnon", vsnon), ("trcrn", trcrn[:, :, :3].reshape(mxb, -1, d.ny_block, d.nx_block))):
            got = np.ascontiguousarray(arr[:d.nblocks]).copy()
            want = mpi_semantics(z[f"{case}/bound/{name}"].reshape(d.nblocks, -1, d.ny_block, d.nx_block), got, 0.0)
            for k in range(got.shape[1]):
                w = np.ascontiguousarray(got[:, k]); orc.halo_r8(d, w, C.LOC_CENTER, C.KIND_SCALAR, 0.0); got[:, k] = w
            assert np.array_equal(got, want), (cfg, case, name, np.argwhere(got != want)[:5])
            n += 1
    assert n >= 8


def test_ice_strength_equals_reference():
    """orc_ice_strength == ice_strength (ice_mechred.F90:2111-2269; asum_ridging, ridge_itd) for Rothrock with both
    participation / redistribution functions and for Hibler's formula.  The Fortran evaluates exp() with the compiler's
    intrinsic, the restatement (and the kernel) with the fixed algorithm of orc_exp (< 1 ulp): wherever exp enters
    (krdg_partic = 1, krdg_redist = 1, kstrength = 0) the result may differ in the last bits; everything else is bit-exact."""
    cfg = "g24x16_b24x16"
    z = load(cfg)
    nx, ny, bx, by, _ = rv.CONFIGS[cfg]
    nxb, nyb = bx + 2, by + 2
    case = "cyclic_open"
    report = {}
    for ks, kp, kr in rv.STRENGTH_CASES:
        for rep in (0, 1):
            tag = f"k{ks}{kp}{kr}_{rep}"
            s = rv.strength_input(cfg, tag, nyb, nxb)
            want = z[f"{case}/strength/{tag}"]
            p = orc.make_params(3600.0, 120, 1.0e4, strength_mode=1, kstrength=ks, krdg_partic=kp, krdg_redist=kr, ncat=rv.NCAT,
                                mu_rdg=rv.MU_RDG, Cf=rv.CF)
            got = orc.ice_strength_block(nxb, nyb, 2, nxb - 1, 2, nyb - 1, s["indxi"], s["indxj"], s["aice"], s["vice"], s["aice0"],
                                         s["aicen"], s["vicen"], p)
            assert len(s["indxi"]) > 200
            if ks == 1:
                on = np.zeros((nyb, nxb), dtype=bool); on[s["indxj"] - 1, s["indxi"] - 1] = True
                assert np.all(want[~on] == 0.0) and np.all(got[~on] == 0.0)
            else:
                on = np.zeros((nyb, nxb), dtype=bool); on[1:-1, 1:-1] = True
            assert np.all(np.isfinite(want))
            nz = want[on] != 0
            assert nz.sum() > 50, (tag, nz.sum())
            rel = np.abs(got[on] - want[on]) / np.maximum(np.abs(want[on]), 1e-300)
            report[tag] = (int((got[on] != want[on]).sum()), int(on.sum()), float(rel.max()))
            uses_exp = (ks == 0) or kp == 1 or kr == 1
            if not uses_exp:
                assert np.array_equal(got, want), (tag, report[tag])
            else:
                assert rel.max() <= 4.5e-16, (tag, report[tag])      # 2 ulp
    print("ice_strength vs reference (cells differing, cells, max rel diff):", report)


@pytest.mark.parametrize("cfg", list(rv.CONFIGS))
def test_global_minval_equals_reference(cfg):
    """set_evp_parameters' xmin = global_minval(dxt, distrb_info, tmask) (ice_dyn_shared.F90:221): physical cells only"""
    z = load(cfg)
    for ew, ns, land, case in cases(cfg, z):
        d = decomp(cfg, z, ew, ns, case)
        dx = 1000.0 * (1.0 + rv.halo_r8_input(cfg, case, "minval", d.nblocks, d.ny_block, d.nx_block, 0) ** 2)
        msk = rv.halo_i4_input(cfg, case, "minval_mask", d.nblocks, d.ny_block, d.nx_block) % 10 > 3
        phys = np.zeros_like(msk)
        for n, b in enumerate(d.local_blocks):
            phys[n, b.jlo - 1:b.jhi, b.ilo - 1:b.ihi] = True
        assert dx[msk & phys].min() == float(z[f"{case}/global_minval"])


def test_compute_tracers_equals_reference():
    """orc_compute_tracers == compute_tracers (ice_itd.F90:1359-1501, the arithmetic of work_to_state in transport_upwind):
    tracers on the area, the ice and snow volumes, on the level-ice / pond / brine fractions; Tsfc takes Tocnfrz and fbri 1
    where there is no ice, bit for bit."""
    import ctypes as ct
    cfg = "g24x16_b24x16"
    z = load(cfg)
    nx, ny, bx, by, _ = rv.CONFIGS[cfg]
    nxb, nyb = bx + 2, by + 2
    L = orc.lib()
    for tag, (dep, n_tsfc, n_alvl, n_apnd, n_fbri, pond) in rv.TRACER_CASES.items():
        nt = len(dep)
        a, v, sn, atr = rv.tracers_input(cfg, tag, nyb, nxb, nt)
        want = z[f"cyclic_open/tracers/{tag}"]
        got = np.full((nt, nyb, nxb), 7.0)
        dp = np.asarray(dep, dtype=np.int32)
        L.orc_compute_tracers.argtypes = [ct.c_int] * 3 + [orc.c_i32p] + [ct.c_int] * 7 + [ct.c_double] + [orc.c_f64p] * 5
        L.orc_compute_tracers(nxb, nyb, nt, orc._p32(dp), n_tsfc, n_alvl, n_apnd, n_fbri, *pond, rv.TOCNFRZ,
                              orc._p64(atr), orc._p64(a), orc._p64(v), orc._p64(sn), orc._p64(got))
        assert np.array_equal(got, want), (tag, np.argwhere(got != want)[:5])
        assert (want[0] == rv.TOCNFRZ).any() and np.abs(want).max() > 1.0
