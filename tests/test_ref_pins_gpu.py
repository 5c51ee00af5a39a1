"""The HIP path against the REFERENCE'S OWN outputs (tests/golden/ref_*.npz: the reference's routines compiled unmodified,
oracle/ref/Makefile; inputs tests/golden/refvec.py) -- the device counterparts of tests/test_ref_pins.py:

  * the halo / tripole-fold kernels and the exchange machinery behind them, through the C ABI entry evpk_halo_update /
    evpk_halo_update_stress, for every field location x field type the reference updates, on cyclic / open / closed /
    tripole grids in 1 and 16 (padded) blocks, with an eliminated land block -- one rank, the forced-exchange path, and
    2 - 4 x-slab ranks;
  * k_ice_strength (evpk_run with strength == NULL) against ice_strength;
  * bound_state inside evpk_transport_remap_state's scatter against bound_state.
"""
from __future__ import annotations

import os
import sys
import traceback
import uuid

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ctx(cfg, z, ew, ns, case, nprocs=1, rank=0, unique_id=None):
    from cice5_amd import constants as C, evpk, synth
    from tests import test_ref_pins as P
    from tests.golden import refvec as rv
    nx, ny, bx, by, _ = rv.CONFIGS[cfg]
    d = P.decomp(cfg, z, ew, ns, case, nprocs=nprocs, rank=rank)
    sc = synth.SynthCase(nx=nx, ny=ny, ns_boundary=C.BND_NAMES[ns], ew_boundary=C.BND_NAMES[ew], land="none")
    f = synth.make_block_fields(sc, d)
    return d, evpk.Context(d, f, device=0, unique_id=unique_id)


def _halo_case(cfg, z, ew, ns, land, case, d, ctx, rows=None):
    """every halo update of the fixture through the device; rows: this rank's blocks as indices into the fixture's block list"""
    from tests import test_ref_pins as P
    from tests.golden import refvec as rv
    bad = []
    nb = len(z[f"{case}/blocks/blocks_ice"])
    sel = (lambda a: a) if rows is None else (lambda a: np.ascontiguousarray(a[rows]))
    for key, nz, loc, typ, fill in rv.HALO_R8:
        inp = sel(rv.halo_r8_input(cfg, case, key, nb, d.ny_block, d.nx_block, nz))
        want = P.mpi_semantics(sel(z[f"{case}/halo_r8/{key}"]), inp, 0.0 if fill is None else fill)
        got = inp.copy()
        ctx.halo_update(got, loc, typ, 0.0 if fill is None else fill)
        if not np.array_equal(got, want):
            k = np.argwhere(got != want)
            bad.append((case, key, len(k), [(tuple(int(v) for v in q), float(got[tuple(q)]), float(want[tuple(q)]), float(inp[tuple(q)])) for q in k[:4]]))
    a1 = sel(rv.halo_r8_input(cfg, case, "stress1", nb, d.ny_block, d.nx_block, 0))
    a2 = sel(rv.halo_r8_input(cfg, case, "stress2", nb, d.ny_block, d.nx_block, 0))
    got = a1.copy()
    ctx.halo_update_stress(got, a2)
    want = sel(z[f"{case}/halo_stress/center_scalar"])
    if not np.array_equal(got, want):
        k = np.argwhere(got != want)
        bad.append((case, "stress", len(k), [(tuple(int(v) for v in q), float(got[tuple(q)]), float(want[tuple(q)]), float(a1[tuple(q)])) for q in k[:4]]))
    return bad


@pytest.mark.parametrize("force", ["", "1"])
@pytest.mark.parametrize("cfg", ["g24x16_b24x16", "g24x16_b6x4", "g26x18_b8x5"])
def test_device_halo_update_equals_reference(cfg, force, monkeypatch):
    """evpk_halo_update == ice_HaloUpdate 2DR8 / 3DR8, evpk_halo_update_stress == ice_HaloUpdate_stress, every cell of every
    block; force = "1": the single rank takes the multi-rank code path (pack -> copies in place of the transport -> unpack)"""
    from tests import test_ref_pins as P
    if force:
        monkeypatch.setenv("EVPK_FORCE_EXCHANGE", force)
    z = P.load(cfg)
    n = 0
    for ew, ns, land, case in P.cases(cfg, z):
        d, ctx = _ctx(cfg, z, ew, ns, case)
        try:
            bad = _halo_case(cfg, z, ew, ns, land, case, d, ctx)
        finally:
            ctx.close()
        assert not bad, bad[:3]
        n += 1
    assert n >= 7


def _worker(rank, world, tag, xp, cfg, q):
    try:
        sys.path.insert(0, ROOT)
        from tests import test_ref_pins as P
        z = P.load(cfg)
        uid = ({"shm": b"EVPKSHM:", "ipc": b"EVPKIPC:"}[xp] + tag.encode()).ljust(128, b"\0")
        bad = []
        for ew, ns, land, case in P.cases(cfg, z):
            if any(P.decomp(cfg, z, ew, ns, case, nprocs=world, rank=r).nblocks == 0 for r in range(world)):
                continue            # a rank whose whole share is eliminated land (closed E-W rim): every rank must own a block
            d, ctx = _ctx(cfg, z, ew, ns, case, nprocs=world, rank=rank, unique_id=uid)
            ice = list(z[f"{case}/blocks/blocks_ice"])
            rows = [ice.index(b.block_id) for b in d.local_blocks]
            try:
                bad += _halo_case(cfg, z, ew, ns, land, case, d, ctx, rows=rows)
            finally:
                ctx.close()
        q.put((rank, bad[:4]))
    except Exception:
        q.put((rank, ["EXC " + traceback.format_exc()]))


@pytest.mark.parametrize("world,xp,cfg", [(2, "ipc", "g24x16_b6x4"), (4, "shm", "g24x16_b6x4"), (2, "shm", "g26x18_b8x5"), (4, "ipc", "g26x18_b8x5")])
def test_device_halo_update_across_slabs_equals_reference(world, xp, cfg):
    """the same on 2 / 4 x-slab ranks (one process each, sharing the GPU): the E-W ring and the fold with the mirror ranks
    deliver what the reference's one-task run has in every block"""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    tag = "evpk_p_" + uuid.uuid4().hex[:12]
    procs = [ctx.Process(target=_worker, args=(r, world, tag, xp, cfg, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in procs:
            res.append(q.get(timeout=600))
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
        try:
            os.unlink("/dev/shm/" + tag)
        except OSError:
            pass
    for rank, bad in res:
        assert not bad, f"rank {rank}: {bad}"


def test_device_ice_strength_equals_reference():
    """k_ice_strength (evpk_run with strength == NULL) == the reference's ice_strength (ice_mechred.F90:2111-2269) on the
    cells both evaluate: bit for bit where no exp() enters (krdg_partic = 0), within 2 ulp where one does -- the kernel's
    exp is a fixed algorithm (csrc/evpk_fmath.h), the Fortran's the compiler's intrinsic (tests/test_ref_pins.py)."""
    from cice5_amd import constants as C, dyn, synth
    from tests import test_ref_pins as P, util
    from tests.golden import refvec as rv
    cfg = "g24x16_b24x16"
    z = P.load(cfg)
    nx, ny, bx, by, _ = rv.CONFIGS[cfg]
    nxb, nyb = bx + 2, by + 2
    case, d, f0 = util.make_case(nx, ny, bx, by, land="none")
    xmin = synth.global_min_dx(case)
    report = {}
    for ks, kp, kr in rv.STRENGTH_CASES:
        for rep in (0, 1):
            tag = f"k{ks}{kp}{kr}_{rep}"
            s = rv.strength_input(cfg, tag, nyb, nxb)
            want = z[f"cyclic_open/strength/{tag}"]
            f = util.clone(f0)
            f["tmask"][...] = 1; f["umask"][...] = 1
            for k in ("aice", "vice", "aice0"):
                f[k] = np.ascontiguousarray(s[k][None])
            f["aicen"], f["vicen"] = np.ascontiguousarray(s["aicen"][None]), np.ascontiguousarray(s["vicen"][None])
            f["vsno"] = 0.1 * f["aice"]
            f["aice_init"] = f["aice"].copy()
            f["strength"][...] = -7.0
            sw = dict(kstrength=ks, krdg_partic=kp, krdg_redist=kr)
            solver = dyn.EvpDynamics(d, f, ndte=2, xmin=xmin, device_strength=sw)
            solver.init_evp(3600.0)
            solver.evp(3600.0)
            solver.close()
            listed = np.zeros((nyb, nxb), dtype=bool)
            if ks == 1:
                listed[s["indxj"] - 1, s["indxi"] - 1] = True
            else:
                listed[1:-1, 1:-1] = True                       # Hibler's formula runs over ilo..ihi, jlo..jhi
            both = listed & (f["icetmask"][0] == 1) if ks == 1 else listed
            got = f["strength"][0]
            assert both.sum() > 150, (tag, int(both.sum()))
            rel = np.abs(got[both] - want[both]) / np.maximum(np.abs(want[both]), 1e-300)
            report[tag] = (int((got[both] != want[both]).sum()), int(both.sum()), float(rel.max()))
            if kp == 0 and ks == 1:
                assert np.array_equal(got[both], want[both]), (tag, report[tag])
            else:
                assert rel.max() <= 4.5e-16, (tag, report[tag])
    print("device ice_strength vs reference (cells differing, cells, max rel diff):", report)


@pytest.mark.parametrize("cfg", ["g24x16_b6x4", "g26x18_b8x5"])
def test_device_bound_state_equals_reference(cfg):
    """bound_state inside evpk_transport_remap_state's scatter (k_state_scatter after the ghost-ring update of the new areas and
    tracers) against the reference's bound_state (ice_state.F90:173-238).  The fixture says which physical cell every ghost cell
    of every block takes its value from (every input cell carries its own value); the device must deliver exactly that
    cell's NEW value there -- with zero velocity the new state is tracers_to_state(state_to_tracers(old)), which differs
    from the old one in the last bits, so a ghost cell that merely kept its input would show."""
    from cice5_amd import constants as C, evpk, synth
    from oracle import orc
    from tests import test_ref_pins as P
    from tests.golden import refvec as rv
    z = P.load(cfg)
    nx, ny, bx, by, mxb = rv.CONFIGS[cfg]
    ntrcr, checked = 3, 0
    for ew, ns, land, case in P.cases(cfg, z):
        if f"{case}/bound/vicen" not in z.files:
            continue
        d = P.decomp(cfg, z, ew, ns, case)
        nb, nyb, nxb = d.nblocks, d.ny_block, d.nx_block
        a_in, v_in, s_in, t_in = [np.ascontiguousarray(x[:nb]) for x in rv.state_input(cfg, case, mxb, nyb, nxb, ntrcr)]
        ref = {k: P.mpi_semantics(z[f"{case}/bound/{k}"].reshape(nb, -1, nyb, nxb), x.reshape(nb, -1, nyb, nxb)[:, :z[f"{case}/bound/{k}"].reshape(nb, -1, nyb, nxb).shape[1]], 0.0)
               for k, x in (("aicen", a_in), ("vicen", v_in), ("vsnon", s_in))}
        # source of every ghost cell, from category 1 of vicen (unique values)
        phys = np.zeros((nb, nyb, nxb), dtype=bool)
        for n, b in enumerate(d.local_blocks):
            phys[n, b.jlo - 1:b.jhi, b.ilo - 1:b.ihi] = True
        vin0, vout0 = v_in[:, 0], ref["vicen"][:, 0]
        where = {float(vin0[k]): k for k in map(tuple, np.argwhere(phys))}
        src = {}
        for k in map(tuple, np.argwhere(~phys)):
            if vout0[k] != vin0[k] and float(vout0[k]) in where:
                src[k] = where[float(vout0[k])]
        assert len(src) > 0.8 * (~phys).sum() * (0.5 if "landblock" in case or cfg == "g26x18_b8x5" else 1.0), (case, len(src))
        # the device: the reference's halo-updated state as input (ghost cells current, as the entry point requires)
        sc = synth.SynthCase(nx=nx, ny=ny, ns_boundary=C.BND_NAMES[ns], ew_boundary=C.BND_NAMES[ew], land="none")
        f = synth.make_block_fields(sc, d)
        synth.add_remap_grid(sc, d, f)
        f["uvel"][...] = 0.0; f["vvel"][...] = 0.0
        aicen = np.ascontiguousarray(ref["aicen"]); vicen = np.ascontiguousarray(ref["vicen"]); vsnon = np.ascontiguousarray(ref["vsnon"])
        trcrn = np.ascontiguousarray(np.concatenate([z[f"{case}/bound/trcrn"], t_in[:, :, ntrcr:ntrcr + 1]], axis=2))
        for q in range(nb):                                            # (the fixture's trcrn went through the serial backend: ring -> 0)
            trcrn[q] = P.mpi_semantics(trcrn[q], np.concatenate([t_in[q, :, :ntrcr], t_in[q, :, ntrcr:ntrcr + 1]], axis=1), 0.0)
        aice0 = np.ascontiguousarray(1.0 - aicen.sum(axis=1))
        before = [x.copy() for x in (aicen, vicen, vsnon, trcrn)]
        ctx = evpk.Context(d, f, device=0)
        try:
            from cice5_amd import dyn
            ctx.set_params(dyn.set_evp_parameters(3600.0, 2, False, synth.global_min_dx(sc)))
            ctx.upload(f)
            ctx.remap_init(f["dxu"], f["dyu"], f["hm"])
            tables = orc.remap_tables([0, 1, 2])
            rc = ctx.transport_remap_state(600.0, aice0, aicen, vicen, vsnon, trcrn, ntrcr, 3, 1, 330.0 * 3.34e5, *tables)
        finally:
            ctx.close()
        assert rc == 0
        changed = sum(int((x != y).sum()) for x, y in zip((vicen, vsnon, trcrn), before[1:]))
        assert changed > 100, changed                                   # the round trip did move last bits
        for g, s in src.items():
            for name, arr in (("aicen", aicen), ("vicen", vicen), ("vsnon", vsnon)):
                assert np.array_equal(arr[g[0], :, g[1], g[2]], arr[s[0], :, s[1], s[2]]), (case, name, g, s)
            assert np.array_equal(trcrn[g[0], :, :ntrcr, g[1], g[2]], trcrn[s[0], :, :ntrcr, s[1], s[2]]), (case, "trcrn", g, s)
            assert aice0[g] == aice0[s], (case, "aice0", g, s)
        assert np.array_equal(trcrn[:, :, ntrcr:], before[3][:, :, ntrcr:])          # tracers not in use stay
        checked += 1
    assert checked >= 2
