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
