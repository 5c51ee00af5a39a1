"""Real multi-rank runs of libevpk on ONE GPU: K processes, each with its own x-slab and context, exchanging through
  * "ipc": the peer-mapped transport (unique id "EVPKIPC:<name>") -- the ranks map each other's receive buffers with
    hipIpcOpenMemHandle, pack kernels store straight into the neighbour's buffer, completion by flags the consumer's
    stream waits on; between GPUs the same stores go over xGMI, here all ranks share the one GPU;
  * "shm": the host-staged shared-memory relay (unique id "EVPKSHM:<name>") with the semantics of the RCCL
    point-to-point calls, which refuse several ranks on one device.
Everything else is the production multi-rank path: slabs with i0 > 1, edge-column / ghost-zone exchange, the edge-first
overlap on two streams, the tripole fold (mirror slab, or the packed exchange with the mirror ranks).  Each rank compares its
slab with the single-process oracle.

On a node with SEVERAL GPUs (none was available to the builder: the cross-device visibility of the peer-mapped buffers, the
flags and RCCL itself are unverified) the same cases run one rank per device BY THEMSELVES -- a case whose world size fits
torch.cuda.device_count() puts rank r on device r -- and the parametrised cases also run over RCCL ("rccl": rank 0 makes the
ncclUniqueId, the harness hands it to the other ranks; refused by RCCL when two ranks share a device, so those cases are skipped
where the devices do not suffice).  EVPK_TEST_DEVICES=shared|per_rank and EVPK_TEST_XPS=ipc,shm,rccl override both choices."""
import os
import sys
import traceback
import uuid

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _wander(d, base, ff, call, ns):
    """Inputs of call `call`: the ice cover of `base` under a moving smooth mask of the global cell index (so every
    decomposition and every ghost cell sees the same field; ghost cells wrap E-W and mirror across a tripole fold)."""
    from cice5_amd import blocks
    I, J = blocks.block_index_windows(d)
    nxg, nyg = d.nx_global, d.ny_global
    rng = np.random.default_rng(77 + call)
    kx, ky, ph, thin = rng.uniform(0.02, 0.09), rng.uniform(0.03, 0.12), rng.uniform(0, 6.28), rng.uniform(0.0, 1.0)
    for n in range(d.nblocks):
        Ig = np.broadcast_to(((I[n] - 1) % nxg + 1)[None, :], (d.ny_block, d.nx_block)).copy()
        Jg = np.broadcast_to(J[n][:, None], (d.ny_block, d.nx_block)).copy()
        if ns == "tripole":
            top = Jg > nyg
            Ig[top] = nxg - Ig[top] + 1
            Jg = np.minimum(Jg, nyg)
        w = np.sin(kx * Ig + ph) * np.cos(ky * Jg - 0.5 * ph) + 0.35 * np.sin(0.7 * call)
        keep = np.where(w > 0.0, 1.0, 0.0) * np.where(w > 0.6, 1.0, thin * 0.01 + 0.001)
        for name in ("aice", "vice", "vsno", "aice_init", "strength"):
            ff[name][n] = base[name][n] * keep
        ff["strairxT"][n] = base["strairxT"][n] * np.cos(0.5 * call) - base["strairyT"][n] * np.sin(0.5 * call)
        ff["strairyT"][n] = base["strairyT"][n] * np.cos(0.5 * call) + base["strairxT"][n] * np.sin(0.5 * call)


def _worker(rank, world, tag, ns, nx, ny, bsx, bsy, ndte, env, q, uidq=None):
    try:
        import time
        t0 = time.time()
        sys.path.insert(0, ROOT)
        os.environ.update(env)
        os.environ["OMP_NUM_THREADS"] = "2"
        from cice5_amd import blocks, constants as C, dyn, synth
        from oracle import orc
        from tests import util

        band = tuple(int(v) for v in env["TEST_LAND_BAND"].split(",")) if env.get("TEST_LAND_BAND") else ()
        case = synth.SynthCase(nx=nx, ny=ny, ns_boundary=C.BND_NAMES[ns], land="continents", land_band=band)
        work = None
        if band:       # land-block elimination (ice_domain.F90:387-441): blocks without an ocean cell are dropped, on every rank alike
            dfull = blocks.create_distrb_cart(nx, ny, bsx, bsy, ns_boundary_type=ns)
            tm = synth.make_block_fields(case, dfull)["tmask"]
            work = [int(tm[n, b.jlo - 1:b.jhi, b.ilo - 1:b.ihi].any()) for n, b in enumerate(dfull.local_blocks)]
        d = blocks.create_distrb_cart(nx, ny, bsx, bsy, nprocs=world, rank=rank, ns_boundary_type=ns, work_per_block=work)
        f = synth.make_block_fields(case, d)
        xmin = synth.global_min_dx(case)
        xp = env.get("TEST_XP", "shm")
        dev = 0
        if env.get("TEST_PER_RANK"):
            dev = rank % max(1, _device_count())
        if xp == "rccl":                          # rank 0 made the id (see _run); every rank takes one copy from the queue
            uid = uidq.get(timeout=120)
        else:
            uid = ({"shm": b"EVPKSHM:", "ipc": b"EVPKIPC:"}[xp] + tag.encode()).ljust(128, b"\0")
        eap = bool(env.get("TEST_EAP"))
        if eap:                                   # kdyn = 2: the same slabs, exchange and fold around stress_eap / stepa
            from cice5_amd.eap_tables import eap_tables
            T = eap_tables()
            synth.add_eap_state(f)
        s = dyn.EvpDynamics(d, f, ndte=ndte, xmin=xmin, device=dev, unique_id=uid)
        if eap:
            s.init_eap(3600.0, T)
        else:
            s.init_evp(3600.0)
        # reference: whole domain, same block size, in this process
        d1 = blocks.create_distrb_cart(nx, ny, bsx, bsy, ns_boundary_type=ns, work_per_block=work)
        f1 = synth.make_block_fields(case, d1)
        if eap:
            synth.add_eap_state(f1)
        p = orc.make_params(3600.0, ndte, xmin)
        bad = []
        ncalls = int(env.get("TEST_WANDER_CALLS", "0"))
        base, base1 = util.clone(f), util.clone(f1)
        for call in range(ncalls or 2):
            if ncalls:
                _wander(d, base, f, call, ns)
                _wander(d1, base1, f1, call, ns)
            elif call:
                for ff in (f, f1):
                    ff["aice"] *= 0.97
                    ff["vice"] *= 0.97
                    ff["strairxT"], ff["strairyT"] = ff["strairyT"].copy(), -ff["strairxT"]
            if eap:
                s.eap(3600.0)
                orc.eap(d1, p, f1, T)
            else:
                s.evp(3600.0)
                orc.evp(d1, p, f1)
            ref = {}
            for n, b in enumerate(d.local_blocks):
                n1 = next(k for k, bb in enumerate(d1.local_blocks) if bb.block_id == b.block_id)
                for name in util.ALL_CELLS + util.NE_CELLS + util.PHYS_CELLS:
                    ref.setdefault(name, np.zeros_like(f[name]))[n] = f1[name][n1]
            if d.nblocks:                          # (a rank without a block has nothing to compare: it only took part in the start)
                bad += [(call,) + x for x in util.compare(d, f, ref)]
            if env.get("TEST_DEBUG_LOC") and bad:
                for name in ("uvel", "stressp_1"):
                    k = np.argwhere(f[name] != ref[name])
                    loc = [(int(d.local_blocks[b].iglob_lo + i - (d.local_blocks[b].ilo - 1)), int(d.local_blocks[b].jglob_lo + j - (d.local_blocks[b].jlo - 1))) for b, j, i in k[:40]]
                    print(f"rank {rank} call {call} {name}: {len(k)} cells differ at global (i, j): {sorted(set(loc))[:40]}", flush=True)
            if eap:
                ne = util.cell_mask(d, "ne")
                for name in (synth.EAP_STATE + synth.EAP_HISTORY) if d.nblocks else []:
                    r = np.stack([f1[name][next(k for k, bb in enumerate(d1.local_blocks) if bb.block_id == b.block_id)] for b in d.local_blocks])
                    if not np.array_equal(f[name][ne], r[ne]):
                        bad.append((call, name, int((f[name][ne] != r[ne]).sum())))
        if env.get("TEST_UPWIND"):
            # transport_upwind (row f-3) on the slab's resident velocities: its edge-velocity halos go through the same
            # exchange machinery (E-W ring, tripole fold of E-face / N-face fields with the mirror ranks)
            synth.add_thickness_distribution(f1)
            planes = [f1["aice0"]] + [a for n in range(f1["aicen"].shape[1]) for a in (f1["aicen"][:, n], f1["vicen"][:, n])]
            w1 = np.ascontiguousarray(np.stack(planes, axis=1))
            for k in range(w1.shape[1]):
                w = np.ascontiguousarray(w1[:, k]); orc.halo_r8(d1, w, C.LOC_CENTER, C.KIND_SCALAR, 0.0); w1[:, k] = w
            loc = [next(k for k, bb in enumerate(d1.local_blocks) if bb.block_id == b.block_id) for b in d.local_blocks]
            wl = np.ascontiguousarray(w1[loc])
            orc.transport_upwind(d1, 3600.0, f1, w1)
            s.ctx.transport_upwind(3600.0, wl)
            if not np.array_equal(wl, w1[loc]):
                bad.append(("transport_upwind", int((wl != w1[loc]).sum())))
            if not np.abs(w1[loc] - np.stack(planes, axis=1)[loc]).max() > 0:
                bad.append(("transport_upwind did nothing", 0))
        if env.get("TEST_UPWIND_STATE"):
            # transport_upwind whole: bound_state's ghost cells of the new state come through the exchange machinery
            from tests.test_parity_gpu import _upwind_state
            st1, kw = _upwind_state(d1, f1, 11, "lvl_ponds")
            loc = [next(k for k, bb in enumerate(d1.local_blocks) if bb.block_id == b.block_id) for b in d.local_blocks]
            stl = [np.ascontiguousarray(a[loc]) for a in st1]
            orc.transport_upwind_state(d1, 3600.0, f1, *st1, **kw)
            s.ctx.transport_upwind_state(3600.0, *stl, **kw)
            every = util.cell_mask(d, "all")
            for name, a, r in zip(("aice0", "aicen", "vicen", "vsnon", "trcrn"), stl, st1):
                m = np.broadcast_to(every if a.ndim == 3 else (every[:, None] if a.ndim == 4 else every[:, None, None]), a.shape)
                if not np.array_equal(a[m], r[loc][m]):
                    bad.append(("transport_upwind_state " + name, int((a[m] != r[loc][m]).sum())))
        if env.get("TEST_REMAP"):
            # transport_remap (row f-3): the ghost-cell updates of its centre fields go through the exchange machinery in
            # groups of planes (E-W ring, tripole fold with the mirror ranks), dpx / dpy as NE-corner vectors
            synth.add_thickness_distribution(f1)
            synth.add_remap_grid(case, d1, f1)
            synth.add_remap_grid(case, d, f)
            ncat = f1["aicen"].shape[1]
            tables = orc.remap_tables([0, 1, 2 + 1, 2 + 2])
            ntrace = len(tables[0])
            m1 = np.zeros((d1.nblocks, ncat + 1) + f1["aice0"].shape[1:])
            t1 = np.zeros((d1.nblocks, ncat, ntrace) + f1["aice0"].shape[1:])
            m1[:, 0] = f1["aice0"]
            for n in range(ncat):
                a, v = f1["aicen"][:, n], f1["vicen"][:, n]
                m1[:, n + 1] = a
                h = np.where(a > 1e-11, v / np.where(a > 1e-11, a, 1.0), 0.0)
                t1[:, n, 0], t1[:, n, 1] = h, 0.2 * h
                for k in range(2, ntrace):
                    t1[:, n, k] = np.where(a > 1e-11, -3.0 - n - 0.5 * k + 0.1 * h, 0.0)
            for arr in (m1.reshape(d1.nblocks, -1, *m1.shape[2:]), t1.reshape(d1.nblocks, -1, *m1.shape[2:])):
                for k in range(arr.shape[1]):
                    w = np.ascontiguousarray(arr[:, k]); orc.halo_r8(d1, w, C.LOC_CENTER, C.KIND_SCALAR, 0.0); arr[:, k] = w
            loc = [next(k for k, bb in enumerate(d1.local_blocks) if bb.block_id == b.block_id) for b in d.local_blocks]
            ml, tl = np.ascontiguousarray(m1[loc]), np.ascontiguousarray(t1[loc])
            m0 = m1.copy()
            dtr = 0.4 * xmin / max(np.abs(f1["uvel"]).max(), np.abs(f1["vvel"]).max())
            s.ctx.remap_init(f["dxu"], f["dyu"], f["hm"])
            rc1 = orc.horizontal_remap(d1, dtr, f1, m1, t1, *tables)
            rc = s.ctx.transport_remap(dtr, ml, tl, *tables)
            if rc1 or rc:
                bad.append(("transport_remap rc", rc1 * 100 + rc))
            if not (np.array_equal(ml, m1[loc]) and np.array_equal(tl, t1[loc])):
                bad.append(("transport_remap", int((ml != m1[loc]).sum() + (tl != t1[loc]).sum())))
            if not np.abs(m1 - m0).max() > 1e-4:
                bad.append(("transport_remap did nothing", 0))
        if env.get("TEST_REMAP_STATE"):
            # transport_remap with the state transforms: bound_state's ghost cells come through the exchange machinery too
            from tests.test_parity_gpu import _ice_state
            synth.add_remap_grid(case, d1, f1)
            synth.add_remap_grid(case, d, f)
            ntrcr, ntrcr_dim, nt_qsno, nslyr = 4, 5, 3, 1
            tables = orc.remap_tables([0, 1, 2, 0])
            st1 = _ice_state(d1, f1, ntrcr, ntrcr_dim, nt_qsno, nslyr)
            loc = [next(k for k, bb in enumerate(d1.local_blocks) if bb.block_id == b.block_id) for b in d.local_blocks]
            stl = [np.ascontiguousarray(a[loc]) for a in st1]
            dtr = 0.4 * xmin / max(np.abs(f1["uvel"]).max(), np.abs(f1["vvel"]).max())
            s.ctx.remap_init(f["dxu"], f["dyu"], f["hm"])
            rc1 = orc.transport_remap_state(d1, dtr, f1, *st1, ntrcr, nt_qsno, nslyr, 1.1e8, *tables)
            rc = s.ctx.transport_remap_state(dtr, *stl, ntrcr, nt_qsno, nslyr, 1.1e8, *tables)
            if rc1 or rc:
                bad.append(("transport_remap_state rc", rc1 * 100 + rc))
            every = util.cell_mask(d, "all")
            for name, a, r in zip(("aice0", "aicen", "vicen", "vsnon", "trcrn"), stl, st1):
                m = np.broadcast_to(every if a.ndim == 3 else (every[:, None] if a.ndim == 4 else every[:, None, None]), a.shape)
                if not np.array_equal(a[m], r[loc][m]):
                    bad.append(("transport_remap_state " + name, int((a[m] != r[loc][m]).sum())))
        st = s.ctx.stats()
        s.close()
        if int(st.transport) != {"shm": 2, "ipc": 3, "rccl": 1}[xp]:
            bad.append(("transport", int(st.transport)))
        q.put((rank, bad[:6], int(st.icellu), int(st.kernel2_launches), float(np.abs(f["uvel"]).max()) if d.nblocks else 0.0, time.time() - t0,
               int(st.zone_cols), int(st.zone_exchanges), int(st.band_row_exchanges)))
    except Exception:
        q.put((rank, ["EXC " + traceback.format_exc()], 0, 0, 0.0, 0.0, 0, 0, 0))


def _device_count():
    """GPUs of the box (torch.cuda.device_count() does not initialise the GPU in the calling process on this image)"""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 1


def _per_rank(world):
    mode = os.environ.get("EVPK_TEST_DEVICES", "auto")
    return mode == "per_rank" or (mode == "auto" and _device_count() >= world)


def _run(world, ns, nx, ny, bsx, bsy, ndte, env=None, xp="shm"):
    import multiprocessing as mp
    env = dict(env or {}, TEST_XP=xp)
    if _per_rank(world):
        env["TEST_PER_RANK"] = "1"
    elif xp == "rccl":
        pytest.skip(f"RCCL refuses {world} ranks on {_device_count()} device(s)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    tag = "evpk_t_" + uuid.uuid4().hex[:12]
    uidq = None
    if xp == "rccl":
        from cice5_amd import evpk
        uidq = ctx.Queue()
        uid = evpk.get_unique_id()                # (ncclGetUniqueId: no device is touched by this process)
        for _ in range(world):
            uidq.put(uid)
    procs = [ctx.Process(target=_worker, args=(r, world, tag, ns, nx, ny, bsx, bsy, ndte, env or {}, q, uidq)) for r in range(world)]
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
    assert len(res) == world
    for rank, bad, icellu, k2, umax, secs, zc, zx, bx in res:
        assert not bad, f"rank {rank}: {bad}"
    assert max(r[4] for r in res) > 1e-3
    print(f"[{world} ranks {ns}] worker seconds: {[round(r[5], 1) for r in res]}")
    return res


# transports of the parametrised cases: RCCL joins by itself on a box with at least two GPUs
XPS = [x for x in os.environ.get("EVPK_TEST_XPS", "ipc,shm,rccl" if _device_count() >= 2 else "ipc,shm").split(",") if x]


@pytest.mark.parametrize("xp", XPS)
@pytest.mark.parametrize("world", [2, 3, 4])
def test_x_slabs_open(world, xp):
    res = _run(world, "open", 240, 64, 20, 32, ndte=31, xp=xp)
    assert all(r[3] > 0 for r in res)          # the two-subcycle kernel ran on every rank (ghost-zone mode)
    assert all(r[6] == 8 for r in res)         # default: zones of 8 columns, an exchange every 4th launch


@pytest.mark.parametrize("m", [1, 2, 3])
def test_ghost_zone_depth(m, xp="ipc"):
    """EVPK_ZONE_M launches of the two-subcycle kernel per exchange (zones of 2*m columns, default 4): the redundant
    zone columns and the compacted row lists must leave every rank bit-identical to the oracle.  Odd ndte and a
    second call cover the one-subcycle tail and the re-made row lists."""
    res = _run(3, "open", 240, 64, 20, 32, ndte=31, env={"EVPK_ZONE_M": str(m)}, xp=xp)
    assert all(r[3] > 0 for r in res)
    # 15 two-subcycle launches + the one-subcycle tail (its own one-column halo): an exchange after every m-th launch
    assert all(r[6] == 2 * m and r[7] == 15 // m for r in res), [(r[6], r[7]) for r in res]
    _run(2, "open", 240, 64, 40, 64, ndte=26, env={"EVPK_ZONE_M": str(m), "EVPK_OVERLAP": "0"}, xp=xp)


def test_narrow_slabs_limit_the_zone_depth():
    """Slabs of 6 columns can only feed zones of 6 (m = 3); slabs of 4 columns zones of 4."""
    assert all(r[6] == 6 for r in _run(4, "open", 24, 40, 6, 20, ndte=22, xp="ipc"))
    assert all(r[6] == 4 for r in _run(4, "open", 16, 40, 4, 20, ndte=22, xp="shm"))


@pytest.mark.parametrize("xp", XPS)
@pytest.mark.parametrize("ns", ["open", "tripole"])
def test_soak_wandering_ice_across_slabs(ns, xp):
    """Eight consecutive evp() calls on three x-slabs while the ice cover wanders across the slab boundaries (and, on the
    tripole grid, across the fold): the ghost-zone row lists, strip lists and tile flags change every call."""
    res = _run(3, ns, 240, 72, 20, 36, ndte=14, env={"TEST_WANDER_CALLS": "8"}, xp=xp)
    assert all(r[3] > 0 for r in res)


def test_cfg4_1440x1080_on_four_ranks():
    """BASELINE config 4 shape: 1440x1080 in 30x27 blocks (bld/config.nci.auscom.1440x1080) on 4 x-slabs;
    a short loop (ndte = 10) keeps the single-process oracle cheap."""
    _run(4, "open", 1440, 1080, 30, 27, ndte=10, xp="ipc")


def test_cfg4_1440x1080_tripole_on_four_ranks():
    """the 0.25-degree ACCESS-OM2 grid is tripolar in production: the same decomposition with the fold between mirror ranks"""
    _run(4, "tripole", 1440, 1080, 30, 27, ndte=10, xp="ipc")


@pytest.mark.parametrize("ns", ["open", "tripole"])
def test_cfg4_1440x1080_ndte120_on_four_ranks(ns):
    """BASELINE config 4 as BASELINE.json states it -- 1440x1080, ndte = 120, 4 x-slab ranks (here sharing the one GPU): two
    whole evp calls, every output of every rank against the single-process oracle (full ghost-zone cycles, the LAST2 ending
    across ranks, finish)"""
    res = _run(4, ns, 1440, 1080, 30, 27, ndte=120, xp="ipc")
    assert all(r[3] > 0 for r in res)


@pytest.mark.parametrize("what", ["TEST_REMAP_STATE", "TEST_EAP"])
def test_cfg4_shape_rows_f3_f4_on_four_ranks(what):
    """transport_remap (with the state transforms) and eap(dt) across four x-slabs at config 4's shape and block size, tripole"""
    _run(4, "tripole", 1440, 1080, 30, 27, ndte=22 if what == "TEST_EAP" else 12, env={what: "1"}, xp="ipc")


def test_cfg5_shape_ndte240_on_four_ranks():
    """BASELINE config 5's subcycle count (ndte = 240) on a config-5-shaped tripole grid: 3600 columns in four 900-column
    slabs, the whole evp (120 pairs with their band launches and folds between mirror ranks, the stress folds, finish), twice"""
    res = _run(4, "tripole", 3600, 96, 450, 48, ndte=240, xp="ipc")
    assert all(r[3] > 0 for r in res)


@pytest.mark.parametrize("xp", XPS)
def test_cfg5_tripole_slabs_eight_wide_grid(xp):
    """BASELINE config 5 decomposition in miniature: tripole grid, 4 slabs of a 3600-column grid, few rows."""
    _run(4, "tripole", 3600, 64, 450, 32, ndte=12, xp=xp)


@pytest.mark.parametrize("xp", XPS)
def test_x_slabs_tripole(xp):
    _run(2, "tripole", 240, 64, 20, 32, ndte=24, xp=xp)
    _run(4, "tripole", 240, 64, 30, 16, ndte=13, xp=xp)
    _run(3, "tripole", 240, 64, 20, 32, ndte=10, xp=xp)          # uneven mirror: 3 slabs


@pytest.mark.parametrize("world,nx,bsx", [(3, 240, 20), (2, 240, 48), (5, 240, 48), (3, 240, 48), (4, 400, 40)])
def test_mirror_slab_for_any_rank_count(world, nx, bsx):
    """mpi/ice_boundary.F90:2737-2913 folds whatever blocks hold the mirrored columns.  The mirror slab of band_pair does the
    same since round 4: an odd number of ranks (the middle one mirrors onto itself), slabs of unequal width (5 block columns on
    2 or 3 ranks: 3 + 2, 2 + 2 + 1; 10 on 4: 3 + 3 + 3 + 1) -- one message per partner and zone refresh instead of an exchange
    after every subcycle"""
    res = _run(world, "tripole", nx, 64, bsx, 32, ndte=120, xp="ipc")
    for r in res:
        assert r[6] >= 3, r                      # ghost zones deeper than one launch: the mirror-slab path ran
        assert 0 < r[8] <= 20, r                 # band_row_exchanges per evp(dt) of 120 subcycles (two calls: the last one's count)


@pytest.mark.parametrize("ns,world", [("open", 3), ("tripole", 2), ("tripole", 4)])
def test_transport_upwind_across_slabs(ns, world):
    _run(world, ns, 240, 64, 20, 32, ndte=12, env={"TEST_UPWIND": "1"}, xp="ipc")


@pytest.mark.parametrize("ns,world,xp", [("open", 3, "ipc"), ("tripole", 2, "shm"), ("tripole", 4, "ipc")])
def test_transport_upwind_state_across_slabs(ns, world, xp):
    _run(world, ns, 240, 64, 20, 32, ndte=12, env={"TEST_UPWIND_STATE": "1"}, xp=xp)


@pytest.mark.parametrize("ns,world,xp", [("open", 3, "ipc"), ("tripole", 2, "ipc"), ("tripole", 4, "shm"), ("tripole", 3, "ipc")])
def test_transport_remap_across_slabs(ns, world, xp):
    _run(world, ns, 240, 64, 20, 32, ndte=12, env={"TEST_REMAP": "1"}, xp=xp)


@pytest.mark.parametrize("ns,world,xp", [("open", 3, "ipc"), ("tripole", 4, "ipc"), ("tripole", 2, "shm")])
def test_transport_remap_state_across_slabs(ns, world, xp):
    _run(world, ns, 240, 64, 20, 32, ndte=12, env={"TEST_REMAP_STATE": "1"}, xp=xp)


@pytest.mark.parametrize("ns,world,xp", [("open", 3, "ipc"), ("tripole", 2, "ipc"), ("tripole", 4, "shm")])
def test_eap_across_slabs(ns, world, xp):
    """kdyn = 2 on x-slabs: the velocity halo of every subcycle through the exchange machinery, T-cell state (stresses,
    structure tensor) redundant on the shared ghost T column, as the reference's blocks have it"""
    res = _run(world, ns, 240, 64, 20, 32, ndte=22, env={"TEST_EAP": "1"}, xp=xp)
    assert all(r[3] == 0 for r in res)          # no two-subcycle EVP launches


def test_x_slabs_one_subcycle_kernel_and_serial_exchange():
    _run(3, "open", 240, 64, 40, 64, ndte=20, env={"EVPK_DOUBLE": "0"}, xp="ipc")
    _run(2, "open", 240, 64, 40, 64, ndte=20, env={"EVPK_OVERLAP": "0"}, xp="ipc")
    _run(2, "tripole", 240, 64, 40, 64, ndte=20, env={"EVPK_DOUBLE": "0"}, xp="ipc")
    _run(2, "tripole", 240, 64, 40, 64, ndte=20, env={"EVPK_DOUBLE": "0"}, xp="shm")


@pytest.mark.parametrize("seed", range(int(os.environ.get("EVPK_FUZZ_MR_N", "8"))))
def test_random_multirank_configuration(seed):
    """Seeded random x-slab runs through the relay: 2-4 ranks, slab widths from 4 columns up, several blocks per slab,
    open / tripole, ghost-zone depth, overlap on / off, one- or two-subcycle kernel, odd / even ndte, wandering ice."""
    rng = np.random.default_rng(500 + seed + int(os.environ.get("EVPK_FUZZ_BASE", "0")))
    world = int(rng.choice([2, 3, 4]))
    bsx = int(rng.choice([4, 6, 10, 20, 31]))
    nx = world * int(rng.choice([1, 2, 3])) * bsx
    ns = str(rng.choice(["open", "tripole"]))
    if ns == "tripole" and nx % 2:
        bsx += 1
        nx = world * (nx // (bsx - 1) // world) * bsx
    ny = int(rng.choice([24, 40, 64]))
    bsy = int(rng.choice([ny, ny // 2]))
    env = {"EVPK_ZONE_M": str(int(rng.integers(1, 5)))}
    if rng.random() < 0.3:
        env["EVPK_OVERLAP"] = "0"
    if rng.random() < 0.2:
        env["EVPK_DOUBLE"] = "0"
    if rng.random() < 0.5:
        env["TEST_WANDER_CALLS"] = "3"
    ndte = int(rng.choice([5, 8, 13]))
    xp = XPS[seed % 2]
    print("config:", world, ns, nx, ny, bsx, bsy, ndte, env, xp)
    _run(world, ns, nx, ny, bsx, bsy, ndte=ndte, env=env, xp=xp)


@pytest.mark.parametrize("seed", range(int(os.environ.get("EVPK_FUZZ_MR_N", "8"))))
def test_random_unequal_slabs_on_a_tripole_grid(seed):
    """Seeded random tripole runs whose block columns do NOT divide evenly over the ranks (create_distrb_cart then gives the
    last rank a narrower slab) on 2-5 ranks: the mirror slab's columns come from two or three ranks, or the rank itself."""
    rng = np.random.default_rng(9100 + seed + int(os.environ.get("EVPK_FUZZ_BASE", "0")))
    while True:
        world = int(rng.choice([2, 3, 4, 5]))
        bsx = int(rng.choice([8, 12, 18, 20, 30]))
        nbx = int(rng.integers(world, 3 * world + 1))
        pp = (nbx - 1) // world + 1
        idle_ok = rng.random() < 0.3         # (round 5: the last ranks may be left without a block column, ice_distribution.F90:603-640)
        if (idle_ok or (world - 1) * pp < nbx) and (nbx % world or world % 2) and nbx * bsx % 2 == 0:
            break
    nx = nbx * bsx
    ny = int(rng.choice([24, 40, 64]))
    bsy = int(rng.choice([ny, ny // 2]))
    env = {"EVPK_ZONE_M": str(int(rng.integers(1, 5)))}
    if rng.random() < 0.5:
        env["TEST_WANDER_CALLS"] = "3"
    ndte = int(rng.choice([8, 13, 24]))
    xp = XPS[seed % 2]
    print("config:", world, nx, ny, bsx, bsy, ndte, env, xp)
    _run(world, "tripole", nx, ny, bsx, bsy, ndte=ndte, env=env, xp=xp)


@pytest.mark.parametrize("ns,world", [("tripole", 2), ("open", 3)])
def test_bench_line_of_a_multirank_run_verifies_itself(ns, world):
    """`python bench.py --gpus N`: after the timed steps every rank runs one untimed evp from rest and rank 0 compares all of it,
    bit for bit, with a ONE-rank device run of the whole grid on its own GPU (`verify.against = "1-rank device run"`); what the
    line says about its ranks (`devices`, `rccl_ranks`) and the comparison are part of the exit code.  Here: the ranks share the
    box's GPU(s) as the other cases of this file do, over the peer-mapped transport."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    shared = not _per_rank(world)
    if shared:
        env["EVPK_FORCE_DEVICE"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--transport", "ipc",
                        "--grid", "720x400", "--xblocks", "6", "--yblocks", "2", "--ndte", "24", "--ns", ns, "--dt", "1800"],
                       capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    line = json.loads(lines[-1])
    v = line["verify"]
    assert r.returncode == 0, (r.returncode, line.get("checks"), v)
    assert v["bit_identical"] is True and v["against"] == "1-rank device run" and v["blocks_compared"] == 12 and v["max_abs_u"] > 1e-3
    assert v["values_compared"] > 18 * 720 * 400
    assert line["checks"] == {"devices": True, "rccl_ranks": True, "bit_identical": True}
    assert line["n_gpus"] == world and line["config"]["devices"] == (1 if shared else world) and line["value"] > 0


@pytest.mark.parametrize("xp", XPS)
@pytest.mark.parametrize("ns,world,nx,bsx", [("open", 4, 200, 40), ("tripole", 4, 200, 40), ("open", 3, 128, 64), ("tripole", 5, 240, 80)])
def test_ranks_without_a_block_column(ns, world, nx, bsx, xp):
    """create_distrb_cart deals ceil(nblocks_x / nprocs) block columns to a rank and the last ranks may get none (ice_distribution.F90:
    603-640: 5 columns on 4 ranks = 2, 2, 1, 0; 2 on 3 = 1, 1, 0; 3 on 5 = 1, 1, 1, 0, 0) and carries on.  Such a rank takes part in
    evpk_connect and returns from everything else; the others form the ring (uneven slabs; the tripole fold from a virtual mirror slab)."""
    res = _run(world, ns, nx, 64, bsx, 32, ndte=14, xp=xp)
    idle = [r for r in res if r[2] == 0 and r[3] == 0]
    nbx = nx // bsx
    assert len(idle) == world - -(-nbx // (-(-nbx // world)))


@pytest.mark.parametrize("ns", ["open", "tripole"])
def test_a_rank_whose_blocks_are_all_eliminated_land(ns):
    """three ranks, three block columns, the middle one land on every row: rank 1 is handed NO block (land-block elimination,
    ice_domain.F90:387-441) but its columns exist -- its slab is land, its neighbours' ghost zones read land from it, it takes part in
    every exchange of the ring; also with the transports of rows f-3 (bound_state next to the eliminated blocks)"""
    res = _run(3, ns, 120, 64, 40, 32, ndte=14, env={"TEST_LAND_BAND": "41,80", "TEST_UPWIND_STATE": "1"}, xp="ipc")
    assert sorted(r[2] > 0 for r in res) == [False, True, True]
