"""world_size-2 CPU tests of the sharded path (gloo): x-slab decomposition, ring neighbours from evpk_slab_layout, and the
exchange protocols the multi-GPU library speaks (cice5_amd/csrc/evpk_api.hip), modelled with the CPU oracle as the
arithmetic of each rank and torch.distributed point-to-point calls as the transport:

  * `test_two_rank_ghost_zones_gloo` -- the communication-avoiding E-W protocol of the two-subcycle kernel
    (exchange_cols): every rank carries W = 2 m ghost-zone columns per side, advances them redundantly, loses two valid
    columns per launch (= two subcycles) and receives the neighbour's W edge columns of the whole prognostic state (u, v,
    twelve stresses, all rows) once per m launches; m = 1, 2, 4.  No message inside the 2 m subcycles in between.
  * `test_two_rank_slab_exchange_gloo` -- the one-subcycle protocol (halo()): E-W edge columns over ALL rows (ghost rows
    included, which carries the corners) after every subcycle; on tripole grids the fold as POINT-TO-POINT messages with
    the mirror ranks (fold_p2p: each rank sends its two top rows to the ranks that own the mirror images of its columns and
    folds its own columns only; the ghost columns of the two rows come with the E-W exchange that follows).

The result must equal the single-process oracle bit for bit (the reference is decomposition invariant).
"""
import os
import socket
import sys
import traceback

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _edge_column(d, a, which):
    """Full-height (ny+2) column of the slab: 'W' = first physical column, 'E' = last."""
    i0, i1, _, _ = d.slab()
    col = np.zeros(d.ny_global + 2)
    for n, b in enumerate(d.local_blocks):
        if which == "W" and b.iglob_lo != i0:
            continue
        if which == "E" and b.iglob_lo + (b.ihi - b.ilo) != i1:
            continue
        ic = (b.ilo if which == "W" else b.ihi) - 1
        for j in range(b.jlo - 1, b.jhi + 2):          # 1-based j = jlo-1 .. jhi+1 -> 0-based j-1
            gj = b.jglob_lo + (j - b.jlo)
            if 0 <= gj <= d.ny_global + 1:
                col[gj] = a[n, j - 1, ic]
    return col


def _set_ghost_column(d, a, which, col):
    i0, i1, _, _ = d.slab()
    for n, b in enumerate(d.local_blocks):
        if which == "W" and b.iglob_lo != i0:
            continue
        if which == "E" and b.iglob_lo + (b.ihi - b.ilo) != i1:
            continue
        ic = (b.ilo - 1 if which == "W" else b.ihi + 1) - 1
        for j in range(b.jlo - 1, b.jhi + 2):
            gj = b.jglob_lo + (j - b.jlo)
            a[n, j - 1, ic] = col[gj]


def _fold_rows(nx, B1, B2, necorner, sgn):
    """numpy mirror of k_fold_apply: returns (top_row or None, north_ghost_row), index 1..nx."""
    g = np.arange(1, nx + 1)          # B1, B2 are 1-based (index 0 unused); results are 0-based over g = 1..nx
    if not necorner:
        return None, sgn * B2[nx - g + 1]
    src = nx - g
    src[src == 0] = nx
    sym = B2.copy()
    h = nx // 2
    for i in range(1, h):
        x = 0.5 * (B2[i] + sgn * B2[nx - i])
        sym[i] = x
        sym[nx - i] = sgn * x
    return sgn * sym[src], sgn * B1[src]


def _worker(rank, world, port, ns, q, bsx=12):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          OMP_NUM_THREADS="2")
        import torch
        import torch.distributed as dist
        from cice5_amd import blocks, constants as C, evpk, synth
        from oracle import orc
        from tests import util

        dist.init_process_group("gloo", rank=rank, world_size=world)
        nx, ny, bsy, ndte = 48, 40, 10, 30
        case = synth.SynthCase(nx=nx, ny=ny, ns_boundary=C.BND_NAMES[ns], land="continents")
        d = blocks.create_distrb_cart(nx, ny, bsx, bsy, nprocs=world, rank=rank, ns_boundary_type=ns)
        f = synth.make_block_fields(case, d)
        # the ring is made of the ranks that own block columns: create_distrb_cart deals ceil(nbx / nprocs) columns to a rank and
        # the last ranks may get none (ice_distribution.F90:603-640) -- such a rank joins the start (process group here, evpk_connect
        # in the library) and the global reductions of the host, and no exchange
        nbx = nx // bsx
        nact = -(-nbx // (-(-nbx // world)))
        if rank >= nact:
            assert d.nblocks == 0 and d.slab() == (1, 0, 1, 0)
            tot = torch.tensor([0, 0], dtype=torch.int64)
            dist.all_reduce(tot)
            dist.barrier()
            dist.destroy_process_group()
            q.put((rank, []))
            return
        world_host, world = world, nact
        i0, i1, j0, j1 = d.slab()
        assert (j0, j1) == (1, ny) and i1 - i0 + 1 == nx // world
        lay = evpk.slab_layout(nx, world, rank, d.ew_boundary, i0, i1)
        assert lay["west"] == lay["east"] == 1 - rank
        shape = (d.nblocks, d.ny_block, d.nx_block)
        tripole = ns == "tripole"

        saved = {}

        def patch(ptr, loc, kind, fill, phase):
            a = np.ctypeslib.as_array(ptr, shape=shape)
            sgn = -1.0 if kind == C.KIND_VECTOR else 1.0
            if phase == 0:
                if tripole:     # the two top physical rows as they are BEFORE any update
                    loc_rows = np.zeros((2, i1 - i0 + 1))
                    for n, b in enumerate(d.local_blocks):
                        if b.tripole:
                            c0 = b.iglob_lo - i0
                            w = b.ihi - b.ilo + 1
                            loc_rows[0, c0:c0 + w] = a[n, b.jhi - 2, b.ilo - 1:b.ihi]
                            loc_rows[1, c0:c0 + w] = a[n, b.jhi - 1, b.ilo - 1:b.ihi]
                    saved["rows"] = loc_rows
                return
            if tripole:
                # point-to-point with the mirror ranks (fold_p2p): my columns g = i0..i1 read the rows at nx - g (NE corner)
                # and nx - g + 1 (centre); every rank works the partner set out from the slab starts alone
                w = nx // world
                owner = lambda g: ((g - 1) % nx) // w
                need = sorted({owner(nx - g + k) for g in range(i0, i1 + 1) for k in (0, 1)})
                gives = sorted(r for r in range(world) if any(owner(nx - g + k) == rank for g in range(r * w + 1, (r + 1) * w + 1) for k in (0, 1)))
                mine = torch.from_numpy(saved.pop("rows"))
                got = {}
                reqs = [dist.isend(mine.clone(), r) for r in gives if r != rank]
                for r in need:
                    if r == rank:
                        got[r] = mine.numpy()
                    else:
                        t = torch.zeros_like(mine)
                        dist.recv(t, r)
                        got[r] = t.numpy()
                for q in reqs:
                    q.wait()
                B1 = np.full(nx + 1, np.nan); B2 = np.full(nx + 1, np.nan)      # only the partners' columns are known
                for r, rows in got.items():
                    B1[r * w + 1:(r + 1) * w + 1] = rows[0]
                    B2[r * w + 1:(r + 1) * w + 1] = rows[1]
                top, north = _fold_rows(nx, B1, B2, loc == C.LOC_NECORNER, sgn)
                for n, b in enumerate(d.local_blocks):
                    if b.tripole:
                        for i in range(1, d.nx_block + 1):                      # the slab's own columns only (a block's ghost
                            g = (b.iglob_lo + (i - b.ilo) - 1) % nx + 1         # column inside the slab is one of them)
                            if not (i0 <= g <= i1):
                                continue
                            a[n, b.jhi, i - 1] = north[g - 1]
                            if top is not None:
                                a[n, b.jhi - 1, i - 1] = top[g - 1]
            # E-W: one message each way with both edges (two ranks, cyclic ring)
            send = torch.from_numpy(np.concatenate([_edge_column(d, a, "W"), _edge_column(d, a, "E")]))
            recv = torch.zeros_like(send)
            peer = lay["west"]
            if rank == 0:
                dist.send(send, peer); dist.recv(recv, peer)
            else:
                dist.recv(recv, peer); dist.send(send, peer)
            r = recv.numpy()
            half = ny + 2
            _set_ghost_column(d, a, "E", r[:half])      # the peer's W edge is my east ghost
            _set_ghost_column(d, a, "W", r[half:])      # the peer's E edge is my west ghost

        keep = orc.set_halo_callback(patch)
        xmin = synth.global_min_dx(case)
        p = orc.make_params(3600.0, ndte, xmin)
        nt, nu, _ = orc.evp(d, p, f)
        orc.set_halo_callback(None)
        del keep
        # reference: the whole domain in this process
        d1 = blocks.create_distrb_cart(nx, ny, bsx, bsy, ns_boundary_type=ns)
        f1 = synth.make_block_fields(case, d1)
        nt1, nu1 = orc.evp(d1, p, f1)[:2]
        tot = torch.tensor([nt, nu], dtype=torch.int64)
        dist.all_reduce(tot)
        assert (int(tot[0]), int(tot[1])) == (nt1, nu1)
        bad = []
        for name in ["uvel", "vvel", "divu", "strocnxT", "strintx", "prs_sig"] + util.SIGMA:
            G = blocks.gather_global(d1, f1[name])[:, i0 - 1:i1]
            L = blocks.gather_global(d, f[name])[:, i0 - 1:i1]
            if not np.array_equal(G, L):
                bad.append(name)
        # ghost columns of the velocity too
        for n, b in enumerate(d.local_blocks):
            n1 = next(k for k, bb in enumerate(d1.local_blocks) if bb.block_id == b.block_id)
            if not np.array_equal(f["uvel"][n], f1["uvel"][n1]):
                bad.append(f"uvel ghosts of block {b.block_id}")
        assert np.abs(f["uvel"]).max() > 1e-3
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, bad))
    except Exception:
        q.put((rank, ["EXC " + traceback.format_exc()]))


@pytest.mark.parametrize("ns,world,bsx", [("open", 2, 12), ("tripole", 2, 12), ("open", 3, 24), ("tripole", 3, 24)])
def test_two_rank_slab_exchange_gloo(ns, world, bsx):
    """world = 3 with two block columns: ranks 0 and 1 form the ring, rank 2 owns nothing and only joins the collectives"""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ns, q, bsx)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    for _ in procs:
        res.append(q.get(timeout=300))
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.terminate()
    for rank, bad in res:
        assert not bad, f"rank {rank}: {bad}"


def _zone_worker(rank, world, port, m, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          OMP_NUM_THREADS="2")
        import torch
        import torch.distributed as dist
        from cice5_amd import blocks, constants as C, evpk, synth
        from oracle import orc
        from tests import util

        dist.init_process_group("gloo", rank=rank, world_size=world)
        nx, ny, ndte = 48, 40, 24
        W = 2 * m                                    # ghost-zone columns per side; one exchange per m launches of two subcycles
        nxl = nx // world
        i0 = rank * nxl + 1
        lay = evpk.slab_layout(nx, world, rank, C.BND_CYCLIC, i0, i0 + nxl - 1)
        west, east = lay["west"], lay["east"]
        assert nxl >= W and west == east == 1 - rank
        case = synth.SynthCase(nx=nx, ny=ny, land="continents")
        # the rank's slab with its zones as ONE block of an open-ended strip: columns i0-W .. i0+nxl-1+W of the global grid
        # (cyclic), every field a function of the global cell index, so zone columns start out as the neighbour's own
        next_ = nxl + 2 * W
        dx = blocks.create_distrb_cart(next_, ny, next_, ny, ew_boundary_type="open")
        Iw, Jw = blocks.block_index_windows(dx)
        I, J = np.broadcast_arrays((Iw[0] + i0 - W - 1)[None, :], Jw[0][:, None])
        f = synth.make_block_fields(synth.SynthCase(nx=next_, ny=ny), dx)
        cache = {}
        for name in synth.GRID_FIELDS + synth.INPUT_FIELDS + synth.MASK_FIELDS:
            f[name][0] = case.field(name, I, J, cache)
        xmin = synth.global_min_dx(case)
        p = orc.make_params(3600.0, ndte, xmin)
        state = ["uvel", "vvel"] + util.SIGMA
        nex = 0
        for launch in range(ndte // 2):
            orc.evp(dx, p, f, nsub=2)                # one launch of the two-subcycle kernel: the zones lose two columns per side
            if (launch + 1) % m:
                continue
            # zones used up: my W edge columns of the state, all rows, to each neighbour; theirs into my zones
            # (block array column index = strip column, the west ghost column of the block being index 0)
            sendW = torch.from_numpy(np.stack([f[n][0][:, W + 1:2 * W + 1] for n in state]).copy())
            sendE = torch.from_numpy(np.stack([f[n][0][:, nxl + 1:nxl + W + 1] for n in state]).copy())
            recvE, recvW = torch.zeros_like(sendW), torch.zeros_like(sendE)
            if rank == 0:
                dist.send(sendW, west); dist.send(sendE, east); dist.recv(recvE, east); dist.recv(recvW, west)
            else:
                dist.recv(recvE, east); dist.recv(recvW, west); dist.send(sendW, west); dist.send(sendE, east)
            for k, n in enumerate(state):
                f[n][0][:, nxl + W + 1:nxl + 2 * W + 1] = recvE[k].numpy()      # the east neighbour's west edge
                f[n][0][:, 1:W + 1] = recvW[k].numpy()                          # the west neighbour's east edge
            nex += 1
        assert nex == ndte // (2 * m)
        # reference: the whole domain in this process, one evp of ndte subcycles, another block size
        d1 = blocks.create_distrb_cart(nx, ny, 12, 10)
        f1 = synth.make_block_fields(case, d1)
        orc.evp(d1, p, f1)
        bad = []
        for name in state:
            G = blocks.gather_global(d1, f1[name])[:, i0 - 1:i0 - 1 + nxl]
            L = f[name][0][1:ny + 1, W + 1:W + nxl + 1]
            if not np.array_equal(G, L):
                bad.append((name, int((G != L).sum())))
        assert np.abs(f["uvel"]).max() > 1e-3
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, bad))
    except Exception:
        q.put((rank, ["EXC " + traceback.format_exc()]))


@pytest.mark.parametrize("m", [1, 2, 4])
def test_two_rank_ghost_zones_gloo(m):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_zone_worker, args=(r, 2, port, m, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    for _ in procs:
        res.append(q.get(timeout=300))
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.terminate()
    for rank, bad in res:
        assert not bad, f"rank {rank}: {bad}"


@pytest.mark.parametrize("m", [1, 2, 3])
def test_band_validity_between_mirror_ranks_needs_one_more_zone_column(m):
    """Dependency model of band_pair between x-slab ranks (evpk_kernels.hip): which top-row U columns of a slab of width w
    are still correct after m two-subcycle launches without a ghost-zone exchange, when the NE-corner fold takes column c
    from the mirror rank's column w - c (serial/ice_boundary.F90:801-888).  T(c) reads U(c-1), U(c); U(c) reads T(c),
    T(c+1); the fold after each subcycle reads the partner's column w - c.  With zones of 2m columns the slab's own
    column w is lost (the image of the zone [1-zW, w+zW] is [-zW, w+zW-1]); 2m+1 columns keep 0 .. w (evpk_connect)."""
    w = 40

    def run(zW):
        lo = 1 - zW
        n = w + 2 * zW
        ok = np.ones(n, bool)                       # U columns lo .. w+zW, both ranks alike (equal widths)

        def sub(u):
            t = np.zeros(n, bool); t[1:] = u[1:] & u[:-1]
            un = np.zeros(n, bool); un[:-1] = t[:-1] & t[1:]
            f = np.zeros(n, bool)
            for k in range(n):
                q = (w - (lo + k)) - lo             # index of the partner's column w - c
                f[k] = un[k] and 0 <= q < n and un[q]
            return f
        for _ in range(2 * m):
            ok = sub(ok)
        return [lo + k for k in range(n) if ok[k]]
    good = run(2 * m + 1)
    assert set(range(0, w + 1)) <= set(good)
    assert w not in run(2 * m)


@pytest.mark.parametrize("m", [1, 2, 3])
def test_mirror_slab_depth_for_m_pairs_without_a_refresh(m):
    """Dependency model in y of the mirror slab M of band_pair (evpk_connect): M holds the mirror rank's rows N-nylM .. N+1
    (local rows 0 .. nylM+1) and is advanced here between two refreshes.  T(j) reads U(j-1), U(j); U(j) reads T(j), T(j+1);
    the top rows are kept by the band.  Every pair of subcycles costs two rows from the bottom, and the band of pair k+1 needs
    the rows nylM-3 .. nylM+1 the pair k left: nylM = 2m+1 rows carry m pairs, 2m do not."""
    def pairs_supported(nylM):
        ok = np.ones(nylM + 2, bool)                                 # rows 0 .. nylM+1 after a refresh
        n = 0
        while ok[max(nylM - 3, 0):].all():                           # the band can run this pair
            n += 1
            for _ in range(2):
                t = np.zeros_like(ok); t[1:] = ok[1:] & ok[:-1]
                u = np.zeros_like(ok); u[:-1] = t[:-1] & t[1:]
                u[nylM - 1:] = True                                  # rows N-1 .. N+1: band_pair (both ranks compute them)
                ok = u
            if n > 10:
                break
        return n
    assert pairs_supported(max(4, 2 * m + 1)) >= m
    if m >= 2:
        assert pairs_supported(2 * m) < m


def _xband_worker(rank, world, port, m, q):
    """The tripole protocol of two x-slab ranks (evpk_connect / subcycle_impl with xband), restated with the oracle over gloo:
    every rank advances ONE composite domain [ my strip with its zones | the mirror rank's strip with its zones ] of width
    2 L, L = w + 2 W, closed by a tripole fold -- with equal slab widths my local column c and the mirror rank's w - c are
    images of each other under the fold of that composite (positions c + W and 2 L - c - W), exactly as band_pair pairs
    them.  Zones are W = 2 m + 1 columns; after m launches of two subcycles the E-W zones of my strip come from the neighbour
    and the TOP nylM + 2 = 2 m + 3 rows of the mirror strip (the mirror slab M) from the mirror rank; the rows of the mirror
    strip below them are never refreshed -- they go stale from the bottom exactly as the rows below M would if M had them.
    (What this case can and cannot see: a bit difference that enters at a zone's edge fades below one ulp within about five
    columns or rows here -- the EVP update is strongly damped -- so too narrow a zone shows (2m columns: late and in a few
    cells), too shallow a mirror slab does not; the depth rules are the dependency models below, the protocol -- who sends
    what when, the fold's pairing c <-> w - c, its axis columns -- is what this test holds.)"""
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          OMP_NUM_THREADS="2")
        import torch
        import torch.distributed as dist
        from cice5_amd import blocks, constants as C, synth
        from oracle import orc
        from tests import util

        dist.init_process_group("gloo", rank=rank, world_size=world)
        nx, ny, ndte = 48, 40, 24
        W = 2 * m + 1
        nylM = max(4, 2 * m + 1)
        w = nx // world
        other = 1 - rank                             # two ranks: the mirror rank P-1-r is also both E-W neighbours
        L = w + 2 * W
        case = synth.SynthCase(nx=nx, ny=ny, land="continents", ns_boundary=C.BND_TRIPOLE)
        dx = blocks.create_distrb_cart(2 * L, ny, 2 * L, ny, ew_boundary_type="open", ns_boundary_type="tripole")
        Iw, Jw = blocks.block_index_windows(dx)
        # global column of every composite column (the block's own ghost columns 0 and 2L+1 continue the strips)
        col = Iw[0].astype(np.int64)                 # 0 .. 2L+1
        gcolumn = np.where(col <= L, rank * w + 1 - W - 1 + col, other * w + 1 - W - 1 + (col - L))
        I, J = np.broadcast_arrays(gcolumn[None, :], Jw[0][:, None])
        f = synth.make_block_fields(synth.SynthCase(nx=2 * L, ny=ny, ns_boundary=C.BND_TRIPOLE), dx)
        cache = {}
        for name in synth.GRID_FIELDS + synth.INPUT_FIELDS + synth.MASK_FIELDS:
            f[name][0] = case.field(name, I, J, cache)
        xmin = synth.global_min_dx(case)
        p = orc.make_params(3600.0, ndte, xmin)
        state = ["uvel", "vvel"] + util.SIGMA
        # The columns on the fold's axes (global nx/2 and nx) are their own images: the fold leaves sgn * (their own top-row
        # value) there (serial/ice_boundary.F90:818-824 skips them in the symmetrisation, the copy :3752-3776 then negates a
        # vector).  In the composite each of them sits twice -- as an own column of one strip and as the ghost column 0 of the
        # other -- and the composite's fold would average the two copies instead; band_pair knows the GLOBAL column of every
        # lane (gcol) and applies the axis rule, so does this model, through the oracle's halo callback.
        gw = (gcolumn - 1) % nx + 1
        axis = [int(k) for k in np.nonzero((gw == nx // 2) | (gw == nx))[0] if 1 <= k <= 2 * L]
        shape = (1, ny + 2, 2 * L + 2)
        pre = {}

        def patch(ptr, loc, kind, fill, phase):
            if loc != C.LOC_NECORNER or kind != C.KIND_VECTOR:
                return
            a = np.ctypeslib.as_array(ptr, shape=shape)
            if phase == 0:
                pre["top"] = a[0, ny, axis].copy()
            else:
                a[0, ny, axis] = -pre["top"]

        keep = orc.set_halo_callback(patch)
        nex = 0
        for launch in range(ndte // 2):
            orc.evp(dx, p, f, nsub=2)
            if (launch + 1) % m:
                continue
            # 1. ghost zones of my strip: the neighbour's own edge columns, all rows (two ranks: one partner for both sides)
            sendW = torch.from_numpy(np.stack([f[n][0][:, W + 1:2 * W + 1] for n in state]).copy())          # my first W own columns
            sendE = torch.from_numpy(np.stack([f[n][0][:, w + 1:w + W + 1] for n in state]).copy())          # my last W own columns
            recvE, recvW = torch.zeros_like(sendW), torch.zeros_like(sendE)
            if rank == 0:
                dist.send(sendW, other); dist.send(sendE, other); dist.recv(recvE, other); dist.recv(recvW, other)
            else:
                dist.recv(recvE, other); dist.recv(recvW, other); dist.send(sendW, other); dist.send(sendE, other)
            for k, n in enumerate(state):
                f[n][0][:, w + W + 1:L + 1] = recvE[k].numpy()
                f[n][0][:, 1:W + 1] = recvW[k].numpy()
            # 2. the mirror slab: the top rows of the mirror rank's strip, its (fresh) zones included
            rows = slice(ny - nylM, ny + 2)
            sendM = torch.from_numpy(np.stack([f[n][0][rows, 1:L + 1] for n in state]).copy())
            recvM = torch.zeros_like(sendM)
            if rank == 0:
                dist.send(sendM, other); dist.recv(recvM, other)
            else:
                dist.recv(recvM, other); dist.send(sendM, other)
            for k, n in enumerate(state):
                f[n][0][rows, L + 1:2 * L + 1] = recvM[k].numpy()
            nex += 1
        assert nex == ndte // (2 * m)
        orc.set_halo_callback(None)
        del keep
        # reference: the whole domain in this process, another block size, the SAME sequence of calls (the oracle's evp runs
        # its preparation -- with a velocity halo update, whose fold negates the axis columns once more -- at every call)
        d1 = blocks.create_distrb_cart(nx, ny, 12, 10, ns_boundary_type="tripole")
        f1 = synth.make_block_fields(case, d1)
        for launch in range(ndte // 2):
            orc.evp(d1, p, f1, nsub=2)
        bad = []
        for name in state:
            G = blocks.gather_global(d1, f1[name])[:, rank * w:rank * w + w]
            Lc = f[name][0][1:ny + 1, W + 1:W + w + 1]
            if not np.array_equal(G, Lc):
                jj, ii = np.nonzero(G != Lc)
                bad.append((name, int((G != Lc).sum()), int(jj.min()) + 1, int(ii.min()) + 1, int(ii.max()) + 1))
        assert np.abs(f["uvel"]).max() > 1e-3
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, bad))
    except Exception:
        q.put((rank, ["EXC " + traceback.format_exc()]))


@pytest.mark.parametrize("m", [1, 2, 3])
def test_two_rank_tripole_mirror_slab_gloo(m):
    """world size 2 over gloo: zones of 2m+1 columns, a mirror slab of 2m+1 rows, both refreshed once per m launches -- bit for
    bit the single-domain evp on every rank's own columns"""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_xband_worker, args=(r, 2, port, m, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    for _ in procs:
        res.append(q.get(timeout=300))
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.terminate()
    for rank, bad in res:
        assert not bad, f"rank {rank}: {bad}"


def _vmirror_worker(rank, world, port, m, widths, q):
    """The round-4 form of the protocol above for ANY decomposition: the mirror strip of rank r is a VIRTUAL slab of r's own width
    w that starts at global column nx - i0 - w + 2 (evpk_connect), so that the image of my local column c is its local column
    w - c whatever the slab widths; its columns -- like the columns of my own ghost zones -- are fetched from whichever ranks own
    them (cyclic in x).  Every rank publishes its own physical columns (an all-gather stands for the point-to-point messages
    the library sends: one per partner); the E-W zones take all rows, the mirror strip its top 2m + 3 rows only."""
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          OMP_NUM_THREADS="2")
        import torch
        import torch.distributed as dist
        from cice5_amd import blocks, constants as C, synth
        from oracle import orc
        from tests import util

        dist.init_process_group("gloo", rank=rank, world_size=world)
        nx, ny, ndte = int(sum(widths)), 40, 24
        W = 2 * m + 1
        nylM = max(4, 2 * m + 1)
        starts = np.concatenate([[1], 1 + np.cumsum(widths)]).astype(np.int64)        # slab_i0, 1-based
        w, i0, wmax = int(widths[rank]), int(starts[rank]), int(max(widths))
        m0 = nx - i0 - w + 2                                                            # the mirror slab's first global column
        L = w + 2 * W
        case = synth.SynthCase(nx=nx, ny=ny, land="continents", ns_boundary=C.BND_TRIPOLE)
        dx = blocks.create_distrb_cart(2 * L, ny, 2 * L, ny, ew_boundary_type="open", ns_boundary_type="tripole")
        Iw, Jw = blocks.block_index_windows(dx)
        col = Iw[0].astype(np.int64)                                                    # 0 .. 2L+1
        gcolumn = np.where(col <= L, i0 - W - 1 + col, m0 - W - 1 + (col - L))
        gw = (gcolumn - 1) % nx + 1                                                     # cyclic
        owner = np.searchsorted(starts, gw, side="right") - 1
        local = gw - starts[owner]                                                      # 0-based column inside its owner's slab
        I, J = np.broadcast_arrays(gcolumn[None, :], Jw[0][:, None])
        f = synth.make_block_fields(synth.SynthCase(nx=2 * L, ny=ny, ns_boundary=C.BND_TRIPOLE), dx)
        cache = {}
        for name in synth.GRID_FIELDS + synth.INPUT_FIELDS + synth.MASK_FIELDS:
            f[name][0] = case.field(name, I, J, cache)
        xmin = synth.global_min_dx(case)
        p = orc.make_params(3600.0, ndte, xmin)
        state = ["uvel", "vvel"] + util.SIGMA
        axis = [int(k) for k in np.nonzero((gw == nx // 2) | (gw == nx))[0] if 1 <= k <= 2 * L]
        shape = (1, ny + 2, 2 * L + 2)
        pre = {}

        def patch(ptr, loc, kind, fill, phase):      # the fold's axis columns: see _xband_worker
            if loc != C.LOC_NECORNER or kind != C.KIND_VECTOR:
                return
            a = np.ctypeslib.as_array(ptr, shape=shape)
            if phase == 0:
                pre["top"] = a[0, ny, axis].copy()
            else:
                a[0, ny, axis] = -pre["top"]

        keep = orc.set_halo_callback(patch)
        zone_cols = [k for k in range(1, L + 1) if not (W + 1 <= k <= W + w)]            # my strip's two ghost zones
        mirror_cols = list(range(L + 1, 2 * L + 1))                                      # every column of the mirror strip
        rows = slice(ny - nylM, ny + 2)
        partners = set()
        for launch in range(ndte // 2):
            orc.evp(dx, p, f, nsub=2)
            if (launch + 1) % m:
                continue
            mine = torch.zeros((len(state), ny + 2, wmax), dtype=torch.float64)
            for k, n in enumerate(state):
                mine[k, :, :w] = torch.from_numpy(f[n][0][:, W + 1:W + w + 1].copy())
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            for k, n in enumerate(state):
                for c in zone_cols:
                    f[n][0][:, c] = every[owner[c]][k, :, local[c]].numpy()
                for c in mirror_cols:
                    f[n][0][rows, c] = every[owner[c]][k, rows, local[c]].numpy()
            partners |= {int(owner[c]) for c in mirror_cols}
        orc.set_halo_callback(None)
        del keep
        d1 = blocks.create_distrb_cart(nx, ny, 12, 10, ns_boundary_type="tripole")
        f1 = synth.make_block_fields(case, d1)
        for launch in range(ndte // 2):
            orc.evp(d1, p, f1, nsub=2)
        bad = []
        for name in state:
            G = blocks.gather_global(d1, f1[name])[:, i0 - 1:i0 - 1 + w]
            Lc = f[name][0][1:ny + 1, W + 1:W + w + 1]
            if not np.array_equal(G, Lc):
                jj, ii = np.nonzero(G != Lc)
                bad.append((name, int((G != Lc).sum()), int(jj.min()) + 1, int(ii.min()) + 1, int(ii.max()) + 1))
        assert np.abs(f["uvel"]).max() > 1e-3
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, bad, sorted(partners)))
    except Exception:
        q.put((rank, ["EXC " + traceback.format_exc()], []))


@pytest.mark.parametrize("widths,m", [((28, 20), 2), ((16, 16, 16), 3), ((20, 16, 12), 1)])
def test_virtual_mirror_slab_for_any_decomposition_gloo(widths, m):
    """unequal slabs on two ranks, an odd rank count (the middle rank mirrors onto itself), both: the virtual mirror slab of
    evpk_connect (round 4), zones and mirror rows fetched from their owners -- bit for bit the single-domain evp on every
    rank's own columns; and who supplies a mirror slab: more than one rank as soon as the slabs are not mirror images"""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = len(widths)
    procs = [ctx.Process(target=_vmirror_worker, args=(r, world, port, m, widths, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    for _ in procs:
        res.append(q.get(timeout=300))
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.terminate()
    for rank, bad, partners in res:
        assert not bad, f"rank {rank}: {bad}"
    by_rank = {r: p for r, _, p in res}
    if widths == (16, 16, 16):
        assert 1 in by_rank[1] and len(by_rank[1]) == 3        # the middle rank's own columns, its zones from both neighbours
    if widths == (28, 20):
        assert by_rank[0] == [0, 1] and by_rank[1] == [0, 1]
