"""Shared helpers of the parity tests: build a synthetic case in block layout, run the
CPU oracle and the HIP path on identical bytes, compare cell sets the reference defines."""
from __future__ import annotations

import copy
from typing import Dict, Iterable, Optional

import numpy as np

from cice5_amd import blocks, constants as C, synth

SIGMA = synth.STRESS_FIELDS
# cells the reference leaves defined after evp():
ALL_CELLS = ["uvel", "vvel", "strength"]           # ghost cells halo-updated (ice_dyn_evp.F90:392-407, :311-312)
NE_CELLS = SIGMA                                   # physical + N/E ghost T cells (ice_dyn_shared.F90:528-537)
PHYS_CELLS = ["divu", "shear", "rdg_conv", "rdg_shear", "prs_sig", "strintx", "strinty", "strocnx", "strocny",
              "strocnxT", "strocnyT", "strairx", "strairy", "strtltx", "strtlty", "fm", "tmass", "aiu", "umass",
              "uvel_init", "vvel_init", "iceumask", "icetmask"]


def make_case(nx, ny, bsx, bsy, *, nprocs=1, rank=0, ns="open", **kw):
    case = synth.SynthCase(nx=nx, ny=ny, ns_boundary=C.BND_NAMES[ns], **kw)
    d = blocks.create_distrb_cart(nx, ny, bsx, bsy, nprocs=nprocs, rank=rank, ns_boundary_type=ns)
    f = synth.make_block_fields(case, d)
    return case, d, f


def clone(f: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    return {k: v.copy() for k, v in f.items()}


def cell_mask(d, kind: str) -> np.ndarray:
    m = np.zeros((d.nblocks, d.ny_block, d.nx_block), dtype=bool)
    for n, b in enumerate(d.local_blocks):
        if kind == "all":
            m[n, :b.jhi + 1, :b.ihi + 1] = True
        elif kind == "ne":
            m[n, b.jlo - 1:b.jhi + 1, b.ilo - 1:b.ihi + 1] = True
        else:
            m[n, b.jlo - 1:b.jhi, b.ilo - 1:b.ihi] = True
    return m


def compare(d, got: Dict[str, np.ndarray], ref: Dict[str, np.ndarray], names: Optional[Iterable[str]] = None):
    """Returns a list of (name, n_mismatch, max_abs_diff) for fields that are not bit-identical
    (signed zeros compare equal, NaN equals NaN)."""
    bad = []
    masks = {"all": cell_mask(d, "all"), "ne": cell_mask(d, "ne"), "phys": cell_mask(d, "phys")}
    for grp, kind in ((ALL_CELLS, "all"), (NE_CELLS, "ne"), (PHYS_CELLS, "phys")):
        for n in grp:
            if names is not None and n not in names:
                continue
            a, b = got[n][masks[kind]], ref[n][masks[kind]]
            neq = ~((a == b) | (np.isnan(a) & np.isnan(b))) if a.dtype.kind == "f" else (a != b)
            if neq.any():
                diff = np.abs(a[neq].astype(np.float64) - b[neq].astype(np.float64))
                bad.append((n, int(neq.sum()), float(np.nanmax(diff))))
    return bad


def remap_case(nx, ny, bsx, bsy, ns="open", ncat=3, seed=4, dt=3600.0, trcr_depend=(0, 1, 2 + 1), ew="cyclic", land="continents"):
    """a state for horizontal_remap: areas aim(0:ncat) summing to 1 over ocean, hice / hsno (type 1), a surface tracer on
    the area (depend 0), one on the ice volume (depend 1: type 2 on hice) and one on the first tracer (type 2 / 3), and a
    smooth velocity field that vanishes on land; ghost cells current (halo updates of the oracle)"""
    from oracle import orc
    case = synth.SynthCase(nx=nx, ny=ny, ns_boundary=C.BND_NAMES[ns], ew_boundary=C.BND_NAMES[ew], land=land)
    d = blocks.create_distrb_cart(nx, ny, bsx, bsy, ew_boundary_type=ew, ns_boundary_type=ns)
    f = synth.make_block_fields(case, d)
    synth.add_remap_grid(case, d, f)
    I, J = blocks.block_index_windows(d)
    ttype, depend, has = orc.remap_tables(list(trcr_depend))
    ntrace = len(ttype)
    mm = np.zeros((d.nblocks, ncat + 1, d.ny_block, d.nx_block))
    tm = np.zeros((d.nblocks, ncat, ntrace, d.ny_block, d.nx_block))
    for b in range(d.nblocks):
        Ig = ((I[b] - 1) % nx + 1)[None, :] + 0 * J[b][:, None]
        Jg = J[b][:, None] + 0 * I[b][None, :]
        x, y = 2 * np.pi * Ig / nx, np.pi * Jg / ny
        ocean = f["tmask"][b] > 0
        ice = ocean & (np.sin(3 * x + 0.5) * np.cos(2 * y) > -0.3)
        tot = np.zeros_like(x)
        for n in range(1, ncat + 1):
            a = np.where(ice, 0.25 * (1 + 0.8 * np.sin(n * x + y)) / ncat * 2.0, 0.0)
            a = np.where(np.sin(5 * x * n + 2 * y) > 0.7, 0.0, a)          # holes: categories without ice
            mm[b, n] = a
            tot += a
            tm[b, n - 1, 0] = np.where(a > 0, n * (0.5 + 0.3 * np.cos(2 * x - y)), 0.0)        # hice
            tm[b, n - 1, 1] = np.where(a > 0, 0.1 * (1 + 0.5 * np.sin(x + 3 * y)), 0.0)        # hsno
            for k in range(2, ntrace):
                tm[b, n - 1, k] = np.where(a > 0, -2.0 - k + np.sin(k * x) * np.cos(y + n), 0.0)
        mm[b, 0] = np.where(ocean, 1.0 - tot, 0.0)
        f["uvel"][b] = 0.3 * np.sin(2 * x) * np.cos(y) * f["umask"][b]
        f["vvel"][b] = 0.2 * np.cos(3 * x + 1.0) * np.sin(2 * y) * f["umask"][b]
    for arr in (mm.reshape(d.nblocks, -1, d.ny_block, d.nx_block), tm.reshape(d.nblocks, -1, d.ny_block, d.nx_block)):
        for k in range(arr.shape[1]):
            w = np.ascontiguousarray(arr[:, k]); orc.halo_r8(d, w, C.LOC_CENTER, C.KIND_SCALAR, 0.0); arr[:, k] = w
    for n in ("uvel", "vvel"):
        orc.halo_r8(d, f[n], C.LOC_NECORNER, C.KIND_VECTOR, 0.0)
    return case, d, f, mm, tm, (ttype, depend, has)
