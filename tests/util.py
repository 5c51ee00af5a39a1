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
