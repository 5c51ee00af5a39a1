"""Block decomposition of the global grid: the host-side mirror of ice_blocks / ice_distribution.

* `create_blocks` follows source/ice_blocks.F90:111-316: the global nx x ny grid is cut into
  block_size_x x block_size_y blocks, each stored with one ghost ring (nghost = 1, :43), the
  last block in a direction is padded when the size does not divide (:148-150, :178-179).
* `create_distrb_cart` follows source/ice_distribution.F90:535-680: blocks are dealt to
  ranks as a Cartesian product of contiguous block ranges; blocks with zero work are
  eliminated (land-block elimination, ice_domain.F90:387-441).
* processor shape `slenderX1` (nprocs_x = nprocs, nprocs_y = 1) is the ACCESS-OM2 default
  and is what shards the path over the GPUs of a node: every rank owns a contiguous
  x-slab of whole columns.

Arrays in "block layout" have shape (nblocks, ny_block, nx_block) in C order, which is
the Fortran (nx_block, ny_block, max_blocks) layout of source/ice_state.F90:141-147.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from .constants import BND_CYCLIC, BND_NAMES, BND_TRIPOLE

nghost = 1  # ice_blocks.F90:43


@dataclass
class Block:
    """source/ice_blocks.F90:22-35 (indices are Fortran 1-based, as in the reference)."""

    block_id: int          # global block number, 1-based, x fastest
    iblock: int
    jblock: int
    ilo: int
    ihi: int
    jlo: int
    jhi: int
    tripole: bool
    i_glob: np.ndarray     # (nx_block,) global i of every column, 0 = padding / closed
    j_glob: np.ndarray     # (ny_block,)
    local_id: int = 0

    @property
    def iglob_lo(self) -> int:
        return int(self.i_glob[self.ilo - 1])

    @property
    def jglob_lo(self) -> int:
        return int(self.j_glob[self.jlo - 1])


def _bnd(b) -> int:
    return BND_NAMES[b] if isinstance(b, str) else int(b)


def create_blocks(nx_global: int, ny_global: int, block_size_x: int, block_size_y: int,
                  ew_boundary_type="cyclic", ns_boundary_type="open") -> List[Block]:
    ew, ns = _bnd(ew_boundary_type), _bnd(ns_boundary_type)
    nx_block = block_size_x + 2 * nghost
    ny_block = block_size_y + 2 * nghost
    nblocks_x = (nx_global - 1) // block_size_x + 1
    nblocks_y = (ny_global - 1) // block_size_y + 1
    blocks: List[Block] = []
    n = 0
    for jblock in range(1, nblocks_y + 1):
        js = (jblock - 1) * block_size_y + 1
        for iblock in range(1, nblocks_x + 1):
            n += 1
            is_ = (iblock - 1) * block_size_x + 1
            ilo, jlo = nghost + 1, nghost + 1
            ihi, jhi = nx_block - nghost, ny_block - nghost
            j_glob = np.zeros(ny_block, dtype=np.int32)
            for j in range(1, ny_block + 1):
                jg = js - nghost + j - 1
                if jg < 1:                                   # southern ghost cells (:199-216)
                    if ns == BND_CYCLIC:
                        jg = jg + ny_global
                    elif ns in (1, BND_TRIPOLE):            # open (tripole south is open)
                        jg = nghost - j + 1
                    else:
                        jg = 0
                if jg > ny_global + nghost:                  # padding (:220-221)
                    jg = 0
                elif jg > ny_global:                         # northern ghost cells (:225-240)
                    if ns == BND_CYCLIC:
                        jg = jg - ny_global
                    elif ns == 1:
                        jg = 2 * ny_global - jg + 1
                    elif ns == BND_TRIPOLE:
                        jg = -jg
                    else:
                        jg = 0
                elif jg == ny_global and jlo <= j <= jhi:    # last physical point in padded domain
                    jhi = j
                j_glob[j - 1] = jg
            i_glob = np.zeros(nx_block, dtype=np.int32)
            for i in range(1, nx_block + 1):
                ig = is_ - nghost + i - 1
                if ig < 1:                                   # western ghost cells (:256-269)
                    if ew == BND_CYCLIC:
                        ig = ig + nx_global
                    elif ew == 1:
                        ig = nghost - i + 1
                    else:
                        ig = 0
                if ig > nx_global + nghost:
                    ig = 0
                elif ig > nx_global:                         # eastern ghost cells (:278-289)
                    if ew == BND_CYCLIC:
                        ig = ig - nx_global
                    elif ew == 1:
                        ig = 2 * nx_global - ig + 1
                    else:
                        ig = 0
                elif ig == nx_global and ilo <= i <= ihi:
                    ihi = i
                i_glob[i - 1] = ig
            blocks.append(Block(block_id=n, iblock=iblock, jblock=jblock, ilo=ilo, ihi=ihi, jlo=jlo, jhi=jhi,
                                tripole=(jblock == nblocks_y and ns == BND_TRIPOLE),
                                i_glob=i_glob, j_glob=j_glob))
    return blocks


def proc_decomposition(nprocs: int, processor_shape: str = "slenderX1"):
    """source/ice_distribution.F90 proc_decomposition, the two shapes the BASELINE configs use."""
    if processor_shape == "slenderX1":
        return nprocs, 1
    if processor_shape == "slenderX2":
        if nprocs % 2:
            raise ValueError("slenderX2 needs an even number of ranks")
        return nprocs // 2, 2
    raise ValueError(f"unsupported processor_shape {processor_shape!r}")


@dataclass
class Decomp:
    """One rank's view of the decomposition (ice_domain.F90: nblocks, blocks_ice, distrb_info)."""

    nx_global: int
    ny_global: int
    block_size_x: int
    block_size_y: int
    ew_boundary: int
    ns_boundary: int
    nprocs: int
    rank: int
    all_blocks: List[Block]
    block_location: np.ndarray            # (nblocks_tot,) owning rank + 1, 0 = eliminated
    local_blocks: List[Block] = field(default_factory=list)

    @property
    def nx_block(self) -> int:
        return self.block_size_x + 2 * nghost

    @property
    def ny_block(self) -> int:
        return self.block_size_y + 2 * nghost

    @property
    def nblocks(self) -> int:
        return len(self.local_blocks)

    def geom_arrays(self) -> Dict[str, np.ndarray]:
        g = lambda a: np.ascontiguousarray(np.array(a, dtype=np.int32))
        lb = self.local_blocks
        return dict(ilo=g([b.ilo for b in lb]), ihi=g([b.ihi for b in lb]),
                    jlo=g([b.jlo for b in lb]), jhi=g([b.jhi for b in lb]),
                    iglob_lo=g([b.iglob_lo for b in lb]), jglob_lo=g([b.jglob_lo for b in lb]))

    def slab(self):
        """(i0, i1, j0, j1): 1-based inclusive global extent of this rank's physical cells."""
        lb = self.local_blocks
        if not lb:
            return (1, 0, 1, 0)
        i0 = min(b.iglob_lo for b in lb)
        i1 = max(b.iglob_lo + (b.ihi - b.ilo) for b in lb)
        j0 = min(b.jglob_lo for b in lb)
        j1 = max(b.jglob_lo + (b.jhi - b.jlo) for b in lb)
        return (i0, i1, j0, j1)


def create_distrb_cart(nx_global: int, ny_global: int, block_size_x: int, block_size_y: int,
                       nprocs: int = 1, rank: int = 0, ew_boundary_type="cyclic", ns_boundary_type="open",
                       processor_shape: str = "slenderX1",
                       work_per_block: Optional[Sequence[int]] = None) -> Decomp:
    blocks = create_blocks(nx_global, ny_global, block_size_x, block_size_y, ew_boundary_type, ns_boundary_type)
    nblocks_x = (nx_global - 1) // block_size_x + 1
    nblocks_y = (ny_global - 1) // block_size_y + 1
    npx, npy = proc_decomposition(nprocs, processor_shape)
    nbx_pp = (nblocks_x - 1) // npx + 1          # ice_distribution.F90:603-604
    nby_pp = (nblocks_y - 1) // npy + 1
    loc = np.zeros(len(blocks), dtype=np.int32)
    local: List[Block] = []
    for j in range(1, npy + 1):
        for i in range(1, npx + 1):
            processor = (j - 1) * npx + i
            is_, ie = (i - 1) * nbx_pp + 1, min(i * nbx_pp, nblocks_x)
            js, je = (j - 1) * nby_pp + 1, min(j * nby_pp, nblocks_y)
            lid = 0
            for jb in range(js, je + 1):
                for ib in range(is_, ie + 1):
                    gid = (jb - 1) * nblocks_x + ib
                    if work_per_block is None or work_per_block[gid - 1] != 0:
                        lid += 1
                        loc[gid - 1] = processor
                        if processor == rank + 1:
                            b = blocks[gid - 1]
                            local.append(Block(**{**b.__dict__, "local_id": lid}))
    return Decomp(nx_global, ny_global, block_size_x, block_size_y, _bnd(ew_boundary_type), _bnd(ns_boundary_type),
                  nprocs, rank, blocks, loc, local)


FieldFn = Callable[[np.ndarray, np.ndarray], np.ndarray]


def block_index_windows(d: Decomp):
    """For every local block the unwrapped extended-global (I, J) index of every cell
    (ghost ring included): I = iglob_lo + (i - ilo), so ghosts are 0 / nx+1 style indices
    and padding cells run past the domain.  Shapes (nblocks, nx_block) / (nblocks, ny_block)."""
    nb = d.nblocks
    I = np.zeros((nb, d.nx_block), dtype=np.int64)
    J = np.zeros((nb, d.ny_block), dtype=np.int64)
    for n, b in enumerate(d.local_blocks):
        I[n] = b.iglob_lo + (np.arange(1, d.nx_block + 1) - b.ilo)
        J[n] = b.jglob_lo + (np.arange(1, d.ny_block + 1) - b.jlo)
    return I, J


def to_blocks(d: Decomp, fn: FieldFn, dtype=np.float64) -> np.ndarray:
    """Evaluate a field given as fn(I, J) on extended-global indices into block layout."""
    I, J = block_index_windows(d)
    out = np.zeros((d.nblocks, d.ny_block, d.nx_block), dtype=dtype)
    for n in range(d.nblocks):
        out[n] = fn(I[n][None, :], J[n][:, None])
    return out


def gather_global(d: Decomp, a: np.ndarray, fill=0.0) -> np.ndarray:
    """Physical cells of the local blocks -> (ny_global, nx_global) array (test helper)."""
    G = np.full((d.ny_global, d.nx_global), fill, dtype=a.dtype)
    for n, b in enumerate(d.local_blocks):
        ni, nj = b.ihi - b.ilo + 1, b.jhi - b.jlo + 1
        G[b.jglob_lo - 1:b.jglob_lo - 1 + nj, b.iglob_lo - 1:b.iglob_lo - 1 + ni] = \
            a[n, b.jlo - 1:b.jhi, b.ilo - 1:b.ihi]
    return G
