"""The EVP state as the reference's binary restart carries it.

Reference: `dumpfile` / `restartfile`, source/ice_restart_driver.F90:118-176 (write) and :290-412 (read);
`write_restart_field` -> `ice_write(..., 'ruf8', ...)`, io_binary/ice_restart.F90:641-690: every field is ONE Fortran
sequential unformatted record holding the gathered (nx_global, ny_global) real*8 array (land blocks filled), written
big-endian under the production flag `-convert big_endian` (bld/Macros.nci:17-25).  The records of the dynamics are

    uvel, vvel, [radiation fields], strocnxT, strocnyT,
    stressp_1, stressp_3, stressp_2, stressp_4, stressm_1, stressm_3, stressm_2, stressm_4,
    stress12_1, stress12_3, stress12_2, stress12_4,                 (order 1,3,2,4: pairs across the tripole cut)
    iceumask as real 0/1 (read back as  > 0.5)

This module writes and reads that subsequence so that a run can be restarted bit-for-bit across the CPU reference and
this library.  Physical cells only: after a read the host model fills ghost cells as the reference does
(`scatter_global` + `ice_HaloUpdate_stress`, ice_restart_driver.F90:370-395); for cyclic / open boundaries
`read_dynamics_records` fills them itself.
"""
from __future__ import annotations

import struct
from typing import BinaryIO, Dict

import numpy as np

from . import constants as C
from .blocks import Decomp, block_index_windows, gather_global

VELOCITY = ["uvel", "vvel"]
OCEAN_STRESS = ["strocnxT", "strocnyT"]
INTERNAL_STRESS = [f"{k}_{c}" for k in ("stressp", "stressm", "stress12") for c in (1, 3, 2, 4)]
DYNAMICS_RECORDS = VELOCITY + OCEAN_STRESS + INTERNAL_STRESS + ["iceumask"]


def _write_record(fh: BinaryIO, a: np.ndarray, byteorder: str):
    raw = np.ascontiguousarray(a, dtype=byteorder + "f8").tobytes()
    mark = struct.pack(byteorder + "i", len(raw))
    fh.write(mark + raw + mark)


def _read_record(fh: BinaryIO, shape, byteorder: str) -> np.ndarray:
    n = int(np.prod(shape)) * 8
    head = fh.read(4)
    if len(head) != 4 or struct.unpack(byteorder + "i", head)[0] != n:
        raise ValueError("restart record: unexpected length marker (wrong grid size or byte order?)")
    a = np.frombuffer(fh.read(n), dtype=byteorder + "f8").reshape(shape).astype(np.float64)
    if struct.unpack(byteorder + "i", fh.read(4))[0] != n:
        raise ValueError("restart record: trailing length marker does not match")
    return a


def write_dynamics_records(fh: BinaryIO, d: Decomp, f: Dict[str, np.ndarray], byteorder: str = ">"):
    """Single-rank writer (the reference gathers to master_task first): one record per field of DYNAMICS_RECORDS."""
    if d.nprocs != 1:
        raise ValueError("gather the blocks on one rank first (ice_gather_scatter.F90:gather_global)")
    for name in DYNAMICS_RECORDS:
        a = f[name]
        if name == "iceumask":
            a = np.where(a != 0, 1.0, 0.0)                                  # ice_restart_driver.F90:163-174
        _write_record(fh, gather_global(d, np.asarray(a, dtype=np.float64)), byteorder)


def read_dynamics_records(fh: BinaryIO, d: Decomp, f: Dict[str, np.ndarray], byteorder: str = ">"):
    """Reads DYNAMICS_RECORDS into the block arrays of `f` (every rank reads the whole file and keeps its blocks).
    Ghost cells: cyclic E-W wrap, zero outside open / closed boundaries; on a tripole grid the north ghost row is left
    to the caller's halo update."""
    nx, ny = d.nx_global, d.ny_global
    I, J = block_index_windows(d)
    for name in DYNAMICS_RECORDS:
        G = _read_record(fh, (ny, nx), byteorder)
        out = np.zeros((d.nblocks, d.ny_block, d.nx_block))
        for n in range(d.nblocks):
            ii, jj = I[n].copy(), J[n].copy()
            okx = (ii >= 1) & (ii <= nx)
            if d.ew_boundary == C.BND_CYCLIC:
                ii = (ii - 1) % nx + 1
                okx = np.ones_like(okx)
            oky = (jj >= 1) & (jj <= ny)
            blk = G[np.clip(jj, 1, ny)[:, None] - 1, np.clip(ii, 1, nx)[None, :] - 1]
            out[n] = np.where(oky[:, None] & okx[None, :], blk, 0.0)
        if name == "iceumask":
            f[name][...] = (out > 0.5).astype(f[name].dtype)                 # :399-409
        else:
            f[name][...] = out
