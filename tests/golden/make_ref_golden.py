"""Generates tests/golden/ref_<config>.npz: outputs of the REFERENCE'S OWN routines (oracle/_ref, built from the unmodified
sources under /root/reference by oracle/ref/Makefile) on the deterministic inputs of tests/golden/refvec.py.

Runs in the build container only (the reference does not travel); the fixtures it writes are data -- block descriptors,
distributions, halo-updated arrays, bound_state and ice_strength results -- and are committed.  Re-run after changing
refvec.py:     python tests/golden/make_ref_golden.py
"""
from __future__ import annotations

import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import refvec as rv  # noqa: E402

REFDIR = os.path.join(ROOT, "oracle", "ref")
NTRCR_BOUND = 3


def build(cfg):
    nx, ny, bx, by, mxb = rv.CONFIGS[cfg]
    exe = os.path.join(ROOT, "oracle", "_ref", cfg, "ref_harness")
    subprocess.check_call(["make", "-C", REFDIR, f"CFG={cfg}", f"NX={nx}", f"NY={ny}", f"BX={bx}", f"BY={by}", f"MXB={mxb}",
                           f"NCAT={rv.NCAT}"], stdout=subprocess.DEVNULL)
    return exe


class Writer:
    def __init__(self):
        self.parts = []

    def i4(self, *v):
        self.parts.append(np.asarray(v, dtype=np.int32).tobytes())

    def r8(self, *v):
        self.parts.append(np.asarray(v, dtype=np.float64).tobytes())

    def arr(self, a):
        self.parts.append(np.ascontiguousarray(a).tobytes())

    def raw(self, b):
        self.parts.append(b)


class Reader:
    def __init__(self, buf):
        self.b, self.o = buf, 0

    def take(self, dtype, shape):
        n = int(np.prod(shape)) if shape else 1
        a = np.frombuffer(self.b, dtype=dtype, count=n, offset=self.o).reshape(shape).copy()
        self.o += a.nbytes
        return a


def run_case(exe, cfg, ew, ns, land, with_strength):
    nx, ny, bx, by, mxb = rv.CONFIGS[cfg]
    nxb, nyb = bx + 2, by + 2
    case = rv.case_name(ew, ns, land)
    kmt, ulat = rv.kmt_ulat(nx, ny, bx, by, ew, ns, land)
    nbx, nby = (nx - 1) // bx + 1, (ny - 1) // by + 1
    nbt = nbx * nby
    # blocks without an ocean cell are eliminated from the distribution (ice_domain.F90:387-441)
    work = np.array([int(kmt[jb * by:(jb + 1) * by, ib * bx:(ib + 1) * bx].sum() > 0) for jb in range(nby) for ib in range(nbx)], dtype=np.int32)
    nblocks = int(work.sum())
    w = Writer()
    w.arr(kmt); w.arr(ulat)
    plan = []
    for key, nz, loc, typ, fill in rv.HALO_R8:
        a = rv.halo_r8_input(cfg, case, key, nblocks, nyb, nxb, nz)
        w.i4(1, nz, loc, typ, 0 if fill is None else 1); w.r8(0.0 if fill is None else fill); w.arr(a)
        plan.append((f"halo_r8/{key}", np.float64, a.shape))
    for key, loc, typ, fill in rv.HALO_I4:
        a = rv.halo_i4_input(cfg, case, key, nblocks, nyb, nxb)
        w.i4(2, loc, typ, 0 if fill is None else 1, 0 if fill is None else fill); w.arr(a)
        plan.append((f"halo_i4/{key}", np.int32, a.shape))
    a1 = rv.halo_r8_input(cfg, case, "stress1", nblocks, nyb, nxb, 0)
    a2 = rv.halo_r8_input(cfg, case, "stress2", nblocks, nyb, nxb, 0)
    w.i4(3, rv.LOC["center"], rv.TYPE["scalar"]); w.arr(a1); w.arr(a2)
    plan.append(("halo_stress/center_scalar", np.float64, a1.shape))
    do_bound = ew == "cyclic" and ns in ("open", "tripole")
    if do_bound:
        aicen, vicen, vsnon, trcrn = rv.state_input(cfg, case, mxb, nyb, nxb, NTRCR_BOUND)
        w.i4(4, NTRCR_BOUND); w.arr(aicen); w.arr(vicen); w.arr(vsnon); w.arr(trcrn)
        plan += [("bound/aicen", np.float64, aicen.shape), ("bound/vicen", np.float64, vicen.shape),
                 ("bound/vsnon", np.float64, vsnon.shape), ("bound/trcrn", np.float64, trcrn.shape)]
    if nbt > 1:
        for npz_, shape in rv.DISTRIBUTIONS:
            sb = shape.encode()
            w.i4(5, npz_, len(sb)); w.raw(sb); w.arr(work)
            plan += [(f"distrb/{npz_}_{shape}/blockLocation", np.int32, (nbt,)), (f"distrb/{npz_}_{shape}/blockLocalID", np.int32, (nbt,))]
    if with_strength:
        for ks, kp, kr in rv.STRENGTH_CASES:
            for rep in (0, 1):
                tag = f"k{ks}{kp}{kr}_{rep}"
                s = rv.strength_input(cfg, tag, nyb, nxb)
                idx = np.zeros((2, nxb * nyb), dtype=np.int32)
                n = len(s["indxi"])
                idx[0, :n], idx[1, :n] = s["indxi"], s["indxj"]
                w.i4(6, ks, kp, kr); w.r8(rv.MU_RDG, rv.CF)
                w.i4(2, nxb - 1, 2, nyb - 1, n); w.arr(idx[0]); w.arr(idx[1])
                for k in ("aice", "vice", "aice0", "aicen", "vicen"):
                    w.arr(s[k])
                plan.append((f"strength/{tag}", np.float64, (nyb, nxb)))
    if with_strength:
        for tag, (dep, n_tsfc, n_alvl, n_apnd, n_fbri, pond) in rv.TRACER_CASES.items():
            nt = len(dep)
            a, v, sn, atr = rv.tracers_input(cfg, tag, nyb, nxb, nt)
            idx = np.zeros((2, nxb * nyb), dtype=np.int32)
            jj, ii = np.meshgrid(np.arange(1, nyb + 1), np.arange(1, nxb + 1), indexing="ij")     # work_to_state's list: every cell, i fastest
            idx[0], idx[1] = ii.ravel(), jj.ravel()
            w.i4(8, nt); w.arr(np.asarray(dep, dtype=np.int32)); w.i4(n_tsfc, n_alvl, n_apnd, n_fbri, *pond); w.r8(rv.TOCNFRZ); w.i4(nxb * nyb)
            w.arr(idx[0]); w.arr(idx[1])
            w.arr(atr.reshape(nt, nyb * nxb))                   # Fortran atrcrn(icells, ntrcr)
            w.arr(a); w.arr(v); w.arr(sn)
            plan.append((f"tracers/{tag}", np.float64, (nt, nyb, nxb)))
    # global_minval over the ocean T cells (set_evp_parameters, ice_dyn_shared.F90:221-222)
    dx = 1000.0 * (1.0 + rv.halo_r8_input(cfg, case, "minval", nblocks, nyb, nxb, 0) ** 2)
    msk = (rv.halo_i4_input(cfg, case, "minval_mask", nblocks, nyb, nxb) % 10 > 3).astype(np.int32)
    w.i4(7); w.arr(dx); w.arr(msk)
    plan.append(("global_minval", np.float64, ()))
    w.i4(0)

    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "in.bin"), "wb") as f:
            f.write(b"".join(w.parts))
        with open(os.path.join(td, "cice_in.nml"), "w") as f:
            f.write("&domain_nml\n  nprocs = 1\n  processor_shape = 'slenderX1'\n  distribution_type = 'cartesian'\n"
                    "  distribution_wght = 'latitude'\n"
                    f"  ew_boundary_type = '{ew}'\n  ns_boundary_type = '{ns}'\n"
                    "  maskhalo_dyn = .false.\n  maskhalo_remap = .false.\n  maskhalo_bound = .false.\n/\n")
        p = subprocess.run([exe, "in.bin", "out.bin"], cwd=td, capture_output=True, text=True)
        if p.returncode != 0 or not os.path.exists(os.path.join(td, "out.bin")):
            return None, (p.stdout[-600:] + p.stderr[-600:])
        buf = open(os.path.join(td, "out.bin"), "rb").read()
    r = Reader(buf)
    out = {}
    hdr = r.take(np.int32, (7,))
    assert tuple(hdr[:5]) == (nx, ny, nxb, nyb, mxb) and hdr[5] == rv.NCAT and hdr[6] == rv.MAX_NTRCR, hdr
    nbt_r, nbx_r, nby_r, nbl = r.take(np.int32, (4,))
    assert nbt_r == nbt and nbl == nblocks and (nbx_r, nby_r) == (nbx, nby), (nbt_r, nbt, nbl, nblocks)
    desc = np.zeros((nbt, 8), dtype=np.int32)
    ig = np.zeros((nbt, nxb), dtype=np.int32)
    jg = np.zeros((nbt, nyb), dtype=np.int32)
    for n in range(nbt):
        desc[n] = r.take(np.int32, (8,)); ig[n] = r.take(np.int32, (nxb,)); jg[n] = r.take(np.int32, (nyb,))
    out["blocks/desc"] = desc          # block_id, iblock, jblock, ilo, ihi, jlo, jhi, tripole
    out["blocks/i_glob"], out["blocks/j_glob"] = ig, jg
    out["blocks/nblocks_xy"] = np.array([nbx, nby], dtype=np.int32)
    out["blocks/blocks_ice"] = r.take(np.int32, (nblocks,))
    out["blocks/blockLocation"] = r.take(np.int32, (nbt,))
    out["blocks/blockLocalID"] = r.take(np.int32, (nbt,))
    for key, dt, shape in plan:
        a = r.take(dt, shape)
        if key == "bound/trcrn":
            untouched = np.array_equal(a[:, :, NTRCR_BOUND:], trcrn[:, :, NTRCR_BOUND:])
            out["bound/trcrn_beyond_ntrcr_untouched"] = np.array(untouched)
            a = np.ascontiguousarray(a[:, :, :NTRCR_BOUND])
        if key.startswith("bound/"):
            a = np.ascontiguousarray(a[:nblocks])
        out[key] = a
    assert r.o == len(buf), (r.o, len(buf))
    return out, p.stdout[-300:]


def main():
    for ci, cfg in enumerate(rv.CONFIGS):
        exe = build(cfg)
        allout = {}
        nbt = ((rv.CONFIGS[cfg][0] - 1) // rv.CONFIGS[cfg][2] + 1) * ((rv.CONFIGS[cfg][1] - 1) // rv.CONFIGS[cfg][3] + 1)
        first = True
        for ew, ns, land in rv.BOUNDARIES:
            if land == "landblock" and nbt == 1:
                continue
            case = rv.case_name(ew, ns, land)
            out, log = run_case(exe, cfg, ew, ns, land, with_strength=first)
            if out is None:
                # the reference itself refuses this combination (abort_ice): recorded, so that the tests know it is not a gap
                print(f"{cfg} {case}: REFERENCE ABORTS: {log.strip().splitlines()[-3:]}")
                allout[f"{case}/aborted"] = np.array(True)
                continue
            first = False
            for k, v in out.items():
                allout[f"{case}/{k}"] = v
            print(f"{cfg} {case}: {len(out)} arrays")
        path = os.path.join(HERE, f"ref_{cfg}.npz")
        np.savez_compressed(path, **allout)
        print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
