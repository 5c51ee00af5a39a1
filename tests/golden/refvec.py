"""Deterministic test inputs shared by tests/golden/make_ref_golden.py (which feeds them to the reference build,
oracle/_ref) and by the tests (which feed the same bytes to the C restatement and to the HIP path).  Only the OUTPUTS of
the reference are stored in tests/golden/ref_*.npz; the inputs are regenerated from (seed, shape) by the integer hash
below, which does not depend on any library's random stream.
"""
from __future__ import annotations

import numpy as np

# the builds of oracle/ref/Makefile the fixtures come from: name -> (NX, NY, BX, BY, MXB)
CONFIGS = {
    "g24x16_b24x16": (24, 16, 24, 16, 1),       # one block
    "g24x16_b6x4": (24, 16, 6, 4, 16),          # 4 x 4 blocks
    "g26x18_b8x5": (26, 18, 8, 5, 16),          # 4 x 4 blocks, the last ones padded in x and in y
}
NCAT = 5
MAX_NTRCR = 20              # ice_domain_size.F90:38-52 with the defines of oracle/ref/Makefile

# (ew_boundary_type, ns_boundary_type, land pattern)
BOUNDARIES = [
    ("cyclic", "open", "none"), ("cyclic", "closed", "rim"), ("cyclic", "tripole", "none"),
    ("open", "open", "none"), ("open", "closed", "rim"),
    ("closed", "open", "rim"), ("closed", "closed", "rim"),
    # (tripole grids are cyclic E-W: with 'open' / 'closed' the reference's copy out of the tripole buffer follows mirrored
    #  ghost indices resp. reads column nx_global + 1 of the buffer, serial/ice_boundary.F90:3752-3776, :3420-3424)
    ("cyclic", "open", "landblock"), ("cyclic", "tripole", "landblock"),
]

LOC = {"center": 1, "necorner": 2, "nface": 3, "eface": 4}           # ice_constants.F90 field_loc_*
TYPE = {"scalar": 1, "vector": 2, "angle": 3}                        # field_type_*

# the halo updates every case runs: (key, nz, loc, type, fill or None)
HALO_R8 = [(f"{l}_{t}", 0, LOC[l], TYPE[t], None) for l in LOC for t in ("scalar", "vector")] + [
    ("center_angle", 0, LOC["center"], TYPE["angle"], None),
    ("center_scalar_fill", 0, LOC["center"], TYPE["scalar"], -9.5),
    ("necorner_vector_fill", 0, LOC["necorner"], TYPE["vector"], 3.25),
    ("center_scalar_3d", 3, LOC["center"], TYPE["scalar"], None),
    ("necorner_vector_3d", 2, LOC["necorner"], TYPE["vector"], None),
]
HALO_I4 = [("center_scalar", LOC["center"], TYPE["scalar"], None), ("center_scalar_fill", LOC["center"], TYPE["scalar"], 7)]


def case_name(ew, ns, land):
    return f"{ew}_{ns}" + ("" if land in ("none", "rim") else f"_{land}")


def _splitmix(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def hash01(shape, seed: int) -> np.ndarray:
    """float64 in [0, 1) with full 53-bit mantissas, a pure function of (seed, flat index)"""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        k = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001B3)
        h = _splitmix(_splitmix(k))
    return ((h >> np.uint64(11)).astype(np.float64) * 2.0 ** -53).reshape(shape)


def seed_of(*parts) -> int:
    s = 1469598103934665603
    for p in parts:
        for ch in str(p):
            s = ((s ^ ord(ch)) * 1099511628211) % (1 << 63)
    return s % (1 << 40)


def halo_r8_input(cfg, case, key, nblocks, ny_block, nx_block, nz):
    """every cell, ghost cells included, gets its own value in (-1, 1): a ghost cell the update leaves alone is then visible"""
    shape = (nblocks, nz, ny_block, nx_block) if nz else (nblocks, ny_block, nx_block)
    return np.ascontiguousarray(2.0 * hash01(shape, seed_of(cfg, case, "r8", key)) - 1.0)


def halo_i4_input(cfg, case, key, nblocks, ny_block, nx_block):
    n = nblocks * ny_block * nx_block          # every cell its own value (a cell the update leaves alone stays recognisable)
    return np.ascontiguousarray((hash01((nblocks, ny_block, nx_block), seed_of(cfg, case, "i4", key)) * 5).astype(np.int32) - 1
                                + 10 * np.arange(1, n + 1, dtype=np.int32).reshape(nblocks, ny_block, nx_block))


def kmt_ulat(nx, ny, bx, by, ew, ns, land):
    """KMTG (1 ocean / 0 land) and ULATG (radians) handed to init_domain_distribution (ice_domain.F90:248)"""
    kmt = np.ones((ny, nx))
    if land in ("rim",):
        if ns == "closed":
            kmt[:2, :] = 0; kmt[-2:, :] = 0
        if ew == "closed":
            kmt[:, :2] = 0; kmt[:, -2:] = 0
    if land == "landblock":          # block (iblock, jblock) = (2, 2) is all land: eliminated from the distribution
        kmt[by:2 * by, bx:2 * bx] = 0
    ulat = np.deg2rad(np.linspace(-80.0, 88.0, ny))[:, None] + np.zeros((1, nx))
    return kmt, ulat


def state_input(cfg, case, nblocks, ny_block, nx_block, ntrcr):
    """aicen, vicen, vsnon (nb, ncat, ny, nx), trcrn (nb, ncat, MAX_NTRCR, ny, nx) for bound_state: every cell its own value"""
    s = lambda k, shape: np.ascontiguousarray(hash01(shape, seed_of(cfg, case, "state", k)))
    a = s("aicen", (nblocks, NCAT, ny_block, nx_block)) * 0.2
    a[a < 0.05] = 0.0                                                  # categories without ice
    v = s("vicen", (nblocks, NCAT, ny_block, nx_block)) * 2.0
    sn = s("vsnon", (nblocks, NCAT, ny_block, nx_block)) * 0.3
    t = s("trcrn", (nblocks, NCAT, MAX_NTRCR, ny_block, nx_block)) * 4.0 - 2.0
    return a, v, sn, t


def strength_input(cfg, tag, ny_block, nx_block):
    """a thickness distribution per cell that exercises ice_strength's branches: open-water fractions on both sides of
    Gstar = 0.15, empty categories, thin and thick ice; plus the list of cells it is evaluated on"""
    h = lambda k, shape: hash01(shape, seed_of(cfg, "strength", tag, k))
    hin = np.array([0.0, 0.64, 1.39, 2.47, 4.57, 9.0])
    aicen = h("a", (NCAT, ny_block, nx_block))
    aicen[h("hole", (NCAT, ny_block, nx_block)) < 0.25] = 0.0
    tot = aicen.sum(axis=0)
    target = h("tot", (ny_block, nx_block)) ** 0.3                      # mostly compact ice, some open cells
    target[h("full", (ny_block, nx_block)) < 0.15] = 1.0
    with np.errstate(invalid="ignore", divide="ignore"):
        aicen = np.where(tot > 0, aicen * (target / tot), 0.0)
    frac = h("hh", (NCAT, ny_block, nx_block))
    hi = hin[:-1, None, None] + frac * (hin[1:, None, None] - hin[:-1, None, None])
    vicen = aicen * hi
    aicen[aicen < 1e-11] = 0.0                                          # at or below puny: both branches of `aicen > puny`
    tiny = h("tiny", (NCAT, ny_block, nx_block)) < 0.03
    aicen[tiny] = 0.5e-11
    aice = aicen.sum(axis=0)
    vice = vicen.sum(axis=0)
    aice0 = np.maximum(1.0 - aice, 0.0)
    sel = (h("sel", (ny_block, nx_block)) < 0.9) & (aice > 1e-3)
    sel[0, :] = sel[-1, :] = False; sel[:, 0] = sel[:, -1] = False       # icetmask lives on physical cells
    jj, ii = np.nonzero(sel)                                            # j outer, i inner: the order of the reference's list
    c = np.ascontiguousarray
    return dict(aice=c(aice), vice=c(vice), aice0=c(aice0), aicen=c(aicen), vicen=c(vicen),
                indxi=(ii + 1).astype(np.int32), indxj=(jj + 1).astype(np.int32))


STRENGTH_CASES = [(1, 1, 1), (1, 0, 1), (1, 1, 0), (1, 0, 0), (0, 1, 1)]       # (kstrength, krdg_partic, krdg_redist)
MU_RDG, CF = 3.0, 17.0                                                          # ice_init.F90:273-277 defaults
DISTRIBUTIONS = [(2, "slenderX1"), (4, "slenderX1"), (3, "slenderX1"), (4, "slenderX2"), (8, "slenderX2")]


# compute_tracers (ice_itd.F90:1359): tracer tables (trcr_depend, nt_Tsfc, nt_alvl, nt_apnd, nt_fbri, (cesm, lvl, topo))
# Tsfc, qice, qsno, alvl, vlvl, apnd (on alvl), hpnd (on apnd), fbri, a brine tracer (on fbri)
TRACER_CASES = {
    "lvl_ponds": ([0, 1, 2, 0, 1, 2 + 4, 2 + 6, 1, 2 + 8], 1, 4, 6, 8, (0, 1, 0)),
    "cesm_ponds": ([0, 1, 2, 0, 1, 0, 2 + 6, 1, 2 + 8], 1, 4, 6, 8, (1, 0, 0)),
    "plain": ([0, 1, 1, 2, 0], 1, 0, 0, 0, (0, 0, 0)),
}
TOCNFRZ = -1.8


def tracers_input(cfg, tag, ny_block, nx_block, ntrcr):
    h = lambda k, shape: hash01(shape, seed_of(cfg, "tracers", tag, k))
    a = h("a", (ny_block, nx_block)); a[a < 0.3] = 0.0; a[(a > 0.3) & (a < 0.35)] = 0.5e-11
    v = h("v", (ny_block, nx_block)) * 2.0; v[h("v0", (ny_block, nx_block)) < 0.25] = 0.0
    sn = h("s", (ny_block, nx_block)) * 0.4; sn[h("s0", (ny_block, nx_block)) < 0.3] = 0.0
    atr = h("atr", (ntrcr, ny_block, nx_block)) * 2.0 - 0.7
    atr[3] = np.abs(atr[3]) if ntrcr > 3 else 0        # (alvl, apnd, fbri products: positive where there is ice)
    if ntrcr > 7:
        atr[5] = np.abs(atr[5]); atr[7] = np.abs(atr[7])
        atr[5][h("p0", (ny_block, nx_block)) < 0.2] = 0.0
    c = np.ascontiguousarray
    return c(a), c(v), c(sn), c(atr)
