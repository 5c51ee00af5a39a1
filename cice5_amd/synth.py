"""Synthetic EVP workloads: grid metrics, land mask, ice state and forcing
given as functions of the extended-global cell index (I, J), so any rank can evaluate
exactly its own blocks (ghost ring included) and every decomposition sees the same bytes.

Grid: analytic lat-lon-like lengths HTN (north face) / HTE (east face); the derived
metric terms use the formulas of source/ice_grid.F90:338-369 (tarea..cxm),
:1436-1465 (dxu, dxt from HTN) and :1506-1541 (dyu, dyt from HTE).  Masks follow
makemask, :1555-1625 (uvm = min of the four surrounding hm).

Random parts come from an integer hash of (I, J, field) instead of a sequential PCG64
stream, so that a rank never has to generate the whole 3600x2700 grid.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import numpy as np

from . import constants as C
from .blocks import Decomp, to_blocks

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _hash01(I, J, salt: int, seed: int):
    """Uniform [0,1) from a splitmix64-style mix of the global index."""
    with np.errstate(over="ignore"):
        x = (I.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ \
            (J.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)) ^ \
            np.uint64((salt * 0x165667B19E3779F9 + seed * 0x27D4EB2F165667C5) & 0xFFFFFFFFFFFFFFFF)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


@dataclass
class SynthCase:
    nx: int
    ny: int
    ew_boundary: int = C.BND_CYCLIC
    ns_boundary: int = C.BND_OPEN
    seed: int = 20261003
    land: str = "rows"         # "rows": 2 land rows top & bottom (rectgrid, ice_grid.F90:1248-1254); "continents": + ~30 % land
    ice: str = "polar"         # "polar": ice where |lat| > ~55 deg;  "full": ice on every ocean cell
    dt: float = 3600.0
    ndte: int = 120
    lat_s: float = -78.0
    lat_n: float = 88.0
    land_band: tuple = ()      # (i_lo, i_hi): these global columns are land on every row (a rank whose block columns are all eliminated)

    # ---- index helpers -------------------------------------------------------------
    def _wrap(self, I):
        if self.ew_boundary == C.BND_CYCLIC:
            return (I - 1) % self.nx + 1
        return I

    def _inside(self, I, J):
        ok = (J >= 1) & (J <= self.ny)
        if self.ew_boundary != C.BND_CYCLIC:
            ok = ok & (I >= 1) & (I <= self.nx)
        return ok

    def _latT(self, J):
        return self.lat_s + (J - 0.5) * ((self.lat_n - self.lat_s) / self.ny)

    def _latU(self, J):
        return self.lat_s + J * ((self.lat_n - self.lat_s) / self.ny)

    # ---- primary grid lengths (m) --------------------------------------------------
    def HTN(self, I, J):
        dx0 = 111.0e3 * 360.0 / self.nx
        I, J = np.broadcast_arrays(I, J)
        return dx0 * np.maximum(np.cos(np.deg2rad(self._latU(J))), 0.05) + 0.0 * I

    def HTE(self, I, J):
        dy0 = 111.0e3 * (self.lat_n - self.lat_s) / self.ny
        I, J = np.broadcast_arrays(I, J)
        Iw = self._wrap(I)
        return dy0 * (1.0 + 0.1 * np.sin(2.0 * np.pi * Iw / self.nx) * np.cos(np.deg2rad(self._latT(J))))

    # ---- masks ---------------------------------------------------------------------
    def _fold(self, I, J):
        """tripole: a cell-centre index beyond the north edge is the 180-degree image of a physical
        cell: (i, ny+k) -> (nx-i+1, ny-k+1)  (centre fold of serial/ice_boundary.F90:804-807)."""
        if self.ns_boundary != C.BND_TRIPOLE:
            return I, J
        I, J = np.broadcast_arrays(I, J)
        up = J > self.ny
        Iw = self._wrap(I)
        return np.where(up, self.nx - Iw + 1, I), np.where(up, 2 * self.ny + 1 - J, J)

    def hm(self, I, J):
        I, J = np.broadcast_arrays(I, J)
        I, J = self._fold(I, J)
        Iw = self._wrap(I)
        ocean = self._inside(I, J)
        if self.ns_boundary == C.BND_TRIPOLE:
            # tripole: the north edge is the fold, ocean reaches it; only the two grid poles
            # (i = nx/2 and i = nx on the top row) sit on land, as on real tripole grids
            pole = (J >= self.ny - 1) & ((np.abs(Iw - self.nx // 2) <= 1) | (Iw >= self.nx - 1) | (Iw <= 1))
            ocean = ocean & (J > 2) & ~pole
        else:
            ocean = ocean & (J > 2) & (J < self.ny - 1)
        if self.land == "continents":
            x = 2.0 * np.pi * Iw / self.nx
            y = np.pi * J / self.ny
            f = (np.sin(3.0 * x + 1.3) * np.cos(2.0 * y) + 0.6 * np.sin(5.0 * x - 0.7 + 2.0 * np.sin(3.0 * y))
                 + 0.4 * np.cos(7.0 * y + x))
            ocean = ocean & (f < 0.55)
        if self.land_band:
            ocean = ocean & ~((Iw >= self.land_band[0]) & (Iw <= self.land_band[1]))
        return ocean.astype(np.float64)

    def uvm(self, I, J):
        I, J = np.broadcast_arrays(I, J)
        v = np.minimum(np.minimum(self.hm(I, J), self.hm(I + 1, J)),
                       np.minimum(self.hm(I, J + 1), self.hm(I + 1, J + 1)))
        return np.where(self._inside(I, J), v, 0.0)

    # ---- ice state / forcing -------------------------------------------------------
    def _icy(self, I, J):
        Iw = self._wrap(I)
        ocean = self.hm(I, J) > 0.5
        if self.ice == "full":
            return ocean
        edge = 55.0 + 5.0 * np.sin(2.0 * np.pi * 4.0 * Iw / self.nx)
        return ocean & (np.abs(self._latT(J)) > edge)

    def field(self, name: str, I, J, cache=None):
        """Field `name` on the index window (I, J).  `cache` (a dict) memoises sub-results between
        calls on the SAME window, which make_block_fields uses to evaluate ~35 fields per block."""
        if cache is not None:
            if name in cache:
                return cache[name]
            cache[name] = r = self._field(name, I, J, cache)
            return r
        return self._field(name, I, J, None)

    def _field(self, name: str, I, J, cache):
        I, J = np.broadcast_arrays(np.asarray(I, dtype=np.int64), np.asarray(J, dtype=np.int64))
        _self_field = self.field
        class _S:   # route nested lookups through the cache
            field = staticmethod(lambda n, I_, J_: _self_field(n, I_, J_, cache))
        if cache is not None:
            def memo(key, fn):
                if key not in cache:
                    cache[key] = fn()
                return cache[key]
            HTN = lambda I_, J_: memo(("HTN", int(J_.flat[0]) - int(J.flat[0])), lambda: self.HTN(I_, J_))
            HTE = lambda I_, J_: memo(("HTE", int(I_.flat[0]) - int(I.flat[0]), int(J_.flat[0]) - int(J.flat[0])), lambda: self.HTE(I_, J_))
            return self._eval(name, I, J, _S, HTN, HTE, lambda: memo("_icy", lambda: self._icy(I, J)),
                              lambda: memo("_hm", lambda: self.hm(I, J)), lambda: memo("_uvm", lambda: self.uvm(I, J)))
        return self._eval(name, I, J, self, self.HTN, self.HTE, lambda: self._icy(I, J), lambda: self.hm(I, J), lambda: self.uvm(I, J))

    def _eval(self, name, I, J, self_, HTN, HTE, icy_fn, hm_fn, uvm_fn):
        if name in ("aice", "aice_init", "vice", "vsno", "strength") and self.ns_boundary == C.BND_TRIPOLE:
            If, Jf = self._fold(I, J)          # T-cell scalars: ghost row north of the fold = image cells
            I, J = If, Jf
            icy_fn = lambda: self._icy(I, J)
            self_ = self
        Iw = self._wrap(I)
        s = self.seed
        if name == "HTN":
            return HTN(I, J)
        if name == "HTE":
            return HTE(I, J)
        if name == "dxt":
            return 0.5 * (HTN(I, J) + HTN(I, J - 1))
        if name == "dyt":
            return 0.5 * (HTE(I, J) + HTE(I - 1, J))
        if name == "dxu":
            return 0.5 * (HTN(I, J) + HTN(I + 1, J))
        if name == "dyu":
            return 0.5 * (HTE(I, J) + HTE(I, J + 1))
        if name == "tarea":
            return self_.field("dxt", I, J) * self_.field("dyt", I, J)
        if name == "uarea":
            return self_.field("dxu", I, J) * self_.field("dyu", I, J)
        if name == "tarear":
            return 1.0 / self_.field("tarea", I, J)
        if name == "uarear":
            return 1.0 / self_.field("uarea", I, J)
        if name == "tinyarea":
            return C.puny * self_.field("tarea", I, J)
        if name == "dxhy":
            return 0.5 * (HTE(I, J) - HTE(I - 1, J))
        if name == "dyhx":
            return 0.5 * (HTN(I, J) - HTN(I, J - 1))
        if name == "cyp":
            return 1.5 * HTE(I, J) - 0.5 * HTE(I - 1, J)
        if name == "cxp":
            return 1.5 * HTN(I, J) - 0.5 * HTN(I, J - 1)
        if name == "cym":
            return -(1.5 * HTE(I - 1, J) - 0.5 * HTE(I, J))
        if name == "cxm":
            return -(1.5 * HTN(I, J - 1) - 0.5 * HTN(I, J))
        if name == "fcor":
            return 2.0 * C.omega * np.sin(np.deg2rad(self._latU(J))) + 0.0 * I
        if name == "tmask":
            return (hm_fn() > 0.5).astype(np.int32)
        if name == "umask":
            return (uvm_fn() > 0.5).astype(np.int32)

        icy = icy_fn()
        x = 2.0 * np.pi * Iw / self.nx
        y = np.pi * J / self.ny
        if name in ("aice", "aice_init"):
            return np.where(icy, 0.6 + 0.4 * _hash01(Iw, J, 1, s), 0.0)
        if name == "vice":
            hi = 0.5 + 2.5 * _hash01(Iw, J, 2, s)
            return np.where(icy, self_.field("aice", I, J) * hi, 0.0)
        if name == "vsno":
            return np.where(icy, self_.field("aice", I, J) * 0.3 * _hash01(Iw, J, 3, s), 0.0)
        ins = self._inside(I, J)
        if name == "uocn":
            return np.where(ins, 0.05 * np.sin(3.0 * x + 0.4) * np.cos(4.0 * y) + 0.01 * (_hash01(Iw, J, 4, s) - 0.5), 0.0)
        if name == "vocn":
            return np.where(ins, 0.05 * np.cos(2.0 * x - 1.1) * np.sin(5.0 * y) + 0.01 * (_hash01(Iw, J, 5, s) - 0.5), 0.0)
        if name == "strairxT":
            return self_.field("aice", I, J) * (0.1 * np.sin(2.0 * x + 3.0 * y) + 0.05 * (_hash01(Iw, J, 6, s) - 0.5))
        if name == "strairyT":
            return self_.field("aice", I, J) * (0.1 * np.cos(3.0 * x - 2.0 * y) + 0.05 * (_hash01(Iw, J, 7, s) - 0.5))
        if name == "ss_tltx":
            return np.where(ins, 1.0e-6 * np.sin(4.0 * x + y), 0.0)
        if name == "ss_tlty":
            return np.where(ins, 1.0e-6 * np.cos(3.0 * x - y), 0.0)
        if name == "Cdn_ocn":
            # the ocean drag coefficient is a per-cell field in the reference (Cw = Cdn_ocn(:,:,iblk), ice_dyn_evp.F90:383;
            # form drag makes it vary in space, ice_atmo.F90:40): +-30 % around dragio, smooth plus cell-scale noise
            return C.dragio * (1.0 + 0.25 * np.sin(5.0 * x + 0.3) * np.cos(3.0 * y - 0.2) + 0.1 * (_hash01(Iw, J, 8, s) - 0.5))
        if name == "strength":   # Hibler (1979), ice_mechred.F90:2258-2265
            a, v = self_.field("aice", I, J), self_.field("vice", I, J)
            return C.Pstar * v * np.exp(-C.Cstar * (1.0 - a))
        if name in ("strax", "stray"):   # ACCESS: wind stress already on the U grid
            return self_.field("strairxT" if name == "strax" else "strairyT", I, J)
        raise KeyError(name)


GRID_FIELDS = ["dxt", "dyt", "dxhy", "dyhx", "cxp", "cyp", "cxm", "cym",
               "tarear", "uarear", "tinyarea", "tarea", "uarea", "fcor", "HTN", "HTE"]
MASK_FIELDS = ["tmask", "umask"]
INPUT_FIELDS = ["aice", "vice", "vsno", "aice_init", "strairxT", "strairyT", "strax", "stray",
                "uocn", "vocn", "ss_tltx", "ss_tlty", "Cdn_ocn", "strength"]
STRESS_FIELDS = [f"{k}_{c}" for k in ("stressp", "stressm", "stress12") for c in (1, 2, 3, 4)]
STATE_FIELDS = ["uvel", "vvel"] + STRESS_FIELDS
OUTPUT_FIELDS = ["divu", "shear", "rdg_conv", "rdg_shear", "prs_sig", "strintx", "strinty",
                 "strocnx", "strocny", "strocnxT", "strocnyT", "strairx", "strairy",
                 "strtltx", "strtlty", "fm", "tmass", "aiu", "umass", "uvel_init", "vvel_init"]


def make_block_fields(case: SynthCase, d: Decomp) -> Dict[str, np.ndarray]:
    """All arrays evp(dt) touches, in block layout, for the local blocks of `d`.
    Prognostic state starts at rest (init_evp, ice_dyn_shared.F90:133-172)."""
    from .blocks import block_index_windows
    f: Dict[str, np.ndarray] = {}
    shp = (d.nblocks, d.ny_block, d.nx_block)
    for n in GRID_FIELDS + INPUT_FIELDS:
        f[n] = np.zeros(shp, dtype=np.float64)
    for n in MASK_FIELDS:
        f[n] = np.zeros(shp, dtype=np.int32)
    Iw, Jw = block_index_windows(d)
    for b in range(d.nblocks):
        I, J = np.broadcast_arrays(Iw[b][None, :], Jw[b][:, None])
        cache: dict = {}
        for n in GRID_FIELDS + INPUT_FIELDS + MASK_FIELDS:
            f[n][b] = case.field(n, I, J, cache)
    for n in STATE_FIELDS + OUTPUT_FIELDS:
        f[n] = np.zeros(shp, dtype=np.float64)
    f["iceumask"] = np.zeros(shp, dtype=np.int32)
    f["icetmask"] = np.zeros(shp, dtype=np.int32)
    return f


def add_remap_grid(case: SynthCase, d: Decomp, f: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """the grid arrays horizontal_remap reads beyond those of evp (ice_transport_remap.F90:340-343): dxu, dyu, hm (real land mask)"""
    from .blocks import block_index_windows
    shp = (d.nblocks, d.ny_block, d.nx_block)
    for n in ("dxu", "dyu", "hm"):
        f[n] = np.zeros(shp, dtype=np.float64)
    Iw, Jw = block_index_windows(d)
    for b in range(d.nblocks):
        I, J = np.broadcast_arrays(Iw[b][None, :], Jw[b][:, None])
        f["dxu"][b] = case.field("dxu", I, J)
        f["dyu"][b] = case.field("dyu", I, J)
        f["hm"][b] = case.hm(I, J)
    return f


def global_min_dx(case: SynthCase) -> float:
    """min(global_minval(dxt, tmask), global_minval(dyt, tmask)) of ice_dyn_shared.F90:221-223."""
    I = np.arange(1, case.nx + 1)[None, :]
    J = np.arange(1, case.ny + 1)[:, None]
    m = case.field("tmask", I, J) > 0
    if not m.any():
        return 1.0
    return float(min(case.field("dxt", I, J)[m].min(), case.field("dyt", I, J)[m].min()))


def add_thickness_distribution(f: Dict[str, np.ndarray], ncat: int = 5) -> Dict[str, np.ndarray]:
    """aicen, vicen (nblocks, ncat, ny_block, nx_block) and aice0 consistent with f['aice'], f['vice'] (the inputs of
    ice_strength with kstrength = 1, ice_state.F90).  The split over the categories is a function of the cell's own
    (aice, vice) only, so ghost cells agree with their owners on every decomposition; some categories are left empty
    (aicen <= puny branches of ridge_itd)."""
    aice, vice = f["aice"], f["vice"]
    base = np.array([0.3, 1.0, 2.0, 3.5, 6.0, 9.0, 13.0, 18.0])[:ncat]
    w = np.zeros((ncat,) + aice.shape)
    for n in range(ncat):
        r = np.modf(aice * (997.0 + 131.0 * n) + vice * (113.0 + 17.0 * n))[0]
        w[n] = np.where(r < 0.2, 0.0, r)
    tot = w.sum(axis=0)
    w[0] = np.where(tot == 0.0, 1.0, w[0])
    tot = np.where(tot == 0.0, 1.0, tot)
    aicen = aice[None] * (w / tot[None])
    den = (aicen * base[:, None, None, None]).sum(axis=0)
    fac = np.where(den > 0.0, vice / np.where(den > 0.0, den, 1.0), 0.0)
    vicen = aicen * base[:, None, None, None] * fac[None]
    f["aicen"] = np.ascontiguousarray(np.moveaxis(aicen, 0, 1))        # (nblocks, ncat, ny, nx) = Fortran (nx,ny,ncat,nblocks)
    f["vicen"] = np.ascontiguousarray(np.moveaxis(vicen, 0, 1))
    f["aice0"] = np.maximum(1.0 - aice, 0.0)
    return f


EAP_STATE = [f"a11_{c}" for c in (1, 2, 3, 4)] + [f"a12_{c}" for c in (1, 2, 3, 4)]
EAP_HISTORY = ["a11", "a12", "e11", "e12", "e22", "yieldstress11", "yieldstress12", "yieldstress22", "s11", "s12", "s22"]


def add_eap_state(f: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """the module arrays of ice_dyn_eap (source/ice_dyn_eap.F90:36-57) as init_eap leaves them (:529-551): isotropic
    structure tensor a11 = 1/2, a12 = 0 at the four corners, history fields zero"""
    shp = f["uvel"].shape
    for n in EAP_STATE:
        f[n] = np.full(shp, 0.5 if n.startswith("a11") else 0.0)
    for n in EAP_HISTORY:
        f[n] = np.zeros(shp)
    return f
