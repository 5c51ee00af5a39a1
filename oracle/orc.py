"""ctypes front end of the CPU oracle (oracle/evp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the cice5_amd product path.
Parity: halo updates / ice_strength pinned by reference output (tests/golden/ref_*.npz), the rest UNPINNED (oracle/evp_oracle.h).
"""
from __future__ import annotations

import ctypes as ct
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libevp_oracle.so")

c_i32p = ct.POINTER(ct.c_int32)
c_f64p = ct.POINTER(ct.c_double)


class OrcGeom(ct.Structure):
    _fields_ = [("nx_global", ct.c_int32), ("ny_global", ct.c_int32),
                ("nx_block", ct.c_int32), ("ny_block", ct.c_int32), ("nblocks", ct.c_int32),
                ("ew_boundary", ct.c_int32), ("ns_boundary", ct.c_int32),
                ("ilo", c_i32p), ("ihi", c_i32p), ("jlo", c_i32p), ("jhi", c_i32p),
                ("iglob_lo", c_i32p), ("jglob_lo", c_i32p)]


class OrcParams(ct.Structure):
    _fields_ = [("dt", ct.c_double), ("ndte", ct.c_int32), ("revised_evp", ct.c_int32),
                ("revp", ct.c_double), ("ecci", ct.c_double), ("dtei", ct.c_double), ("dte2T", ct.c_double),
                ("denom1", ct.c_double), ("arlx1i", ct.c_double), ("brlx", ct.c_double),
                ("cosw", ct.c_double), ("sinw", ct.c_double), ("dragio", ct.c_double),
                ("rhow", ct.c_double), ("rhoi", ct.c_double), ("rhos", ct.c_double), ("gravit", ct.c_double),
                ("a_min", ct.c_double), ("m_min", ct.c_double),
                ("tilt_from_slope", ct.c_int32), ("wind_on_ugrid", ct.c_int32),
                ("strength_mode", ct.c_int32), ("kstrength", ct.c_int32), ("krdg_partic", ct.c_int32),
                ("krdg_redist", ct.c_int32), ("ncat", ct.c_int32), ("pad_", ct.c_int32),
                ("mu_rdg", ct.c_double), ("Cf", ct.c_double)]


_F64_IN = ["dxt", "dyt", "dxhy", "dyhx", "cxp", "cyp", "cxm", "cym",
           "tarear", "uarear", "tinyarea", "tarea", "uarea", "fcor"]
_I32_IN = ["tmask", "umask"]
_F64_IN2 = ["aice", "vice", "vsno", "aice_init", "strairxT", "strairyT", "strax", "stray",
            "uocn", "vocn", "ss_tltx", "ss_tlty", "Cdn_ocn"]
_F64_OUT = ["divu", "shear", "rdg_conv", "rdg_shear", "prs_sig", "strintx", "strinty",
            "strocnx", "strocny", "strocnxT", "strocnyT", "strairx", "strairy",
            "strtltx", "strtlty", "fm", "tmass", "aiu", "umass", "uvel_init", "vvel_init"]


class OrcFields(ct.Structure):
    _fields_ = ([(n, c_f64p) for n in _F64_IN] + [(n, c_i32p) for n in _I32_IN] +
                [(n, c_f64p) for n in _F64_IN2] +
                [("aicen", c_f64p), ("vicen", c_f64p), ("aice0", c_f64p)] +
                [("strength", c_f64p), ("uvel", c_f64p), ("vvel", c_f64p),
                 ("stressp", c_f64p * 4), ("stressm", c_f64p * 4), ("stress12", c_f64p * 4),
                 ("iceumask", c_i32p)] +
                [(n, c_f64p) for n in _F64_OUT] + [("icetmask", c_i32p)])


EAP_OUT = ["e11", "e12", "e22", "yieldstress11", "yieldstress12", "yieldstress22", "s11", "s12", "s22"]


class OrcEapState(ct.Structure):
    _fields_ = ([("nx_yield", ct.c_int32), ("ny_yield", ct.c_int32), ("na_yield", ct.c_int32), ("pad_", ct.c_int32)] +
                [(n, c_f64p) for n in ("s11r", "s12r", "s22r", "s11s", "s12s", "s22s")] +
                [("a11", c_f64p * 4), ("a12", c_f64p * 4), ("a11m", c_f64p), ("a12m", c_f64p)] + [(n, c_f64p) for n in EAP_OUT])


HALO_CB = ct.CFUNCTYPE(None, c_f64p, ct.c_int, ct.c_int, ct.c_double, ct.c_int, ct.c_void_p)


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, n)) for n in ("evp_oracle.c", "remap_oracle.c", "eap_oracle.c", "evp_oracle.h")) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "..", "cice5_amd", "csrc", "evpk_fmath.h")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libevp_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
last_halo_seconds = 0.0


def _limit_threads():
    """OpenMP over blocks (as the reference's THRD build): use the cores this process may run on,
    capped, and do not spin -- GPU boxes expose far more hardware threads than our share."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    n = max(1, min(n, int(os.environ.get("EVP_ORACLE_THREADS", "16"))))
    os.environ.setdefault("OMP_NUM_THREADS", str(n))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    return int(os.environ["OMP_NUM_THREADS"])


_lib_libm = None


def lib_libm():
    """the restatement built with the host's libm for the EAP angles (oracle/Makefile: libevp_oracle_libm.so)"""
    global _lib_libm
    if _lib_libm is None:
        build()
        path = os.path.join(_HERE, "libevp_oracle_libm.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "eap_oracle.c")):
            subprocess.check_call(["make", "-C", _HERE, "-B", "libevp_oracle_libm.so"], stdout=subprocess.DEVNULL)
        L = ct.CDLL(path)
        L.orc_set_num_threads.argtypes = [ct.c_int]
        L.orc_set_num_threads(_limit_threads())
        L.orc_eap.argtypes = lib().orc_eap.argtypes
        _lib_libm = L
    return _lib_libm


def eap_lookup_counts(enable: bool):
    """(lookups, lookups whose index triple differs under libm, calc_ffrac decisions, decisions that differ) since the last
    call; enable / disable the counting mode of eap_oracle.c"""
    out = (ct.c_longlong * 4)()
    lib().orc_eap_lookup_counts(out, int(enable))
    return tuple(int(v) for v in out)


def lib():
    global _lib
    if _lib is None:
        build()
        nthr = _limit_threads()
        _lib = ct.CDLL(_LIB_PATH)
        # the environment variable is only read when the OpenMP runtime starts, which a host process (torch) may have done
        # long ago with its own default: set the team size explicitly
        _lib.orc_set_num_threads.argtypes = [ct.c_int]
        _lib.orc_set_num_threads(nthr)
        _lib.orc_set_evp_parameters.argtypes = [ct.c_double, ct.c_int32, ct.c_int32, ct.c_double, ct.POINTER(OrcParams)]
        _lib.orc_evp.argtypes = [ct.POINTER(OrcGeom), ct.POINTER(OrcParams), ct.POINTER(OrcFields), ct.c_int,
                                 ct.POINTER(ct.c_int64), ct.POINTER(ct.c_double)]      # loop_seconds[2]
        _lib.orc_eap.argtypes = [ct.POINTER(OrcGeom), ct.POINTER(OrcParams), ct.POINTER(OrcFields), ct.POINTER(OrcEapState), ct.c_int,
                                 ct.POINTER(ct.c_int64), ct.POINTER(ct.c_double)]
        for n in ("orc_fm_sin", "orc_fm_cos"):
            getattr(_lib, n).argtypes = [ct.c_double]
            getattr(_lib, n).restype = ct.c_double
        _lib.orc_fm_atan2.argtypes = [ct.c_double, ct.c_double]
        _lib.orc_fm_atan2.restype = ct.c_double
        _lib.orc_halo_r8.argtypes = [ct.POINTER(OrcGeom), c_f64p, ct.c_int, ct.c_int, ct.c_double]
        _lib.orc_halo_i4.argtypes = [ct.POINTER(OrcGeom), c_i32p, ct.c_int32]
        _lib.orc_transport_upwind.argtypes = [ct.POINTER(OrcGeom), ct.c_double, ct.c_int] + [c_f64p] * 6
        _lib.orc_horizontal_remap.argtypes = ([ct.POINTER(OrcGeom), ct.c_double, ct.c_int, ct.c_int] + [c_f64p] * 4 + [ct.c_int] +
                                              [c_i32p] * 3 + [ct.c_int, ct.c_int] + [c_f64p] * 6)
        _lib.orc_horizontal_remap.restype = ct.c_int
        _lib.orc_transport_remap_state.argtypes = ([ct.POINTER(OrcGeom), ct.c_double] + [ct.c_int] * 5 + [ct.c_double] + [c_f64p] * 7 +
                                                   [c_i32p] * 3 + [ct.c_int] * 2 + [c_f64p] * 6)
        _lib.orc_halo_stress.argtypes = [ct.POINTER(OrcGeom), c_f64p, c_f64p]
        _lib.orc_principal_stress.argtypes = [ct.c_int, ct.c_int] + [c_f64p] * 6
        _lib.orc_set_halo_callback.argtypes = [HALO_CB, ct.c_void_p]
        _lib.orc_eap_lookup_counts.argtypes = [ct.POINTER(ct.c_longlong), ct.c_int]
        _lib.orc_exp.argtypes = [ct.c_double]
        _lib.orc_exp.restype = ct.c_double
        _lib.orc_ice_strength.argtypes = [ct.c_int] * 7 + [c_i32p] * 2 + [c_f64p] * 6 + [ct.POINTER(OrcParams)]
        _lib.orc_stress.argtypes = ([ct.c_int] * 5 + [c_i32p] * 2 + [c_f64p] * 13 +
                                    [ct.POINTER(c_f64p)] * 3 + [c_f64p] * 6 + [ct.POINTER(OrcParams)])
    return _lib


def _p64(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(c_f64p)


def _p32(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(c_i32p)


def make_geom(d):
    """d: cice5_amd.blocks.Decomp (duck-typed).  Returns (OrcGeom, keepalive)."""
    ga = d.geom_arrays()
    g = OrcGeom(d.nx_global, d.ny_global, d.nx_block, d.ny_block, d.nblocks, d.ew_boundary, d.ns_boundary,
                _p32(ga["ilo"]), _p32(ga["ihi"]), _p32(ga["jlo"]), _p32(ga["jhi"]),
                _p32(ga["iglob_lo"]), _p32(ga["jglob_lo"]))
    return g, ga


def make_params(dt: float, ndte: int, xmin: float, revised_evp: bool = False, cosw: float = 1.0, sinw: float = 0.0,
                dragio: float = 0.00536, tilt_from_slope: bool = False, wind_on_ugrid: bool = False,
                strength_mode: int = 0, kstrength: int = 1, krdg_partic: int = 1, krdg_redist: int = 1, ncat: int = 5,
                mu_rdg: float = 3.0, Cf: float = 17.0) -> OrcParams:
    """strength_mode = 1: evp computes the ice strength itself (ice_mechred.F90:2111) from aice, vice and, for
    kstrength = 1, the thickness distribution aicen / vicen / aice0 in `f`; the namelist defaults are those of
    ice_init.F90:273-277."""
    p = OrcParams()
    lib().orc_set_evp_parameters(dt, ndte, int(revised_evp), xmin, ct.byref(p))
    p.cosw, p.sinw, p.dragio = cosw, sinw, dragio
    p.rhow, p.rhoi, p.rhos, p.gravit = 1026.0, 917.0, 330.0, 9.80616
    p.a_min, p.m_min = 0.001, 0.01
    p.tilt_from_slope, p.wind_on_ugrid = int(tilt_from_slope), int(wind_on_ugrid)
    p.strength_mode, p.kstrength, p.krdg_partic, p.krdg_redist, p.ncat = strength_mode, kstrength, krdg_partic, krdg_redist, ncat
    p.mu_rdg, p.Cf = mu_rdg, Cf
    return p


def make_fields(f: Dict[str, np.ndarray]) -> OrcFields:
    o = OrcFields()
    for n in _F64_IN + _F64_IN2 + _F64_OUT + ["strength", "uvel", "vvel"]:
        setattr(o, n, _p64(f[n]))
    for n in ("aicen", "vicen", "aice0"):
        setattr(o, n, _p64(f[n]) if n in f else None)
    for n in _I32_IN + ["iceumask", "icetmask"]:
        setattr(o, n, _p32(f[n]))
    for k in ("stressp", "stressm", "stress12"):
        setattr(o, k, (c_f64p * 4)(*[_p64(f[f"{k}_{c}"]) for c in (1, 2, 3, 4)]))
    return o


def evp(d, params: OrcParams, f: Dict[str, np.ndarray], nsub: int = 0):
    """Run the oracle's evp(dt) in place on the block-layout dict `f`.
    Returns (icellt_phys, icellu, loop_seconds)."""
    g, keep = make_geom(d)
    of = make_fields(f)
    counts = (ct.c_int64 * 2)()
    secs = (ct.c_double * 2)(0.0, 0.0)
    lib().orc_evp(ct.byref(g), ct.byref(params), ct.byref(of), int(nsub), counts, secs)
    del keep
    global last_halo_seconds
    last_halo_seconds = float(secs[1])          # the halo updates' share of the loop time returned below
    return int(counts[0]), int(counts[1]), float(secs[0])


def eap(d, params: OrcParams, f: Dict[str, np.ndarray], tables, nsub: int = 0, libm: bool = False):
    """orc_eap: eap(dt) (ice_dyn_eap.F90:66-486) in place on `f`, which also holds a11_1..4, a12_1..4 (in/out), a11, a12 and
    the nine history fields (cice5_amd.synth.add_eap_state); tables = cice5_amd.eap_tables.eap_tables()"""
    g, keep = make_geom(d)
    of = make_fields(f)
    e = OrcEapState()
    e.na_yield, e.ny_yield, e.nx_yield = tables[0].shape
    for n, t in zip(("s11r", "s12r", "s22r", "s11s", "s12s", "s22s"), tables):
        assert t.flags["C_CONTIGUOUS"] and t.dtype == np.float64
        setattr(e, n, _p64(t))
    e.a11 = (c_f64p * 4)(*[_p64(f[f"a11_{c}"]) for c in (1, 2, 3, 4)])
    e.a12 = (c_f64p * 4)(*[_p64(f[f"a12_{c}"]) for c in (1, 2, 3, 4)])
    e.a11m, e.a12m = _p64(f["a11"]), _p64(f["a12"])
    for n in EAP_OUT:
        setattr(e, n, _p64(f[n]))
    counts = (ct.c_int64 * 2)()
    secs = (ct.c_double * 2)(0.0, 0.0)
    (lib_libm() if libm else lib()).orc_eap(ct.byref(g), ct.byref(params), ct.byref(of), ct.byref(e), int(nsub), counts, secs)
    del keep
    return int(counts[0]), int(counts[1]), float(secs[0])


def transport_upwind(d, dt: float, f: Dict[str, np.ndarray], works: np.ndarray):
    """orc_transport_upwind: works (nblocks, narr, ny_block, nx_block) advected in place with f["uvel"], f["vvel"]"""
    g, keep = make_geom(d)
    lib().orc_transport_upwind(ct.byref(g), float(dt), int(works.shape[1]), _p64(f["uvel"]), _p64(f["vvel"]),
                               _p64(f["HTE"]), _p64(f["HTN"]), _p64(f["tarea"]), _p64(works))
    del keep


def transport_upwind_state(d, dt: float, f: Dict[str, np.ndarray], aice0, aicen, vicen, vsnon, trcrn, ntrcr: int, trcr_depend, nt_Tsfc=1, nt_alvl=0,
                           nt_apnd=0, nt_fbri=0, ponds=(0, 0, 0), Tocnfrz=-1.8):
    """orc_transport_upwind_state: transport_upwind whole (state_to_work, upwind_field, work_to_state, bound_state), in place"""
    g, keep = make_geom(d)
    L = lib()
    L.orc_transport_upwind_state.argtypes = ([ct.POINTER(OrcGeom), ct.c_double] + [ct.c_int] * 3 + [c_i32p] + [ct.c_int] * 7 + [ct.c_double] + [c_f64p] * 10)
    dep = np.ascontiguousarray(trcr_depend, dtype=np.int32)
    L.orc_transport_upwind_state(ct.byref(g), float(dt), int(aicen.shape[1]), int(ntrcr), int(trcrn.shape[2]), _p32(dep), int(nt_Tsfc), int(nt_alvl),
                                 int(nt_apnd), int(nt_fbri), *[int(p) for p in ponds], float(Tocnfrz), _p64(f["uvel"]), _p64(f["vvel"]),
                                 _p64(f["HTE"]), _p64(f["HTN"]), _p64(f["tarea"]), _p64(aice0), _p64(aicen), _p64(vicen), _p64(vsnon), _p64(trcrn))
    del keep


def remap_tables(trcr_depend):
    """tracer_type / depend / has_dependents of init_transport (ice_transport_driver.F90:88-125) for hice, hsno and the
    tracers whose `trcr_depend` (0 area, 1 ice volume, 2 snow volume, 2 + nt: tracer nt) is given"""
    ntrace = 2 + len(trcr_depend)
    depend = np.zeros(ntrace, dtype=np.int32)
    ttype = np.ones(ntrace, dtype=np.int32)
    for nt, dep in enumerate(trcr_depend, start=1):
        depend[2 + nt - 1] = dep
        ttype[2 + nt - 1] = 2
        if dep == 0:
            ttype[2 + nt - 1] = 1
        elif dep > 2 and trcr_depend[dep - 2 - 1] > 0:
            ttype[2 + nt - 1] = 3
    has = np.zeros(ntrace, dtype=np.int32)
    for nt in range(ntrace):
        if depend[nt] > 0:
            assert depend[nt] - 1 < nt, "a tracer must come after the tracer it depends on"
            has[depend[nt] - 1] = 1
    return ttype, depend, has


def horizontal_remap(d, dt: float, f: Dict[str, np.ndarray], mm: np.ndarray, tm: np.ndarray, tracer_type, depend, has_dependents,
                     integral_order: int = 3, l_dp_midpt: bool = True, l_fixed_area: bool = False) -> int:
    """orc_horizontal_remap: mm (nblocks, ncat+1, ny, nx), tm (nblocks, ncat, ntrace, ny, nx) in place; velocities and grid from f"""
    g, keep = make_geom(d)
    ncat, ntrace = mm.shape[1] - 1, tm.shape[2]
    rc = lib().orc_horizontal_remap(ct.byref(g), float(dt), ncat, ntrace, _p64(f["uvel"]), _p64(f["vvel"]), _p64(mm), _p64(tm),
                                    int(l_fixed_area), _p32(np.ascontiguousarray(tracer_type, dtype=np.int32)),
                                    _p32(np.ascontiguousarray(depend, dtype=np.int32)),
                                    _p32(np.ascontiguousarray(has_dependents, dtype=np.int32)), int(integral_order), int(l_dp_midpt),
                                    _p64(f["HTE"]), _p64(f["HTN"]), _p64(f["dxu"]), _p64(f["dyu"]), _p64(f["tarear"]), _p64(f["hm"]))
    del keep
    return int(rc)


def transport_remap_state(d, dt: float, f: Dict[str, np.ndarray], aice0, aicen, vicen, vsnon, trcrn, ntrcr: int, nt_qsno: int, nslyr: int,
                          rhos_lfresh: float, tracer_type, depend, has_dependents, integral_order: int = 3, l_dp_midpt: bool = True) -> int:
    """orc_transport_remap_state: aice0 (nb, ny, nx), aicen / vicen / vsnon (nb, ncat, ny, nx), trcrn (nb, ncat, ntrcr_dim, ny, nx), in place"""
    g, keep = make_geom(d)
    ncat, ntrcr_dim = aicen.shape[1], trcrn.shape[2]
    rc = lib().orc_transport_remap_state(ct.byref(g), float(dt), ncat, int(ntrcr), ntrcr_dim, int(nt_qsno), int(nslyr), float(rhos_lfresh),
                                         _p64(f["uvel"]), _p64(f["vvel"]), _p64(aice0), _p64(aicen), _p64(vicen), _p64(vsnon), _p64(trcrn),
                                         _p32(np.ascontiguousarray(tracer_type, dtype=np.int32)), _p32(np.ascontiguousarray(depend, dtype=np.int32)),
                                         _p32(np.ascontiguousarray(has_dependents, dtype=np.int32)), int(integral_order), int(l_dp_midpt),
                                         _p64(f["HTE"]), _p64(f["HTN"]), _p64(f["dxu"]), _p64(f["dyu"]), _p64(f["tarear"]), _p64(f["hm"]))
    del keep
    return int(rc)


def halo_r8(d, a: np.ndarray, loc: int, kind: int, fill: float = 0.0):
    g, keep = make_geom(d)
    lib().orc_halo_r8(ct.byref(g), _p64(a), loc, kind, fill)
    del keep


def halo_i4(d, a: np.ndarray, fill: int = 0):
    """orc_halo_i4: centre / scalar update of an int32 block array (ice_HaloUpdate2DI4)"""
    g, keep = make_geom(d)
    lib().orc_halo_i4(ct.byref(g), _p32(a), int(fill))
    del keep


def halo_stress(d, a1: np.ndarray, a2: np.ndarray):
    """orc_halo_stress: ice_HaloUpdate_stress(array1 = a1, array2 = a2, field_loc_center, field_type_scalar)"""
    g, keep = make_geom(d)
    lib().orc_halo_stress(ct.byref(g), _p64(a1), _p64(a2))
    del keep


def set_halo_callback(fn):
    """fn(array_ptr, loc, kind, fill, phase) -- phase 0 before, 1 after the local update; None clears.
    Returns the ctypes callback object, which the caller must keep alive."""
    if fn is None:
        lib().orc_set_halo_callback(ct.cast(None, HALO_CB), None)
        return None
    cb = HALO_CB(lambda a, loc, kind, fill, phase, user: fn(a, loc, kind, fill, phase))
    lib().orc_set_halo_callback(cb, None)
    return cb


def stress_block(nx, ny, ksub, ndte, indxti, indxtj, arrs: Dict[str, np.ndarray], params: OrcParams):
    """orc_stress on one (ny, nx) block; `arrs` holds uvel..strength, stressp_1.. and the outputs.
    Returns str as an (8, ny, nx) array."""
    L = lib()
    strv = np.zeros((8, ny, nx))
    sp = (c_f64p * 4)(*[_p64(arrs[f"stressp_{c}"]) for c in (1, 2, 3, 4)])
    sm = (c_f64p * 4)(*[_p64(arrs[f"stressm_{c}"]) for c in (1, 2, 3, 4)])
    s12 = (c_f64p * 4)(*[_p64(arrs[f"stress12_{c}"]) for c in (1, 2, 3, 4)])
    L.orc_stress(nx, ny, ksub, ndte, len(indxti), _p32(indxti), _p32(indxtj),
                 *[_p64(arrs[n]) for n in ("uvel", "vvel", "dxt", "dyt", "dxhy", "dyhx", "cxp", "cyp", "cxm", "cym",
                                           "tarear", "tinyarea", "strength")],
                 sp, sm, s12,
                 *[_p64(arrs[n]) for n in ("shear", "divu", "prs_sig", "rdg_conv", "rdg_shear")],
                 _p64(strv), ct.byref(params))
    return strv


def exp(x: float) -> float:
    """exp() as the restatement evaluates it (orc_exp)."""
    return float(lib().orc_exp(float(x)))


def ice_strength_block(nx, ny, ilo, ihi, jlo, jhi, indxi, indxj, aice, vice, aice0, aicen, vicen, params: OrcParams):
    """orc_ice_strength on one (ny, nx) block (aicen, vicen: (ncat, ny, nx)); returns the strength plane."""
    out = np.zeros((ny, nx))
    lib().orc_ice_strength(nx, ny, ilo, ihi, jlo, jhi, len(indxi), _p32(indxi), _p32(indxj),
                           _p64(aice), _p64(vice), _p64(aice0), _p64(aicen), _p64(vicen), _p64(out), ct.byref(params))
    return out
