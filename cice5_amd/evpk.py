"""ctypes binding of libevpk.so (include/evpk.h) -- the same C ABI the Fortran shim binds.

No torch types cross this boundary: arrays are numpy block-layout buffers
(nblocks, ny_block, nx_block), i.e. the reference's (nx_block, ny_block, max_blocks).
There is no CPU fallback: loading fails loudly if the library is missing, and
evpk_create fails if no gfx950 device is usable.
"""
from __future__ import annotations

import ctypes as ct
import os
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EVPK_LIB") or os.path.join(_HERE, "libevpk.so")      # (EVPK_LIB: A/B builds of the kernels, scripts/)

c_i32p = ct.POINTER(ct.c_int32)
c_f64p = ct.POINTER(ct.c_double)

UNIQUE_ID_BYTES = 128

GEOM_F64 = ["dxt", "dyt", "dxhy", "dyhx", "cxp", "cyp", "cxm", "cym",
            "tarear", "uarear", "tinyarea", "tarea", "uarea", "fcor"]
STEP_IN_F64 = ["aice", "vice", "vsno", "aice_init", "strairxT", "strairyT", "strax", "stray",
               "uocn", "vocn", "ss_tltx", "ss_tlty", "Cdn_ocn", "strength", "aicen", "vicen", "aice0"]
STATE_OUT_F64 = ["divu", "shear", "rdg_conv", "rdg_shear", "prs_sig", "strintx", "strinty",
                 "strocnx", "strocny", "strocnxT", "strocnyT", "strairx", "strairy",
                 "strtltx", "strtlty", "fm", "tmass", "aiu", "umass", "uvel_init", "vvel_init"]


class Geom(ct.Structure):
    _fields_ = ([("nx_global", ct.c_int32), ("ny_global", ct.c_int32),
                 ("nx_block", ct.c_int32), ("ny_block", ct.c_int32), ("nblocks", ct.c_int32),
                 ("ew_boundary", ct.c_int32), ("ns_boundary", ct.c_int32),
                 ("ilo", c_i32p), ("ihi", c_i32p), ("jlo", c_i32p), ("jhi", c_i32p),
                 ("iglob_lo", c_i32p), ("jglob_lo", c_i32p),
                 ("rank", ct.c_int32), ("nranks", ct.c_int32), ("device", ct.c_int32),
                 ("unique_id", ct.c_void_p)] +
                [(n, c_f64p) for n in GEOM_F64] + [("tmask", c_i32p), ("umask", c_i32p), ("HTN", c_f64p), ("HTE", c_f64p)])


class Params(ct.Structure):
    _fields_ = [("dt", ct.c_double), ("ndte", ct.c_int32), ("revised_evp", ct.c_int32),
                ("revp", ct.c_double), ("ecci", ct.c_double), ("denom1", ct.c_double),
                ("arlx1i", ct.c_double), ("brlx", ct.c_double),
                ("cosw", ct.c_double), ("sinw", ct.c_double),
                ("rhow", ct.c_double), ("rhoi", ct.c_double), ("rhos", ct.c_double), ("gravit", ct.c_double),
                ("a_min", ct.c_double), ("m_min", ct.c_double),
                ("tilt_from_slope", ct.c_int32), ("wind_on_ugrid", ct.c_int32),
                ("kstrength", ct.c_int32), ("krdg_partic", ct.c_int32), ("krdg_redist", ct.c_int32), ("ncat", ct.c_int32),
                ("mu_rdg", ct.c_double), ("Cf", ct.c_double), ("sparse_io", ct.c_int32), ("reserved_", ct.c_int32)]


class StepIn(ct.Structure):
    _fields_ = [(n, c_f64p) for n in STEP_IN_F64]


class State(ct.Structure):
    _fields_ = ([("uvel", c_f64p), ("vvel", c_f64p),
                 ("stressp", c_f64p * 4), ("stressm", c_f64p * 4), ("stress12", c_f64p * 4),
                 ("iceumask", c_i32p)] +
                [(n, c_f64p) for n in STATE_OUT_F64] + [("icetmask", c_i32p), ("strength", c_f64p)])


EAP_HISTORY = ["a11", "a12", "e11", "e12", "e22", "yieldstress11", "yieldstress12", "yieldstress22", "s11", "s12", "s22"]


class EapState(ct.Structure):
    _fields_ = [("a11_c", c_f64p * 4), ("a12_c", c_f64p * 4)] + [(n, c_f64p) for n in EAP_HISTORY]


class Stats(ct.Structure):
    _fields_ = [("icellt", ct.c_int64), ("icellu", ct.c_int64), ("ncell_slab", ct.c_int64),
                ("nstrips", ct.c_int32), ("nstrips_total", ct.c_int32), ("subcycles_done", ct.c_int32),
                ("loop_ms", ct.c_float), ("kernel_ms", ct.c_float), ("kernel_launches", ct.c_int32),
                ("kernel2_ms", ct.c_float), ("kernel2_launches", ct.c_int32),
                ("strip_rows", ct.c_int32), ("strip_rows2", ct.c_int32), ("nstrips2", ct.c_int32),
                ("zone_cols", ct.c_int32), ("zone_exchanges", ct.c_int32), ("zone_bytes", ct.c_int64),
                ("overlap_split", ct.c_int32), ("tile_kernel", ct.c_int32), ("kernel_timed", ct.c_int32),
                ("kernel2_timed", ct.c_int32), ("bound_ms", ct.c_float), ("bound_updates", ct.c_int32),
                ("compact_metrics", ct.c_int32), ("transport", ct.c_int32), ("band_row_exchanges", ct.c_int32),
                ("kernel3_ms", ct.c_float), ("kernel3_launches", ct.c_int32), ("kernel3_timed", ct.c_int32),
                ("strip_rows3", ct.c_int32), ("nstrips3", ct.c_int32),
                ("rccl_ranks", ct.c_int32), ("device", ct.c_int32), ("device_pci", ct.c_int32),
                ("delivery_checked", ct.c_int64), ("delivery_bad", ct.c_int64)]

XP_NAMES = {0: "none", 1: "rccl", 2: "shm relay", 3: "ipc peer-mapped", 4: "self (forced exchange)"}


EXPORTS = ["evpk_get_unique_id", "evpk_create", "evpk_set_params", "evpk_run", "evpk_upload", "evpk_prep",
           "evpk_subcycle", "evpk_finish", "evpk_download", "evpk_sync", "evpk_get_stats", "evpk_destroy",
           "evpk_last_error", "evpk_slab_layout", "evpk_calibrate", "evpk_principal_stress", "evpk_pin_host",
           "evpk_unpin_host", "evpk_connect", "evpk_device_check", "evpk_restart_write", "evpk_restart_read",
           "evpk_transport_upwind", "evpk_remap_init", "evpk_transport_remap", "evpk_transport_remap_state",
           "evpk_eap_init", "evpk_eap_upload", "evpk_eap_download", "evpk_halo_update", "evpk_halo_update_stress",
           "evpk_transport_upwind_state", "evpk_host_alloc", "evpk_host_free", "evpk_host_is_mapped", "evpk_experimental_built"]

REMAP_BAD_DEPARTURE, REMAP_NEGATIVE_MASS = 11, 12        # include/evpk.h

_lib = None


class EvpkError(RuntimeError):
    pass


def lib():
    """Load libevpk.so; raises if it has not been built (python __graft_entry__.py / make -C cice5_amd/csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EvpkError(f"{LIB_PATH} not found: build it with `make -C cice5_amd/csrc` "
                            "(there is no CPU fallback for the EVP kernels)")
        L = ct.CDLL(LIB_PATH)
        ctxp = ct.c_void_p
        L.evpk_get_unique_id.argtypes = [ct.c_void_p]
        L.evpk_create.argtypes = [ct.POINTER(Geom), ct.POINTER(ctxp)]
        L.evpk_set_params.argtypes = [ctxp, ct.POINTER(Params)]
        L.evpk_run.argtypes = [ctxp, ct.POINTER(StepIn), ct.POINTER(State)]
        L.evpk_upload.argtypes = [ctxp, ct.POINTER(StepIn), ct.POINTER(State)]
        L.evpk_prep.argtypes = [ctxp]
        L.evpk_subcycle.argtypes = [ctxp, ct.c_int32]
        L.evpk_finish.argtypes = [ctxp]
        L.evpk_download.argtypes = [ctxp, ct.POINTER(State)]
        L.evpk_sync.argtypes = [ctxp]
        L.evpk_get_stats.argtypes = [ctxp, ct.POINTER(Stats)]
        L.evpk_destroy.argtypes = [ctxp]
        L.evpk_calibrate.argtypes = [ctxp, ct.c_int32]
        L.evpk_principal_stress.argtypes = [ctxp, c_f64p, c_f64p]
        L.evpk_last_error.argtypes = [ctxp]
        L.evpk_last_error.restype = ct.c_char_p
        L.evpk_slab_layout.argtypes = [ct.c_int32] * 6 + [c_i32p]
        L.evpk_pin_host.argtypes = [ct.c_void_p, ct.c_size_t]
        L.evpk_unpin_host.argtypes = [ct.c_void_p]
        L.evpk_host_alloc.argtypes = [ct.c_size_t, ct.POINTER(ct.c_void_p)]
        L.evpk_host_free.argtypes = [ct.c_void_p]
        L.evpk_host_is_mapped.argtypes = [ct.c_void_p, ct.c_size_t]
        L.evpk_experimental_built.argtypes = []
        L.evpk_connect.argtypes = [ctxp, ct.c_void_p]
        L.evpk_device_check.argtypes = [ct.c_int32]
        L.evpk_transport_upwind.argtypes = [ctxp, ct.c_double, ct.c_int32, c_f64p]
        L.evpk_transport_upwind_state.argtypes = ([ctxp, ct.c_double] + [ct.c_int32] * 3 + [c_i32p] + [ct.c_int32] * 7 + [ct.c_double] + [c_f64p] * 5)
        L.evpk_halo_update.argtypes = [ctxp, c_f64p, ct.c_int32, ct.c_int32, ct.c_int32, ct.c_double]
        L.evpk_halo_update_stress.argtypes = [ctxp, c_f64p, c_f64p]
        L.evpk_remap_init.argtypes = [ctxp, c_f64p, c_f64p, c_f64p]
        L.evpk_transport_remap.argtypes = [ctxp, ct.c_double, ct.c_int32, ct.c_int32, c_f64p, c_f64p, c_i32p, c_i32p, c_i32p,
                                           ct.c_int32, ct.c_int32, ct.c_int32]
        L.evpk_transport_remap_state.argtypes = ([ctxp, ct.c_double] + [ct.c_int32] * 5 + [ct.c_double] + [c_f64p] * 5 + [c_i32p] * 3 +
                                                 [ct.c_int32] * 2)
        L.evpk_eap_init.argtypes = [ctxp, ct.c_int32, ct.c_int32, ct.c_int32] + [c_f64p] * 6
        L.evpk_eap_upload.argtypes = [ctxp, ct.POINTER(EapState)]
        L.evpk_eap_download.argtypes = [ctxp, ct.POINTER(EapState)]
        L.evpk_restart_write.argtypes = [ctxp, ct.c_char_p, ct.c_int32, ct.c_int32]
        L.evpk_restart_read.argtypes = [ctxp, ct.c_char_p, ct.c_int64, ct.c_int32]
        for n in EXPORTS:
            if n != "evpk_last_error":
                getattr(L, n).restype = ct.c_int
        _lib = L
    return _lib


def pin_host(a: np.ndarray) -> bool:
    """Page-lock a host array for in-place transfers (evpk_pin_host); False if the runtime refuses."""
    assert a.flags["C_CONTIGUOUS"]
    return lib().evpk_pin_host(ct.c_void_p(a.ctypes.data), a.nbytes) == 0


def unpin_host(a: np.ndarray) -> bool:
    return lib().evpk_unpin_host(ct.c_void_p(a.ctypes.data)) == 0


class _HostBlock:
    """one evpk_host_alloc allocation; freed when the last array on it is gone"""

    def __init__(self, nbytes: int):
        self.ptr = ct.c_void_p()
        if lib().evpk_host_alloc(max(nbytes, 8), ct.byref(self.ptr)) != 0:
            raise EvpkError("evpk_host_alloc failed (no device, or out of page-locked memory)")
        self.nbytes = nbytes

    def __del__(self):
        if self.ptr and _lib is not None:
            _lib.evpk_host_free(self.ptr)
            self.ptr = ct.c_void_p()


class _HostMem:
    """array-interface owner of a _HostBlock: np.asarray(_HostMem) is a view whose base keeps the block alive"""

    def __init__(self, blk, shape, dt):
        self.blk = blk
        self.__array_interface__ = {"data": (blk.ptr.value, False), "shape": tuple(int(v) for v in shape), "typestr": dt.str, "version": 3}


def host_empty(shape, dtype=np.float64) -> np.ndarray:
    """An array on page-locked memory that the DRIVER allocated and pinned (evpk_host_alloc = hipHostMalloc, mapped into the
    device address space): the library reads / writes it in place, and its pages cannot move under a kernel.  Freed with the
    last view of it."""
    dt = np.dtype(dtype)
    shape = (shape,) if np.isscalar(shape) else tuple(shape)
    return np.asarray(_HostMem(_HostBlock(int(np.prod(shape)) * dt.itemsize), shape, dt))


def host_copy(a: np.ndarray) -> np.ndarray:
    b = host_empty(a.shape, a.dtype)
    b[...] = a
    return b


def experimental() -> bool:
    """True if the loaded library was built with -DEVPK_EXPERIMENTAL (make -C cice5_amd/csrc exp -> libevpk_exp.so, EVPK_LIB selects it):
    it then contains k_subcycle2 and k_subcycle3w, the kernels that were measured and not adopted"""
    return lib().evpk_experimental_built() == 1


def host_is_mapped(a: np.ndarray) -> bool:
    return lib().evpk_host_is_mapped(ct.c_void_p(a.ctypes.data), a.nbytes) == 1


def _p64(a: Optional[np.ndarray]):
    if a is None:
        return None
    if a.dtype != np.float64 or not a.flags.c_contiguous:
        raise TypeError("expected a C-contiguous float64 block array")
    return a.ctypes.data_as(c_f64p)


def _p32(a: Optional[np.ndarray]):
    if a is None:
        return None
    if a.dtype != np.int32 or not a.flags.c_contiguous:
        raise TypeError("expected a C-contiguous int32 block array")
    return a.ctypes.data_as(c_i32p)


def get_unique_id() -> bytes:
    buf = ct.create_string_buffer(UNIQUE_ID_BYTES)
    if lib().evpk_get_unique_id(buf):
        raise EvpkError("evpk_get_unique_id failed")
    return buf.raw


def device_check(device: int) -> Optional[str]:
    """None if `device` is a usable gfx950 part, else the reason (evpk_device_check)."""
    L = lib()
    if L.evpk_device_check(int(device)):
        return L.evpk_last_error(None).decode()
    return None


def slab_layout(nx_global, nranks, rank, ew_boundary, i0, i1):
    out = (ct.c_int32 * 5)()
    if lib().evpk_slab_layout(nx_global, nranks, rank, ew_boundary, i0, i1, out):
        raise EvpkError("evpk_slab_layout: bad arguments")
    return dict(west=out[0], east=out[1], i0=out[2], i1=out[3], wmax=out[4])


class Context:
    """Owns one evpk_ctx (one GPU, one rank)."""

    def __init__(self, decomp, fields: Dict[str, np.ndarray], device: int = 0, unique_id: Optional[bytes] = None,
                 defer_connect: bool = False):
        """defer_connect (nprocs > 1): evpk_create without a unique id -- nothing collective happens until connect(),
        so the host can first agree across ranks that every create succeeded."""
        L = lib()
        self._L = L
        self._ctx = ct.c_void_p()
        ga = decomp.geom_arrays()
        g = Geom()
        g.nx_global, g.ny_global = decomp.nx_global, decomp.ny_global
        g.nx_block, g.ny_block, g.nblocks = decomp.nx_block, decomp.ny_block, decomp.nblocks
        g.ew_boundary, g.ns_boundary = decomp.ew_boundary, decomp.ns_boundary
        for n in ("ilo", "ihi", "jlo", "jhi", "iglob_lo", "jglob_lo"):
            setattr(g, n, _p32(ga[n]))
        g.rank, g.nranks, g.device = decomp.rank, decomp.nprocs, device
        if defer_connect:
            unique_id = None
        self._uid = ct.create_string_buffer(unique_id, UNIQUE_ID_BYTES) if unique_id is not None else None
        g.unique_id = ct.cast(self._uid, ct.c_void_p) if self._uid is not None else None
        for n in GEOM_F64:
            setattr(g, n, _p64(fields[n]))
        g.tmask, g.umask = _p32(fields["tmask"]), _p32(fields["umask"])
        g.HTN, g.HTE = _p64(fields.get("HTN")), _p64(fields.get("HTE"))      # optional
        if L.evpk_create(ct.byref(g), ct.byref(self._ctx)):
            raise EvpkError("evpk_create: " + L.evpk_last_error(None).decode())
        self.decomp = decomp
        self.device_strength = False     # True: evpk_step_in.strength = NULL, ice_strength runs on the device

    def _chk(self, rc, what):
        if rc:
            raise EvpkError(f"{what}: " + self._L.evpk_last_error(self._ctx).decode())

    def connect(self, unique_id: bytes):
        """evpk_connect: the collective half of the start (communicator / peer mapping)."""
        self._uid = ct.create_string_buffer(unique_id, UNIQUE_ID_BYTES)
        self._chk(self._L.evpk_connect(self._ctx, ct.cast(self._uid, ct.c_void_p)), "evpk_connect")

    def set_params(self, p: Params):
        self._chk(self._L.evpk_set_params(self._ctx, ct.byref(p)), "evpk_set_params")

    def _step_in(self, f) -> StepIn:
        si = StepIn()
        for n in STEP_IN_F64:
            setattr(si, n, _p64(f.get(n)))
        if self.device_strength:            # strength == NULL: ice_strength runs on the device
            si.strength = None
        return si

    def _state(self, f) -> State:
        """evpk_state from the arrays present in `f` (an absent output is a NULL pointer: skipped by evpk_download)"""
        st = State()
        st.uvel, st.vvel = _p64(f.get("uvel")), _p64(f.get("vvel"))
        for k in ("stressp", "stressm", "stress12"):
            setattr(st, k, (c_f64p * 4)(*[_p64(f.get(f"{k}_{c}")) for c in (1, 2, 3, 4)]))
        st.iceumask = _p32(f.get("iceumask"))
        for n in STATE_OUT_F64:
            setattr(st, n, _p64(f.get(n)))
        st.icetmask = _p32(f.get("icetmask"))
        st.strength = _p64(f.get("strength"))        # comes back with its ghost cells halo-updated (ice_dyn_evp.F90:311-312)
        return st

    def run(self, f):
        si, st = self._step_in(f), self._state(f)
        self._chk(self._L.evpk_run(self._ctx, ct.byref(si), ct.byref(st)), "evpk_run")

    def upload(self, f):
        si, st = self._step_in(f), self._state(f)
        self._chk(self._L.evpk_upload(self._ctx, ct.byref(si), ct.byref(st)), "evpk_upload")

    def upload_inputs(self, f):
        """inputs only; the prognostic state stays resident on the device (evpk_upload with state == NULL)"""
        si = self._step_in(f)
        self._chk(self._L.evpk_upload(self._ctx, ct.byref(si), None), "evpk_upload")

    def prep(self):
        self._chk(self._L.evpk_prep(self._ctx), "evpk_prep")

    def subcycle(self, nsub: int):
        self._chk(self._L.evpk_subcycle(self._ctx, int(nsub)), "evpk_subcycle")

    def finish(self):
        self._chk(self._L.evpk_finish(self._ctx), "evpk_finish")

    def download(self, f):
        st = self._state(f)
        self._chk(self._L.evpk_download(self._ctx, ct.byref(st)), "evpk_download")

    def sync(self):
        self._chk(self._L.evpk_sync(self._ctx), "evpk_sync")

    def stats(self) -> Stats:
        s = Stats()
        self._chk(self._L.evpk_get_stats(self._ctx, ct.byref(s)), "evpk_get_stats")
        return s

    def principal_stress(self, sig1: np.ndarray, sig2: np.ndarray):
        self._chk(self._L.evpk_principal_stress(self._ctx, _p64(sig1), _p64(sig2)), "evpk_principal_stress")

    def halo_update(self, a: np.ndarray, field_loc: int, field_type: int, fill: float = 0.0):
        """evpk_halo_update (ice_HaloUpdate): a is (nblocks, ny_block, nx_block) or (nblocks, nz, ny_block, nx_block), in place"""
        assert a.ndim in (3, 4)
        self._chk(self._L.evpk_halo_update(self._ctx, _p64(a), 0 if a.ndim == 3 else int(a.shape[1]), int(field_loc), int(field_type),
                                           float(fill)), "evpk_halo_update")

    def halo_update_stress(self, a1: np.ndarray, a2: np.ndarray):
        """evpk_halo_update_stress (ice_HaloUpdate_stress(array1 = a1, array2 = a2, centre, scalar)): a1 in place"""
        self._chk(self._L.evpk_halo_update_stress(self._ctx, _p64(a1), _p64(a2)), "evpk_halo_update_stress")

    def transport_upwind(self, dt: float, works: np.ndarray):
        """evpk_transport_upwind: works is (nblocks, narr, ny_block, nx_block), advected in place"""
        assert works.ndim == 4
        self._chk(self._L.evpk_transport_upwind(self._ctx, float(dt), int(works.shape[1]), _p64(works)), "evpk_transport_upwind")

    def transport_upwind_state(self, dt: float, aice0, aicen, vicen, vsnon, trcrn, ntrcr: int, trcr_depend, nt_Tsfc=1, nt_alvl=0, nt_apnd=0,
                               nt_fbri=0, ponds=(0, 0, 0), Tocnfrz=-1.8):
        """evpk_transport_upwind_state: aice0 (nb, ny, nx), aicen / vicen / vsnon (nb, ncat, ny, nx), trcrn (nb, ncat, ntrcr_dim, ny, nx), in place"""
        dep = np.ascontiguousarray(trcr_depend, dtype=np.int32)
        self._chk(self._L.evpk_transport_upwind_state(self._ctx, float(dt), int(aicen.shape[1]), int(ntrcr), int(trcrn.shape[2]), _p32(dep),
                                                      int(nt_Tsfc), int(nt_alvl), int(nt_apnd), int(nt_fbri), *[int(p) for p in ponds], float(Tocnfrz),
                                                      _p64(aice0), _p64(aicen), _p64(vicen), _p64(vsnon), _p64(trcrn)), "evpk_transport_upwind_state")

    def remap_init(self, dxu: np.ndarray, dyu: np.ndarray, hm: np.ndarray):
        """evpk_remap_init: the grid arrays of horizontal_remap beyond the geometry's (block arrays)"""
        self._chk(self._L.evpk_remap_init(self._ctx, _p64(dxu), _p64(dyu), _p64(hm)), "evpk_remap_init")

    def transport_remap(self, dt: float, mm: np.ndarray, tm: Optional[np.ndarray], tracer_type, depend, has_dependents,
                        integral_order: int = 3, l_dp_midpt: bool = True, l_fixed_area: bool = False) -> int:
        """evpk_transport_remap: mm (nblocks, ncat+1, ny_block, nx_block), tm (nblocks, ncat, ntrace, ny_block, nx_block) in
        place.  Returns 0, REMAP_BAD_DEPARTURE or REMAP_NEGATIVE_MASS (the reference's two l_stop cases); raises otherwise."""
        assert mm.ndim == 4 and mm.flags["C_CONTIGUOUS"]
        ncat = mm.shape[1] - 1
        ntrace = 0 if tm is None else int(tm.shape[2])
        if ntrace:
            assert tm.ndim == 5 and tm.shape[1] == ncat and tm.flags["C_CONTIGUOUS"]
        tt, dp, hd = (np.ascontiguousarray(a, dtype=np.int32) for a in (tracer_type, depend, has_dependents))
        rc = self._L.evpk_transport_remap(self._ctx, float(dt), ncat, ntrace, _p64(mm), _p64(tm) if ntrace else None,
                                          _p32(tt) if ntrace else None, _p32(dp) if ntrace else None, _p32(hd) if ntrace else None,
                                          int(integral_order), int(l_dp_midpt), int(l_fixed_area))
        if rc not in (0, REMAP_BAD_DEPARTURE, REMAP_NEGATIVE_MASS):
            self._chk(rc, "evpk_transport_remap")
        return int(rc)

    def transport_remap_state(self, dt: float, aice0, aicen, vicen, vsnon, trcrn, ntrcr: int, nt_qsno: int, nslyr: int, rhos_lfresh: float,
                              tracer_type, depend, has_dependents, integral_order: int = 3, l_dp_midpt: bool = True) -> int:
        """evpk_transport_remap_state: aice0 (nblocks, ny, nx), aicen / vicen / vsnon (nblocks, ncat, ny, nx), trcrn (nblocks, ncat, ntrcr_dim,
        ny, nx) in place -- state_to_tracers, horizontal_remap, tracers_to_state and bound_state on the device"""
        ncat, ntrcr_dim = aicen.shape[1], (trcrn.shape[2] if trcrn is not None else 0)
        tt, dp, hd = (np.ascontiguousarray(a, dtype=np.int32) for a in (tracer_type, depend, has_dependents))
        rc = self._L.evpk_transport_remap_state(self._ctx, float(dt), ncat, int(ntrcr), ntrcr_dim, int(nt_qsno), int(nslyr), float(rhos_lfresh),
                                                _p64(aice0), _p64(aicen), _p64(vicen), _p64(vsnon), _p64(trcrn), _p32(tt), _p32(dp), _p32(hd),
                                                int(integral_order), int(l_dp_midpt))
        if rc not in (0, REMAP_BAD_DEPARTURE, REMAP_NEGATIVE_MASS):
            self._chk(rc, "evpk_transport_remap_state")
        return int(rc)

    def eap_init(self, tables):
        """evpk_eap_init: the six lookup tables of init_eap, each [na_yield][ny_yield][nx_yield]; the context then runs eap(dt)"""
        na, ny, nx = tables[0].shape
        assert all(t.shape == (na, ny, nx) and t.dtype == np.float64 and t.flags["C_CONTIGUOUS"] for t in tables) and len(tables) == 6
        self._chk(self._L.evpk_eap_init(self._ctx, nx, ny, na, *[_p64(t) for t in tables]), "evpk_eap_init")

    def _eap_state(self, f, names) -> EapState:
        st = EapState()
        st.a11_c = (c_f64p * 4)(*[_p64(f.get(f"a11_{c}") if f"a11_{c}" in names else None) for c in (1, 2, 3, 4)])
        st.a12_c = (c_f64p * 4)(*[_p64(f.get(f"a12_{c}") if f"a12_{c}" in names else None) for c in (1, 2, 3, 4)])
        for n in EAP_HISTORY:
            setattr(st, n, _p64(f.get(n) if n in names else None))
        return st

    def eap_upload(self, f):
        """evpk_eap_upload: a11_1..4, a12_1..4 of `f` (those present)"""
        st = self._eap_state(f, [n for n in f if n[:4] in ("a11_", "a12_")])
        self._chk(self._L.evpk_eap_upload(self._ctx, ct.byref(st)), "evpk_eap_upload")

    def eap_download(self, f, names=None):
        """evpk_eap_download: the structure tensor and the EAP history fields present in `f` (or only `names`)"""
        st = self._eap_state(f, list(f) if names is None else names)
        self._chk(self._L.evpk_eap_download(self._ctx, ct.byref(st)), "evpk_eap_download")

    def restart_write(self, path: str, append: bool = False, big_endian: bool = True):
        self._chk(self._L.evpk_restart_write(self._ctx, path.encode(), int(append), int(big_endian)), "evpk_restart_write")

    def restart_read(self, path: str, byte_offset: int = 0, big_endian: bool = True):
        self._chk(self._L.evpk_restart_read(self._ctx, path.encode(), int(byte_offset), int(big_endian)), "evpk_restart_read")

    def calibrate(self, nrep: int = 3):
        self._chk(self._L.evpk_calibrate(self._ctx, int(nrep)), "evpk_calibrate")

    def close(self):
        if self._ctx:
            self._L.evpk_destroy(self._ctx)
            self._ctx = ct.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
