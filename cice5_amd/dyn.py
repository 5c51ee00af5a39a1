"""Host-side mirror of the reference's EVP module interface (same names and argument meaning):

    ice_dyn_shared: set_evp_parameters(dt), init_evp(dt)      source/ice_dyn_shared.F90:99-259
    ice_dyn_evp:    evp(dt)                                   source/ice_dyn_evp.F90:68

The module-global arrays of ice_state / ice_flux / ice_grid are one dict of block-layout
numpy arrays (`fields`), keyed by the reference's variable names.  All arithmetic of the
path runs in libevpk (HIP); this file only computes the handful of scalars that
set_evp_parameters derives on the host in the reference too.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import constants as C
from . import evpk
from .blocks import Decomp


def set_evp_parameters(dt: float, ndte: int, revised_evp: bool, xmin: float, *,
                       cosw: float = C.cosw, sinw: float = C.sinw,
                       tilt_from_slope: bool = False, wind_on_ugrid: bool = False,
                       kstrength: int = 1, krdg_partic: int = 1, krdg_redist: int = 1, ncat: int = 5,
                       mu_rdg: float = 3.0, Cf: float = 17.0, sparse_io: bool = False) -> evpk.Params:
    """ice_dyn_shared.F90:185-259.  `xmin` = min(global_minval(dxt,tmask), global_minval(dyt,tmask)) (:221-223)."""
    p = evpk.Params()
    dte = dt / float(ndte)                       # :209
    dtei = 1.0 / dte                             # :210
    p.dt, p.ndte, p.revised_evp = dt, ndte, int(revised_evp)
    p.ecci = 0.25                                # :214
    tdamp2 = 2.0 * C.eyc * dt                    # :217
    dte2T = dte / tdamp2                         # :218
    Se, xi = 0.86, 5.5e-3                        # :226-227
    gamma = 0.25 * 1.0e11 * dt                   # :228
    if revised_evp:                              # :230-233
        p.revp = 1.0
        p.arlx1i = 2.0 * xi / Se
        p.brlx = 2.0 * Se * xi * gamma / xmin ** 2
    else:                                        # :239-242
        p.revp = 0.0
        p.arlx1i = dte2T
        p.brlx = dt * dtei
    p.denom1 = 1.0 / (1.0 + p.arlx1i)            # :257
    p.cosw, p.sinw = cosw, sinw
    p.rhow, p.rhoi, p.rhos, p.gravit = C.rhow, C.rhoi, C.rhos, C.gravit
    p.a_min, p.m_min = C.a_min, C.m_min
    p.tilt_from_slope, p.wind_on_ugrid = int(tilt_from_slope), int(wind_on_ugrid)
    # ice_strength switches (ice_mechred.F90:54-64; defaults ice_init.F90:273-277), used when the strength is not an input
    p.kstrength, p.krdg_partic, p.krdg_redist, p.ncat = kstrength, krdg_partic, krdg_redist, ncat
    p.mu_rdg, p.Cf = mu_rdg, Cf
    p.sparse_io = int(sparse_io)       # sparse transfers in resident-state mode (include/evpk.h)
    return p


def local_min_dx(fields: Dict[str, np.ndarray], decomp: Decomp) -> float:
    """This rank's contribution to global_minval(dxt/dyt, tmask) over physical cells."""
    m = np.inf
    for n, b in enumerate(decomp.local_blocks):
        sl = (n, slice(b.jlo - 1, b.jhi), slice(b.ilo - 1, b.ihi))
        mk = fields["tmask"][sl] > 0
        if mk.any():
            m = min(m, float(fields["dxt"][sl][mk].min()), float(fields["dyt"][sl][mk].min()))
    return m


EVERY_STEP_OUTPUTS = ["uvel", "vvel", "rdg_conv", "rdg_shear", "divu", "shear", "strocnxT", "strocnyT"]


class EvpDynamics:
    """One rank's EVP solver.  Usage mirrors the reference call order:
    `init_evp(dt)` once (CICE_InitMod.F90:103-104), then `evp(dt)` every dynamics step
    (ice_step_mod.F90:1119)."""

    def __init__(self, decomp: Decomp, fields: Dict[str, np.ndarray], *, ndte: int = 120, revised_evp: bool = False,
                 device: int = 0, unique_id: Optional[bytes] = None, cosw: float = C.cosw, sinw: float = C.sinw,
                 tilt_from_slope: bool = False, wind_on_ugrid: bool = False, xmin: Optional[float] = None,
                 pin_host: bool = False, device_strength: Optional[dict] = None, defer_connect: bool = False,
                 resident: bool = False, outputs: Optional[list] = None, sparse_io: bool = False):
        """pin_host: page-lock the arrays of `fields` (evpk_pin_host) as a host model does once for its module arrays;
        evp() then moves them in place over PCIe.  The arrays must stay the same objects until close() (this object keeps
        a reference to each, so none can be freed while it is registered)."""
        self.decomp, self.fields = decomp, fields
        self.ndte, self.revised_evp = ndte, revised_evp
        self._opts = dict(cosw=cosw, sinw=sinw, tilt_from_slope=tilt_from_slope, wind_on_ugrid=wind_on_ugrid)
        if device_strength is not None:
            # ice_strength (ice_mechred.F90:2111) on the device instead of fields["strength"] as an input: a dict of
            # kstrength / krdg_partic / krdg_redist / ncat / mu_rdg / Cf; kstrength = 1 reads fields["aicen", "vicen", "aice0"]
            self._opts.update(device_strength)
        # resident: uvel, vvel, sigma, iceumask stay on the device between evp() calls (only evp writes them: the host
        # re-uploads inputs only, evpk_upload(in, NULL)); outputs: the arrays to bring back every call (None = all) -- what a
        # host model reads every step is uvel, vvel (transport), rdg_conv, rdg_shear (ridging), strocnxT/yT (coupler), divu,
        # shear; the rest on the steps that write history or a restart; sparse_io: evpk_params.sparse_io
        self._resident, self._outputs, self._ncalls = resident, outputs, 0
        self._opts["sparse_io"] = int(sparse_io)      # 0, 1 (aice, vice, vsno whole), 2 (aice whole): evpk_params.sparse_io
        self._xmin = xmin
        self.ctx = evpk.Context(decomp, fields, device=device, unique_id=unique_id, defer_connect=defer_connect)
        self.ctx.device_strength = device_strength is not None
        self.params: Optional[evpk.Params] = None
        # pin_host = True / "register": evpk_pin_host of the caller's own arrays (hipHostRegister; what a host with static module
        # arrays can do); "alloc": the arrays of `fields` are MOVED to page-locked memory the driver allocates (evpk_host_alloc:
        # the dict's entries are replaced by arrays there, same contents) -- memory that cannot migrate under a kernel
        if pin_host == "alloc":
            for k, a in list(fields.items()):
                if isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"] and not evpk.host_is_mapped(a):
                    fields[k] = evpk.host_copy(a)
            self._pinned = []
        else:
            self._pinned = [a for a in fields.values() if isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"]
                            and evpk.pin_host(a)] if pin_host else []

    def connect(self, unique_id: bytes):
        """second phase of a multi-rank start (EvpDynamics(..., defer_connect=True)): collective over the ranks"""
        self.ctx.connect(unique_id)

    def set_evp_parameters(self, dt: float):
        xmin = self._xmin if self._xmin is not None else local_min_dx(self.fields, self.decomp)
        self.params = set_evp_parameters(dt, self.ndte, self.revised_evp, xmin, **self._opts)
        self.ctx.set_params(self.params)

    def init_evp(self, dt: float):
        """ice_dyn_shared.F90:99-174: parameters, then velocities / stresses / masks at rest."""
        self.set_evp_parameters(dt)
        f = self.fields
        for n in ["uvel", "vvel", "divu", "shear", "rdg_conv", "rdg_shear"] + \
                 [f"{k}_{c}" for k in ("stressp", "stressm", "stress12") for c in (1, 2, 3, 4)]:
            f[n][...] = 0.0
        f["iceumask"][...] = 0

    def evp(self, dt: float):
        """ice_dyn_evp.F90:68: one call of the dynamics, in place on `fields`."""
        if self.params is None or self.params.dt != dt:
            self.set_evp_parameters(dt)        # :153-154 ("needed only if dt changes during runtime")
        if not self._resident:
            self.ctx.run(self.fields)
            return
        if self._ncalls == 0:
            self.ctx.upload(self.fields)
        else:
            self.ctx.upload_inputs(self.fields)
        self._ncalls += 1
        self.ctx.prep()
        self.ctx.subcycle(self.ndte)
        self.ctx.finish()
        f = self.fields if self._outputs is None else {n: self.fields[n] for n in self._outputs}
        self.ctx.download(f)

    # ---- kdyn = 2: the elastic-anisotropic-plastic rheology (source/ice_dyn_eap.F90) ----
    def init_eap(self, dt: float, tables=None):
        """ice_dyn_eap.F90:493-621: init_evp, isotropic structure tensor, the lookup tables (eap_tables.py, or `tables` as a
        host's own init_eap made them, [na_yield][ny_yield][nx_yield] each).  evp()/eap() of this object then run eap(dt)."""
        if tables is None:
            from .eap_tables import eap_tables
            tables = eap_tables()
        self.init_evp(dt)
        f = self.fields
        for c in (1, 2, 3, 4):
            f[f"a11_{c}"][...] = 0.5
            f[f"a12_{c}"][...] = 0.0
        for n in evpk.EAP_HISTORY:
            if n in f:
                f[n][...] = 0.0
        self.ctx.eap_init(tables)
        self._eap = True

    def eap(self, dt: float):
        """ice_dyn_eap.F90:66: one call of the EAP dynamics, in place on `fields` (the structure tensor a11_1..4, a12_1..4 is
        resident on the device between calls and comes back with the history fields every call)"""
        assert getattr(self, "_eap", False), "init_eap has not been called"
        self.evp(dt)
        self.ctx.eap_download(self.fields)

    def principal_stress(self):
        """ice_dyn_shared.F90:853: (sig1, sig2) block arrays from the state of the last evp() (physical cells)."""
        shp = self.fields["uvel"].shape
        sig1, sig2 = np.zeros(shp), np.zeros(shp)
        self.ctx.principal_stress(sig1, sig2)
        return sig1, sig2

    def close(self):
        self.ctx.close()
        for a in self._pinned:
            evpk.unpin_host(a)
        self._pinned = []
