"""The Fortran host side: fortran/ice_dyn_evp.F90 (drop-in module exporting evp(dt)) driven by
fortran/evp_driver.F90 through ISO_C_BINDING into libevpk.  The driver stands in for
ice_step_mod.F90:1119 `call evp (dt)`; the module arrays come from test-double modules
(fortran/mock) because the reference's own modules cannot be built here (netCDF)."""
import os
import subprocess

import numpy as np
import pytest

from cice5_amd import constants as C, dyn, synth
from oracle import orc
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "fortran", "evp_driver")

IN_F64 = ["dxt", "dyt", "dxhy", "dyhx", "cxp", "cyp", "cxm", "cym", "tarear", "uarear", "tinyarea", "tarea", "uarea", "fcor",
          "HTN", "HTE",
          "aice", "vice", "vsno", "aice_init", "strairxT", "strairyT", "strax", "stray", "uocn", "vocn", "ss_tltx", "ss_tlty",
          "Cdn_ocn", "strength", "uvel", "vvel"] + util.SIGMA
OUT_F64 = ["uvel", "vvel"] + util.SIGMA + ["divu", "shear", "rdg_conv", "rdg_shear", "prs_sig", "strintx", "strinty",
                                            "strocnx", "strocny", "strocnxT", "strocnyT", "strairx", "strairy", "strtltx",
                                            "strtlty", "fm", "uvel_init", "vvel_init"]


def write_fixture(path, d, f, p, ncalls, device_strength=None, eap_tables=None, remap=None):
    ga = d.geom_arrays()
    with open(path, "wb") as fh:
        np.array([d.nx_global, d.ny_global, d.nx_block, d.ny_block, d.nblocks, d.ew_boundary, d.ns_boundary,
                  p.ndte, p.revised_evp, ncalls], dtype=np.int32).tofile(fh)
        np.array([p.dt, p.revp, p.ecci, p.denom1, p.arlx1i, p.brlx, p.cosw, p.sinw], dtype=np.float64).tofile(fh)
        # geo(nb,6) in Fortran order: column-major -> six contiguous vectors
        for n in ("ilo", "ihi", "jlo", "jhi", "iglob_lo", "jglob_lo"):
            ga[n].astype(np.int32).tofile(fh)
        for n in IN_F64:
            f[n].tofile(fh)
        for n in ("tmask", "umask", "iceumask"):
            f[n].astype(np.int32).tofile(fh)
        if device_strength is not None:      # trailer: ice_strength on the device
            sw = device_strength
            np.array([1, f["aicen"].shape[1], sw.get("kstrength", 1), sw.get("krdg_partic", 1), sw.get("krdg_redist", 1), 0],
                     dtype=np.int32).tofile(fh)
            for n in ("aicen", "vicen", "aice0"):
                f[n].tofile(fh)
        if remap is not None:                # trailer: horizontal_remap after the last evp
            mm, tm, (ttype, depend, has), dt_r, order, midpt = remap
            np.array([3, mm.shape[1] - 1, tm.shape[2], order, int(midpt), 0], dtype=np.int32).tofile(fh)
            np.array([dt_r], dtype=np.float64).tofile(fh)
            for n in ("dxu", "dyu", "hm"):
                f[n].tofile(fh)
            for a in (ttype, depend, has):
                np.ascontiguousarray(a, dtype=np.int32).tofile(fh)
            mm.tofile(fh)
            tm.tofile(fh)
        if eap_tables is not None:           # trailer: kdyn = 2 -- table extents, tables, structure tensor
            na, ny, nx = eap_tables[0].shape
            np.array([2, nx, ny, na, 0, 0], dtype=np.int32).tofile(fh)
            for t in eap_tables:
                t.tofile(fh)
            for n in synth.EAP_STATE:
                f[n].tofile(fh)


def read_output(path, d, with_strength=False, with_eap=False, remap_shapes=None):
    shp = (d.nblocks, d.ny_block, d.nx_block)
    n = int(np.prod(shp))
    out = {}
    with open(path, "rb") as fh:
        for name in OUT_F64:
            out[name] = np.fromfile(fh, dtype=np.float64, count=n).reshape(shp)
        out["iceumask"] = np.fromfile(fh, dtype=np.int32, count=n).reshape(shp)
        if with_strength:
            out["strength"] = np.fromfile(fh, dtype=np.float64, count=n).reshape(shp)
        if remap_shapes is not None:
            for name, shp_ in zip(("aim", "trm"), remap_shapes):
                out[name] = np.fromfile(fh, dtype=np.float64, count=int(np.prod(shp_))).reshape(shp_)
        if with_eap:
            for name in synth.EAP_STATE + synth.EAP_HISTORY:
                out[name] = np.fromfile(fh, dtype=np.float64, count=n).reshape(shp)
    return out


def test_fortran_driver_is_built_and_links_libevpk():
    assert os.path.exists(DRIVER) and os.path.exists(DRIVER + "_auscom"), "run __graft_entry__.build() (make -C fortran)"
    out = subprocess.run(["ldd", DRIVER], capture_output=True, text=True).stdout
    assert "libevpk.so" in out and "not found" not in out.split("libevpk.so")[1].split("\n")[0]
    src = open(os.path.join(ROOT, "fortran", "ice_dyn_evp.F90")).read()
    assert "module ice_dyn_evp" in src and "public :: evp" in src and "subroutine evp (dt)" in src


def test_fortran_driver_aborts_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    case, d, f = util.make_case(24, 20, 12, 10)
    p = dyn.set_evp_parameters(3600.0, 4, False, synth.global_min_dx(case))
    write_fixture(tmp_path / "in.bin", d, f, p, 1)
    r = subprocess.run([DRIVER, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)     # abort_ice with evpk_last_error


@pytest.mark.gpu
@pytest.mark.parametrize("ns,bs,ncalls,driver", [("open", (100, 116), 1, DRIVER), ("open", (25, 29), 2, DRIVER),
                                                 ("tripole", (25, 29), 2, DRIVER), ("open", (50, 58), 2, DRIVER + "_auscom")])
def test_fortran_host_matches_oracle(tmp_path, ns, bs, ncalls, driver):
    """BASELINE config 1 shape (gx3-size 100x116, ndte=120) through the Fortran host; the `_auscom` driver is the
    same source compiled with -DAusCOM -DACCESS, the cpp flags the COSIMA fork needs (sicemass = tmass hook)."""
    case, d, f = util.make_case(100, 116, *bs, ns=ns, land="continents")
    xmin = synth.global_min_dx(case)
    p = dyn.set_evp_parameters(3600.0, 120, False, xmin)
    write_fixture(tmp_path / "in.bin", d, f, p, ncalls)
    r = subprocess.run([driver, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "evp_driver:" in r.stdout
    print(r.stdout)
    npin = int(r.stdout.split("page-locked host arrays = ")[1].split()[0])
    assert npin > 0          # the module arrays are moved in place over PCIe (some may share a page and stay staged)
    if driver.endswith("_auscom"):
        assert "sicemass max" in r.stdout
    got = read_output(tmp_path / "out.bin", d)
    fo = util.clone(f)
    po = orc.make_params(3600.0, 120, xmin)
    for _ in range(ncalls):
        orc.evp(d, po, fo)
    bad = util.compare(d, got, fo, names=list(got.keys()))
    assert not bad, bad[:6]
    assert np.abs(got["uvel"]).max() > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("sw", [dict(kstrength=1, krdg_partic=1, krdg_redist=1), dict(kstrength=1, krdg_partic=0, krdg_redist=0)])
def test_fortran_host_with_ice_strength_on_the_device(tmp_path, sw):
    """evpk_device_strength = .true.: the module hands aicen / vicen / aice0 and the ice_mechred switches over instead of
    calling evp_prep1 + ice_HaloUpdate + ice_strength on the host; oracle: evp with strength_mode = 1."""
    case, d, f = util.make_case(100, 116, 25, 29, land="continents")
    synth.add_thickness_distribution(f)
    f["strength"][...] = 0.0
    xmin = synth.global_min_dx(case)
    p = dyn.set_evp_parameters(3600.0, 60, False, xmin)
    write_fixture(tmp_path / "in.bin", d, f, p, 2, device_strength=sw)
    r = subprocess.run([DRIVER, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    got = read_output(tmp_path / "out.bin", d, with_strength=True)
    fo = util.clone(f)
    po = orc.make_params(3600.0, 60, xmin, strength_mode=1, **sw)
    for _ in range(2):
        orc.evp(d, po, fo)
    names = [n for n in got if n != "strength"]
    bad = util.compare(d, got, fo, names=names)
    assert not bad, bad[:6]
    m = util.cell_mask(d, "all")
    assert np.array_equal(got["strength"][m], fo["strength"][m]) and fo["strength"].max() > 1e3


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(int(os.environ.get("EVPK_FUZZ_F_N", "6"))))
def test_fortran_host_random_configuration(tmp_path, seed):
    """Seeded random shapes through the Fortran module: padded edge blocks, boundary types, 1-3 calls (the second call on
    takes the inputs-only upload with the state resident on the device), host or device ice_strength."""
    rng = np.random.default_rng(900 + seed)
    nx, ny = int(rng.choice([24, 61, 96, 130])), int(rng.choice([12, 23, 40, 57]))
    ns = str(rng.choice(["open", "tripole", "closed"]))
    nx += (nx & 1) if ns == "tripole" else 0
    bsx = min(nx, int(rng.choice([7, 20, max(3, nx // 3 + 1), nx])))
    bsy = min(ny, int(rng.choice([5, 11, max(3, ny // 2), ny])))
    if ns == "tripole" and ny % bsy == 1 and bsy < ny:
        bsy += 1
    ndte, ncalls = int(rng.choice([1, 2, 7, 12])), int(rng.choice([1, 2, 3]))
    sw = None if rng.random() < 0.5 else dict(kstrength=int(rng.integers(0, 2)), krdg_partic=int(rng.integers(0, 2)),
                                              krdg_redist=int(rng.integers(0, 2)))
    case, d, f = util.make_case(nx, ny, bsx, bsy, ns=ns, land="continents", ice=str(rng.choice(["polar", "full"])))
    if sw:
        synth.add_thickness_distribution(f)
    xmin = synth.global_min_dx(case)
    p = dyn.set_evp_parameters(3600.0, ndte, False, xmin)
    write_fixture(tmp_path / "in.bin", d, f, p, ncalls, device_strength=sw)
    r = subprocess.run([DRIVER, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    got = read_output(tmp_path / "out.bin", d, with_strength=sw is not None)
    fo = util.clone(f)
    po = orc.make_params(3600.0, ndte, xmin, **(dict(strength_mode=1, **sw) if sw else {}))
    for _ in range(ncalls):
        orc.evp(d, po, fo)
    bad = util.compare(d, got, fo, names=[n for n in got if n != "strength"])
    assert not bad, ((nx, ny, bsx, bsy, ns, ndte, ncalls, sw), bad[:4])


@pytest.mark.gpu
@pytest.mark.parametrize("ns,bs,ncalls", [("open", (50, 58), 1), ("tripole", (25, 29), 2)])
def test_fortran_host_eap(tmp_path, ns, bs, ncalls):
    """kdyn = 2 from Fortran: `call evpk_eap (dt, nx_yield, ..., s11r, ..., a11_1, ..., s22)` -- the call that replaces the body
    of the reference's `subroutine eap (dt)` -- through ISO_C_BINDING (evpk_eap_state, evpk_eap_init / upload / download) into
    the HIP kernels, against the oracle"""
    from cice5_amd.eap_tables import eap_tables
    T = eap_tables()
    case, d, f = util.make_case(100, 116, *bs, ns=ns, land="continents")
    synth.add_eap_state(f)
    rng = np.random.default_rng(8)
    for n in synth.EAP_STATE:                  # a structure tensor "from a restart"
        f[n] = rng.uniform(0.35, 0.65, f["uvel"].shape) if n.startswith("a11") else rng.uniform(-0.1, 0.1, f["uvel"].shape)
        orc.halo_r8(d, f[n], C.LOC_CENTER, C.KIND_SCALAR, 0.0)
    xmin = synth.global_min_dx(case)
    write_fixture(str(tmp_path / "in.bin"), d, f, dyn.set_evp_parameters(3600.0, 21, False, xmin), ncalls, eap_tables=T)
    p = orc.make_params(3600.0, 21, xmin)
    r = subprocess.run([DRIVER, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    out = read_output(str(tmp_path / "out.bin"), d, with_eap=True)
    fo = util.clone(f)
    for _ in range(ncalls):
        orc.eap(d, p, fo, T)
    assert not util.compare(d, out, fo, names=OUT_F64 + ["iceumask"])
    ne = util.cell_mask(d, "ne")
    for n in synth.EAP_STATE + synth.EAP_HISTORY:
        assert np.array_equal(out[n][ne], fo[n][ne]), n
    assert np.abs(fo["a11"]).max() > 0.3


@pytest.mark.gpu
@pytest.mark.parametrize("ns,bs,order,midpt", [("open", (50, 58), 3, True), ("tripole", (25, 29), 2, False)])
def test_fortran_host_horizontal_remap(tmp_path, ns, bs, order, midpt):
    """transport_remap's `call horizontal_remap (dt, ntrace, uvel, vvel, aim, trm, l_fixed_area, tracer_type, depend,
    has_dependents, integral_order, l_dp_midpt)` (ice_transport_driver.F90:475-481) through the shim's evpk_horizontal_remap,
    after an evp from the same Fortran host: aim / trm against the oracle's horizontal_remap on the oracle's velocities"""
    case, d, f = util.make_case(100, 116, *bs, ns=ns, land="continents")
    synth.add_thickness_distribution(f)
    synth.add_remap_grid(case, d, f)
    xmin = synth.global_min_dx(case)
    ncat = f["aicen"].shape[1]
    tables = orc.remap_tables([0, 1, 2 + 1, 2 + 2])
    ntrace = len(tables[0])
    mm = np.zeros((d.nblocks, ncat + 1) + f["aice0"].shape[1:])
    tm = np.zeros((d.nblocks, ncat, ntrace) + f["aice0"].shape[1:])
    mm[:, 0] = f["aice0"]
    for n in range(ncat):
        a, v = f["aicen"][:, n], f["vicen"][:, n]
        mm[:, n + 1] = a
        h = np.where(a > 1e-11, v / np.where(a > 1e-11, a, 1.0), 0.0)
        tm[:, n, 0], tm[:, n, 1] = h, 0.2 * h
        for k in range(2, ntrace):
            tm[:, n, k] = np.where(a > 1e-11, -3.0 - n - 0.5 * k + 0.1 * h, 0.0)
    for arr in (mm.reshape(d.nblocks, -1, *mm.shape[2:]), tm.reshape(d.nblocks, -1, *mm.shape[2:])):
        for k in range(arr.shape[1]):
            w = np.ascontiguousarray(arr[:, k]); orc.halo_r8(d, w, C.LOC_CENTER, C.KIND_SCALAR, 0.0); arr[:, k] = w
    fo = util.clone(f)
    orc.evp(d, orc.make_params(3600.0, 30, xmin), fo)
    dt_r = 0.35 * xmin / max(np.abs(fo["uvel"]).max(), np.abs(fo["vvel"]).max())
    mo, to = mm.copy(), tm.copy()
    assert orc.horizontal_remap(d, dt_r, fo, mo, to, *tables, integral_order=order, l_dp_midpt=midpt) == 0
    write_fixture(str(tmp_path / "in.bin"), d, f, dyn.set_evp_parameters(3600.0, 30, False, xmin), 1,
                  remap=(mm, tm, tables, dt_r, order, midpt))
    r = subprocess.run([DRIVER, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    out = read_output(str(tmp_path / "out.bin"), d, remap_shapes=(mm.shape, tm.shape))
    assert not util.compare(d, out, fo, names=["uvel", "vvel"])
    assert np.abs(mo - mm).max() > 1e-4
    assert np.array_equal(out["aim"], mo) and np.array_equal(out["trm"], to)

