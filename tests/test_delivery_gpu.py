"""The in-place (zero-copy) transfers into page-locked caller arrays -- the path the Fortran drop-in uses for the host model's
module arrays -- and the fences round 5 put around it (profiles/r04_v6/fuzz.txt: one value of one plane arrived as the array's old
content once in ~17 000 draws):
  * the library moves an array in place only inside ranges IT registered (evpk_pin_host) or allocated (evpk_host_alloc);
  * EVPK_VERIFY_DELIVERY delivers every in-place plane a second time through the staged path and compares;
  * arrays on evpk_host_alloc memory (hipHostMalloc: cannot migrate) give the oracle's bits."""
import ctypes as ct

import numpy as np
import pytest

from cice5_amd import dyn, evpk, synth
from oracle import orc
from tests import util

pytestmark = pytest.mark.gpu


def _case(ns="open"):
    nx, ny = (100, 116) if ns == "open" else (96, 64)
    case, d, f = util.make_case(nx, ny, 25 if ns == "open" else 24, 29 if ns == "open" else 32, ns=ns, land="continents")
    return case, d, f, synth.global_min_dx(case)


@pytest.mark.parametrize("pin", [True, "alloc"])
@pytest.mark.parametrize("ns", ["open", "tripole"])
def test_verified_delivery_into_page_locked_arrays(ns, pin, monkeypatch):
    """every plane evpk_download writes in place is checked against a staged copy of the same device state: all equal, and the
    result is the oracle's -- for registered numpy arrays and for arrays on driver-allocated page-locked memory"""
    monkeypatch.setenv("EVPK_VERIFY_DELIVERY", "1")
    case, d, f, xmin = _case(ns)
    fo, fg = util.clone(f), util.clone(f)
    orc.evp(d, orc.make_params(3600.0, 20, xmin), fo)
    s = dyn.EvpDynamics(d, fg, ndte=20, xmin=xmin, pin_host=pin)
    assert all(evpk.host_is_mapped(fg[n]) for n in ("uvel", "stressm_4", "aice", "iceumask"))
    s.init_evp(3600.0)
    s.evp(3600.0)
    st = s.ctx.stats()
    s.close()
    assert not util.compare(d, fg, fo)
    assert st.delivery_bad == 0 and st.delivery_checked > 30 * d.nblocks * 20 * 20


@pytest.mark.parametrize("mode", [1, 2])
def test_delivery_check_sees_a_lost_write(mode, monkeypatch):
    """the detector itself: a test hook makes one in-place value differ from what the device holds; mode 1 fails the download and names
    plane, block, cell and page, mode 2 repairs the caller's array from the staged copy and counts the event"""
    monkeypatch.setenv("EVPK_VERIFY_DELIVERY", str(mode))
    monkeypatch.setenv("EVPK_VERIFY_INJECT", "9")
    case, d, f, xmin = _case()
    fo, fg = util.clone(f), util.clone(f)
    orc.evp(d, orc.make_params(3600.0, 8, xmin), fo)
    s = dyn.EvpDynamics(d, fg, ndte=8, xmin=xmin, pin_host=True)
    s.init_evp(3600.0)
    if mode == 1:
        with pytest.raises(evpk.EvpkError, match=r"delivery check: field \d+, block \d+, \(i,j\) = \(\d+,\d+\).*4-KiB page 0x[0-9a-f]+.*1 of \d+ delivered values"):
            s.evp(3600.0)
    else:
        s.evp(3600.0)
        st = s.ctx.stats()
        assert st.delivery_bad == 1 and not util.compare(d, fg, fo)
    s.close()


def test_only_ranges_the_library_registered_are_moved_in_place(monkeypatch):
    """an array the HIP runtime knows from a registration the library did not make (here: hipHostRegister called behind its back)
    takes the staged path -- mapped_alias trusts its own table, not hipPointerGetAttributes"""
    monkeypatch.setenv("EVPK_VERIFY_DELIVERY", "1")
    hip = ct.CDLL("libamdhip64.so")
    hip.hipHostRegister.argtypes = [ct.c_void_p, ct.c_size_t, ct.c_uint]
    hip.hipHostUnregister.argtypes = [ct.c_void_p]
    case, d, f, xmin = _case()
    fo, fg = util.clone(f), util.clone(f)
    orc.evp(d, orc.make_params(3600.0, 6, xmin), fo)
    s = dyn.EvpDynamics(d, fg, ndte=6, xmin=xmin)
    foreign = [a for a in fg.values() if a.flags["C_CONTIGUOUS"] and hip.hipHostRegister(ct.c_void_p(a.ctypes.data), a.nbytes, 0x2 | 0x1) == 0]
    assert len(foreign) > 40 and not any(evpk.host_is_mapped(a) for a in foreign)
    s.init_evp(3600.0)
    s.evp(3600.0)
    st = s.ctx.stats()
    for a in foreign:
        hip.hipHostUnregister(ct.c_void_p(a.ctypes.data))
    s.close()
    assert st.delivery_checked == 0            # nothing went in place
    assert not util.compare(d, fg, fo)


def test_pin_registry_rules():
    """overlapping registrations are refused, unpin needs the registered pointer, a sub-range of a registered array is mapped, a range
    that sticks out is not; host_empty arrays are mapped for their whole life and their memory goes back when the last view dies"""
    a = np.zeros(300000)
    assert evpk.pin_host(a)
    assert not evpk.pin_host(a) and not evpk.pin_host(a[1000:2000])
    assert evpk.host_is_mapped(a) and evpk.host_is_mapped(a[10:20])
    L = evpk.lib()
    assert L.evpk_host_is_mapped(ct.c_void_p(a.ctypes.data + 8), a.nbytes) == 0
    assert not evpk.unpin_host(a[5:])
    assert evpk.unpin_host(a) and not evpk.host_is_mapped(a) and not evpk.unpin_host(a)
    h = evpk.host_empty((7, 33))
    h[...] = 3.0
    v = h[2:4]
    addr = h.ctypes.data
    assert evpk.host_is_mapped(h) and evpk.host_is_mapped(v)
    del h
    assert L.evpk_host_is_mapped(ct.c_void_p(addr), 8) == 1 and v[0, 0] == 3.0
    del v
    import gc
    gc.collect()
    assert L.evpk_host_is_mapped(ct.c_void_p(addr), 8) == 0
