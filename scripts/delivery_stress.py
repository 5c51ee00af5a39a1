#!/usr/bin/env python3
"""Stress the delivery of evpk_download into page-locked caller arrays (the zero-copy path) WITHOUT the oracle in between:
the same device state is delivered twice -- in place into the page-locked arrays (k_scatter_* over PCIe) and through the staged
path (scatter into the staging buffer + hipMemcpy into plain arrays) -- and the two sets of arrays are compared bit for bit.
A lost or misdirected in-place write shows as a difference whatever the kernels computed.  (profiles/r04_v6/fuzz.txt: one 8-byte
value of one plane in ~17 000 draws arrived as the array's old content.)

usage: python scripts/delivery_stress.py [--iters N] [--seconds S] [--variant heap|aligned|alloc|heap_raw|stale] [--grid NXxNY] [--blocks BXxBY]
                                         [--ndte K] [--churn] [--oracle]
  heap      numpy heap arrays registered with evpk_pin_host (hardened: MADV_NOHUGEPAGE + mlock), as tests/test_fuzz_gpu.py pins them
  heap_raw  the same with EVPK_PIN_HARDEN=0 (round 4's registration)
  aligned   page-aligned arrays (one mmap each), registered
  alloc     arrays on evpk_host_alloc memory (hipHostMalloc)
  stale     hypothesis (c): arrays are registered, dropped WITHOUT unpinning, new arrays are allocated (often at the same
            addresses) and NOT registered; the library must take the staged path for them (its own table of live ranges has
            the old entries, which still cover the addresses -- the check is that results stay right either way)
  --churn   allocate / free unrelated numpy arrays of assorted sizes between iterations (heap reuse, trim, mmap thresholds)
  --oracle  also run the C oracle each iteration (its OpenMP threads and allocations, as in the fuzz test)
Prints one line per difference and a summary; exit code 1 if anything differed."""
import argparse, mmap, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--seconds", type=float, default=0.0)
ap.add_argument("--variant", default="heap")
ap.add_argument("--grid", default="200x40")
ap.add_argument("--blocks", default="20x20")
ap.add_argument("--ndte", type=int, default=2)
ap.add_argument("--churn", action="store_true")
ap.add_argument("--oracle", action="store_true")
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
if a.variant == "heap_raw":
    os.environ["EVPK_PIN_HARDEN"] = "0"
import numpy as np
from cice5_amd import blocks, dyn, evpk, synth
from tests import util


def vm():
    out = {}
    try:
        for line in open("/proc/vmstat"):
            k, v = line.split()
            if k in ("numa_pages_migrated", "numa_hint_faults", "numa_pte_updates", "thp_collapse_alloc", "thp_fault_alloc", "pgmigrate_success",
                     "compact_migrate_scanned", "thp_split_page", "pgfault", "pgmajfault"):
                out[k] = int(v)
    except OSError:
        pass
    return out


def setting(path):
    try:
        return open(path).read().strip()
    except OSError:
        return "?"


def aligned_copy(x):
    n = x.nbytes
    m = mmap.mmap(-1, max(n, 1) + mmap.PAGESIZE)
    y = np.frombuffer(m, dtype=x.dtype, count=x.size).reshape(x.shape)
    assert y.ctypes.data % mmap.PAGESIZE == 0
    y[...] = x
    return y


nx, ny = (int(v) for v in a.grid.split("x"))
bx, by = (int(v) for v in a.blocks.split("x"))
case = synth.SynthCase(nx=nx, ny=ny, ns_boundary="open", ew_boundary="closed", land="continents", ice="full")
d = blocks.create_distrb_cart(nx, ny, bx, by, ew_boundary_type="closed", ns_boundary_type="open")
f0 = synth.make_block_fields(case, d)
xmin = synth.global_min_dx(case)
rng = np.random.default_rng(a.seed)
print(f"delivery_stress: variant={a.variant} grid={nx}x{ny} blocks={bx}x{by} ({d.nblocks}) ndte={a.ndte} churn={a.churn} oracle={a.oracle}")
print("  THP:", setting("/sys/kernel/mm/transparent_hugepage/enabled"), "| khugepaged defrag:", setting("/sys/kernel/mm/transparent_hugepage/khugepaged/defrag"),
      "| numa_balancing:", setting("/proc/sys/kernel/numa_balancing"), "| compact_unevictable_allowed:", setting("/proc/sys/vm/compact_unevictable_allowed"),
      "| kernel:", os.uname().release)
v0, t0 = vm(), time.time()
nbad = nplanes = it = nstale_err = 0
junk = []
stale_keep = []
while True:
    if a.seconds > 0:
        if time.time() - t0 > a.seconds:
            break
    elif it >= a.iters:
        break
    it += 1
    if a.churn:
        junk = [np.empty(int(rng.integers(1, 400000))) for _ in range(int(rng.integers(0, 12)))]
        for j in junk[::2]:
            j[...] = 1.0
        del junk[::3]
    fg = util.clone(f0)
    keep = (np.sin(0.3 * np.arange(fg["aice"].size) + it) > 0.9).reshape(fg["aice"].shape).astype(np.float64)
    for name in ("aice", "vice", "vsno", "aice_init", "strength"):
        fg[name] *= keep
    pin = True
    if a.variant == "aligned":
        fg = {k: aligned_copy(v) for k, v in fg.items()}
    elif a.variant == "alloc":
        pin = "alloc"
    elif a.variant == "stale":
        # register a throw-away clone, drop it without unpinning, then run on NEW arrays that are not registered
        tmp = util.clone(f0)
        ok = [x for x in tmp.values() if evpk.pin_host(x)]
        stale_keep.append([(x.ctypes.data, x.nbytes) for x in ok])
        del tmp, ok
        fg = util.clone(fg)
        pin = False
    s = None
    try:
        s = dyn.EvpDynamics(d, fg, ndte=a.ndte, xmin=xmin, pin_host=pin)
        s.init_evp(3600.0)
        ref = util.clone(fg)                  # plain arrays: the staged delivery lands here
        s.evp(3600.0)
        s.ctx.download(ref)
    except evpk.EvpkError as e:
        if a.variant != "stale":
            raise
        # a registration that outlived its array: the HIP runtime refuses even a plain copy from memory that overlaps it
        nstale_err += 1
        if nstale_err <= 3:
            print(f"stale registration, iter {it}: {str(e)[:200]}", flush=True)
        if s is not None:
            s.close()
        continue
    if a.oracle:
        from oracle import orc
        fo = util.clone(f0)
        orc.evp(d, orc.make_params(3600.0, a.ndte, xmin), fo)
    for name in fg:
        x, y = fg[name], ref[name]
        if x.dtype.kind not in "fi" or x.shape != y.shape:
            continue
        nplanes += 1
        xb, yb = x.view(np.uint64 if x.itemsize == 8 else np.uint32), y.view(np.uint64 if y.itemsize == 8 else np.uint32)
        if not np.array_equal(xb, yb):
            idx = np.argwhere(xb != yb)
            nbad += len(idx)
            for ix in idx[:8]:
                off = int(np.ravel_multi_index(tuple(ix), x.shape)) * x.itemsize
                addr = x.ctypes.data + off
                print(f"DIFF iter {it} {name} {tuple(int(v) for v in ix)}: in-place {x[tuple(ix)]!r} staged {y[tuple(ix)]!r}  array base {x.ctypes.data:#x} "
                      f"(page offset {x.ctypes.data % 4096}) element address {addr:#x} page {addr >> 12:#x} elapsed {time.time() - t0:.1f} s", flush=True)
    s.close()
if a.variant == "stale":
    for lst in stale_keep:                    # release the registrations this test leaked on purpose
        for ptr, _ in lst:
            import ctypes as ct
            evpk.lib().evpk_unpin_host(ct.c_void_p(ptr))
v1 = vm()
print(f"  {it} iterations, {nplanes} planes compared, {nbad} values differ, {time.time() - t0:.1f} s" + (f"; {nstale_err} iterations refused by the runtime (copy from memory under a stale registration)" if a.variant == "stale" else ""))
print("  vmstat deltas:", {k: v1[k] - v0.get(k, 0) for k in v1})
sys.exit(1 if nbad else 0)
