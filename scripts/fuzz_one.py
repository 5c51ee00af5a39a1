#!/usr/bin/env python3
"""Re-run one draw of tests/test_fuzz_gpu.py::test_random_configuration and say where device and oracle differ.
usage: EVPK_FUZZ_BASE=<base> python scripts/fuzz_one.py <seed>[-<seed>][xREPS] [ENV=VALUE ...]   (a range runs in ONE process, like pytest)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FIXED, MODE_KEYS = [], []
for kv in sys.argv[2:]:
    a, b = kv.split("=", 1); os.environ[a] = b; FIXED.append(a)
import numpy as np
from cice5_amd import blocks, constants as C, dyn, synth
from oracle import orc
from tests import util, test_fuzz_gpu as t

def run(seed, quiet):
    k = t._config(seed)
    k["sparse"] = k["sparse"] and k["resident"] and not k["ugrid_wind"]
    k["pin"] = k["pin"] or k["sparse"] or bool(os.environ.get("EVPK_FUZZ_PIN"))      # (EVPK_FUZZ_PIN=1: every draw through page-locked arrays)
    for name in MODE_KEYS:                      # (pytest's monkeypatch restores the environment between draws)
        os.environ.pop(name, None)
    MODE_KEYS.clear()
    for name, v in k["mode"].items():
        if name not in FIXED:
            os.environ[name] = v; MODE_KEYS.append(name)
    if not quiet: print({a: b for a, b in k.items() if a != "rng"}, {a: os.environ[a] for a in os.environ if a.startswith("EVPK_")})
    case = synth.SynthCase(nx=k["nx"], ny=k["ny"], ns_boundary=C.BND_NAMES[k["ns"]], ew_boundary=C.BND_NAMES[k["ew"]],
                           land=k["land"], ice="full" if k["ice"] != "polar" else "polar")
    d = blocks.create_distrb_cart(k["nx"], k["ny"], k["bsx"], k["bsy"], ew_boundary_type=k["ew"], ns_boundary_type=k["ns"])
    f = synth.make_block_fields(case, d)
    base = util.clone(f)
    xmin = synth.global_min_dx(case)
    cosw, sinw = (np.cos(0.4), np.sin(0.4)) if k["turn"] else (1.0, 0.0)
    fo, fg = util.clone(f), util.clone(f)
    skw = dict(strength_mode=1, **k["strength"]) if k["strength"] else {}
    dt = k["dt"]
    p = orc.make_params(dt, k["ndte"], xmin, revised_evp=k["revised"], cosw=cosw, sinw=sinw, tilt_from_slope=k["tilt"], wind_on_ugrid=k["ugrid_wind"], **skw)
    s = dyn.EvpDynamics(d, fg, ndte=k["ndte"], revised_evp=k["revised"], xmin=xmin, cosw=cosw, sinw=sinw, tilt_from_slope=k["tilt"],
                        wind_on_ugrid=k["ugrid_wind"], device_strength=k["strength"], pin_host=k["pin"], sparse_io=k["sparse"])
    s.init_evp(dt)
    I, J = blocks.block_index_windows(d)
    nxg, nyg = d.nx_global, d.ny_global
    rng = k["rng"]
    for call in range(k["ncalls"]):
        if k["ice"] in ("patches", "dots"):
            kx, ky, ph = rng.uniform(0.05, 0.5), rng.uniform(0.05, 0.5), rng.uniform(0, 6.28)
            for ff in (fo, fg):
                for n in range(d.nblocks):
                    Ig = np.broadcast_to(I[n][None, :], (d.ny_block, d.nx_block)).copy()
                    Jg = np.broadcast_to(J[n][:, None], (d.ny_block, d.nx_block)).copy()
                    if k["ew"] == "cyclic":
                        Ig = (Ig - 1) % nxg + 1
                    if k["ns"] == "tripole":
                        top = Jg > nyg
                        Ig[top] = nxg - Ig[top] + 1
                        Jg = np.minimum(Jg, nyg)
                    w = np.sin(kx * Ig + ph) * np.cos(ky * Jg - ph)
                    keep = (w > (0.97 if k["ice"] == "dots" else 0.2)).astype(np.float64)
                    for name in ("aice", "vice", "vsno", "aice_init", "strength"):
                        ff[name][n] = base[name][n] * keep
        elif call:
            for ff in (fo, fg):
                ff["aice"] *= 0.9; ff["vice"] *= 0.9
        if k["strength"]:
            for ff in (fo, fg):
                synth.add_thickness_distribution(ff)
        if k["sparse"]:
            for ff in (fo, fg):
                for name in ("strairxT", "strairyT"):
                    ff[name][...] = np.where(ff["aice"] > 0.0, base[name], 0.0)
        fin = util.clone(fo) if call == 0 else None
        nt, nu, _ = orc.evp(d, p, fo)
        if k["resident"] and call:
            s.ctx.upload_inputs(fg); s.ctx.prep(); s.ctx.subcycle(k["ndte"]); s.ctx.finish(); s.ctx.download(fg)
        else:
            s.evp(dt)
        st = s.ctx.stats()
        bad = util.compare(d, fg, fo)
        bad_any = bool(bad)
        if not quiet or bad_any: print("seed", seed, "call", call, "counts", (st.icellt, st.icellu), (nt, nu), "tile", st.tile_kernel, "R2", st.strip_rows2, "k2", st.kernel2_launches, "k1", st.kernel_launches)
        if not quiet or bad_any: print("bad:", bad)
        if bad_any and call == 0:
            # which side moved?  the oracle once more from the same inputs, and the device once more in a fresh context
            f2 = util.clone(fin); orc.evp(d, p, f2)
            print("   oracle run twice, same result:", not util.compare(d, f2, fo), " second oracle run vs device:", util.compare(d, fg, f2)[:3])
            f3 = util.clone(fin)
            s3 = dyn.EvpDynamics(d, f3, ndte=k["ndte"], revised_evp=k["revised"], xmin=xmin, cosw=cosw, sinw=sinw, tilt_from_slope=k["tilt"],
                                 wind_on_ugrid=k["ugrid_wind"], device_strength=k["strength"], pin_host=k["pin"], sparse_io=k["sparse"])
            s3.init_evp(dt); s3.evp(dt); s3.close()
            print("   device run twice, same result:", not util.compare(d, f3, fg), " second device run vs oracle:", util.compare(d, f3, fo)[:3])
        for name, _, _ in bad:
            a, b = fg[name], fo[name]
            for (n, j, i) in zip(*np.nonzero(a != b)):
                bl = d.local_blocks[n]
                print(f"  {name} block {n} (i,j)=({i+1},{j+1}) global=({I[n][i]},{J[n][j]}) device={a[n,j,i]!r} oracle={b[n,j,i]!r}  tmask={f['tmask'][n,j,i]} icetmask dev/orc={fg['icetmask'][n,j,i]}/{fo['icetmask'][n,j,i]} "
                      f"aice={fo['aice'][n,j,i]:.4f} block ilo..ihi={bl.ilo}..{bl.ihi} jlo..jhi={bl.jlo}..{bl.jhi}")
                for q in ("stressm_1", "stressm_2", "stressm_3", "stressm_4", "stressp_4", "stress12_4"):
                    print("     ", q, fg[q][n, j, i], fo[q][n, j, i])
                print("      iceumask around (dev):", fg["iceumask"][n, max(j-1,0):j+1, max(i-1,0):i+1].tolist(), " uvel:", fg["uvel"][n, max(j-1,0):j+1, max(i-1,0):i+1].tolist())
    s.close()
    return bad_any


spec = sys.argv[1]
reps = 1
if "x" in spec:
    spec, r = spec.split("x"); reps = int(r)
lo, hi = (spec.split("-") + [spec])[:2] if "-" in spec else (spec, spec)
nbad = 0
for rep in range(reps):
    for seed in range(int(lo), int(hi) + 1):
        nbad += bool(run(seed, quiet=(int(hi) > int(lo) or reps > 1)))
print("draws", reps * (int(hi) - int(lo) + 1), "bad", nbad)
