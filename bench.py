#!/usr/bin/env python3
"""bench.py -- EVP subcycle throughput on N MI355X of one node.

Metric (BASELINE.json): EVP subcycle cell-updates/s + % HBM roofline, 3600x2700, ndte=120.
A step = one device-resident evp(dt): evp_prep1/2 + ndte x (fused stress+stepu kernel,
velocity halo) + evp_finish, inputs already in HBM.  N > 1 shards the SAME grid into x-slabs
(one ice_blocks block of 450x2700 per eighth of the grid), i.e. strong scaling, with the
per-subcycle halo exchange over RCCL.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

ALG_BYTES_STRESS = 360      # 24 reads + 21 writes, fp64
ALG_BYTES_STEPU = 232       # 23 reads + 6 writes
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", default="3600x2700")
    ap.add_argument("--ndte", type=int, default=120)
    ap.add_argument("--dt", type=float, default=450.0)
    ap.add_argument("--ice", default="polar", choices=["polar", "full"])
    ap.add_argument("--land", default="continents", choices=["continents", "rows"])
    ap.add_argument("--ns", default="open", choices=["open", "tripole"])
    ap.add_argument("--xblocks", type=int, default=8, help="blocks across x (one slab each at 8 GPUs)")
    ap.add_argument("--yblocks", type=int, default=10, help="blocks across y")
    ap.add_argument("--cpu-subcycles", type=int, default=120, help="subcycles of the CPU baseline sample (0 = skip)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "shm"],
                    help="shm: host-staged shared-memory relay (functional check of the multi-rank path on one GPU, not a measurement)")
    ap.add_argument("--calib", type=int, default=0, help="untimed calibration copies for rocprofv3 --pmc runs")
    ap.add_argument("--no-tripole-variant", action="store_true",
                    help="skip the informational run of the same grid with ns_boundary_type='tripole' (N = 1, default flags only)")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per k_subcycle launch from a separate rocprofv3 --pmc pass (profiles/)")
    return ap.parse_args()


def main():
    t_start = time.perf_counter()
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL across processes needs dmabuf IPC on this pool
    import torch
    import torch.distributed as dist
    from cice5_amd import blocks, constants as C, dyn, evpk, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the EVP path has no CPU fallback)")
    if os.environ.get("EVPK_FORCE_DEVICE") is not None:      # debugging only: several ranks on one GPU
        local_rank = int(os.environ["EVPK_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    uid = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # torch.distributed is only the control plane here (rendezvous, barrier, scalar reductions, the 128-byte
        # RCCL id): gloo on CPU tensors.  The data plane is libevpk's own RCCL communicator (ncclSend/ncclRecv over xGMI).
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        if a.transport == "shm":
            box = [(b"EVPKSHM:evpk_bench_%d" % os.getpid()).ljust(128, b"\0") if rank == 0 else None]
        else:
            box = [evpk.get_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]

    nx, ny = (int(v) for v in a.grid.split("x"))
    if nx % a.xblocks or a.xblocks % world or ny % a.yblocks:
        raise SystemExit("grid / xblocks / yblocks / gpus do not divide")
    bsx, bsy = nx // a.xblocks, ny // a.yblocks
    case = synth.SynthCase(nx=nx, ny=ny, ns_boundary=C.BND_NAMES[a.ns], land=a.land, ice=a.ice, dt=a.dt, ndte=a.ndte)
    d = blocks.create_distrb_cart(nx, ny, bsx, bsy, nprocs=world, rank=rank, ns_boundary_type=a.ns)
    t_gen = time.time()
    f = synth.make_block_fields(case, d)
    t_gen = time.time() - t_gen
    # global_minval(dxt/dyt) of set_evp_parameters: local minimum, then MIN over ranks
    xmin = dyn.local_min_dx(f, d)
    if world > 1:
        t = torch.tensor([xmin], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        xmin = float(t[0])

    transport = a.transport if world > 1 else "none"
    try:
        solver, err = dyn.EvpDynamics(d, f, ndte=a.ndte, xmin=xmin, device=local_rank, unique_id=uid), ""
    except evpk.EvpkError as e:
        solver, err = None, str(e)
    if world > 1:
        # every rank must end up on the same transport: if RCCL could not be brought up anywhere, all ranks fall back to
        # the host-staged shared-memory relay (correct, slow) instead of losing the run
        okv = torch.tensor([1 if solver is not None else 0], dtype=torch.int32)
        dist.all_reduce(okv, op=dist.ReduceOp.MIN)
        if int(okv[0]) == 0:
            if a.transport == "shm":
                raise SystemExit(f"rank {rank}: evpk_create failed: {err}")
            if solver is not None:
                solver.close()
            box = [(b"EVPKSHM:evpk_bench_fb_%d" % os.getpid()).ljust(128, b"\0") if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            solver = dyn.EvpDynamics(d, f, ndte=a.ndte, xmin=xmin, device=local_rank, unique_id=box[0])
            transport = "shm relay (fallback: RCCL communicator could not be created" + (": " + err[:200] if err else "") + ")"
    elif solver is None:
        raise SystemExit("evpk_create failed: " + err)
    solver.init_evp(a.dt)
    ctx = solver.ctx
    ctx.upload(f)                       # inputs resident in HBM from here on

    def step():
        ctx.prep()
        ctx.subcycle(a.ndte)
        ctx.finish()

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    if a.calib:
        ctx.calibrate(a.calib)
    t_setup = time.perf_counter() - t_start
    for _ in range(a.warmup):
        step()
    fence()
    loop_ms, k1_ms, k1_n, k2_ms, k2_n = 0.0, 0.0, 0, 0.0, 0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
        st = ctx.stats()                # HIP events on the library's compute stream: the ndte loop and every kernel launch
        loop_ms += st.loop_ms
        k1_ms += st.kernel_ms; k1_n += st.kernel_launches
        k2_ms += st.kernel2_ms; k2_n += st.kernel2_launches
    fence()
    dt_wall = time.perf_counter() - t0
    st = ctx.stats()

    vals = torch.tensor([dt_wall, loop_ms, k1_ms, k2_ms, float(st.icellt), float(st.icellu)], dtype=torch.float64)
    if world > 1:
        tmax = vals[:4].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = vals[4:].clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_wall, loop_ms, k1_ms, k2_ms = (float(v) for v in tmax)
        icellt, icellu = float(tsum[0]), float(tsum[1])
    else:
        icellt, icellu = float(st.icellt), float(st.icellu)

    n_active = 0.5 * (icellt + icellu)                  # one cell-update = one T stress + one U stepu update
    updates = n_active * a.ndte * a.steps
    value = updates / dt_wall
    alg_bytes_sub = (ALG_BYTES_STRESS * icellt + ALG_BYTES_STEPU * icellu) / world   # per GPU per subcycle
    # dominant kernel: the two-subcycle kernel when it ran (one launch = two subcycles of algorithmic work)
    if k2_n > 0:
        k2 = "evpk::k_subcycle2" if os.environ.get("EVPK_PREFETCH") == "0" else "evpk::k_subcycle2p"    # name in the rocprofv3 trace
        kname, nsub_per_launch, kern_ms, launches = k2 + " (stress+stepu fused, two subcycles per launch)", 2, k2_ms / k2_n, k2_n
    else:
        kname, nsub_per_launch, kern_ms, launches = "evpk::k_subcycle (stress+stepu fused)", 1, k1_ms / max(k1_n, 1), k1_n
    alg_bytes_launch = nsub_per_launch * alg_bytes_sub
    achieved = alg_bytes_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0

    traffic = a.traffic_bytes
    workload = (f"{nx}x{ny} ndte={a.ndte} ice={a.ice} land={a.land} ns={a.ns} "
                f"({a.xblocks * a.yblocks} ice_blocks blocks of {bsx}x{bsy}, x-slabs over {world} GPU)")
    tfile = os.path.join(ROOT, "profiles", f"traffic_n{world}.json")
    if traffic is None and os.path.exists(tfile):       # measured in a separate rocprofv3 --pmc pass of this same command
        t = json.load(open(tfile))
        if t.get("workload") == workload:
            traffic = t["hbm_bytes_per_launch"]
    out = {
        "metric": "EVP subcycle cell-updates/sec", "value": value, "unit": "cell-updates/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt_wall / a.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload,
                   "active_T_cells": int(icellt), "active_U_cells": int(icellu), "grid_cells": nx * ny,
                   "grid_cell_updates_per_s": nx * ny * a.ndte * a.steps / dt_wall,
                   "strips_per_launch_rank0": int(st.nstrips2 or st.nstrips),
                   "strip_rows_rank0": int(st.strip_rows2 or st.strip_rows), "transport": transport,
                   "ghost_zone_cols": int(st.zone_cols), "zone_exchanges_per_evp": int(st.zone_exchanges),
                   "zone_bytes_sent_rank0": int(st.zone_bytes), "overlap_split_rank0": int(st.overlap_split), "step": "prep + ndte x (stress+stepu, halo) + finish"},
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic,
                     # real HBM rate of that kernel: PMC bytes per launch (profiles/) over the launch time measured here
                     "traffic_GBps": (traffic / (kern_ms * 1e-3) / 1e9) if (traffic and kern_ms > 0) else None,
                     "traffic_frac_of_peak": (traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and kern_ms > 0) else None,
                     "alg_bytes_per_launch": alg_bytes_launch,
                     "avg_launch_ms": kern_ms, "launches_timed": int(launches), "subcycles_per_launch": nsub_per_launch,
                     "loop_ms_per_step": loop_ms / a.steps,
                     "other_kernel": {"name": "k_subcycle", "launches": int(k1_n), "avg_launch_ms": k1_ms / max(k1_n, 1)}},
    }

    t_cpu = time.perf_counter()
    if rank == 0 and world == 1 and a.cpu_subcycles > 0:
        solver.init_evp(a.dt)          # the host arrays are still the uploaded inputs; state back at rest
        out["cpu_baseline"] = cpu_baseline(d, f, a, xmin)
    t_cpu = time.perf_counter() - t_cpu
    t_extra = time.perf_counter()
    # (profiling runs pass --cpu-subcycles 0 and skip both extras, so that their traces hold the timed workload only)
    if world == 1 and a.ns == "open" and a.cpu_subcycles > 0 and not a.no_tripole_variant:
        # informational: the same grid closed by the tripole fold at the north (BASELINE config 5's boundary; two extra
        # band launches + two fold updates per launch pair).  Not part of `value`.
        solver.close()
        solver = None
        try:
            out["config"]["tripole_variant"] = tripole_variant(a, nx, ny, bsx, bsy, local_rank)
        except Exception as e:           # never lose the bench line over the extra
            out["config"]["tripole_variant"] = {"error": str(e)[:200]}
        # informational: one whole evp(dt) through evpk_run INCLUDING the PCIe transfers of all arrays (never `value`)
        try:
            out["config"]["evp_incl_pcie_ms"] = evp_incl_pcie(d, f, a, xmin, local_rank)
        except Exception as e:
            out["config"]["evp_incl_pcie_ms"] = {"error": str(e)[:200]}
    # where the wall time of this process went (the timed region is `steps` x ms_per_step; the rest is imports, synthetic
    # inputs, context creation, warm-up, the CPU baseline and the informational extras)
    out["wall_s"] = {"setup_imports_inputs_create_upload": t_setup, "timed_steps": dt_wall, "cpu_baseline": t_cpu,
                     "extras_tripole_and_pcie": time.perf_counter() - t_extra, "total": time.perf_counter() - t_start}
    # RCCL prints a version banner through C stdio, which is block-buffered on a pipe and would otherwise surface AFTER
    # the JSON line at exit: flush it first so that the JSON line is the last line of stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        print(json.dumps(out), flush=True)
    if solver is not None:
        solver.close()
    if world > 1:
        dist.destroy_process_group()


def evp_incl_pcie(d, f, a, xmin, device):
    from cice5_amd import dyn
    res = {}
    for key, pin in (("pageable_host_arrays", False), ("page_locked_host_arrays", True)):
        s = dyn.EvpDynamics(d, f, ndte=a.ndte, xmin=xmin, device=device, pin_host=pin)
        s.init_evp(a.dt)
        s.evp(a.dt)
        t = time.perf_counter()
        s.evp(a.dt)
        res[key] = 1e3 * (time.perf_counter() - t)
        s.close()
    return res


def tripole_variant(a, nx, ny, bsx, bsy, device):
    from cice5_amd import blocks, constants as C, dyn, synth
    case = synth.SynthCase(nx=nx, ny=ny, ns_boundary=C.BND_NAMES["tripole"], land=a.land, ice=a.ice, dt=a.dt, ndte=a.ndte)
    d = blocks.create_distrb_cart(nx, ny, bsx, bsy, ns_boundary_type="tripole")
    f = synth.make_block_fields(case, d)
    s = dyn.EvpDynamics(d, f, ndte=a.ndte, xmin=dyn.local_min_dx(f, d), device=device)
    s.init_evp(a.dt)
    s.ctx.upload(f)
    ts = []
    for k in range(2 + 3):
        t = time.perf_counter()
        s.ctx.prep(); s.ctx.subcycle(a.ndte); s.ctx.finish(); s.ctx.sync()
        if k >= 2:
            ts.append(time.perf_counter() - t)
    st = s.ctx.stats()
    s.close()
    sec = sum(ts) / len(ts)
    return {"ns": "tripole", "ms_per_step": 1e3 * sec, "value": 0.5 * (st.icellt + st.icellu) * a.ndte / sec,
            "active_T_cells": int(st.icellt), "steps": len(ts)}


def cpu_baseline(d, f, a, xmin):
    """The oracle (C restatement of the reference loop, OpenMP over blocks like the reference's
    THRD build) on the host cores: the first few subcycles of the same workload, same blocks."""
    from oracle import orc
    p = orc.make_params(a.dt, a.ndte, xmin)
    threads = orc._limit_threads()
    nt, nu, secs = orc.evp(d, p, f, nsub=a.cpu_subcycles)
    return {"value": 0.5 * (nt + nu) * a.cpu_subcycles / secs, "unit": "cell-updates/s", "cores": min(threads, d.nblocks),
            "kind": "port",
            "sample": f"first {a.cpu_subcycles} of {a.ndte} subcycles of the same {d.nx_global}x{d.ny_global} state "
                      f"({d.nblocks} blocks of {d.block_size_x}x{d.block_size_y}, OpenMP over blocks: "
                      f"stress + stepu + halo copies), {secs:.2f} s of CPU wall time"}


if __name__ == "__main__":
    main()
