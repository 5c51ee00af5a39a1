#!/usr/bin/env python3
"""bench.py -- EVP subcycle throughput on N MI355X of one node.

Metric (BASELINE.json): EVP subcycle cell-updates/s + % HBM roofline, 3600x2700, ndte=120.
Workload: BASELINE config 5's grid and boundary -- 3600x2700, ns_boundary_type = 'tripole' (the only 3600x2700 entry of
BASELINE.json's configs is the tripolar one) -- at the metric's ndte = 120.  A step = one device-resident evp(dt):
evp_prep1/2 + ndte x (fused stress+stepu kernel, velocity halo / tripole fold) + stress folds + evp_finish, inputs
already in HBM.  N > 1 shards the SAME grid into x-slabs (one ice_blocks block column of 450x2700 per eighth of the
grid), i.e. strong scaling, with the halo exchange between the ranks' libevpk contexts.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

ALG_BYTES_STRESS = 360      # 24 reads + 21 writes, fp64 -- the reference's UNFUSED per-subcycle traffic
ALG_BYTES_STEPU = 232       # 23 reads + 6 writes
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s is what a streaming copy reaches)


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
    ap.add_argument("--ns", default="tripole", choices=["open", "tripole"])
    ap.add_argument("--xblocks", type=int, default=8,
                    help="block columns (default 8: equal x-slabs on 1, 2, 4, 8 GPUs); 0 = the multiple of --gpus that divides the grid with "
                         "blocks closest to 450 columns.  create_distrb_cart deals ceil(xblocks / gpus) columns to a rank: counts that "
                         "do not divide leave the last ranks short or idle (the reference's 40x30 blocks = 90 columns on 8 ranks: "
                         "12, ..., 12, 6 -- legal, 2x imbalanced)")
    ap.add_argument("--yblocks", type=int, default=10, help="blocks across y")
    ap.add_argument("--cpu-subcycles", type=int, default=120, help="subcycles of the CPU baseline sample (0 = skip)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "shm", "ipc"],
                    help="rccl: ncclSend/ncclRecv over xGMI (the measurement); ipc: peer-mapped buffers; shm: host-staged "
                         "shared-memory relay (functional check of the multi-rank path on one GPU, never a measurement)")
    ap.add_argument("--calib", type=int, default=0, help="untimed calibration copies for rocprofv3 --pmc runs")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the informational extras (same grid with ns_boundary_type='open', PCIe-inclusive evp; N = 1 only)")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch of the dominant kernel from a separate rocprofv3 --pmc pass (profiles/)")
    return ap.parse_args()


def source_sha():
    """identifies the kernel build a PMC pass was made with: the sources libevpk.so is compiled from"""
    h = hashlib.sha256()
    for n in ("evpk_kernels.hip", "evpk_api.hip", "evpk_internal.h", "evpk_remap.hip", "evpk_eap.hip", "evpk_fmath.h"):
        h.update(open(os.path.join(ROOT, "cice5_amd", "csrc", n), "rb").read())
    return h.hexdigest()[:16]


def fused_bytes(icellt, icellu, st, revp, fused):
    """Bytes one launch of the fused kernel MUST move (fp64, each distinct array element once, perfect stencil reuse):
    per active T cell: 12 sigma read + 12 written, the (tinyarea, strength) pair, the metrics the variant reads -- the
    (HTN, HTE) pair with compact metrics, else the four pairs cxp/cyp, cxm/cym, dxt/dyt, dxhy/dyhx -- the (u, v) pair and
    the mask byte; per active U cell: the four stepu input pairs (vrelc/uarear, uocn/vocn, forcex/forcey, umassdti/fm),
    (u, v) written, + the uvel_init pair under revised EVP.  The two-subcycle kernel moves this ONCE per two subcycles, the
    three-subcycle pipeline kernel ONCE per three."""
    metrics = 16 if (st.compact_metrics and fused) else 64
    per_t = 12 * 8 * 2 + 16 + metrics + 16 + 1
    per_u = 4 * 16 + 16 + (16 if revp else 0)
    return per_t * icellt + per_u * icellu


def self_launch(a):
    """`python bench.py --gpus N` from a bare command line: start the N ranks as CHILD processes through torch.distributed.run
    before this process has made any GPU call (it never makes one), relay rank 0's JSON line and exit with the children's status."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout:                      # everything the ranks print is passed on; the JSON line is repeated last
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        else:
            sys.stdout.write(ln)
    rc = p.wait()
    sys.stdout.flush()
    if line is None:
        line = json.dumps({"metric": "EVP subcycle cell-updates/sec", "value": None, "unit": "cell-updates/s", "n_gpus": a.gpus,
                           "failed": "launch", "error": f"torch.distributed.run exited with {rc} and no rank printed a result"})
    print(line, flush=True)
    sys.exit(rc if rc else (0 if '"value": null' not in line else 3))


def main():
    t_start = time.perf_counter()
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        self_launch(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL / IPC handles across processes need dmabuf IPC on this pool
    import torch
    import torch.distributed as dist
    from cice5_amd import blocks, constants as C, dyn, evpk, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the EVP path has no CPU fallback)")
    if os.environ.get("EVPK_FORCE_DEVICE") is not None:      # debugging only: several ranks on one GPU
        local_rank = int(os.environ["EVPK_FORCE_DEVICE"])
    if local_rank < torch.cuda.device_count():
        torch.cuda.set_device(local_rank)      # (else: evpk.device_check below reports the missing device on every rank's behalf)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # torch.distributed is only the control plane here (rendezvous, barrier, scalar reductions, the 128-byte
        # unique id): gloo on CPU tensors.  The data plane is libevpk's own transport (ncclSend/ncclRecv over xGMI).
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    def all_ok(ok: bool) -> bool:
        if world == 1:
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t[0]) == 1

    def fail(stage, err):
        """A failed start is a failed measurement: no fallback to a slower transport under the same metric name."""
        if world > 1:
            errs = [None] * world
            dist.all_gather_object(errs, err)
            err = "; ".join(f"rank {r}: {e}" for r, e in enumerate(errs) if e)
        if rank == 0:
            print(json.dumps({"metric": "EVP subcycle cell-updates/sec", "value": None, "unit": "cell-updates/s", "n_gpus": world,
                              "failed": stage, "error": err[:600], "transport_requested": a.transport}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        sys.exit(3)

    nx, ny = (int(v) for v in a.grid.split("x"))
    # (block columns need not divide over the ranks: create_distrb_cart gives every rank ceil(xblocks / world) of them, the last
    #  ones the rest or none at all, ice_distribution.F90:603-640; a rank without a column joins the start and no exchange)
    if a.xblocks == 0:
        cand = [k * world for k in range(1, nx // world + 1) if nx % (k * world) == 0]
        if not cand:
            raise SystemExit(f"no block-column count divides {nx} columns over {world} ranks evenly")
        a.xblocks = min(cand, key=lambda q: abs(nx // q - 450))
    if nx % a.xblocks or ny % a.yblocks:
        raise SystemExit("grid / xblocks / yblocks do not divide")
    bsx, bsy = nx // a.xblocks, ny // a.yblocks
    case = synth.SynthCase(nx=nx, ny=ny, ns_boundary=C.BND_NAMES[a.ns], land=a.land, ice=a.ice, dt=a.dt, ndte=a.ndte)
    d = blocks.create_distrb_cart(nx, ny, bsx, bsy, nprocs=world, rank=rank, ns_boundary_type=a.ns)
    f = synth.make_block_fields(case, d)
    # global_minval(dxt/dyt) of set_evp_parameters: local minimum, then MIN over ranks
    xmin = dyn.local_min_dx(f, d)
    if world > 1:
        t = torch.tensor([xmin], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        xmin = float(t[0])

    # two-phase start: every rank checks its device and creates its context WITHOUT touching the others; only when all
    # of them succeeded do the ranks enter the collective part (communicator / peer mapping)
    err = evpk.device_check(local_rank)
    if not all_ok(err is None):
        fail("device check", err)
    solver, err = None, None
    try:
        solver = dyn.EvpDynamics(d, f, ndte=a.ndte, xmin=xmin, device=local_rank, defer_connect=(world > 1))
    except evpk.EvpkError as e:
        err = str(e)
    if not all_ok(solver is not None):
        fail("evpk_create", err)
    if world > 1:
        tag = b"evpk_bench_%d" % os.getpid()
        box = [{"rccl": lambda: evpk.get_unique_id(), "shm": lambda: (b"EVPKSHM:" + tag).ljust(128, b"\0"),
                "ipc": lambda: (b"EVPKIPC:" + tag).ljust(128, b"\0")}[a.transport]() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        try:
            solver.connect(box[0])
        except evpk.EvpkError as e:
            err = str(e)
        if not all_ok(err is None):
            fail("evpk_connect (" + a.transport + ")", err)
    solver.init_evp(a.dt)
    ctx = solver.ctx
    ctx.upload(f)                       # inputs resident in HBM from here on

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    if a.calib:
        ctx.calibrate(a.calib)
    t_setup = time.perf_counter() - t_start
    r = timed_steps(ctx, a.ndte, a.steps, a.warmup, fence)
    st = r["stats"]

    vals = torch.tensor([r["wall_s"], r["loop_ms"], float(st.icellt), float(st.icellu)], dtype=torch.float64)
    devs = [(int(st.device), int(st.device_pci))]
    rccl_ranks = int(st.rccl_ranks)
    if world > 1:
        tmax = vals[:2].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = vals[2:].clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_wall, loop_ms = (float(v) for v in tmax)
        icellt, icellu = float(tsum[0]), float(tsum[1])
        got = [None] * world
        dist.all_gather_object(got, (devs[0], rccl_ranks))
        devs = [g[0] for g in got]
        rccl_ranks = min(g[1] for g in got)
    else:
        dt_wall, loop_ms = r["wall_s"], r["loop_ms"]
        icellt, icellu = float(st.icellt), float(st.icellu)

    n_active = 0.5 * (icellt + icellu)                  # one cell-update = one T stress + one U stepu update
    value = n_active * a.ndte * a.steps / dt_wall
    workload = (f"{nx}x{ny} ndte={a.ndte} ice={a.ice} land={a.land} ns={a.ns} "
                f"({a.xblocks * a.yblocks} ice_blocks blocks of {bsx}x{bsy}, x-slabs over {world} GPU)")
    n_ring = -(-a.xblocks // (-(-a.xblocks // world)))          # ranks that own block columns
    roof = roofline(r, icellt / n_ring, icellu / n_ring, revp=False)
    sha = source_sha()
    traffic, tnote = a.traffic_bytes, "given on the command line" if a.traffic_bytes else None
    rocprof_ms = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if traffic is None and os.path.exists(tfile):       # measured in separate rocprofv3 --pmc passes of this same command
        ent = [e for e in json.load(open(tfile)).get("entries", []) if e.get("workload") == workload]
        cur = [e for e in ent if e.get("source_sha") == sha]
        if cur:
            traffic, tnote = cur[-1]["hbm_bytes_per_launch"], cur[-1].get("profile")
            rocprof_ms = cur[-1].get("kernel_avg_ms")      # rocprofv3 --kernel-trace --stats average of the same kernel, same build
        elif ent:
            tnote = f"profiles/traffic.json holds this workload for another kernel build ({ent[-1].get('source_sha')}): not reported"
    # the counter bytes belong to the PROFILED run: their rate is taken with that run's own average launch time (the kernel-trace run reads
    # ~7 % slower than the plain one beside it); only without it, with this run's HIP-event time
    kern_s = (rocprof_ms if rocprof_ms else roof["avg_launch_ms"]) * 1e-3
    roof.update({"traffic": traffic, "traffic_source": tnote, "source_sha": sha, "rocprof_avg_launch_ms": rocprof_ms,
                 "traffic_clock": "rocprof_avg_launch_ms" if rocprof_ms else "avg_launch_ms (HIP events of this run)",
                 "traffic_GBps": (traffic / kern_s / 1e9) if (traffic and kern_s > 0) else None,
                 "traffic_frac_of_peak": (traffic / kern_s / 1e9 / HBM_PEAK_GBS) if (traffic and kern_s > 0) else None,
                 "traffic_over_alg": (traffic / roof["alg_bytes_per_launch"]) if traffic else None,
                 "loop_ms_per_step": loop_ms / a.steps})
    out = {
        "metric": "EVP subcycle cell-updates/sec", "value": value, "unit": "cell-updates/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt_wall / a.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload,
                   "active_T_cells": int(icellt), "active_U_cells": int(icellu), "grid_cells": nx * ny,
                   "grid_cell_updates_per_s": nx * ny * a.ndte * a.steps / dt_wall,
                   "strips_per_launch_rank0": int(st.nstrips2 or st.nstrips),
                   "strip_rows_rank0": int(st.strip_rows2 or st.strip_rows),
                   "transport": evpk.XP_NAMES.get(int(st.transport), "?") if world > 1 else "none",
                   # what carried the exchanges, as the library itself sees it: ncclCommCount of every rank's communicator (0 = no
                   # RCCL communicator: one rank, or the ipc / shm transports) and the distinct physical GPUs under the ranks
                   "rccl_ranks": rccl_ranks, "devices": len(set(p for _, p in devs)), "device_ordinals": [o for o, _ in devs],
                   "ranks_with_block_columns": n_ring, "block_columns_per_rank": [max(0, min(-(-a.xblocks // world), a.xblocks - q * -(-a.xblocks // world))) for q in range(world)],
                   "ghost_zone_cols": int(st.zone_cols), "zone_exchanges_per_evp": int(st.zone_exchanges), "band_row_exchanges_per_evp": int(st.band_row_exchanges),
                   "zone_bytes_sent_rank0": int(st.zone_bytes), "overlap_split_rank0": int(st.overlap_split),
                   "step": "prep + ndte x (stress+stepu, halo / fold) + stress folds + finish"},
        "roofline": roof,
    }

    exit_code = 0
    t_cpu = time.perf_counter()
    if rank == 0 and world == 1 and a.cpu_subcycles > 0:
        # verify: the benched context itself against the oracle on the box the number comes from -- state back at rest, the same
        # `cpu_subcycles` subcycles (default: all ndte) + stress folds + evp_finish on the device (untimed) and, as the cpu_baseline
        # leg below, in the oracle, over the same blocks; the prognostic state compared bit for bit on every cell the reference
        # leaves defined.  The oracle is the checker here, never the thing measured.
        nv = min(a.cpu_subcycles, a.ndte)
        names = ["uvel", "vvel"] + list(synth.STRESS_FIELDS) + ["strocnx", "strocny", "strocnxT", "strocnyT"]      # (the last four: evp_finish, u2tgrid_vector)
        solver.init_evp(a.dt)          # the host arrays are still the uploaded inputs; state back at rest
        ctx.upload(f)
        ctx.prep(); ctx.subcycle(nv); ctx.finish()
        got = {n: np.empty_like(f[n]) for n in names}
        ctx.download(got)
        vst = ctx.stats()
        out["cpu_baseline"] = cpu_baseline(d, f, a, xmin)       # (runs the oracle in place on f: nv subcycles from the same state)
        out["verify"] = verify(d, got, f, names, nv, vst, out["cpu_baseline"])
        del got
    elif world > 1 and a.cpu_subcycles > 0:
        # verify, N > 1: every rank puts its slab back at rest and runs one untimed evp; rank 0 runs the WHOLE grid as ONE rank on its
        # own GPU (the path the N = 1 line verifies against the oracle) and compares every rank's blocks with it, bit for bit
        v = verify_multirank(a, case, d, f, solver, xmin, local_rank, rank, world, dist)
        if rank == 0:
            out["verify"] = v
    # what the line claims about its ranks is part of the exit code: one RCCL rank and one physical GPU per process (EVPK_FORCE_DEVICE
    # puts all ranks on one GPU on purpose: tests), and a result that equals the one-rank run
    if rank == 0 and world > 1:
        want_dev = 1 if os.environ.get("EVPK_FORCE_DEVICE") is not None else world
        checks = {"devices": out["config"]["devices"] == want_dev,
                  "rccl_ranks": (out["config"]["rccl_ranks"] == world) if a.transport == "rccl" else True,
                  "bit_identical": out.get("verify", {}).get("bit_identical") is not False}
        out["checks"] = checks
        if not all(checks.values()):
            out["failed"] = "checks"
            exit_code = 4
    t_cpu = time.perf_counter() - t_cpu
    t_extra = time.perf_counter()
    # (profiling runs pass --cpu-subcycles 0 and skip the extras, so that their traces hold the timed workload only)
    if world == 1 and a.cpu_subcycles > 0 and not a.no_variants:
        # informational, never part of `value`: the same grid closed at the north (ns_boundary_type = 'open': no fold, so
        # no band launches), timed the same way -- HIP events on the library's stream, same roofline accounting
        solver.close()
        solver = None
        try:
            out["config"]["open_variant" if a.ns == "tripole" else "tripole_variant"] = other_boundary(a, nx, ny, bsx, bsy, local_rank)
        except Exception as e:           # never lose the bench line over an extra
            out["config"]["open_variant"] = {"error": str(e)[:200]}
        # informational: one whole evp(dt) through evpk_run INCLUDING the PCIe transfers of all arrays (never `value`)
        try:
            out["config"]["evp_incl_pcie_ms"] = evp_incl_pcie(d, f, a, xmin, local_rank)
        except Exception as e:
            out["config"]["evp_incl_pcie_ms"] = {"error": str(e)[:200]}
        # informational: the rows SURVEY.md S8 marks "next" on the same grid and state -- eap(dt) (f-4) in place of evp(dt), then
        # transport_remap's horizontal_remap (f-3) advecting an ice state resident in HBM with the velocities left on the device
        try:
            out["config"]["next_rows"] = next_rows(a, case, d, f, xmin, local_rank)
        except Exception as e:
            out["config"]["next_rows"] = {"error": str(e)[:300]}
    # where the wall time of this process went (the timed region is `steps` x ms_per_step; the rest is imports, synthetic
    # inputs, context creation, warm-up, the CPU baseline and the informational extras)
    out["wall_s"] = {"setup_imports_inputs_create_upload": t_setup, "timed_steps": dt_wall, "cpu_baseline": t_cpu,
                     "extras": time.perf_counter() - t_extra, "total": time.perf_counter() - t_start}
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
        code = torch.tensor([exit_code], dtype=torch.int32)
        dist.broadcast(code, src=0)
        dist.destroy_process_group()
        exit_code = int(code[0])
    if exit_code:
        sys.exit(exit_code)


def verify(d, got, ref, names, nsub, st, cpu):
    """bit comparison of the device result with the oracle's: (u, v) on all cells of every block (ghost cells halo-updated),
    the stresses on physical + N/E ghost T cells (ice_dyn_shared.F90:528-537), the ocean stresses of evp_finish on physical cells,
    as tests/util.compare does"""
    allc = np.zeros((d.nblocks, d.ny_block, d.nx_block), dtype=bool)
    ne, phys = np.zeros_like(allc), np.zeros_like(allc)
    for n, b in enumerate(d.local_blocks):
        allc[n, :b.jhi + 1, :b.ihi + 1] = True
        ne[n, b.jlo - 1:b.jhi + 1, b.ilo - 1:b.ihi + 1] = True
        phys[n, b.jlo - 1:b.jhi, b.ilo - 1:b.ihi] = True
    bad, cells = {}, 0
    for n in names:
        m = allc if n in ("uvel", "vvel") else (phys if n.startswith("strocn") else ne)
        a, b_ = got[n][m], ref[n][m]
        neq = ~((a == b_) | (np.isnan(a) & np.isnan(b_)))
        cells += int(m.sum())
        if neq.any():
            bad[n] = {"cells": int(neq.sum()), "max_abs_diff": float(np.nanmax(np.abs(a[neq] - b_[neq])))}
    return {"bit_identical": not bad, "mismatch": bad, "fields": names, "values_compared": cells, "subcycles": int(nsub),
            "max_abs_u": float(np.abs(ref["uvel"]).max()),
            "what": f"state at rest -> {nsub} subcycles + stress folds + evp_finish on the benched context ({d.nblocks} blocks, the "
                    f"benched decomposition, {int(st.kernel3_launches)} three-subcycle + {int(st.kernel2_launches)} two-subcycle + "
                    f"{int(st.kernel_launches)} one-subcycle launches) vs the oracle (the cpu_baseline run) on the same inputs",
            "counts_match": [int(st.icellt), int(st.icellu)] == [int(cpu["icellt"]), int(cpu["icellu"])]}


def verify_multirank(a, case, d, f, solver, xmin, device, rank, world, dist):
    """N > 1: the benched ranks against a ONE-rank device run of the whole grid.  Every rank: state back at rest, one untimed evp
    (all ndte subcycles, stress folds, evp_finish) on its slab, its blocks' (u, v, sigma x 12, ocean stresses) to rank 0 over the
    control plane (gloo).  Rank 0: the same evp for the whole grid in one context on its own GPU -- the configuration whose result
    the N = 1 bench line compares with the oracle -- and a bit comparison per block on the cells the reference leaves defined."""
    from cice5_amd import blocks, dyn, synth
    names = ["uvel", "vvel"] + list(synth.STRESS_FIELDS) + ["strocnx", "strocny", "strocnxT", "strocnyT"]
    ctx = solver.ctx
    solver.init_evp(a.dt)
    ctx.upload(f)
    ctx.prep(); ctx.subcycle(a.ndte); ctx.finish()
    got = {n: np.empty_like(f[n]) for n in names}
    ctx.download(got)
    mine = ([b.block_id for b in d.local_blocks], got)
    parts = [None] * world if rank == 0 else None
    dist.gather_object(mine, parts, dst=0)
    if rank != 0:
        return None
    d1 = blocks.create_distrb_cart(d.nx_global, d.ny_global, d.block_size_x, d.block_size_y, ns_boundary_type=d.ns_boundary)
    f1 = synth.make_block_fields(case, d1)
    s1 = dyn.EvpDynamics(d1, f1, ndte=a.ndte, xmin=xmin, device=device)
    s1.init_evp(a.dt)
    s1.ctx.upload(f1)
    s1.ctx.prep(); s1.ctx.subcycle(a.ndte); s1.ctx.finish()
    ref = {n: np.empty_like(f1[n]) for n in names}
    s1.ctx.download(ref)
    st1 = s1.ctx.stats()
    s1.close()
    where = {b.block_id: k for k, b in enumerate(d1.local_blocks)}
    bad, cells, nblk = {}, 0, 0
    for r, (ids, arrs) in enumerate(parts):
        for k, gid in enumerate(ids):
            b = d1.local_blocks[where[gid]]
            nblk += 1
            for n in names:
                if n in ("uvel", "vvel"):
                    sl = (slice(0, b.jhi + 1), slice(0, b.ihi + 1))
                elif n.startswith("strocn"):
                    sl = (slice(b.jlo - 1, b.jhi), slice(b.ilo - 1, b.ihi))
                else:
                    sl = (slice(b.jlo - 1, b.jhi + 1), slice(b.ilo - 1, b.ihi + 1))
                x, y = arrs[n][k][sl], ref[n][where[gid]][sl]
                neq = ~((x == y) | (np.isnan(x) & np.isnan(y)))
                cells += x.size
                if neq.any():
                    e = bad.setdefault(n, {"cells": 0, "max_abs_diff": 0.0, "ranks": []})
                    e["cells"] += int(neq.sum())
                    e["max_abs_diff"] = max(e["max_abs_diff"], float(np.nanmax(np.abs(x[neq] - y[neq]))))
                    if r not in e["ranks"]:
                        e["ranks"].append(r)
    return {"bit_identical": not bad and nblk == len(d1.local_blocks), "against": "1-rank device run", "mismatch": bad, "fields": names,
            "values_compared": cells, "blocks_compared": nblk, "blocks_total": len(d1.local_blocks), "subcycles": int(a.ndte),
            "max_abs_u": float(np.abs(ref["uvel"]).max()), "one_rank_counts": [int(st1.icellt), int(st1.icellu)],
            "what": f"every rank: state at rest -> {a.ndte} subcycles + stress folds + evp_finish on its slab (untimed, after the timed steps); "
                    f"rank 0: the same on the whole grid in ONE context on its own GPU (the configuration the N = 1 line verifies against the "
                    f"oracle); (u, v) on all cells of every block, sigma on physical + N/E ghost T cells, the ocean stresses on physical cells"}


def timed_steps(ctx, ndte, steps, warmup, fence):
    """`warmup` untimed and `steps` timed device-resident evp(dt).  Wall clock around the timed steps (fence = sync +
    barrier on both sides); per step the library's own HIP events on ITS streams: the whole ndte loop and the sampled
    kernel launches (torch.cuda.Event would only see torch's current stream)."""
    def step():
        ctx.prep()
        ctx.subcycle(ndte)
        ctx.finish()

    for _ in range(warmup):
        step()
    fence()
    loop_ms = 0.0
    bound = [0.0, 0]                              # timer_bound's share of the loops (ice_dyn_evp.F90:392-400): ms, updates
    k = {q: [0.0, 0, 0] for q in (1, 2, 3)}      # subcycles per launch -> [ms, launches, launches inside timed spans]
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        st = ctx.stats()
        loop_ms += st.loop_ms
        bound[0] += st.bound_ms; bound[1] += st.bound_updates
        for q, (ms, n, t) in ((1, (st.kernel_ms, st.kernel_launches, st.kernel_timed)), (2, (st.kernel2_ms, st.kernel2_launches, st.kernel2_timed)),
                              (3, (st.kernel3_ms, st.kernel3_launches, st.kernel3_timed))):
            k[q][0] += ms; k[q][1] += n; k[q][2] += t
    fence()
    wall = time.perf_counter() - t0
    return {"wall_s": wall, "loop_ms": loop_ms, "k": k, "stats": ctx.stats(), "steps": steps, "bound": bound}


def roofline(r, icellt, icellu, revp):
    """The dominant kernel against the HBM roofline.  `achieved` = the bytes one launch of the fused kernel must move
    (fused_bytes) / its average duration, from HIP events around the sampled launches; `frac` = achieved / 8 TB/s.  The
    SURVEY S8d figure (592 B per cell-update: the reference's unfused traffic, which this kernel never moves) is kept
    as `effective_vs_reference_accounting`."""
    st = r["stats"]
    k = r["k"]
    nsub = max((q for q in (1, 2, 3) if k[q][1] > 0), key=lambda q: k[q][0], default=1)     # the kind the loop spends most time in
    kms, launches, timed = k[nsub]
    kern_ms = kms / max(launches, 1)
    if nsub == 3:
        kname = "evpk::k_subcycle3w (stress+stepu fused, three subcycles per launch, one wave per subcycle stage)"
    elif nsub == 2:
        base = {1: "evpk::k_subcycle2t", 2: "evpk::k_subcycle2r"}.get(int(st.tile_kernel)) or \
               ("evpk::k_subcycle2" if os.environ.get("EVPK_PREFETCH") == "0" else "evpk::k_subcycle2p")
        kname = base + " (stress+stepu fused, two subcycles per launch)"
    else:
        kname = "evpk::k_subcycle (stress+stepu fused)"
    alg = fused_bytes(icellt, icellu, st, revp, nsub >= 2)
    ref = nsub * (ALG_BYTES_STRESS * icellt + ALG_BYTES_STEPU * icellu)
    ach = alg / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    others = [{"subcycles_per_launch": q, "launches": int(k[q][1]), "launches_timed": int(k[q][2]), "avg_launch_ms": k[q][0] / max(k[q][1], 1)}
              for q in (1, 2, 3) if q != nsub and k[q][1] > 0]
    ref_gbps = ref / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    return {"bound": "hbm", "kernel": kname, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            # two accountings of the same launch time: `frac` prices what the FUSED kernel must move (fused_bytes: 160.6 B per cell-update
            # with two subcycles per launch) -- a physical rate, <= 1; `frac_survey_8d` prices SURVEY S8d's 592 B per cell-update, the
            # reference's unfused once-per-subcycle traffic that this kernel never moves -- an EFFECTIVE rate that exceeds 1 because of the
            # temporal fusion, not because work is skipped (the same run verifies every subcycle against the oracle)
            "frac_accounting": "fused-compulsory bytes (fused_bytes) / avg_launch_ms / peak",
            "frac_survey_8d": ref_gbps / HBM_PEAK_GBS, "achieved_survey_8d": ref_gbps,
            "alg_bytes_per_launch": alg, "alg_bytes_per_cell_update": alg / max(0.5 * (icellt + icellu) * nsub, 1.0),
            "avg_launch_ms": kern_ms, "launches": int(launches), "launches_timed": int(timed), "subcycles_per_launch": nsub,
            "timing": "HIP events on the library's stream around runs of six consecutive launches (launches 3..8 of every 20): span "
                      "time / launches in the span, so launches x avg_launch_ms <= the loop time by construction",
            "kernel_ms_per_step_all_kinds": sum(k[q][0] for q in (1, 2, 3)) / max(r["steps"], 1),
            "bound_ms_per_step": r["bound"][0] / max(r["steps"], 1), "bound_updates_per_step": r["bound"][1] / max(r["steps"], 1),
            "compact_metrics": int(st.compact_metrics),
            "effective_vs_reference_accounting": {"bytes_per_cell_update": ALG_BYTES_STRESS + ALG_BYTES_STEPU,
                                                  "GBps": ref / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0,
                                                  "frac_of_peak": ref / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if kern_ms > 0 else 0.0},
            "other_kernels": others}


def evp_incl_pcie(d, f, a, xmin, device):
    """whole evp(dt) calls INCLUDING the PCIe transfers, as a host model sees them (never `value`):
    pageable / page_locked: evpk_run, every array up and down every call (27 + 36 arrays of 78 MB);
    resident_every_step_outputs: the state stays on the device, inputs up, the eight arrays a host model reads every step
    down (uvel, vvel, rdg_conv, rdg_shear, divu, shear, strocnxT/yT), page-locked; + sparse: only the tiles with ice move
    (evpk_params.sparse_io = 1: aice, vice, vsno whole; sparse2 = 2: aice alone whole)."""
    from cice5_amd import dyn
    res = {}
    for key, kw in (("pageable_host_arrays", dict(pin_host=False)), ("page_locked_host_arrays", dict(pin_host=True)),
                    ("resident_every_step_outputs", dict(pin_host=True, resident=True, outputs=dyn.EVERY_STEP_OUTPUTS)),
                    ("resident_every_step_outputs_sparse", dict(pin_host=True, resident=True, outputs=dyn.EVERY_STEP_OUTPUTS, sparse_io=1)),
                    ("resident_every_step_outputs_sparse2", dict(pin_host=True, resident=True, outputs=dyn.EVERY_STEP_OUTPUTS, sparse_io=2))):
        s = dyn.EvpDynamics(d, f, ndte=a.ndte, xmin=xmin, device=device, **kw)
        s.init_evp(a.dt)
        s.evp(a.dt)
        s.evp(a.dt)
        t = time.perf_counter()
        s.evp(a.dt)
        res[key] = 1e3 * (time.perf_counter() - t)
        s.close()
    return res


def other_boundary(a, nx, ny, bsx, bsy, device):
    from cice5_amd import blocks, constants as C, dyn, synth
    import torch
    ns = "open" if a.ns == "tripole" else "tripole"
    case = synth.SynthCase(nx=nx, ny=ny, ns_boundary=C.BND_NAMES[ns], land=a.land, ice=a.ice, dt=a.dt, ndte=a.ndte)
    d = blocks.create_distrb_cart(nx, ny, bsx, bsy, ns_boundary_type=ns)
    f = synth.make_block_fields(case, d)
    s = dyn.EvpDynamics(d, f, ndte=a.ndte, xmin=dyn.local_min_dx(f, d), device=device)
    s.init_evp(a.dt)
    s.ctx.upload(f)

    def fence():
        s.ctx.sync()
        torch.cuda.synchronize()

    r = timed_steps(s.ctx, a.ndte, 3, 2, fence)
    st = r["stats"]
    roof = roofline(r, float(st.icellt), float(st.icellu), revp=False)
    s.close()
    return {"ns": ns, "ms_per_step": 1e3 * r["wall_s"] / r["steps"], "steps": r["steps"],
            "value": 0.5 * (st.icellt + st.icellu) * a.ndte * r["steps"] / r["wall_s"],
            "active_T_cells": int(st.icellt), "loop_ms_per_step": r["loop_ms"] / r["steps"],
            "roofline": {k: roof[k] for k in ("kernel", "achieved", "frac", "avg_launch_ms", "launches_timed", "alg_bytes_per_launch")}}


def next_rows(a, case, d, f, xmin, device, ncat=5, trcr_depend=(0, 1, 1, 1, 1, 2, 1, 1, 1, 1)):
    """eap(dt) and evpk_transport_remap on the bench grid, wall clock around synchronised calls, state resident in HBM.
    trcr_depend: Tsfc, 4 x qice, qsno, 4 x sice beyond hice, hsno (nilyr = 4, nslyr = 1: ntrace = 12)."""
    import ctypes as ct
    import numpy as np
    import torch
    from cice5_amd import dyn, synth
    res = {}
    synth.add_eap_state(f)
    s = dyn.EvpDynamics(d, f, ndte=a.ndte, xmin=xmin, device=device)
    s.init_eap(a.dt)
    ctx = s.ctx
    ctx.upload(f)
    t = []
    for n in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.prep(); ctx.subcycle(a.ndte); ctx.finish(); ctx.sync()
        t.append(time.perf_counter() - t0)
    st = ctx.stats()
    res["eap"] = {"what": "eap(dt), kdyn = 2 (ice_dyn_eap.F90:66), same grid / state / ndte as the headline", "ms_per_eap": 1e3 * min(t[1:]),
                  "value": 0.5 * (st.icellt + st.icellu) * a.ndte / min(t[1:]), "unit": "cell-updates/s", "loop_ms": float(st.loop_ms)}
    # whole-loop HBM roofline of eap: what one subcycle (k_eap_sub: stress_eap + stepu) must move -- per T cell 8 metrics 64 B,
    # tinyarea / strength 16, tarear 8, (u, v) 16, the cached angles (gamma, a' per corner) 64, sigma 96 in + 96 out, mask 1; per U
    # cell 8 inputs 64, (u, v) out 16 -- over the loop time per subcycle (launch gaps, halo and the stepa launches included)
    eap_bytes = 361.0 * st.icellt + 80.0 * st.icellu
    res["eap"]["roofline"] = {"bound": "hbm", "alg_bytes_per_subcycle": eap_bytes, "unit": "GB/s", "peak": 8000.0,
                              "achieved": eap_bytes * a.ndte / (float(st.loop_ms) * 1e-3) / 1e9,
                              "frac": eap_bytes * a.ndte / (float(st.loop_ms) * 1e-3) / 1e9 / 8000.0}
    # horizontal_remap with the velocities eap left: areas from the bench's aice split over ncat categories, tracers functions of it
    synth.add_remap_grid(case, d, f)
    ctx.remap_init(f["dxu"], f["dyu"], f["hm"])
    ctx.download({"uvel": f["uvel"], "vvel": f["vvel"]})
    umax = max(float(np.abs(f["uvel"]).max()), float(np.abs(f["vvel"]).max()), 1e-6)
    dt_r = 0.3 * xmin / umax                                     # departure points up to 0.3 cells away
    ntrace = 2 + len(trcr_depend)
    depend, ttype = np.zeros(ntrace, np.int32), np.ones(ntrace, np.int32)
    for nt, dep in enumerate(trcr_depend):
        depend[2 + nt] = dep
        ttype[2 + nt] = 1 if dep == 0 else (3 if dep > 2 and trcr_depend[dep - 3] > 0 else 2)
    has = np.zeros(ntrace, np.int32)
    has[depend[depend > 0] - 1] = 1
    dev = torch.device("cuda", device)
    aice = torch.from_numpy(f["aice"]).to(dev)
    ocean = torch.from_numpy((f["tmask"] > 0).astype(np.float64)).to(dev)
    mm = torch.zeros((d.nblocks, ncat + 1) + tuple(aice.shape[1:]), dtype=torch.float64, device=dev)
    tm = torch.zeros((d.nblocks, ncat, ntrace) + tuple(aice.shape[1:]), dtype=torch.float64, device=dev)
    w = np.arange(1, ncat + 1, dtype=np.float64); w /= w.sum()
    for n in range(ncat):
        mm[:, n + 1] = aice * float(w[n])
        for k in range(ntrace):
            tm[:, n, k] = torch.where(mm[:, n + 1] > 0, (k + 1.0) * (0.5 + 0.3 * aice) * (1.0 + 0.1 * n), torch.zeros_like(aice))
    mm[:, 0] = (1.0 - aice) * ocean
    m0, t0_ = mm.clone(), tm.clone()
    p64, p32 = ct.POINTER(ct.c_double), ct.POINTER(ct.c_int32)
    t = []
    for n in range(3):
        mm.copy_(m0); tm.copy_(t0_)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = ctx._L.evpk_transport_remap(ctx._ctx, float(dt_r), ncat, ntrace, ct.cast(mm.data_ptr(), p64), ct.cast(tm.data_ptr(), p64),
                                         ttype.ctypes.data_as(p32), depend.ctypes.data_as(p32), has.ctypes.data_as(p32), 3, 1, 0)
        t.append(time.perf_counter() - t0)
        if rc:
            raise RuntimeError(f"evpk_transport_remap rc={rc}: " + ctx._L.evpk_last_error(ctx._ctx).decode())
    cells = d.nx_global * d.ny_global
    nfield = ncat + 1 + ncat * ntrace
    # planes a call moves through HBM, each counted once per kernel that reads or writes it: gather + scatter,
    # construct (its three outputs per tracer only where the category has ice), two flux kernels, update
    ncp, ntp = ncat + 1, ncat * ntrace
    icefrac = float((m0[:, 1:] > 1.0e-11).double().mean())
    # (what is only touched where a category has ice -- tc, tx, ty, the tracer reads of the update -- counted with the ice fraction)
    fused = os.environ.get("EVPK_REMAP_FUSED", "1") != "0" and ntrace <= 14
    planes = (4 * nfield                                             # gather + scatter: block arrays <-> planes
              + (nfield + 1 + 2 * ncp + 3 * ntp * icefrac))          # construct: mm, tm, hm in; mx, my out; tc, tx, ty out where ice
    if fused:   # k_remap_fluxupd: mm, mx, my, (tc, tx, ty), dp, (tm) in; new mm, tm out; the new mm copied over the old one
        planes += 3 * ncp + 3 * ntp * icefrac + 2 + ntp * icefrac + ncp + ntp + 2 * ncp
    else:
        planes += (2 * (3 * ncp + 3 * ntp * icefrac + nfield + 2)    # flux E, N: mm, mx, my, (tc, tx, ty), dp in; fluxes out
                   + (4 * ncp + ntp + 3 * ntp * icefrac))            # update: mm in / out, mass fluxes; tm out, tm + tracer fluxes in where ice
    moved = planes * cells * 8.0
    # counter-backed bytes of the same call (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, scripts/prof_remap.sh), if they were
    # measured on this build of the kernels
    pmc = None
    try:
        db = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        e = [x for x in db.get("remap", []) if x.get("source_sha") == source_sha() and x.get("cells") == cells and x.get("fields") == nfield]
        if e:
            pmc = {"hbm_bytes_per_call": e[-1]["hbm_bytes_per_call"], "GBps": e[-1]["hbm_bytes_per_call"] / min(t[1:]) / 1e9,
                   "frac_of_peak": e[-1]["hbm_bytes_per_call"] / min(t[1:]) / 1e9 / HBM_PEAK_GBS, "over_compulsory": e[-1]["hbm_bytes_per_call"] / (2 * 8 * cells * nfield),
                   "profile": e[-1].get("profile")}
    except (OSError, ValueError, KeyError):
        pmc = None
    res["transport_remap"] = {"what": "horizontal_remap (ice_transport_remap.F90:309), ncat = %d, ntrace = %d, ice state resident in HBM (device arrays), "
                                      "velocities as eap left them" % (ncat, ntrace), "ms_per_call": 1e3 * min(t[1:]),
                              "value": cells * nfield / min(t[1:]), "unit": "field-cell-updates/s", "fields": nfield,
                              "compulsory_GBps": 2 * 8 * cells * nfield / min(t[1:]) / 1e9,
                              "roofline": {"bound": "hbm", "achieved": moved / min(t[1:]) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": moved / min(t[1:]) / 1e9 / HBM_PEAK_GBS, "bytes_per_call": moved,
                                           "note": "whole call; every plane a kernel reads or writes counted once per kernel (hand count)",
                                           "traffic_pmc": pmc},
                              "max_area_change": float((mm - m0).abs().max())}
    s.close()
    return res


def cpu_baseline(d, f, a, xmin):
    """The oracle (C restatement of the reference loop, OpenMP over blocks like the reference's
    THRD build) on the host cores: the first few subcycles of the same workload, same blocks."""
    from oracle import orc
    p = orc.make_params(a.dt, a.ndte, xmin)
    threads = orc._limit_threads()
    nt, nu, secs = orc.evp(d, p, f, nsub=a.cpu_subcycles)
    halo = orc.last_halo_seconds
    cores = min(threads, d.nblocks)
    n = 0.5 * (nt + nu) * a.cpu_subcycles
    return {"value": n / secs, "unit": "cell-updates/s", "cores": cores, "kind": "port", "per_core": n / secs / cores, "icellt": nt, "icellu": nu,
            "loop_only": n / max(secs - halo, 1e-9), "with_halo": n / secs, "halo_share": halo / secs,
            "reference_per_core_survey": 1.1e7,
            "sample": f"first {a.cpu_subcycles} of {a.ndte} subcycles of the same {d.nx_global}x{d.ny_global} state "
                      f"({d.nblocks} blocks of {d.block_size_x}x{d.block_size_y}, OpenMP over blocks as the reference's THRD "
                      f"build): stress + stepu ({secs - halo:.2f} s) + the per-subcycle halo update of (u, v) through a "
                      f"precomputed ghost-cell schedule ({halo:.2f} s); a C PORT of the reference loop (kind = port), "
                      f"value = with_halo; reference_per_core_survey = the reference's own whole evp on gx3 on one "
                      f"Xeon core (SURVEY.md S8c)"}


if __name__ == "__main__":
    main()
