#!/usr/bin/env python3
"""Headline benchmark: Mpix/s per frame pair at 1920x1080 with a fixed SOR schedule (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W

A "step" is one complete coarse-to-fine solve of one frame pair (pyramid, every outer iteration, every SOR sweep,
final bicubic warp) through the C ABI (papof_flow_device) with both frames already resident in HBM when the timed
region starts.  Workload at N=1 = BASELINE.json configs[3]: 1920x1080 pair, 5 pyramid levels, 3 outer / 30 SOR
iterations at every level, exact (reference-order) SOR, fp64.  N>1 = replicas, one independent frame pair per GPU
(weak scaling, no data-path collective; see DESIGN.md "Multi-GPU").

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- for the dominant kernel (the exact-order SOR solve: k_sor_exact, and k_sor_fused on the full-size level): algorithmic bytes (80 B per cell-update)
                  of all SOR launches of a step / their summed duration, measured live with HIP events recorded on
                  the library's own stream right around every solver kernel (behind the memsets that prepare a solve)
                  inside the timed region;
  cpu_baseline -- the same workload run once on ONE host core by the untouched reference (oracle/_ref, kind
                  "reference") when its prebuilt .so travelled with the snapshot, else by our CPU restatement
                  (oracle/, kind "port").  Checker code, used here only as the timed CPU baseline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

SCHEDULES = {  # (n_outer, outer_per_level, n_sor, sor_per_level)
    "cfg4": (3, 0, 30, 0),       # BASELINE.json configs[3]: "3 outer / 30 SOR iters"
    "reference": (7, 1, 30, 3),  # the reference's hard-coded schedule (src/OpticalFlow.cpp:749-751,823)
}
MODES = {"exact": 0, "redblack": 1, "jacobi": 2}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
BYTES_PER_UPDATE = 80  # 8 fp64 reads + 2 fp64 writes per cell-update (SURVEY.md §8d / BASELINE.md §3)


def load_frames(res):
    try:
        import cases
        a, b = cases.load_pair(res)
        return a, b, "HoChiMinhTraffic_10FPS_%s frame_00001->00002 (committed fixture of the reference's set)" % res
    except Exception:
        h, w = {"240": (135, 240), "480": (270, 480), "960": (540, 960), "1920": (1080, 1920)}[res]
        rng = np.random.default_rng(0)
        a = rng.uniform(0, 1, (h // 8 + 2, w // 8 + 2, 3)).repeat(8, 0).repeat(8, 1)[:h, :w]
        b = np.roll(a, (1, 2), (0, 1)) + rng.normal(0, 0.01, a.shape)
        return np.ascontiguousarray(a), np.ascontiguousarray(np.clip(b, 0, 1)), "synthetic"


def level_dims(gpu, h, w, levels):
    import ctypes
    dims = np.zeros(2 * levels, dtype=np.int32)
    n = ctypes.c_long(0)
    rc = gpu.L.papof_stage_pyramid(gpu.h, None, h, w, 3, 0.75, levels, dims.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                   None, ctypes.byref(n))
    assert rc == 0
    return [(int(dims[2 * i]), int(dims[2 * i + 1])) for i in range(levels)]


def cpu_baseline(a, b, levels, sched, mode):
    """[(kind, seconds)]: the same solve on ONE host core by the untouched reference compiled into oracle/_ref (kind
    "reference"; its prebuilt .so travels with the snapshot for exactly this leg and exists for the reference's own sweep
    order only) and by our CPU restatement (kind "port", pinned bit for bit to the reference)."""
    from _libs import OracleLib, RefLib
    libs = []
    if mode == 0 and RefLib.available():
        try:
            libs.append(("reference", RefLib()))
        except OSError:
            pass
    libs.append(("port", OracleLib()))
    out = []
    for kind, lib in libs:
        t0 = time.perf_counter()
        lib.coarse2fine_flow_sched(a, b, levels, 0.012, 0.75, sched[0], sched[1], 1, sched[2], sched[3], mode=mode,
                                   omega=1.8 if mode != 2 else 1.0)
        out.append((kind, time.perf_counter() - t0))
    return out


class stdout_to_stderr:
    """RCCL prints a version banner to STDOUT when a communicator is created; this program's stdout carries exactly one
    JSON line, so file descriptor 1 is pointed at stderr while communicators are being set up."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def measure_tiles(args, gpu, dist, rank, world, d1, d2, h, w, c, sched):
    """Secondary figure for N>1 (never `value`): ONE frame pair sharded as rows x cols tiles over all ranks, red-black
    SOR with ghost-zone halo exchange over RCCL (csrc/tiles.hip; BASELINE.json configs[4]) -- strong scaling of a
    single pair, next to the same red-black solve on one GPU.  A watchdog abandons a stuck collective: every rank
    leaves after --tiles-timeout seconds and rank 0 reports the timeout instead of a number."""
    import threading

    import torch
    from papteam_opticalflow_amd import capi, default_params
    state = {"phase": "init", "done": False}

    def on_timeout():
        if not state["done"]:
            state["timed_out"] = True
    result = {"sor_mode": "redblack", "scaling": "strong", "halo_halfsweeps": args.tile_halo}
    timer = threading.Timer(args.tiles_timeout, on_timeout)
    timer.daemon = True
    box = {}

    def one_mode(tr, sor_mode, res):
        """time the tiled call in one sweep order; rank 0 also runs the same solve on one GPU and compares the bits"""
        P = default_params(n_outer=sched[0], n_outer_per_level=sched[1], n_sor=sched[2], n_sor_per_level=sched[3],
                           sor_mode=sor_mode, omega=1.8, phase_timing=0)
        # own result buffers on rank 0: the headline run's (vx, vy) stay untouched for its parity statistic
        dvx, dvy, dwp = ((gpu.dev_alloc(h * w * 8), gpu.dev_alloc(h * w * 8), gpu.dev_alloc(h * w * c * 8))
                         if rank == 0 else (None, None, None))
        outs = (dvx, dvy, dwp)
        state["phase"] = "warmup (sor_mode %d)" % sor_mode
        tr.flow_device(d1, d2, h, w, c, args.levels, P, *outs)
        torch.cuda.synchronize()
        dist.barrier()
        state["phase"] = "timed (sor_mode %d)" % sor_mode
        t0 = time.perf_counter()
        sor = 0.0
        for _ in range(args.steps):
            sor += tr.flow_device(d1, d2, h, w, c, args.levels, P, *outs)[6]
        torch.cuda.synchronize()
        dist.barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        sec = float(dt.item()) / args.steps
        n_ex, n_bytes = tr.stats()
        info = tr.comm_info()  # what RCCL itself says: ncclCommCount / ncclCommUserRank of this rank's communicator
        per_rank = torch.zeros(world, 3, dtype=torch.float64, device="cuda")
        per_rank[rank, 0] = sor / args.steps * 1e3
        per_rank[rank, 1] = info["nranks_seen"]
        per_rank[rank, 2] = info["rank_seen"]
        dist.all_reduce(per_rank)
        pr = per_rank.cpu().numpy()
        res.update({"ms_per_pair": round(sec * 1e3, 4), "value": round(h * w / 1e6 / sec, 4), "unit": "Mpix/s",
                    "n_ranks_seen": int(pr[:, 1].min()), "ranks_seen": [int(x) for x in pr[:, 2]],
                    "sor_ms_per_pair_by_rank": [round(float(x), 4) for x in pr[:, 0]],
                    "sor_ms_per_pair_rank0": round(sor / args.steps * 1e3, 4),
                    "exchanges_per_pair": n_ex, "exchanged_mb_per_pair_rank0": round(n_bytes / 1e6, 3)})
        if sor_mode == 1:
            res["tile_grid_in_use"] = "%dx%d" % (info["rows"], info["cols"])
        if rank == 0:  # the same solve on one GPU: time, and the sharded result must be bit-identical to it
            tx, ty = np.zeros((h, w)), np.zeros((h, w))
            gpu.dev_download(tx, dvx)
            gpu.dev_download(ty, dvy)
            gpu.flow_device(d1, d2, h, w, c, args.levels, P, dvx, dvy, dwp)
            t1 = time.perf_counter()
            for _ in range(3):
                gpu.flow_device(d1, d2, h, w, c, args.levels, P, dvx, dvy, dwp)
            name = "redblack" if sor_mode == 1 else "exact"
            res["one_gpu_%s_ms_per_pair" % name] = round((time.perf_counter() - t1) / 3 * 1e3, 4)
            sx, sy = np.zeros((h, w)), np.zeros((h, w))
            gpu.dev_download(sx, dvx)
            gpu.dev_download(sy, dvy)
            res["bit_identical_to_one_gpu_%s" % name] = bool(np.array_equal(tx, sx) and np.array_equal(ty, sy))
            for p_ in (dvx, dvy, dwp):
                gpu.dev_free(p_)
        dist.barrier()

    def work():
        try:
            rows, cols = capi.tiles_grid(world)
            result["grid"] = "%dx%d" % (rows, cols)
            idt = torch.zeros(capi.TILES_ID_BYTES, dtype=torch.uint8, device="cuda")
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(capi.tiles_unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            uid = bytes(idt.cpu().numpy().tobytes())
            state["phase"] = "comm_init"
            tr = capi.TileRank.create(gpu, uid, rank, world, rows, cols, args.tile_halo)
            one_mode(tr, 1, result)
            # the same pair in the REFERENCE's sweep order, split into horizontal ranges of solver bands over the ranks
            # (csrc/tiles.hip: bands_flow; over RCCL the staged protocol: one message per solve and cut): the sharded
            # configuration that meets the 1e-4 parity bar -- bit-identical to the one-GPU exact call
            ex = {"sor_mode": "exact", "scaling": "strong", "split": "%d ranges of solver bands" % world,
                  "protocol": "staged over RCCL (kernel-boundary sends; the exact-order solve is one dependency chain: its "
                              "share does not speed up with ranks, the other stages do)"}
            try:
                one_mode(tr, 0, ex)
            except Exception as e:  # noqa: BLE001
                ex["error"] = "%s: %s" % (type(e).__name__, e)
            result["exact_order"] = ex
            tr.close()
        except Exception as e:  # noqa: BLE001 -- reported, never silently replaced by another path
            box["error"] = "%s: %s" % (type(e).__name__, e)
        state["done"] = True

    th = threading.Thread(target=work, daemon=True)
    timer.start()
    th.start()
    while th.is_alive() and not state.get("timed_out"):
        th.join(0.2)
    timer.cancel()
    if state.get("timed_out") and not state["done"]:
        result["error"] = "timeout after %.0f s in phase %s" % (args.tiles_timeout, state["phase"])
        result["abandoned"] = True
    elif "error" in box:
        result["error"] = box["error"]
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--res", default="1920", choices=["240", "480", "960", "1920"])
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--schedule", default="cfg4", choices=sorted(SCHEDULES))
    ap.add_argument("--mode", default="exact", choices=sorted(MODES))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--concurrent", action="store_true",
                    help="also measure several pairs in flight on one GPU (device-resident burst of 4, and a 24-pair "
                         "collection through host buffers).  Off by default: concurrent launches stretch every kernel, "
                         "and the default command is the one whose rocprof kernel statistics are committed")
    ap.add_argument("--no-concurrent", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--no-collection", action="store_true",
                    help="skip the secondary collection figure (24 pairs, 4 sequences in flight, host buffers) that the "
                         "default run reports after the timed headline")
    ap.add_argument("--pairs", type=int, default=1,
                    help="frame pairs solved concurrently per GPU (one handle + stream + host thread each); a step "
                         "is then one solve of EVERY pair and value counts all of them")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend for N>1")
    ap.add_argument("--no-tiles", action="store_true",
                    help="N>1: skip the secondary measurement of ONE pair sharded as 2-D tiles over all ranks")
    ap.add_argument("--force-tiles", action="store_true",
                    help="run the tiled measurement even with one rank (a tile group of one: no sends; exercises the RCCL "
                         "bootstrap and the reporting code on a one-GPU box)")
    ap.add_argument("--tile-halo", type=int, default=10, help="ghost-zone depth of the tiled solve, in half-sweeps")
    ap.add_argument("--tiles-timeout", type=float, default=240.0,
                    help="seconds after which a stuck tiled measurement is abandoned (the JSON line is still printed)")
    ap.add_argument("--simulate-step-ms", type=float, default=0.0,
                    help="CPU-only rehearsal of the N>1 protocol (tests): a step sleeps (rank+1) x this long instead "
                         "of running the GPU path; the printed value is meaningless")
    args = ap.parse_args()
    simulate = args.simulate_step_ms > 0

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    use_cuda = not simulate
    # PyTorch (its process group is only used for barriers / the max over ranks / the tile group's id) must touch the GPU
    # BEFORE libpapof does: then both share PyTorch's HIP runtime; the other order leaves PyTorch without a device
    if world > 1:
        import torch
        import torch.distributed as dist
        if use_cuda:
            torch.cuda.set_device(local_rank)
            with stdout_to_stderr():
                dist.init_process_group(backend=args.backend, device_id=torch.device("cuda", local_rank))
                dist.barrier()  # creates the communicator (and prints RCCL's banner) now, not inside the timed region
        else:
            dist.init_process_group(backend="gloo")
    elif args.force_tiles and use_cuda:  # a process group of one
        import socket
        import torch
        import torch.distributed as dist
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        torch.cuda.set_device(local_rank)
        with stdout_to_stderr():
            dist.init_process_group(backend=args.backend, init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                                    device_id=torch.device("cuda", local_rank))
            dist.barrier()

    sched, mode = SCHEDULES[args.schedule], MODES[args.mode]
    if simulate:
        gpu = None
        h, w = {"240": (135, 240), "480": (270, 480), "960": (540, 960), "1920": (1080, 1920)}[args.res]
        c, data_desc = 3, "none (protocol rehearsal)"

        def one_step():
            time.sleep(args.simulate_step_ms * 1e-3 * (rank + 1))
            return [0.0] * 10
    else:
        from papteam_opticalflow_amd import Papof, default_params
        gpu = Papof(local_rank)
        a, b, data_desc = load_frames(args.res)
        h, w, c = a.shape
        P = default_params(n_outer=sched[0], n_outer_per_level=sched[1], n_sor=sched[2], n_sor_per_level=sched[3],
                           sor_mode=mode, omega=1.8 if mode != 2 else 1.0, phase_timing=2)  # total + SOR kernels only
        P_call = default_params(n_outer=sched[0], n_outer_per_level=sched[1], n_sor=sched[2], n_sor_per_level=sched[3],
                                sor_mode=mode, omega=1.8 if mode != 2 else 1.0, phase_timing=0)  # the drop-in's default
        d1, d2 = gpu.dev_alloc(a.nbytes), gpu.dev_alloc(b.nbytes)
        dvx, dvy, dwp = gpu.dev_alloc(h * w * 8), gpu.dev_alloc(h * w * 8), gpu.dev_alloc(a.nbytes)
        gpu.dev_upload(d1, a)  # inputs resident in HBM before the timed region
        gpu.dev_upload(d2, b)
        extra = []  # further concurrent pairs: own handle (arena + stream) and own output buffers each
        for _ in range(args.pairs - 1):
            g2 = Papof(local_rank)
            extra.append((g2, g2.dev_alloc(h * w * 8), g2.dev_alloc(h * w * 8), g2.dev_alloc(a.nbytes)))

        def one_step():  # returns after every stream has drained
            if not extra:
                return gpu.flow_device(d1, d2, h, w, c, args.levels, P, dvx, dvy, dwp)
            import threading
            res = [None] * (1 + len(extra))

            def run(i, g, ox, oy, ow):
                res[i] = g.flow_device(d1, d2, h, w, c, args.levels, P, ox, oy, ow)
            th = [threading.Thread(target=run, args=(i + 1,) + e) for i, e in enumerate(extra)]
            for t_ in th:
                t_.start()
            run(0, gpu, dvx, dvy, dwp)
            for t_ in th:
                t_.join()
            return res[0]

    def sync_all():
        if dist is not None:
            if use_cuda:
                import torch
                torch.cuda.synchronize()
            dist.barrier()
            if use_cuda:
                import torch
                torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    sync_all()
    t0 = time.perf_counter()
    sor_sec = 0.0
    solve_log = {}  # (h, w, n_sor, kind, depth) -> [solves, launches, seconds] over the timed steps (roofline.by_level)
    for _ in range(args.steps):
        sor_sec += one_step()[6]
        if gpu is not None and args.pairs == 1:  # a host-side read of what the call already recorded: no device work
            for e in gpu.last_sor_solves():
                acc = solve_log.setdefault((e["h"], e["w"], e["n_sor"], e["kind"], e["depth"]), [0, 0, 0.0])
                acc[0] += 1
                acc[1] += e["launches"]
                acc[2] += e["sec"]
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if use_cuda else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)  # the slowest rank defines the job's time
        elapsed = float(tt.item())

    tiles = None
    if use_cuda and not args.no_tiles and (world > 1 or args.force_tiles):
        with stdout_to_stderr():
            tiles = measure_tiles(args, gpu, dist, rank, world, d1, d2, h, w, c, sched)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * args.pairs * (h * w / 1e6) / (elapsed / args.steps)
        dims = level_dims(gpu, h, w, args.levels) if gpu is not None else [(w, h)] * args.levels
        updates = sum(lw * lh * (sched[0] + k * sched[1]) * (sched[2] + k * sched[3]) for k, (lw, lh) in enumerate(dims))
        # solver-kernel launches per step: solves x launches per solve (the exact-order kernels: one per solve unless a
        # solve exceeds the chip's resident capacity; the blocked red-black / Jacobi kernel: one per `depth` half-sweeps)
        launches, depths = 0, []
        for k, (lw, lh) in enumerate(dims):
            nl, dp = gpu.sor_plan(lh, lw, sched[2] + k * sched[3], mode) if gpu is not None else (1, 1)
            launches += (sched[0] + k * sched[1]) * nl
            depths.append(dp)
        if gpu is not None and mode == 0 and args.pairs == 1:  # what the last step really launched (levels solved as strips
            launches = gpu.last_sor_stats()[0] or launches      # of bands issue one launch per strip and solve)
        sor_step = sor_sec / args.steps
        achieved = updates * BYTES_PER_UPDATE / 1e9 / sor_step if sor_step > 0 else 0.0
        traffic, traffic_by_grid = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("%s_%s_%s" % (args.res, args.schedule, args.mode))
                traffic_by_grid = (tj.get("%s_%s_%s_detail" % (args.res, args.schedule, args.mode)) or {}).get("by_grid_finest_first")
            except Exception:
                traffic = None
        # per level and per kernel: the solver kernels' own HIP-event time of every solve, as the library recorded it
        KIND = {0: "k_sor_exact", 1: "k_sor_fused", 2: "k_sor_group", 3: "k_sor_blocked<redblack>",
                4: "k_sor_blocked<jacobi>", 5: "k_sor_redblack/k_sor_jacobi (one launch per half-sweep)",
                6: "k_sor_tiny (whole plane in one workgroup)"}
        by_level, by_kernel = [], {}
        for (lh, lw, ns, kind, depth), (n_solves, n_launch, sec) in sorted(solve_log.items(), key=lambda kv: -kv[0][0] * kv[0][1]):
            if sec <= 0 or n_launch <= 0:
                continue
            gbytes = lh * lw * ns * BYTES_PER_UPDATE * n_solves / 1e9
            name = "%s<%d>" % (KIND.get(kind, "?"), depth) if kind in (0, 1, 2) else KIND.get(kind, "?")
            by_level.append({"level": "%dx%d" % (lw, lh), "sweeps": ns, "kernel": name, "solves": n_solves,
                             "launches": n_launch, "avg_launch_us": round(sec / n_launch * 1e6, 1),
                             "avg_solve_us": round(sec / n_solves * 1e6, 1),
                             "achieved": round(gbytes / sec, 1), "frac": round(gbytes / sec / HBM_PEAK_GBS, 4)})
            k = by_kernel.setdefault(name, {"launches": 0, "sec": 0.0, "gbytes": 0.0})
            k["launches"] += n_launch
            k["sec"] += sec
            k["gbytes"] += gbytes
        # per-level HBM-side traffic from this round's PMC passes (profiles/pmc_traffic.json: by_grid_finest_first; STATIC, as
        # roofline.traffic): the levels finest first are the solver's launch grids largest first -- attached when the counts agree
        if traffic_by_grid and all(e["launches"] == e["solves"] for e in by_level):
            def norm(n):  # "k_sor_exact<8, true, false>" -> "k_sor_exact<8>"
                return n.split("<")[0] + "<" + n.split("<")[1].split(",")[0].rstrip(">") + ">" if "<" in n else n
            pm = {}
            for g in traffic_by_grid:
                pm.setdefault(norm(g["kernel"]), []).append(g)
            lv = {}
            for e in by_level:
                lv.setdefault(e["kernel"], []).append(e)
            for name, es in lv.items():  # same kernel: larger plane = larger launch grid
                gs = sorted(pm.get(name, []), key=lambda g: -g["grid_threads"])
                if len(gs) != len(es):
                    continue
                for e, g in zip(es, gs):  # (by_level is sorted by plane size already)
                    alg = int(e["level"].split("x")[0]) * int(e["level"].split("x")[1]) * e["sweeps"] * BYTES_PER_UPDATE
                    e["traffic"] = g["bytes_per_launch"]
                    e["traffic_over_algorithmic"] = round(g["bytes_per_launch"] / alg, 3)
                    e["frac_on_traffic"] = round(g["bytes_per_launch"] / (e["avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        by_kernel = {n: {"launches": v["launches"], "avg_launch_us": round(v["sec"] / v["launches"] * 1e6, 1),
                         "share_of_sor_time": round(v["sec"] / max(sor_sec, 1e-30), 4),
                         "achieved": round(v["gbytes"] / v["sec"], 1), "frac": round(v["gbytes"] / v["sec"] / HBM_PEAK_GBS, 4)}
                     for n, v in by_kernel.items()}
        # parity statistic of the metric: max-abs delta (u, v) against the untouched reference's golden values
        parity, full_sha_equal = None, None
        try:
            import cases
            gold = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
            key = {("1920", "cfg4", 5): "cfg4_1920_L5", ("1920", "reference", 5): "e2e_1920_L5",
                   ("480", "cfg4", 5): "cfg4_480_L5", ("960", "reference", 5): "e2e_960_L5"}.get(
                       (args.res, args.schedule, args.levels))
            if key and mode == 0 and gpu is not None:
                vx, vy = np.zeros((h, w)), np.zeros((h, w))
                gpu.dev_download(vx, dvx)
                gpu.dev_download(vy, dvy)
                parity = float(max(np.abs(cases.subsample(vx) - gold[key + "|vx"]).max(),
                                   np.abs(cases.subsample(vy) - gold[key + "|vy"]).max()))
                # every value, not the strided subsample: SHA-256 of the full arrays the untouched reference produced
                man = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["cases"][key]
                full_sha_equal = bool(cases.sha(vx) == man["vx"]["sha"] and cases.sha(vy) == man["vy"]["sha"])
        except Exception:
            parity = None
        out = {
            "metric": "Mpix/s per frame-pair (fixed SOR iters) at 1920x1080; max-abs d(u,v) vs ref",
            "value": round(value, 4), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": data_desc,
            "config": {"workload": "%dx%d frame pair, %d-level pyramid, schedule %s (outer %d+%dk, SOR %d+%dk), "
                                   "%s-order SOR" % (w, h, args.levels, args.schedule, sched[0], sched[1], sched[2],
                                                     sched[3], args.mode),
                       "value_is": "device-resident, as the bench contract prescribes (inputs already resident in HBM when "
                                   "the timed region starts; papof_flow_device, timers: total + SOR kernels).  "
                                   "BASELINE.md's caller-side definition -- H*W over the wall time of ONE "
                                   "coarse2fine_flow(im1, im2, levels) call with float64 numpy frames in and out, both "
                                   "PCIe transfers and all ten timers -- is value_call_inclusive on this same line",
                       "pairs_in_flight_per_gpu": args.pairs,
                       "parallelism": "replicas: one independent frame pair per GPU" if world > 1 else "1 GPU"},
            "max_abs_duv_vs_reference": parity, "full_sha_equal": full_sha_equal,
            "roofline": {"bound": "hbm", "kernel": "k_sor_exact + k_sor_fused (exact-order SOR solves)" if mode == 0 else
                         "k_sor_blocked (LDS-tiled, temporally blocked %s solves; %s %s per launch by level)" % (
                             args.mode, "/".join(str(d) for d in depths), "half-sweeps" if mode == 1 else "sweeps"),
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": ("profiles/pmc_traffic.json (STATIC: HBM bytes per launch from this round's "
                                            "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, FETCH_SIZE "
                                            "doubled per the guide's gfx950 correction; not measured in this run)")
                         if traffic is not None else None,
                         "frac_on_traffic": round(traffic * launches / 1e9 / sor_step / HBM_PEAK_GBS, 4)
                         if traffic and sor_step > 0 and launches else None,
                         "by_level": by_level or None, "by_kernel": by_kernel or None,
                         "launches_per_step": launches, "cell_updates_per_step": updates,
                         "avg_launch_ms": round(sor_step * 1e3 / launches, 4) if launches else None,
                         "sor_ms_per_step": round(sor_step * 1e3, 4)},
        }
        if tiles is not None:
            out["tiles"] = tiles
        if world == 1 and args.pairs == 1 and not simulate and args.concurrent:
            # secondary figure (not `value`): the same solve for 4 independent pairs in flight on one GPU -- what a
            # collection of frame pairs (the reference's TestSuite walks 101 per set) gets out of the device
            import threading
            hs = [gpu] + [Papof(local_rank) for _ in range(3)]
            outs = [(dvx, dvy, dwp)] + [(g2.dev_alloc(h * w * 8), g2.dev_alloc(h * w * 8), g2.dev_alloc(a.nbytes))
                                        for g2 in hs[1:]]

            def burst():
                th = [threading.Thread(target=g2.flow_device, args=(d1, d2, h, w, c, args.levels, P) + o)
                      for g2, o in zip(hs, outs)]
                for t_ in th:
                    t_.start()
                for t_ in th:
                    t_.join()
            burst()
            tb = time.perf_counter()
            for _ in range(3):
                burst()
            tb = (time.perf_counter() - tb) / 3
            out["concurrent_pairs"] = {"pairs_in_flight": 4, "value": round(4 * h * w / 1e6 / tb, 2), "unit": "Mpix/s",
                                       "ms_per_burst": round(tb * 1e3, 3)}
            for g2, o in list(zip(hs, outs))[1:]:
                for ptr in o:
                    g2.dev_free(ptr)
                g2.close()
        if world == 1 and not simulate:
            # the drop-in entry point's view (host buffers in and out, pageable numpy arrays): NOT `value`
            gpu.coarse2fine_flow(a, b, args.levels, P_call)
            th = time.perf_counter()
            for _ in range(3):
                gpu.coarse2fine_flow(a, b, args.levels, P_call)
            out["pcie_inclusive_ms_per_pair"] = round((time.perf_counter() - th) / 3 * 1e3, 3)
            # BASELINE.md's definition of the metric on the caller's clock: H*W / wall seconds of ONE
            # coarse2fine_flow(im1, im2, levels) call, float64 numpy frames in, float64 results out (both PCIe transfers)
            out["value_call_inclusive"] = round(h * w / 1e6 / (out["pcie_inclusive_ms_per_pair"] * 1e-3), 4)
            try:  # the callers' side: uint8 frames in, and a video pushed frame by frame
                import cases
                a8, b8 = cases.load_frame_u8(args.res, 1), cases.load_frame_u8(args.res, 2)
                gpu.coarse2fine_flow_u8(a8, b8, args.levels, P)
                th = time.perf_counter()
                for _ in range(3):
                    gpu.coarse2fine_flow_u8(a8, b8, args.levels, P)
                out["pcie_inclusive_u8_ms_per_pair"] = round((time.perf_counter() - th) / 3 * 1e3, 3)
                gpu.seq_reset()
                gpu.seq_push(a8, args.levels, P)
                gpu.seq_push(b8, args.levels, P)
                th = time.perf_counter()
                for i in range(4):
                    gpu.seq_push(a8 if i % 2 == 0 else b8, args.levels, P)
                out["sequence_u8_ms_per_frame"] = round((time.perf_counter() - th) / 4 * 1e3, 3)
                if args.no_collection:
                    raise StopIteration
                # the documented path for COLLECTIONS of frames (the reference's TestSuite walks 101 pairs per set,
                # Code/Serial/TestSuite.py:69-81): several sequences in flight on one GPU, uint8 frames in, results out --
                # secondary figure, measured after (never inside) the timed headline
                from papteam_opticalflow_amd import flow_collection, collection_in_flight
                kw = dict(n_outer=sched[0], n_outer_per_level=sched[1], n_sor=sched[2], n_sor_per_level=sched[3],
                          sor_mode=mode, omega=1.8 if mode != 2 else 1.0)
                nfl = collection_in_flight(h, w)  # 4 at 1080p, 16 for the small frames of the reference's own test matrix
                npairs = 24 if nfl <= 4 else 6 * nfl
                video = [a8, b8] * (npairs // 2) + [a8]
                seen = []
                flow_collection(video[:2 * nfl + 1], args.levels, in_flight=nfl, device=local_rank, batch=0,
                                on_pair=lambda i, *r: None, **kw)
                th = time.perf_counter()
                flow_collection(video, args.levels, in_flight=nfl, device=local_rank, batch=0,
                                on_pair=lambda i, t_, vx_, vy_, w_: seen.append(i), **kw)
                dt = time.perf_counter() - th
                assert sorted(seen) == list(range(npairs))
                out["collection_u8_%d_in_flight" % nfl] = {
                    "pairs": npairs, "in_flight": nfl, "ms_per_pair": round(dt / npairs * 1e3, 3),
                    "value": round(npairs * h * w / 1e6 / dt, 2), "unit": "Mpix/s",
                    "note": "flow_collection(): host uint8 frames in, float64 results out into reused arrays (PCIe-inclusive), "
                            "one stream per handle"}
                # small frames (the reference's own test matrix): the pairs of a collection share every launch of a chain
                # (papof_flow_batch*, csrc/batch.hip) -- flow_collection()'s default there
                from papteam_opticalflow_amd import collection_batch
                nb_ = collection_batch(h, w)
                if nb_ > 1 and mode == 0:
                    npairs = 6 * nb_
                    video = [a8, b8] * (npairs // 2) + [a8]
                    seen = []
                    flow_collection(video, args.levels, device=local_rank, batch=nb_, on_pair=lambda i, *r: None, **kw)
                    th = time.perf_counter()
                    flow_collection(video, args.levels, device=local_rank, batch=nb_,
                                    on_pair=lambda i, t_, vx_, vy_, w_: seen.append(i), **kw)
                    dt = time.perf_counter() - th
                    assert sorted(seen) == list(range(npairs))
                    out["collection_u8_batches_of_%d" % nb_] = {
                        "pairs": npairs, "batch": nb_, "chains_in_flight": 3, "ms_per_pair": round(dt / npairs * 1e3, 3),
                        "value": round(npairs * h * w / 1e6 / dt, 2), "unit": "Mpix/s",
                        "note": "flow_collection(batch=%d): %d consecutive pairs per launch chain (papof_flow_batch_u8), host uint8 "
                                "frames in, float64 results out (PCIe-inclusive); every pair bit-identical to the single call" % (nb_, nb_)}
            except StopIteration:
                pass
            except Exception as e:  # noqa: BLE001 -- secondary figures only
                out["callers_side_error"] = "%s: %s" % (type(e).__name__, e)
        if world == 1 and not args.no_cpu_baseline and not simulate:
            runs = cpu_baseline(a, b, args.levels, sched, mode)
            kind, dt = runs[0]  # the reference itself when its compiled .so is present, else the port
            out["cpu_baseline"] = {"value": round(h * w / 1e6 / dt, 5), "unit": "Mpix/s", "cores": 1, "kind": kind,
                                   "sample": "the same %dx%d pair and schedule, one full solve (%.1f s), single "
                                             "thread (the reference Serial path is single-threaded)" % (w, h, dt),
                                   "host_cpus": os.cpu_count()}
            for k2, dt2 in runs[1:]:  # both kinds side by side when both are present
                out["cpu_baseline"]["also_" + k2] = {"value": round(h * w / 1e6 / dt2, 5), "seconds": round(dt2, 2)}
        print(json.dumps(out), flush=True)
    if tiles is not None and tiles.get("abandoned"):
        sys.stdout.flush()
        os._exit(0)  # a collective of the tiled attempt is stuck: no orderly teardown is possible
    if gpu is not None:
        for p in (d1, d2, dvx, dvy, dwp):
            gpu.dev_free(p)
        gpu.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
