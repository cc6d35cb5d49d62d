#!/usr/bin/env python3
"""bench.py — scans/sec of the registration hot path on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 64×1800-point synthetic LiDAR scans
(cityblock-v1) registered against ONE 10 M-point map with the reference's default point-to-plane ICP
(IcpOptions defaults, icp_registration.hpp:29-37; reference-faithful alpha=0.1 KD-tree search, kdtree.h:128-129).
One "step" = one pass of the hot path over one batch: every rank aligns `--scans-per-gpu` scans (batched many-scans-vs-one-map
mode). The map and its tree are in HBM before the timed region starts. The SCANS are not: scans/sec includes the source deep
copy of every ScanMatch call (SetSource, icp_registration.cpp:221,252-265) — every step aligns a batch that
was copied host → HBM for it through pinned staging on a copy stream, while the previous step's Gauss–Newton loop ran
(depth + 1 batches rotate: one is being copied, the others are being aligned; `--resident` keeps the scans in HBM instead and is
reported as the secondary number). Three alignments are in flight by default (`locgpu_*_align_batch_begin` /
`locgpu_align_batch_end`; `--pipeline 1|2|3`): the first Gauss–Newton iterations of step i+1 fill the chip under the last ones of step i,
which hold a handful of unconverged scans. Every step still begins and ends inside the timed region, which carries no instrumentation;
the kernel durations behind `roofline` and `kernel_ms_per_step` are HIP-event times of `--extra-steps` further steps of the same workload run
one alignment at a time right behind it (launches of different batches overlap inside the timed region), the hardware counters come
from child runs of this script under `rocprofv3 --pmc`, the search loop's lane efficiency from one child run on the library's diagnostic build.
One GPU: BASELINE.json configs[2], every step aligns `--scans-per-gpu` (256) scans. Several GPUs, default (`--scaling strong`):
configs[3] as written — `--total-scans` (256) scans in all, sharded contiguously over the ranks, per-iteration RCCL all-reduce of
the per-scan normal equations inside liblocgpu.so (every rank solves every scan and holds all poses). `--scaling weak`: every
rank aligns its own `--scans-per-gpu` scans, no data-path collective; `value` = scans all ranks completed ÷ max-over-ranks time.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8 ...          # no WORLD_SIZE in the environment: this process starts the 8 ranks itself (self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus 8 ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (≈6.3 TB/s achievable)


def usable_cores():
    """Host threads this process may actually run on: the affinity mask, capped by the cgroup CPU quota when there is one
    (os.cpu_count() reports the machine, not what the container grants)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()  # cgroup v2: "<quota> <period>" or "max <period>"
        if q[0] != "max":
            n = max(1, min(n, int(float(q[0]) / float(q[1]) + 0.5)))
    except (OSError, ValueError, IndexError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = max(1, min(n, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(map_xyz, scans, inits, seconds_budget=20.0, method="p2plane", ndt_kw=None):
    """The oracle (CPU restatement of the reference path, single thread like the reference) on a bounded sample."""
    from oracle import locref  # test infrastructure used here only as the reported baseline

    icp = locref.Ndt(**(ndt_kw or {})) if method == "ndt" else locref.Icp(method=dict(p2p=locref.P2P, p2line=locref.P2LINE, p2plane=locref.P2PLANE)[method])
    t0 = time.time()
    icp.set_target(map_xyz)
    ingest = time.time() - t0
    done, t_align, iters = 0, 0.0, 0
    poses = []
    for s, ip in zip(scans, inits):
        t1 = time.time()
        r = icp.align(s, ip)
        t_align += time.time() - t1
        iters += r["iters"]
        poses.append(r["pose"])
        done += 1
        if t_align > seconds_budget:
            break
    # the same path as a flat-array port (oracle/locref_flat.hpp: packed tree, fixed-size heap, no per-query
    # malloc), one thread, same scans — bit-identical poses, so the R1 figure is not flattered by the reference's allocation style.
    # R3: R2 with whole scans dealt to native threads (std::thread) over every host core — informative upper bound.
    r2 = r3 = None
    if method == "p2plane":
        t1 = time.time()
        p2, _ = icp.align_flat(scans[:done], inits[:done], threads=1)
        t_r2 = time.time() - t1
        cores = usable_cores()
        n_par = min(len(scans), max(done, cores))
        t1 = time.time()
        p3, _ = icp.align_flat(scans[:n_par], inits[:n_par], threads=cores)
        t_r3 = time.time() - t1
        r2 = dict(value=done / t_r2, unit="scans/s", cores=1, scans=done, identical_to_r1=bool(all(np.array_equal(a, b) for a, b in zip(p2, poses))))
        r3 = dict(value=n_par / t_r3, unit="scans/s", cores=cores, machine_threads=os.cpu_count(), scans=n_par,
                  identical_to_r1=bool(all(np.array_equal(a, b) for a, b in zip(p3[:done], poses))))
    return dict(value=done / t_align, unit="scans/s", cores=1, kind="port",
                sample="R1 = %d full %d-pt scans vs the same %.0fM-pt map, oracle/locref.cpp %s (the reference's own style: pointer tree, std::priority_queue "
                       "and std::vector per query), 1 thread like the reference; map ingest %.1fs excluded; %.1f ms per GN iteration"
                       % (done, len(scans[0]), len(map_xyz) / 1e6, method, ingest, 1e3 * t_align / max(iters, 1)),
                r2_flat_port=r2, r3_flat_port_all_cores=r3), poses


def load_traffic(kernel_name, scans_per_gpu, map_points, method):
    """HBM bytes per launch of `kernel_name` from the newest committed PMC collection (tools/collect_traffic.py) made on the
    same workload, or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        c = d.get("config", {})
        if (c.get("scans_per_gpu"), c.get("map_points"), c.get("method")) != (scans_per_gpu, map_points, method):
            continue
        for k, v in d.get("kernels", {}).items():
            if k.split("<")[0] in kernel_name:
                best = v.get("traffic_bytes_per_launch")
    return best


PMC_PASSES = [  # one rocprofv3 --pmc run each, nothing but --pmc (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass)
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64"],
]


def measure_counters_live(passthrough_args, budget_s=300.0, steps=2, pipeline=1, extra=4):
    """Per-kernel hardware counters of this very workload, measured NOW: child runs of this script (2 timed + 4 profiled steps, one
    alignment in flight) under `rocprofv3 --pmc <group>` — separate passes, no tracing, the program directly after `--`, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes. Children are ordinary subprocesses (never an exec of this process).
    Returns ({kernel base name: {counter: total over the child run, "launches": n}, "_steps": steps the child ran}, note)."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return {}, "rocprofv3 not on PATH"
    child = [os.path.realpath(sys.executable), os.path.abspath(__file__)] + list(passthrough_args) + ["--steps", str(steps), "--warmup", "0", "--pipeline", str(pipeline), "--extra-steps", str(extra), "--no-cpu-baseline", "--traffic", "none"]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    t_start = time.time()
    notes = []
    for counters in PMC_PASSES:
        outdir = tempfile.mkdtemp(prefix="locgpu_pmc_", dir="/tmp")
        try:
            left = budget_s - (time.time() - t_start)
            if left < 20:
                notes.append("%s: out of the %.0f s budget" % (counters[0], budget_s))
                break
            subprocess.run(["rocprofv3", "--pmc"] + counters + ["--output-format", "csv", "-d", outdir, "--"] + child, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", timeout=left, check=True)
            for f in glob.glob(os.path.join(outdir, "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("locgpu::", "").strip()
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        except Exception as e:  # a failed profile never fails the bench
            notes.append("%s pass failed: %s" % (counters[0], type(e).__name__))
        finally:
            shutil.rmtree(outdir, ignore_errors=True)
    out = {}
    for k, cs in agg.items():
        out[k] = {c: sum(v) for c, v in cs.items()}  # TOTALS over the child run: the caller divides by the steps the child ran
        out[k]["launches"] = max(len(v) for v in cs.values())
    out["_steps"] = steps + extra  # steps of the workload every hot kernel ran in the child (the fit kernels one more: the instrumented pass)
    return out, "live: rocprofv3 --pmc passes %s, %d timed + the profiled steps each%s" % (" | ".join(" ".join(c) for c in PMC_PASSES), steps, ("; " + "; ".join(notes)) if notes else "")


def measure_lane_efficiency_live(passthrough_args, budget_s=120.0, steps=1, pipeline=1, resident=True):
    """Lane efficiency of the search kernel's main loop on this workload, measured NOW by one child run of this script on the
    library's diagnostic build (LOCGPU_STAMP=1: the same kernel counting, per lane, the rounds the lane needed and the rounds its
    wave ran; its timing is meaningless and not used). Returns (useful / paid | None, note)."""
    import subprocess
    child = [os.path.realpath(sys.executable), os.path.abspath(__file__)] + list(passthrough_args) + ["--steps", str(steps), "--warmup", "0", "--pipeline", str(pipeline), "--no-cpu-baseline", "--traffic", "none"] + (["--resident"] if resident and "--resident" not in passthrough_args else [])
    try:
        out = subprocess.run(child, capture_output=True, text=True, timeout=budget_s, env=dict(os.environ, LOCGPU_STAMP="1"), cwd=ROOT)
        for ln in out.stdout.splitlines():
            if ln.startswith("{") and '"stamp"' in ln:
                st = json.loads(ln)["stamp"]
                if st["paid_rounds"] > 0:
                    return st["lane_rounds"] / st["paid_rounds"], "live: one step on the LOCGPU_STAMP=1 diagnostic build (rounds needed per lane / rounds its wave ran, summed over the step's search launches)"
        return None, "the diagnostic run printed no stamp"
    except Exception as e:
        return None, "diagnostic run failed: %s" % type(e).__name__


def launcher_command(n_ranks, port, argv, python=None):
    """argv of the job that runs this script as `n_ranks` ranks of one node: one process per GPU under torch.distributed.run,
    rendezvous on 127.0.0.1 (the container's hostname may not resolve). `argv` = this script's own arguments, passed through."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launcher_env(env=None):
    """Environment of the rank processes: the caller's, minus any rank variables of an enclosing job, plus what multi-process GPU
    work needs on this image (dmabuf IPC for RCCL)."""
    e = dict(os.environ if env is None else env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e["LOCGPU_BENCH_SELF_LAUNCHED"] = "1"
    return e


def visible_gpus():
    """HIP devices on this node, counted in a CHILD process through the library's own entry point (locgpu_device_count): the parent
    of a multi-rank run must never touch HIP itself."""
    import subprocess
    out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from loc_lib_amd import api; print(api.device_count())" % ROOT],
                         capture_output=True, text=True, timeout=300)
    if out.returncode != 0:
        raise SystemExit("bench.py: cannot count the GPUs (is liblocgpu.so built?): %s" % (out.stderr.strip()[-400:] or out.stdout.strip()[-400:]))
    return int(out.stdout.strip().splitlines()[-1])


def self_launch(n_ranks, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE: start the N ranks (torch.distributed.run, one per GPU), pass rank 0's
    JSON line through to stdout, everything else to stderr, and exit with the job's status. This process imports neither torch nor
    the library and never initialises a GPU."""
    import socket
    import subprocess
    have = visible_gpus()
    if n_ranks > have:
        raise SystemExit("bench.py: --gpus %d but this node has %d GPU(s); refusing to run fewer ranks than asked for" % (n_ranks, have))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = launcher_command(n_ranks, port, argv)
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (n_ranks, " ".join(cmd)))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=launcher_env(), cwd=ROOT)
    line_out = None
    for ln in proc.stdout:
        t = ln.strip()
        is_line = False
        if t.startswith("{") and '"metric"' in t:
            try:
                json.loads(t)
                is_line = True
            except ValueError:
                pass
        if is_line:
            line_out = t
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc != 0:
        raise SystemExit("bench.py: the %d-rank job failed with status %d (its output is above)" % (n_ranks, rc))
    if line_out is None:
        raise SystemExit("bench.py: the %d-rank job ended without a result line" % n_ranks)
    print(line_out, flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=80)  # the default timed region is ≥ 2 s
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scans-per-gpu", type=int, default=256)
    ap.add_argument("--method", choices=["p2plane", "p2line", "p2p", "ndt"], default="p2plane",
                    help="matcher to time; the headline metric is p2plane (others are reported for DESIGN.md tables)")
    ap.add_argument("--ndt-voxel", type=float, default=1.0, help="--method ndt: NdtOptions::voxel_size_ (slam.yaml's localisation node: 1.2)")
    ap.add_argument("--ndt-nearby", choices=["nearby6", "center"], default="nearby6", help="--method ndt: NdtOptions::nearby_type_")
    ap.add_argument("--map-points", type=int, default=10_000_000)
    ap.add_argument("--search", choices=["tree", "tree_exact", "grid"], default="tree",
                    help="tree = the reference's default alpha=0.1 approximate KD-tree search (headline); tree_exact / grid = "
                         "SetEnableANN(false) semantics through the tree or through the exact cell grid")
    ap.add_argument("--resident", action="store_true",
                    help="secondary number: the scans are uploaded once before the timed region (no per-step H2D)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="strong (default with several GPUs): --total-scans in all, sharded over the ranks with the per-iteration RCCL "
                         "all-reduce (BASELINE configs[3]); weak (default with one GPU): --scans-per-gpu scans on every rank, no collective")
    ap.add_argument("--pipeline", type=int, choices=[0, 1, 2, 3], default=0,
                    help="alignments in flight (the library has three compute streams): k = step i+k-1 is begun before step i is ended; "
                         "0 (default) = 3. Kernel launch durations are NOT taken from the timed region (launches of several batches "
                         "overlap there): they come from --extra-steps further steps of the same workload run one at a time behind it")
    ap.add_argument("--extra-steps", type=int, default=4, help="untimed steps behind the timed region, one alignment at a time with HIP events around every stage")
    ap.add_argument("--total-scans", type=int, default=256)
    ap.add_argument("--pool-slots", type=int, default=-1,
                    help="run the steps through the library's open-scan pool (locgpu_pool: one launch sequence per iteration over the open scans of "
                         "all steps begun) with this many scan slots; 0 = plain batches, three alignments in flight; -1 (default) = 256 slots "
                         "whenever a rank's step holds fewer than 256 scans, plain batches otherwise")
    ap.add_argument("--pool-chunk", type=int, default=0, help="pool: iterations between two looks at the flags (0 = the library's default)")
    ap.add_argument("--pool-lanes", type=int, default=2,
                    help="pool: split the slots over this many pools on different streams of the context and deal the steps to them: the launches of "
                         "one fill the tails of the other's (the plain-batch mode's alignments in flight do the same)")
    ap.add_argument("--traffic", choices=["live", "profiles", "none"], default="live",
                    help="roofline counters: live = rocprofv3 --pmc child runs + one diagnostic-build child run now (1 GPU only), profiles = newest committed traffic collection")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not launched as a rank of a job: bring the N ranks up from here (before torch or HIP is touched) and relay rank 0's line
        return self_launch(args.gpus, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:  # never a silent run on fewer (or more) ranks than the line will claim
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.scaling is None:
        args.scaling = "strong" if world > 1 else "weak"

    import torch  # before the library: torch brings its own HIP runtime and must be the first to load one in this process
    n_dev = torch.cuda.device_count()  # counts without initialising the GPU
    if local_rank >= n_dev:
        raise SystemExit("bench.py: rank %d (local rank %d) has no GPU: this node has %d" % (rank, local_rank, n_dev))
    dist = None
    if world > 1 or ("WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ):  # launched by torch.distributed.run (even with one rank)
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", init_method="env://", device_id=torch.device("cuda", local_rank))

    from loc_lib_amd import api, multi_gpu, synth

    strong = args.scaling == "strong"
    ctx = api.Context(local_rank)  # fails loudly without a GPU / without liblocgpu.so
    # The native RCCL communicator (inside liblocgpu.so) is only needed where the path has an exchange step: the strong-scaling
    # mode's per-iteration all-reduce (and its one tree build per node, broadcast over xGMI). Weak scaling has no collective on
    # the data path: every rank ingests the map itself (0.4 s with the level-parallel host build).
    use_comm = strong
    if use_comm:
        multi_gpu.init_comm(ctx, dist)  # one rank when not launched by torchrun

    # ---- inputs (untimed): map, tree ingest — resident in HBM before the timed region
    t0 = time.time()
    map_xyz = synth.make_map(args.map_points) if (rank == 0 or not use_comm or args.method == "ndt") else None
    t_map = time.time() - t0
    t0 = time.time()
    if use_comm:
        ctx.icp_set_target_bcast(map_xyz, root=0)  # ONE host tree build per node, packed tree broadcast over xGMI
    else:
        ctx.icp_set_target(map_xyz)
    t_ingest = time.time() - t0
    tinfo = ctx.icp_target_info()
    if strong:
        n_total = args.total_scans
        lo, hi = multi_gpu.shard_range(n_total, rank, world)
        all_ids = [i % 256 for i in range(n_total)]
    else:
        n_total = args.scans_per_gpu
        lo, hi = 0, n_total
        all_ids = [(rank * n_total + i) % 256 for i in range(n_total)]
    scan_of = {sid: synth.make_scan(sid) for sid in sorted(set(all_ids[lo:hi]))}
    scans = [scan_of[sid] for sid in all_ids[lo:hi]]          # the scans THIS rank holds
    inits = np.stack([synth.make_pose(sid)[1] for sid in all_ids])   # poses of the whole batch (all of it on every rank when strong)
    truth = np.stack([synth.make_pose(sid)[0] for sid in all_ids])
    B_local = len(scans)
    pts_per_scan = len(scans[0])

    method = dict(p2plane=api.P2PLANE, p2line=api.P2LINE, p2p=api.P2P, ndt=-1)[args.method]
    opts = api.icp_opts(method=max(method, 0))  # every other field = reference default
    if args.search == "grid":
        opts.search_mode = api.SEARCH_GRID_EXACT
    elif args.search == "tree_exact":
        opts.approximate = 0
    if method < 0:
        t0 = time.time()
        ctx.ndt_set_target(map_xyz, api.ndt_opts(voxel_size=args.ndt_voxel, nearby_type=(api.NEARBY6 if args.ndt_nearby == "nearby6" else api.CENTER)))  # other NdtOptions: defaults (DIRECT_NDT)
        t_ingest = time.time() - t0

    def new_batch():
        return ctx.batch(scans, first=lo, n_total=n_total) if strong else ctx.batch(scans)

    # The open-scan pool (locgpu_pool, scan_pool.hip): by default whenever a rank's step is small — 32 scans per rank of configs[3] on
    # eight GPUs — with room for about 256 scans of this rank: the steps begun share one launch sequence per Gauss–Newton iteration.
    pool_slots = args.pool_slots
    if pool_slots < 0:
        pool_slots = n_total * (-(-256 // max(B_local, 1))) if (0 < B_local < 256 and args.search != "grid" and not args.resident) else 0
    use_pool = pool_slots > 0
    if use_pool and (args.search == "grid" or args.resident or pool_slots < n_total):
        raise SystemExit("bench.py: --pool-slots needs room for one step (%d scans) and excludes --search grid and --resident" % n_total)
    depth = args.pipeline if args.pipeline else (max(1, pool_slots // n_total) if use_pool else 3)
    # plain batches — resident: `depth` batches hold the same scans; streaming: one more, so that the copy for step g+1 never lands
    # in a batch that an alignment in flight (steps g, g-1) is reading. Pool: one plain batch for the instrumented pass.
    bufs = [new_batch() for _ in range(1 if use_pool else (depth if args.resident else depth + 1))]
    scans_c = api.MarshalledScans(scans) if scans else []  # the (pointer, count) arrays a C caller of locgpu_batch_upload_async already holds
    lanes = max(1, min(args.pool_lanes, 3)) if use_pool else 0
    while lanes > 1 and 2 * (pool_slots // lanes) < n_total:  # a lane's arena (slots + as many prefetch regions) must hold one step
        lanes -= 1
    pools = [api.Pool(ctx, slots=pool_slots // lanes, max_points=max(pts_per_scan, 1), scans_per_job=n_total, chunk=args.pool_chunk, opts=opts, ndt=(method < 0))
             for _ in range(lanes)]

    def align_batch(b):
        return ctx.ndt_align_batch(b, inits) if method < 0 else ctx.icp_align_batch(b, inits, opts)

    g_step = [0]  # steps begun since the start of the process (selects the batch)
    debug_times = [] if os.environ.get("LOCGPU_BENCH_DEBUG") else None  # host seconds per step: (upload_async call, align begin call)

    def begin_step():
        """Start one pass of the hot path over one batch. Streaming (default): begin the alignment of the batch whose copy was
        started one step earlier, then start the host → HBM copy of the NEXT step's scans (returns at once: a worker packs into
        pinned slots, a copy stream moves them). Every step issues exactly one upload and one alignment."""
        g = g_step[0]
        g_step[0] += 1
        b = bufs[g % len(bufs)]
        t_a = time.perf_counter()
        if method < 0:
            ctx.ndt_align_batch_begin(b, inits)
        else:
            ctx.icp_align_batch_begin(b, inits, opts)
        t_b = time.perf_counter()
        # the alignment first: the GPU starts on this step while the host hands the next step's scans to the uploader
        if not args.resident:
            bufs[(g + 1) % len(bufs)].upload_async(scans_c)
        if debug_times is not None:
            debug_times.append((time.perf_counter() - t_b, t_b - t_a))
        return b

    def run_steps(n, pools=pools):
        """Exactly n steps, begun and ended in here; at most `depth` alignments in flight. Pool: a step = one job — its scans are
        submitted (the copy into free source regions starts at once and runs beside the pool's iterations) and collected by ticket."""
        inflight, begun, res = [], 0, None
        if use_pool:
            # a step is submitted as soon as the pool has free source regions for it (its copy then runs ahead of the slots coming free,
            # scan by scan: a job stays open until its slowest scan is done) and collected in order once it is done; at most 4 x depth
            # steps are outstanding.
            # Every decision depends on the flags all ranks see, so the ranks of a sharded run make the same calls in the same order.
            while begun < n or inflight:
                while begun < n and len(inflight) < 4 * depth:
                    p = max(pools, key=lambda q: q.info()["free_regions"])  # the emptier lane
                    if inflight and p.info()["free_regions"] < n_total:
                        break
                    inflight.append((p, p.submit(scans_c, inits, first=lo, n_total=n_total)))
                    begun += 1
                p, t = inflight[0]
                if p.done(t):
                    res = p.wait(t)
                    inflight.pop(0)
                else:
                    for q in pools:
                        q.step(True)
            return res
        while begun < n or inflight:
            while begun < n and len(inflight) < depth:
                inflight.append(begin_step())
                begun += 1
            res = ctx.align_batch_end(inflight.pop(0))
        return res

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()  # device-wide: covers the library's compute and copy streams

    # ---- visit counts of exactly this workload (separate instrumented pass, untimed) → algorithmic bytes
    ctx.visit_count_enable(True)
    out_poses, stats = align_batch(bufs[0])
    vc = ctx.visit_count_read(reset=True)
    ctx.visit_count_enable(False)

    if not args.resident and not use_pool:  # the copy for the first step (every later one is started by the step before it)
        bufs[g_step[0] % len(bufs)].upload_async(scans_c)
    run_steps(args.warmup)

    # ---- timed region: exactly `steps` steps, `depth` alignments in flight, no instrumentation inside it
    stamp_build = bool(os.environ.get("LOCGPU_STAMP"))
    if stamp_build:
        ctx.search_stats_read(reset=True)  # diagnostic build (a child of measure_lane_efficiency_live): switch the counters on
    barrier()
    t0 = time.perf_counter()
    out_poses, stats = run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if debug_times is not None:
        sys.stderr.write("host ms per step (upload_async, begin): %s\n" % [(round(1e3 * a, 2), round(1e3 * b, 2)) for a, b in debug_times[-args.steps:]])
    stamp = ctx.search_stats_read(reset=True) if stamp_build else None
    for b in bufs:
        b.upload_wait()
    # ---- kernel durations: `extra` further steps of the same workload, ONE alignment at a time (launches of different batches
    # overlap in the timed region), HIP events on the library's stream around every stage of every iteration
    n_extra = max(1, args.extra_steps, -(-4 * pool_slots // max(n_total, 1)) if use_pool else 0)  # pool: four pools' worth, so that most launches profiled are steady-state ones
    ctx.profile_read(reset=True)
    ctx.profile_enable(1)
    if use_pool:
        run_steps(n_extra, pools[:1])  # ONE lane: with two, an event interval of one lane would contain the other's kernels
    else:
        for _ in range(n_extra):
            align_batch(bufs[0])
    prof = ctx.profile_read(reset=True)
    ctx.profile_enable(False)
    stage_src = ("HIP events around every stage of %d further steps of the same workload through one lane of the pool, right behind the timed region" % n_extra) if use_pool else \
        "HIP events around every stage of %d further steps of the same workload, one alignment at a time, right behind the timed region" % n_extra

    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    total_scans = (n_total if strong else world * n_total) * args.steps
    value = total_scans / dt
    own = slice(lo, hi)
    gn_iters = sum(s["iterations"] for s in stats[own])   # scan-iterations THIS rank computed per step
    err_t = float(np.median(np.linalg.norm(out_poses[:, 4:] - truth[:, 4:], axis=1)))

    if rank == 0:
        # roofline of the dominant kernel over the timed region (all launches, partially idle ones included)
        k = 1 if args.method == "p2p" else 5
        q = vc["queries"] if method >= 0 else gn_iters * pts_per_scan
        search_bytes = q * 12 + vc["nodes"] * 16 + vc["leaves"] * 12 + q * 4 * k  # N_q·(12 + 16·V̄_n + 12·V̄_l + 4k)
        if args.search == "grid":
            # SURVEY §8(d) exact/grid formula: 16·N_cand + N_q·(12 + 4k); N_cand (distinct leaves in the cells any query's final
            # search block touches) is bounded below by the leaves the queries actually return: use k·q/4 as a conservative stand-in
            search_bytes = q * (16 + 4 * k) + 16 * (k * q // 4)
        accum_bytes = q * (16 + 4 * k + 16 * k) + gn_iters * 29 * 8      # src + indices + 5 gathered leaves; partial sums negligible
        if method < 0:
            nv = ctx.ndt_target_info()["num_voxels"]
            accum_bytes = q * (16 + (7 if args.ndt_nearby == "nearby6" else 1) * 16) + (prof["accum_n"] or 1) * nv * 128  # src + one 16-byte slot probe per voxel looked up; every voxel record (128 B) once per launch
        t_search = prof["search_ms"] * prof["search_n"] / n_extra      # ms per step (profiled steps)
        t_accum = prof["accum_ms"] * prof["accum_n"] / n_extra
        t_solve = prof["solve_ms"] * prof["solve_n"] / n_extra
        search_kernels = (("grid_bin_count_kernel", "grid_bin_scatter_kernel", "grid_tile_search_kernel", "icp_search_walk_list_kernel", "icp_search_redo_kernel")
                          if args.search == "grid" else ("icp_search_walk_kernel", "icp_search_walk_list_kernel", "icp_search_redo_kernel"))
        accum_kernels = ("ndt_accum_kernel",) if method < 0 else ("icp_%s_accum_kernel" % dict(p2plane="plane", p2line="line", p2p="point")[args.method], "icp_plane_refit_kernel")
        if t_search >= t_accum:
            kname, kset, kbytes, kt, kn, kavg = ("grid_tile_search_kernel(+binning+leftover walk+redo)" if args.search == "grid" else "icp_search_walk_kernel(+deep pass+redo)"), search_kernels, search_bytes, t_search, prof["search_n"], prof["search_ms"]
        else:
            kname, kset, kbytes, kt, kn, kavg = accum_kernels[0], accum_kernels, accum_bytes, t_accum, prof["accum_n"], prof["accum_ms"]
        launches_per_step = kn / n_extra
        # ---- hardware counters of the same workload: HBM bytes, VALU / FP64 instruction counts (rocprofv3 --pmc child runs), lane
        # efficiency of the search loop (one child run on the diagnostic build)
        counters, counters_note, lane_eff, lane_note = {}, "not collected", None, "not collected"
        if args.traffic == "live" and world == 1 and dist is None:
            passthrough = ["--scans-per-gpu", str(args.scans_per_gpu), "--map-points", str(args.map_points), "--method", args.method,
                           "--search", args.search, "--scaling", args.scaling, "--total-scans", str(args.total_scans), "--pool-slots", str(pool_slots),
                           "--pool-chunk", str(args.pool_chunk), "--pool-lanes", str(args.pool_lanes), "--ndt-voxel", str(args.ndt_voxel), "--ndt-nearby", args.ndt_nearby] + (["--resident"] if args.resident else [])
            counters, counters_note = measure_counters_live(passthrough, steps=(2 * depth if use_pool else 2), pipeline=(0 if use_pool else 1), extra=n_extra)
            if method >= 0 and args.search != "grid":
                lane_eff, lane_note = measure_lane_efficiency_live(passthrough, steps=(depth if use_pool else 1), pipeline=(0 if use_pool else 1), resident=not use_pool)

        def stage_counter(kernels, name):
            """The counter summed over the stage's kernels, PER LAUNCH of the stage as this process ran it: the child's total ÷ the steps
            the child ran (the same scans and iterations: the same work per step however the launches cut it) ÷ this process's launches
            per step. The fit kernels ran once more in the child (its instrumented pass uses them too)."""
            vals = [counters[kk][name] for kk in counters if kk != "_steps" for want in kernels if kk.startswith(want) and name in counters[kk]]
            if not vals or not counters.get("_steps"):
                return None
            child_steps = counters["_steps"] + (1 if kernels is accum_kernels else 0)
            lps = (prof["accum_n"] if kernels is accum_kernels else prof["search_n"]) / n_extra
            return sum(vals) / child_steps / max(lps, 1e-9)

        def hbm_bytes(kernels):
            f, w = stage_counter(kernels, "FETCH_SIZE"), stage_counter(kernels, "WRITE_SIZE")
            return int((2.0 * f + w) * 1024) if f is not None and w is not None else None  # both count KiB; gfx950 FETCH_SIZE counts half the bytes of wide reads

        # 256 CUs x 4 SIMD-32 units at the nominal 2.4 GHz. /opt/skills/guides/MI355X_MICROARCH.md:53-54 ("A wave (64 lanes) ... issues each
        # VALU instruction over 2 cycles (32 lanes/cycle x 2)"), :473 (`v_fma_f32` (wave64): 2 cyc throughput; 4 only for one wave alone
        # on its SIMD) and :41 (peak FP32 vector 157.3 TF = 1024 SIMDs x 32 lanes x 2 flop x 2.4 GHz): a 32-bit wave64 VALU instruction
        # costs its SIMD 2 issue cycles; an FP64 one 4 (78.6 TF FP64 vector = half the FP32 rate).
        SIMDS, CLOCK_HZ, CYC_VALU32, CYC_FP64 = 1024, 2.4e9, 2.0, 4.0
        FP64_COUNTERS = ["SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64"]

        def issue_frac(kernels, avg_ms):
            """(all VALU instructions, FP64 ones among them, issue cycles they cost / issue cycles the chip has in avg_ms)."""
            v = stage_counter(kernels, "SQ_INSTS_VALU")
            f = [stage_counter(kernels, nm) for nm in FP64_COUNTERS]
            if v is None or avg_ms <= 0:
                return None, None, None
            f64 = sum(x for x in f if x is not None)
            cycles = (v - f64) * CYC_VALU32 + f64 * CYC_FP64
            return v, f64, cycles / (SIMDS * CLOCK_HZ * avg_ms / 1e3)

        traffic, traffic_note = hbm_bytes(kset), counters_note
        if traffic is None and args.traffic != "none":
            t2 = load_traffic(kname, args.scans_per_gpu, args.map_points, args.method)
            if t2 is not None:
                traffic, traffic_note = t2, "copied from the newest committed profiles/*traffic*.json of this workload (" + counters_note + ")"
        nominal_gbs = (kbytes / 1e9) / (kt / 1e3) if kt > 0 else 0.0
        valu_n, valu_f64, valu_frac = issue_frac(kset, kavg)
        ISSUE_PEAK = SIMDS * CLOCK_HZ / CYC_VALU32 / 1e9  # G 32-bit wave64 VALU instructions per second the chip can issue at the nominal clock
        is_search = kset is search_kernels
        # The bound that binds. Both hot kernels are bound by vector-instruction issue, not by HBM (the tree traversal is a
        # cache-resident pointer chase: ~91 % of its node loads hit L1; its HBM traffic is a few per cent of the roofline), so
        # `frac` is the VALU issue fraction — never above 1 by construction. The byte model of SURVEY 8(d) is kept beside it as
        # nominal_bytes_frac (an EFFECTIVE rate: it prices every node visit as HBM bytes and can pass 1); hbm_frac is what the PMC
        # counters say actually left HBM.
        roofline = dict(bound="valu_issue", kernel=kname,
                        achieved=(round(valu_n / (kavg / 1e3) / 1e9, 2) if valu_n and kavg > 0 else None), peak=round(ISSUE_PEAK, 1), unit="G wave64-VALU-inst/s",
                        frac=(round(valu_frac, 4) if valu_frac else None),
                        clock_note="peak = 1024 SIMD-32 units x 2.4 GHz (nominal) / 2 cycles per 32-bit wave64 VALU instruction (MI355X_MICROARCH.md:53-54, :473; "
                                   "FP64 instructions priced at 4 cycles in frac); under this load the chip holds about 1.95 GHz, i.e. frac / 0.81 of what it can "
                                   "issue at the clock it runs at; the kernel's own instruction mix sustains 2.45 cycles per instruction in isolation "
                                   "(tools/ubench/issue_mix.hip)",
                        traffic=traffic,
                        hbm_frac=(round(traffic / (kavg / 1e3) / 1e9 / HBM_PEAK_GBS, 5) if traffic and kavg > 0 else None),
                        hbm_peak_gbs=HBM_PEAK_GBS, traffic_source=traffic_note,
                        nominal_bytes_frac=round(nominal_gbs / HBM_PEAK_GBS, 5), nominal_bytes_gbs=round(nominal_gbs, 2),
                        nominal_bytes_note="SURVEY 8(d)'s algorithmic bytes (one 16-byte load per tree node visited + source + index lists) over launch time and 8 TB/s: an "
                                           "effective rate — those loads are served by L1/L2 — not HBM traffic; it may pass 1. north_star's >= 50 %-of-HBM target is "
                                           "NOT met in HBM's own terms (see hbm_frac): the traversal does not need the bytes",
                        algorithmic_bytes_per_launch=int(kbytes / max(launches_per_step, 1)), avg_launch_ms=round(kavg, 5),
                        launches_per_step=launches_per_step,
                        valu_insts_per_launch=(int(valu_n) if valu_n else None), fp64_insts_per_launch=(int(valu_f64) if valu_f64 else None),
                        nodes_per_query=round(vc["nodes"] / max(q, 1), 2), leaves_per_query=round(vc["leaves"] / max(q, 1), 2))
        if method >= 0 and args.search != "grid" and is_search:
            # what the stage cannot avoid moving: every query's source point and index list once, and every tree slot any query of the
            # launch reads, once (counted by the instrumented pass, per launch)
            roofline["compulsory_bytes"] = int((q * (16 + 4 * k) + vc["distinct_slots"] * 8) / max(launches_per_step, 1))
            roofline["lane_efficiency"] = round(lane_eff, 4) if lane_eff else None
            roofline["lane_efficiency_source"] = lane_note
            roofline["note"] = ("valu_issue frac = (32-bit VALU instructions x 2 + FP64 x 4 cycles) / (SIMDs x launch time) at the nominal clock: the kernel is "
                                "LATENCY-bound (dependent node loads), well below its issue roofline; lane_efficiency = main-loop rounds the "
                                "lanes need / rounds their waves run (a wave runs until its slowest lane is done)")
        # which BASELINE.json configuration the arguments amount to
        if strong:
            cfg_name = "BASELINE configs[3] (%d scans in all, sharded over %d GPU(s), RCCL all-reduce)" % (n_total, world)
        elif args.map_points == 10_000_000:
            cfg_name = "BASELINE configs[2]"
        elif args.map_points == 1_000_000:
            cfg_name = "BASELINE configs[1]"
        else:
            cfg_name = "%d-pt map (not a BASELINE configuration)" % args.map_points
        mode = "scans resident in HBM before the timed region (secondary number)" if args.resident else \
               ("includes scan H2D: every step's scans are copied host->HBM into free pool slots (pinned staging, copy stream, beside the pool's iterations)" if use_pool else
                "includes scan H2D: every step aligns a batch copied host->HBM for it (pinned double buffer, copy stream, overlapped with the previous step)")
        shard = ("strong scaling: %d scans in all sharded over %d rank(s), per-iteration RCCL all-reduce of the per-scan normal equations" % (n_total, world)) if strong \
            else "scans sharded by rank, no collective"
        line = dict(metric="scans/sec (64x1800-pt scan vs 10M-pt map) + ICP iter ms", value=round(value, 3), unit="scans/s",
                    n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * dt / args.steps, 4),
                    higher_is_better=True, scaling=args.scaling, vs_baseline=None, dtype="f64", search_dtype="f32", data="synthetic",
                    config=dict(workload="%s: %d scans/GPU x %d pts (64x1800, cityblock-v1) vs one %d-pt map, "
                                         "%s, reference defaults (alpha=0.1 KD-tree ANN, eps=1e-2, max 20 iters), %s; %s; %s"
                                         % (cfg_name, B_local, pts_per_scan, args.map_points, ("direct NDT (voxel %g, %s)" % (args.ndt_voxel, args.ndt_nearby.upper())) if method < 0 else args.method.upper() + " ICP", shard, mode,
                                            ("open-scan pool of %d slots in %d lane(s), about %d steps in flight" % (pool_slots, lanes, depth)) if use_pool else "%d alignment(s) in flight" % depth),
                                scans_per_gpu=B_local, map_points=args.map_points, scan_h2d_in_timed_region=not args.resident, pipeline_depth=depth,
                                search_mode=dict(tree="tree_faithful_ann", tree_exact="tree_faithful_exact", grid="grid_exact")[args.search], tree_depth=tinfo["depth"],
                                tree_bytes=tinfo["bytes"]),
                    icp_iter_ms=round((t_search + t_accum + t_solve) / max(prof["search_n"] / args.steps, 1), 5),
                    icp_iter_ms_per_scan=round((t_search + t_accum + t_solve) / max(gn_iters, 1), 6),  # kernel ms of one step ÷ scan-iterations of one step
                    gn_iterations_per_scan=round(gn_iters / max(B_local, 1), 2),
                    kernel_ms_per_step=dict(search=round(t_search, 4), fit_accumulate=round(t_accum, 4), solve=round(t_solve, 4)),
                    kernel_ms_source=stage_src,
                    h2d_bytes_per_step=(0 if args.resident else B_local * pts_per_scan * 16),
                    median_translation_error_to_truth_m=round(err_t, 4),
                    setup_s=dict(map_gen=round(t_map, 2), tree_ingest=round(t_ingest, 2)),
                    roofline=roofline)
        if use_pool:
            line["pool"] = dict(slots=pool_slots, lanes=lanes, steps_in_flight=depth, chunk=(args.pool_chunk or 4),
                                iterations=sum(q.info()["iterations"] for q in pools), scan_iterations=sum(q.info()["scan_iterations"] for q in pools))
        line["scans_per_rank"] = B_local
        line["rccl_ranks"] = (ctx.comm_info()[1] if use_comm else (world if dist is not None else 1))  # ranks RCCL joined: the library's communicator (strong), torch's process group (weak)
        line["rccl_use"] = ("per-iteration all-reduce + tree broadcast inside liblocgpu.so (locgpu_comm_info)" if use_comm else
                            ("barrier and max-over-ranks only (torch.distributed nccl)" if dist is not None else "none (one rank, no process group)"))
        if stamp is not None:
            line["stamp"] = dict(lane_rounds=stamp["walked"], paid_rounds=stamp["replayed"])
        if method >= 0 and t_accum > 0:
            # the second kernel (fit + accumulate): algorithmic bytes of SURVEY §8(d) against HBM, its measured HBM traffic, and — what
            # actually bounds it — its VALU and FP64 instruction counts against the issue rate (32-bit wave64 instruction = 2 cycles of its SIMD-32, FP64 = 4)
            a_gbs = (accum_bytes / 1e9) / (t_accum / 1e3)
            k2_traffic = hbm_bytes(accum_kernels)
            k2_avg = prof["accum_ms"]
            v_n, f64_n, v_frac = issue_frac(accum_kernels, k2_avg)
            f64_frac = (f64_n * CYC_FP64 / (SIMDS * CLOCK_HZ * k2_avg / 1e3)) if f64_n and k2_avg > 0 else None
            line["roofline_k2"] = dict(bound="valu_issue (FP64 instructions at 4 cycles, the others at 2)", kernel=accum_kernels[0],
                                       achieved=(round(v_n / (k2_avg / 1e3) / 1e9, 2) if v_n and k2_avg > 0 else None), peak=round(SIMDS * CLOCK_HZ / CYC_FP64 / 1e9, 1),
                                       unit="G wave64-VALU-inst/s (peak: all-FP64 stream)", frac=(round(v_frac, 4) if v_frac else None),
                                       fp64_issue_frac=(round(f64_frac, 4) if f64_frac else None),
                                       fp64_insts_per_launch=(int(f64_n) if f64_n else None), valu_insts_per_launch=(int(v_n) if v_n else None),
                                       ms_per_step=round(t_accum, 4), avg_launch_ms=round(k2_avg, 5), traffic=k2_traffic,
                                       hbm_frac=(round(k2_traffic / (k2_avg / 1e3) / 1e9 / HBM_PEAK_GBS, 5) if k2_traffic and k2_avg > 0 else None),
                                       nominal_bytes_frac=round(a_gbs / HBM_PEAK_GBS, 5),
                                       note="frac at the nominal 2.4 GHz (the chip holds about 1.95 GHz under this load); about two thirds of the instructions are the plane fit; " + stage_src)
        if world == 1 and not args.no_cpu_baseline and args.search == "tree":
            cb, cpu_poses = cpu_baseline(map_xyz, scans, inits, args.cpu_seconds, args.method,
                                         dict(voxel_size=args.ndt_voxel, nearby_type=(1 if args.ndt_nearby == "nearby6" else 0)))
            n = len(cpu_poses)
            d = np.linalg.norm(np.stack(cpu_poses)[:, 4:] - out_poses[:n, 4:], axis=1)
            cb["max_pose_delta_gpu_vs_cpu_m"] = float(d.max())
            cb["gpu_over_cpu"] = round(value / cb["value"], 1)
            line["cpu_baseline"] = cb
        print(json.dumps(line), flush=True)

    for q in pools:
        q.close()
    for b in bufs:
        b.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
