#!/usr/bin/env python3
"""Benchmark of the PatchMatch-MVS sweep on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W           (N > 1 without a launcher: starts its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--workload planesweep` benchmarks the sibling path instead (BASELINE config 2: 8 views of
1280x720, plane-sweep stereo, 64 planes, 5x5 NCC, 6 neighbours; one step = all 8 views).

A "step" is one complete PatchMatch sweep (init, 8 x (2 propagation + 8 refinement) cost
evaluations, confidence) over this rank's batch of reference views, with every image already
resident in HBM.  Workload at N=1: BASELINE config 3 -- 16 views of 1920x1080, 7x7 NCC,
4 sources, 8 iterations.  For N>1 the scene is FIXED (BASELINE config 4: 32 views of 1920x1080;
north_star's scaling target is 8 GPUs vs 1 on a 32-view scene): rank r sweeps its contiguous block
of ceil(32/N) views (all images replicated on every GPU because source sets cross shard boundaries)
and the per-view maps (depth, normal, confidence: 20 B/pixel) are all-gathered over RCCL inside the
timed step, on a second stream: the exchange of step k runs under the sweep of step k+1 (two buffer
sets; from 8 views per rank on, a step is two batches and the first batch's exchange also overlaps the
second batch's sweep) -- "scaling": "strong".  `first_step_ms` is ONE step from an idle pipeline with
its exchange exposed (what a single reconstruct pays); `--gather-per-iteration` issues north_star's
exchange after every iteration instead.  `--scaling weak` restores 16 views per GPU of a 16*N-view scene.

`--mode fast` (default) times the tolerance arithmetic (AMVS_MODE_FAST), `--mode exact` the
bit-exact one; both are parity-tested at this very size and launch shape
(tests/test_hip_fullsize_parity.py).

Prints ONE JSON line (rank 0) with the throughput in Mpixel-hypotheses/s, the HBM-roofline
figure of the dominant kernel (pm_step[_fast]_kernel<7,4>, timed with HIP events on its own stream
through amvs_get_timing), -- at N=1 -- the CPU oracle (same arithmetic mode) timed on the host
cores on a bounded sample of the same workload, and a `planesweep` sub-record (BASELINE config 2).
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def kernel_source_hash(kernel_key):
    """sha1 over the git blob hashes of the source files a sweep kernel is built from: stored with
    every profiles/traffic.json entry (tools/profile_summary.py) so that counters of an older kernel
    are not replayed against a newer build."""
    import hashlib
    csrc = os.path.join(ROOT, "3d-reconstruction-tool_amd", "csrc")
    if "_fast_" in kernel_key:
        names = ("amvs_sweep_fast.hip" if kernel_key.startswith("plane_sweep") else "amvs_kernels_fast.hip", "amvs_fast_common.h")
    else:
        names = ("amvs_sweep_exact.hip" if kernel_key.startswith("plane_sweep") else "amvs_kernels.hip", "amvs_exact_common.h")
    h = hashlib.sha1()
    for name in names + ("amvs_kernel_common.h", "amvs_device.h", "amvs_kernels.h"):
        with open(os.path.join(csrc, name), "rb") as f:
            data = f.read()
        h.update(hashlib.sha1(b"blob %d\0" % len(data) + data).digest())
    return h.hexdigest()


def profiled_traffic(kernel_key, workload_key):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/traffic.json, written by tools/profile_summary.py; FETCH_SIZE and WRITE_SIZE in KiB,
    collected in separate passes).  FETCH_SIZE is corrected as MI355X_MICROARCH.md prescribes: it
    tallies every 128-byte L2 fill at 64 bytes, so it is doubled -- calibrated for THIS access pattern
    with tools/gather_rate.hip (one dword per 128 bytes moves as many bytes as one per 64 bytes, and
    TCC_MISS_sum x 64 B reproduces FETCH_SIZE).  None if no profile of this exact workload AND of the
    kernel sources in this tree (kernel_source_hash) is committed."""
    return profiled_traffic_and_source(kernel_key, workload_key)[0]


def profiled_traffic_and_source(kernel_key, workload_key):
    """(bytes per launch, the committed summary they come from) -- the bench line names the file
    (`roofline.traffic_source`): the figure is REPLAYED from a builder-side rocprofv3 run of the same kernel
    sources on the same workload, not measured inside the driver's run."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        e = t.get(kernel_key)
        if e and e.get("workload") == workload_key and e.get("source_hash") == kernel_source_hash(kernel_key):
            return (int((e["fetch_kib"] * e.get("fetch_correction", 1.0) + e["write_kib"]) * 1024),
                    "profiles/%s.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x 2; replayed, hash-checked "
                    "against the kernel sources)" % e.get("source", "?"))
    except (OSError, ValueError, KeyError):
        pass
    return None, None


# Issue cost of one VALU wave-instruction per SIMD, measured on MI355X at 8 waves per SIMD with
# tools/valu_rate.hip (profiles/r04_logs/valu_rate.log): v_add_f32 / v_fma_f32 0.95-1.16 ns, v_add_f32_dpp
# wave_shl:1 (the row sums' cross-lane adds) 3.26 ns.  1 024 SIMDs per MI355X.  These are the BUILDER'S
# microbenchmark figures (a reading aid, not an independent bound): the line names their source.
VALU_PLAIN_NS, VALU_DPP_NS, N_SIMDS = 0.95, 3.26, 1024
ISSUE_NS_SOURCE = "tools/valu_rate.hip on MI355X, 8 waves per SIMD (profiles/r04_logs/valu_rate.log)"


def valu_issue_roofline(kernel_key, workload_key, dpp_insts, launch_ms):
    """The VALU-ISSUE roofline of a kernel that is not memory-bound (the plane sweep: L2 hit rate 0.985, its
    sources never leave L2): wave-instructions per launch from the committed PMC pass (SQ_INSTS_VALU,
    profiles/traffic.json), the cross-lane DPP adds among them counted from the launch shape, each kind
    priced with its measured issue cost, spread over the chip's SIMDs -> the time the VALU pipes alone
    need, and the fraction of the measured launch it explains.  None without a matching profile."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            e = json.load(f).get(kernel_key)
        if not e or e.get("workload") != workload_key or e.get("source_hash") != kernel_source_hash(kernel_key):
            return None
        valu = float(e["sq_insts_valu"])
    except (OSError, ValueError, KeyError):
        return None
    plain = max(valu - dpp_insts, 0.0)
    floor_ms = (plain * VALU_PLAIN_NS + dpp_insts * VALU_DPP_NS) / N_SIMDS * 1e-6
    return {"bound": "valu-issue", "valu_wave_instructions_per_launch": int(valu), "of_which_dpp": int(dpp_insts),
            "issue_ns": {"plain": VALU_PLAIN_NS, "dpp": VALU_DPP_NS}, "issue_ns_source": ISSUE_NS_SOURCE, "simds": N_SIMDS,
            "floor_ms": round(floor_ms, 3), "frac": round(floor_ms / launch_ms, 3) if launch_ms > 0 else None}


def host_cores():
    """CPU threads this process may really use: the cgroup quota if there is one (a GPU box exposes
    every host core but grants a share), else the affinity mask, else os.cpu_count()."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return int(os.environ.get("AMVS_ORACLE_THREADS", "0")) or n


def cpu_baseline(scene, patch, sources, refs, depth_min, depth_max, iters, samples, mode):
    """Time the CPU oracle (oracle/amvs_oracle.c, OpenMP over the host cores) on a bounded
    sample: the given reference views of the workload with the full iteration schedule."""
    from oracle import oracle
    oracle.set_threads(host_cores())
    K = scene.camera.K.astype(np.float32)
    H, W = scene.grays[refs[0]].shape
    dt = 0.0
    for ref in refs:
        srcs = sources[ref]
        ctx = oracle.ViewContext(K, scene.grays[ref], scene.poses[ref].R, scene.poses[ref].t,
                                 [scene.grays[i] for i in srcs], [scene.poses[i].R for i in srcs],
                                 [scene.poses[i].t for i in srcs], patch, mode=mode)
        t0 = time.time()
        ctx.patchmatch(0, 0, depth_min, depth_max, 1, ref)            # init + confidence only
        t_fixed = time.time() - t0
        t0 = time.time()
        ctx.patchmatch(iters, samples, depth_min, depth_max, 1, ref)
        dt += time.time() - t0 - t_fixed
        ctx.close()
    n_hyp = len(refs) * H * W * iters * (2 + samples)
    return {"value": n_hyp / max(dt, 1e-9) / 1e6, "unit": "Mpx-hyp/s", "cores": oracle.num_threads(),
            "kind": "port",
            "sample": f"{len(refs)} of the views at {W}x{H}, {iters} iterations x (2+{samples}) evaluations = "
                      f"{n_hyp/1e6:.1f} Mpx-hyp in {dt:.1f} s (oracle/amvs_oracle.c, {mode} arithmetic, "
                      f"OpenMP on {oracle.num_threads()} threads)"}


def baseline_label(args, n_views, W, H, world):
    """"BASELINE config N: " only when the run IS that configuration of BASELINE.json (view count, image
    size, 7x7 NCC, 8 iterations x (2 + 8) hypotheses; config 5 additionally the fusion inside the step)."""
    std = (args.patch, args.iters, args.samples) == (7, 8, 8)
    if std and (n_views, W, H) == (16, 1920, 1080) and world == 1 and not args.fusion:
        return "BASELINE config 3: "
    if std and (n_views, W, H) == (32, 1920, 1080) and not args.fusion:
        return "BASELINE config 4 (the 32-view scene; BASELINE shards it over 4 GPUs, here over %d): " % world
    if std and (n_views, W, H) == (64, 3840, 2160) and args.fusion:
        return "BASELINE config 5 (BASELINE shards it over 8 GPUs, here over %d): " % world
    return "custom workload (no BASELINE config): "


def launch_ranks(n):
    """Run this script as n ranks of one node: n fresh child processes with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set (what `python -m torch.distributed.run --nproc-per-node n` would set), the
    same command line.  The parent never imports torch and never touches the GPU; it relays rank 0's
    stdout (the ONE JSON line), sends the other ranks' stdout to stderr and returns non-zero if any rank
    fails -- the remaining ranks (which would wait in a collective for ever) are then terminated by PID."""
    import socket
    import subprocess
    with socket.socket() as sk:                       # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {procs.index(p)} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in alive:
                    q.terminate()
                deadline = time.time() + 10
                for q in alive:
                    try:
                        q.wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--views-per-gpu", type=int, default=16)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--patch", type=int, default=7)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--samples", type=int, default=8)
    ap.add_argument("--tile-rows", type=int, default=0)
    ap.add_argument("--views-per-launch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batches", type=int, default=0,
                    help="sweeps per step (N>1: the all-gather of one batch overlaps the sweep of the next); "
                         "0 = 1 on one GPU, 2 otherwise")
    ap.add_argument("--workload", choices=["patchmatch", "planesweep"], default="patchmatch")
    ap.add_argument("--planes", type=int, default=64)
    ap.add_argument("--sweep-tile-rows", type=int, default=0, help="plane sweep: rows per strip (0 = automatic)")
    ap.add_argument("--mode", choices=["fast", "exact"], default="fast",
                    help="arithmetic of the sweep kernels (include/amvs.h AMVS_MODE_*)")
    ap.add_argument("--split-groups", type=int, default=0, help="split schedule: view groups (0 = automatic)")
    ap.add_argument("--split-lds", type=int, default=0, help="split schedule: unused LDS bytes per sampling workgroup (0 = automatic)")
    ap.add_argument("--split-rows", type=int, default=0, help="split schedule: rows per sampling strip (0 = automatic)")
    ap.add_argument("--schedule", choices=["auto", "view-major", "band-major", "split", "paired"], default="auto",
                    help="strip order of the sweep launches (amvs_pm_params.schedule)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N>1: strong = fixed --scene-views scene split over the ranks; weak = --views-per-gpu each")
    ap.add_argument("--scene-views", type=int, default=32, help="views of the fixed scene for N>1 (BASELINE config 4)")
    ap.add_argument("--no-planesweep", action="store_true", help="skip the plane-sweep sub-record")
    ap.add_argument("--no-subrecords", action="store_true",
                    help="skip the `exact` (configs 3 and 2 in the classes' default arithmetic) and `cli_defaults` sub-records")
    ap.add_argument("--fusion", action="store_true",
                    help="fuse + filter the gathered maps inside every timed step (BASELINE config 5)")
    ap.add_argument("--gather-per-iteration", action="store_true",
                    help="parity mode with north_star's exchange pattern: the sweep runs one iteration per call "
                         "(amvs_pm_params.first_iteration) and the ranks all-gather the depth / normal maps after EVERY "
                         "iteration (the gather of iteration i under the sweep of iteration i+1; the maps do not feed "
                         "back -- the reference has no view propagation -- so the results are those of the default run)")
    ap.add_argument("--as-rank-of", type=int, default=0,
                    help="one GPU, no exchange: sweep only the shard rank 0 of an N-rank run would sweep (of the "
                         "--scene-views scene): the per-rank sweep time of an N-GPU run measured on one GPU")
    ap.add_argument("--config5", action="store_true",
                    help="BASELINE config 5 preset: 64 views of 3840x2160 over the ranks, fusion inside the step")
    args = ap.parse_args()
    # `python bench.py --gpus N` without a launcher (no WORLD_SIZE): start the N ranks here, as fresh child
    # processes, BEFORE anything in this process touches torch or the GPU; this process only waits and
    # relays rank 0's JSON line
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("AMVS_BENCH_FORCE_EXCHANGE") != "1":
        sys.exit(launch_ranks(args.gpus))
    if args.config5:
        args.height, args.width, args.scene_views, args.fusion, args.no_planesweep = 2160, 3840, 64, True, True
        args.no_subrecords = True
        if args.gpus == 1:
            args.views_per_gpu = 64
    if args.workload == "planesweep":
        print(json.dumps(run_planesweep(args, args.steps, args.warmup, not args.no_cpu_baseline)), flush=True)
        return

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # AMVS_BENCH_BACKEND=gloo + AMVS_BENCH_ONE_DEVICE=1: rehearsal of the multi-rank path on a
    # single-GPU box (all ranks on cuda:0, all-gather staged through the host); never set by the
    # driver -- the real run is one rank per GPU over RCCL
    backend = os.environ.get("AMVS_BENCH_BACKEND", "nccl")
    if os.environ.get("AMVS_BENCH_ONE_DEVICE") == "1":
        local = 0
    # AMVS_BENCH_FORCE_EXCHANGE=1: run the N>1 code path (two buffer sets, comm stream, collectives, events)
    # on a ONE-rank process group -- how the RCCL calls are rehearsed on a one-GPU box
    force = os.environ.get("AMVS_BENCH_FORCE_EXCHANGE") == "1"
    if force:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if args.gpus > 1 or world > 1 or force:
        assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import amvs
    from amvs.engine import make_pm_params
    from amvs.parallel import shard
    from amvs.synthetic import make_scene

    H, W = args.height, args.width
    multi = world > 1 or force                       # the exchange is part of the step
    strong = multi and args.scaling == "strong"
    n_views = args.scene_views if (strong or args.as_rank_of > 0) else args.views_per_gpu * world
    vpg = -(-n_views // world)                      # views per GPU (ceil)
    # synthetic calibrated scene rendered on the GPU (data generation, outside the timed region)
    sc = make_scene(n_views, H, W, seed=1234, device=str(dev))
    ids = sorted(sc.poses)
    pm = amvs.PatchMatchMVS.__new__(amvs.PatchMatchMVS)
    sources = {r: pm._select_source_views(r, ids, sc.poses, k=4) for r in ids}
    mine = shard(n_views, rank, world) if args.as_rank_of <= 0 else shard(n_views, 0, args.as_rank_of)

    eng = amvs.Engine(H, W, n_views, sc.camera.K.astype(np.float32), device=local, mode=args.mode)
    stream = torch.cuda.Stream(device=dev)
    eng.set_stream(stream.cuda_stream)
    # 8-bit images, as every real input is (cvtColor(...).astype(float32)/255,
    # mvs_patchmatch.py:177): gray = code/255 in IEEE float32 through a host-built table
    lut = torch.from_numpy(np.arange(256, dtype=np.float32) / np.float32(255.0)).to(dev)
    for i in ids:
        codes = torch.round(torch.from_numpy(sc.grays[i]).to(dev) * 255.0).clamp(0, 255).long()
        g = lut[codes].contiguous()
        sc.grays[i] = g.cpu().numpy()
        eng.set_view_device(i, g.data_ptr(), sc.poses[i].R, sc.poses[i].t)
        torch.cuda.synchronize()
        eng.set_view_colors(i, np.ascontiguousarray(sc.colors[i]))     # for the device fusion (amvs_fuse_filter_views)
    params = make_pm_params(args.patch, args.iters, args.samples, sc.depth_min, sc.depth_max, args.tile_rows,
                            args.views_per_launch, schedule=args.schedule)
    eng.set_split_tuning(args.split_groups, args.split_rows, args.split_lds)
    n_loc = len(mine)
    if multi:
        assert n_views % world == 0, "the bench shards equal blocks: --scene-views must be a multiple of --gpus"
    # N>1: two sets of output maps used alternately, so that the all-gather of step k (which reads the
    # maps in place, no packed copy) can run under the sweep of step k+1
    nbuf = 2 if multi else 1
    depth = torch.empty((nbuf, n_loc, H, W), dtype=torch.float32, device=dev)
    normal = torch.empty((nbuf, n_loc, H, W, 3), dtype=torch.float32, device=dev)
    conf = torch.empty((nbuf, n_loc, H, W), dtype=torch.float32, device=dev)
    refs = list(mine)
    srcs = [sources[r] for r in refs]

    # N>1: the rank's views are swept in `nb` batches (two from 8 views per rank on); the RCCL
    # all-gathers of batch b (three collectives: depth, normal, confidence -- 20 B/pixel, straight from
    # the sweep's output arrays) run on their own stream while the next batch / the next step is swept,
    # so K timed steps expose one batch's exchange once, before the closing barrier.  `first_step_ms` in
    # the output line is ONE step from an idle pipeline with its exchange fully inside: what a single
    # PatchMatchMVS.reconstruct pays (its last group's gather is exposed).
    # (two batches from 8 views per rank on -- the library sweeps a batch in groups of 4 views, its best
    # launch shape, so batches of fewer than 4 views would cost more than their exposed exchange saves:
    # measured on one GPU 39.8 against 42.1 G px-hyp/s)
    nb = args.batches if args.batches > 0 else (2 if (multi and n_loc >= 8) else 1)
    nb = max(1, min(nb, n_loc))
    if args.gather_per_iteration:
        nb = 1                                  # a continuation call resumes the state of ONE batch
    bounds = [(b * n_loc) // nb for b in range(nb + 1)]
    comm_stream = torch.cuda.Stream(device=dev)
    gdev = dev if backend == "nccl" else torch.device("cpu")
    # gathered maps, rows laid out [batch][rank][row of the batch]: the rows a batch's exchange fills are ONE
    # contiguous block, so every exchange is one all_gather_into_tensor per map (no list of output views for
    # the backend to assemble through a scratch buffer); with one batch per step that IS view order
    full = [dict(d=torch.empty((world * n_loc, H * W), dtype=torch.float32, device=gdev),
                 n=torch.empty((world * n_loc, 3 * H * W), dtype=torch.float32, device=gdev),
                 c=torch.empty((world * n_loc, H * W), dtype=torch.float32, device=gdev)) for _ in range(nbuf)] \
        if multi else None
    perm = None
    if multi and nb > 1:
        rows_of = {}
        for b in range(nb):
            lo, hi = bounds[b], bounds[b + 1]
            for r in range(world):
                for i in range(lo, hi):
                    rows_of[r * n_loc + i] = world * lo + r * (hi - lo) + (i - lo)
        perm = torch.tensor([rows_of[v] for v in range(world * n_loc)], dtype=torch.long, device=gdev)

    def in_view_order(t):
        """(world, n_loc, width) view / copy of a gathered map in view order"""
        return (t if perm is None else t.index_select(0, perm)).view(world, n_loc, -1)
    fusion_inputs = None
    if args.fusion:
        # config 5: the fused, filtered cloud is part of the step (on rank 0, which like every rank
        # holds all maps after the gather)
        fusion_inputs = (None if backend != "gloo" or not multi else np.stack([sc.colors[r] for r in ids]),
                         np.linalg.inv(sc.camera.K), [(sc.poses[r].R, sc.poses[r].t) for r in ids])
    in_flight = []
    state = {"step": 0, "cloud": None, "fusion_s": 0.0}
    # event on comm_stream behind the collectives that READ output buffer set k: the sweep that
    # rewrites that set two steps later waits for it on its own stream (the all-gathers read the maps
    # in place, and ProcessGroupNCCL's work.wait() orders only the stream it is called on)
    gathered = [None] * nbuf

    def drain(keep_last_step):
        while len(in_flight) > (1 if keep_last_step else 0):
            for w in in_flight.pop(0):
                w.wait()

    def fuse(k):
        t_f = time.perf_counter()
        cols, K_inv, pose_list = fusion_inputs
        if not multi:
            torch.cuda.synchronize()
            out = eng.fuse_filter_views(ids, depth[k].data_ptr(), conf[k].data_ptr(), K_inv, pose_list, 3, True)
        elif backend == "nccl":
            fd, fc = in_view_order(full[k]["d"]), in_view_order(full[k]["c"])
            torch.cuda.synchronize()
            out = eng.fuse_filter_views(ids, fd.data_ptr(), fc.data_ptr(), K_inv, pose_list, 3, True)
        else:
            out = eng.fuse_filter(in_view_order(full[k]["d"]).numpy().reshape(n_views, H, W),
                                  in_view_order(full[k]["c"]).numpy().reshape(n_views, H, W), cols, K_inv, pose_list, 3, True)
        state["cloud"] = out
        state["fusion_s"] += time.perf_counter() - t_f

    def step():
        k = state["step"] % nbuf
        state["step"] += 1
        works = []
        if multi and gathered[k] is not None:
            stream.wait_event(gathered[k])
        for b in range(nb):
            lo, hi = bounds[b], bounds[b + 1]
            eng.patchmatch_device(refs[lo:hi], srcs[lo:hi], params, 42, depth[k, lo:hi].data_ptr(),
                                  normal[k, lo:hi].data_ptr(), conf[k, lo:hi].data_ptr())
            if multi:
                done = torch.cuda.Event()
                done.record(stream)
                with torch.cuda.stream(comm_stream):
                    comm_stream.wait_event(done)
                    for name, src_t in (("d", depth[k, lo:hi]), ("n", normal[k, lo:hi]), ("c", conf[k, lo:hi])):
                        inp = src_t.reshape(hi - lo, -1)
                        if backend != "nccl":
                            comm_stream.synchronize()
                            inp = inp.cpu()
                        # the batch's block of the gathered array: [rank][row of the batch], contiguous
                        works.append(dist.all_gather_into_tensor(full[k][name][world * lo: world * hi], inp, async_op=True))
                    if backend == "nccl":
                        for w in works:
                            w.wait()                     # comm_stream (the current stream) waits, not the host
                    ev = torch.cuda.Event()
                    ev.record(comm_stream)
                    gathered[k] = ev
        eng.sync()
        if multi:
            in_flight.append(works)
            # the buffers of this step are rewritten two steps from now: the step before this one must
            # be through; with the fusion inside the step, this step's own exchange as well
            drain(keep_last_step=not args.fusion)
        if args.fusion and rank == 0:
            fuse(k)

    # --gather-per-iteration: one call per iteration; intermediate maps (depth, normal) alternate between
    # two buffer sets, each gathered on the comm stream while the next iteration is swept
    if args.gather_per_iteration:
        it_params = [make_pm_params(args.patch, 1, args.samples, sc.depth_min, sc.depth_max, args.tile_rows,
                                    args.views_per_launch, schedule=args.schedule, first_iteration=it,
                                    confidence=it == args.iters - 1) for it in range(args.iters)]
        it_d = torch.empty((2, n_loc, H * W), dtype=torch.float32, device=dev)
        it_n = torch.empty((2, n_loc, 3 * H * W), dtype=torch.float32, device=dev)
        it_full = [dict(d=torch.empty((world * n_loc, H * W), dtype=torch.float32, device=gdev),
                        n=torch.empty((world * n_loc, 3 * H * W), dtype=torch.float32, device=gdev)) for _ in range(2)] \
            if multi else None
        it_gathered = [None, None]
        state["it_gathers"] = 0

    def step_per_iteration():
        k = state["step"] % nbuf
        state["step"] += 1
        works = []
        if multi and gathered[k] is not None:
            stream.wait_event(gathered[k])
        for it in range(args.iters):
            last = it == args.iters - 1
            j = it % 2
            if not last and it_gathered[j] is not None:
                stream.wait_event(it_gathered[j])         # the exchange that read this set two iterations ago
            outs = (depth[k], normal[k], conf[k]) if last else (it_d[j], it_n[j], conf[k])
            eng.patchmatch_device(refs, srcs, it_params[it], 42, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr())
            if not multi:
                continue
            done = torch.cuda.Event()
            done.record(stream)
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(done)
                if last:
                    pairs = [(full[k]["d"], depth[k].reshape(n_loc, -1)), (full[k]["n"], normal[k].reshape(n_loc, -1)),
                             (full[k]["c"], conf[k].reshape(n_loc, -1))]
                else:
                    pairs = [(it_full[j]["d"], it_d[j]), (it_full[j]["n"], it_n[j])]
                mine_works = []
                for out_t, inp in pairs:
                    if backend != "nccl":
                        comm_stream.synchronize()
                        inp = inp.cpu()
                    mine_works.append(dist.all_gather_into_tensor(out_t, inp, async_op=True))
                if backend == "nccl":
                    for w in mine_works:
                        w.wait()                         # comm_stream waits, not the host
                ev = torch.cuda.Event()
                ev.record(comm_stream)
                if last:
                    gathered[k] = ev
                else:
                    it_gathered[j] = ev
                    state["it_gathers"] += 1
                works += mine_works
        eng.sync()
        if multi:
            in_flight.append(works)
            drain(keep_last_step=not args.fusion)
        if args.fusion and rank == 0:
            fuse(k)

    if args.gather_per_iteration:
        step = step_per_iteration                           # noqa: F811

    def fence():
        drain(keep_last_step=False)
        if multi:
            comm_stream.synchronize()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    gc.collect()                 # (no cyclic garbage collection inside the timed region, as timeit does)
    gc.disable()
    for _ in range(args.warmup):
        step()
    fence()
    state["fusion_s"] = 0.0
    t0 = time.perf_counter()
    sweep_ms = conf_ms = 0.0
    launches = 0
    for _ in range(args.steps):
        step()
        t = eng.timing()
        # timing of the last call: batches are equal-sized; with one call per iteration, iterations are
        # not (early ones scatter more), so the roofline figure of that mode is the LAST iteration's
        calls = nb * (args.iters if args.gather_per_iteration else 1)
        sweep_ms += t["sweep_ms"] * calls
        conf_ms += t["confidence_ms"] * nb
        launches += t["sweep_launches"] * calls
    fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    # one step from an idle pipeline, its exchange (and fusion) inside: the latency of one reconstruct
    t1 = time.perf_counter()
    step()
    fence()
    first_step = time.perf_counter() - t1
    k_last = (state["step"] - 1) % nbuf
    if multi:
        tt = torch.tensor([elapsed, first_step], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, first_step = float(tt[0].item()), float(tt[1].item())
        # every rank must now hold every view's maps, in view order: its own block is checked bit for
        # bit and the neighbour's block must be populated
        fd, fc = in_view_order(full[k_last]["d"]), in_view_order(full[k_last]["c"])
        assert torch.equal(fd[rank].to(dev), depth[k_last].reshape(n_loc, -1)), "all-gather: own depth rows differ"
        assert torch.equal(fc[rank].to(dev), conf[k_last].reshape(n_loc, -1)), "all-gather: own confidence rows differ"
        other = fd[(rank + 1) % world]
        assert float(other.min()) >= float(np.float32(sc.depth_min)) - 1e-3, "all-gather: peer rows empty"
        assert float(in_view_order(full[k_last]["n"])[(rank + 1) % world].abs().max()) > 0.0, "all-gather: peer normals empty"

    n_hyp_step = (n_views if args.as_rank_of <= 0 else len(mine)) * H * W * args.iters * (2 + args.samples)
    value = n_hyp_step * args.steps / elapsed / 1e6
    S = len(srcs[0])                                 # 4 unless the scene has fewer than 5 views
    kname = ("pm_step_fast_kernel" if args.mode == "fast" else "pm_step_kernel") + f"<{args.patch},{S}>"
    bytes_per_hyp = 4 * S + 44                       # 60 B at S=4
    launch_ms = sweep_ms / max(launches, 1)
    vpl = eng.last_views_per_launch()
    algo_bytes_launch = bytes_per_hyp * vpl * H * W          # views per launch x pixels x 60 B
    achieved = algo_bytes_launch / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0

    if rank == 0:
        out = {
            "metric": "Mpixel-hypotheses/s (PatchMatch sweep)",
            "value": round(value, 1),
            "unit": "Mpx-hyp/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "first_step_ms": round(first_step * 1e3, 2),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": baseline_label(args, n_views, W, H, world) +
                                   ((f"{n_views}-view {W}x{H} PatchMatch MVS" + (" + fusion, " if args.fusion else ", ")) if world == 1 else
                                    (f"{n_views}-view {W}x{H} PatchMatch MVS scene, "
                                     f"{vpg} views per GPU, {'RCCL' if backend == 'nccl' else backend} all-gather of the maps"
                                     + (", fusion inside the step, " if args.fusion else ", "))) +
                                   f"{args.iters} iters x (2+{args.samples}) hypotheses, {args.patch}x{args.patch} NCC, "
                                   f"{S} sources, {args.mode} arithmetic",
                       "views_per_gpu": vpg, "scene_views": n_views, "width": W, "height": H, "patch": args.patch,
                       "iters": args.iters, "samples": args.samples, "sources": S, "arithmetic": args.mode,
                       "sampling": eng.sampling_mode(), "tile_rows": eng.last_tile_rows(), "views_per_launch": eng.last_views_per_launch(),
                       "schedule": args.schedule, "batches_per_step": nb,
                       "exchange": ("depth + normal maps after every iteration, depth + normal + confidence after the last"
                                    if args.gather_per_iteration else "depth + normal + confidence maps once per sweep"),
                       "pixel_hypotheses_per_step": n_hyp_step},
            "roofline": {"bound": "hbm", "kernel": kname,
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": profiled_traffic(kname, f"{vpl}x{W}x{H}"),
                         "traffic_source": profiled_traffic_and_source(kname, f"{vpl}x{W}x{H}")[1],
                         "algorithmic_bytes_per_launch": algo_bytes_launch,
                         "avg_launch_ms": round(launch_ms, 4), "launches_timed": launches},
            "confidence_ms_per_step": round(conf_ms / args.steps, 3),
        }
        # "dense points/s": fuse + filter the maps of the last step (untimed above) on the device
        # (amvs_fuse_filter = mvs_patchmatch.py:536-588, bit-identical to the NumPy path) and relate
        # the cloud to sweep + fusion time
        if args.fusion and state["cloud"] is not None:
            pts, cols, raw = state["cloud"]
            out["dense_points"] = {"raw": raw, "final": int(len(pts)), "fusion_inside_step": True,
                                   "points_per_s": round(len(pts) / (elapsed / args.steps), 1),
                                   "device_fusion_s": round(state["fusion_s"] / args.steps, 4)}
        elif not multi:
            torch.cuda.synchronize()
            t_f = time.perf_counter()
            pts, cols, raw = eng.fuse_filter_views(refs, depth[k_last].data_ptr(), conf[k_last].data_ptr(),
                                                   np.linalg.inv(sc.camera.K),
                                                   [(sc.poses[r].R, sc.poses[r].t) for r in refs], 3, True)
            t_f = time.perf_counter() - t_f
            out["dense_points"] = {"raw": raw, "final": int(len(pts)), "fusion_inside_step": False,
                                   "points_per_s": round(len(pts) / (elapsed / args.steps + t_f), 1),
                                   "device_fusion_s": round(t_f, 4)}
        elif multi:
            # several ranks: every rank holds all maps after the exchange; rank 0 fuses the gathered maps of
            # the last step (untimed above, like the one-GPU line) -- "dense points/s at 1/2/4/8 GPU"
            torch.cuda.synchronize()
            t_f = time.perf_counter()
            K_inv, pose_list = np.linalg.inv(sc.camera.K), [(sc.poses[r].R, sc.poses[r].t) for r in ids]
            if backend == "nccl":
                torch.cuda.synchronize()
                pts, cols, raw = eng.fuse_filter_views(ids, fd.data_ptr(), fc.data_ptr(), K_inv, pose_list, 3, True)
            else:
                pts, cols, raw = eng.fuse_filter(fd.numpy().reshape(n_views, H, W), fc.numpy().reshape(n_views, H, W),
                                                 np.stack([sc.colors[r] for r in ids]), K_inv, pose_list, 3, True)
            t_f = time.perf_counter() - t_f
            out["dense_points"] = {"raw": raw, "final": int(len(pts)), "fusion_inside_step": False,
                                   "points_per_s": round(len(pts) / (elapsed / args.steps + t_f), 1),
                                   "device_fusion_s": round(t_f, 4)}
        if world == 1 and not args.no_cpu_baseline:
            sample_refs = [n_views // 2, n_views // 2 + 1][: max(1, min(2, n_views))]
            out["cpu_baseline"] = cpu_baseline(sc, args.patch, sources, sample_refs, sc.depth_min, sc.depth_max,
                                               args.iters, args.samples, args.mode)
            out["cpu_baseline"]["value"] = round(out["cpu_baseline"]["value"], 2)

    eng.close()
    del eng, depth, normal, conf, full
    if rank == 0:
        if world == 1 and not args.no_planesweep:
            # BASELINE config 2 beside it: a few steps (about 10 ms each) and a short CPU leg
            torch.cuda.empty_cache()
            # a plane-sweep step is ~6.5 ms: 8 untimed steps (~55 ms, what ONE warm-up step of the main line
            # takes) before the 10 timed ones -- after a single 7 ms step the device is still ramping
            # (launches under the tracer: 7.46 7.01 6.68 6.39 6.37 6.35 ms, profiles/r03_plane_sweep_fast.txt)
            out["planesweep"] = run_planesweep(args, steps=10, warmup=8, with_cpu=not args.no_cpu_baseline, cpu_reps=3)
            if not args.no_subrecords and args.mode == "fast":
                # what a drop-in user runs: the classes default to the EXACT arithmetic (core/mvs_patchmatch.py,
                # core/dense_stereo.py) -- config 3 and config 2 in it -- and the reference's CLI constructs
                # PatchMatchMVS(scale=0.25, num_iterations=3, min_views=3), i.e. patch 11, 3 x (2 + 8)
                exact_args = argparse.Namespace(**dict(vars(args), mode="exact"))
                out["exact"] = {"patchmatch": run_patchmatch_single(sc, sources, ids, H, W, "exact", args.patch, args.iters,
                                                                    args.samples, steps=5, warmup=2),
                                "planesweep": run_planesweep(exact_args, steps=10, warmup=8, with_cpu=False)}
                out["cli_defaults"] = run_cli_defaults()
        print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()


def run_patchmatch_single(sc, sources, ids, H, W, mode, patch, iters, samples, steps, warmup):
    """The one-GPU PatchMatch measurement of the main line (whole batch resident, K timed steps, the sweep
    kernel's roofline from amvs_get_timing) for another arithmetic mode / patch size on the scene the main
    line rendered (sc.grays hold the 8-bit-exact host images by now)."""
    import torch

    import amvs
    from amvs.engine import make_pm_params
    dev = torch.device("cuda", 0)
    n = len(ids)
    eng = amvs.Engine(H, W, n, sc.camera.K.astype(np.float32), device=0, mode=mode)
    stream = torch.cuda.Stream(device=dev)
    eng.set_stream(stream.cuda_stream)
    for i in ids:
        eng.set_view(i, sc.grays[i], sc.poses[i].R, sc.poses[i].t)
    params = make_pm_params(patch, iters, samples, sc.depth_min, sc.depth_max)
    depth = torch.empty((n, H * W), dtype=torch.float32, device=dev)
    normal = torch.empty((n, 3 * H * W), dtype=torch.float32, device=dev)
    conf = torch.empty((n, H * W), dtype=torch.float32, device=dev)
    srcs = [sources[r] for r in ids]
    torch.cuda.synchronize()

    def step():
        eng.patchmatch_device(ids, srcs, params, 42, depth.data_ptr(), normal.data_ptr(), conf.data_ptr())
        eng.sync()

    gc.collect()
    gc.disable()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sweep_ms, launches = 0.0, 0
    for _ in range(steps):
        step()
        t = eng.timing()
        sweep_ms += t["sweep_ms"]
        launches += t["sweep_launches"]
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    S = len(srcs[0])
    n_hyp = n * H * W * iters * (2 + samples)
    vpl = eng.last_views_per_launch()
    kname = ("pm_step_fast_kernel" if mode == "fast" else "pm_step_kernel") + f"<{patch},{S}>"
    launch_ms = sweep_ms / max(launches, 1)
    algo = (4 * S + 44) * vpl * H * W
    achieved = algo / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    traffic, source = profiled_traffic_and_source(kname, f"{vpl}x{W}x{H}")
    rec = {"metric": "Mpixel-hypotheses/s (PatchMatch sweep)", "value": round(n_hyp * steps / elapsed / 1e6, 1), "unit": "Mpx-hyp/s",
           "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 2),
           "higher_is_better": True, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{n}-view {W}x{H} PatchMatch MVS, {iters} iters x (2+{samples}) hypotheses, {patch}x{patch} NCC, "
                                  f"{S} sources, {mode} arithmetic", "arithmetic": mode, "sampling": eng.sampling_mode(),
                      "tile_rows": eng.last_tile_rows(), "views_per_launch": vpl, "pixel_hypotheses_per_step": n_hyp},
           "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": source,
                        "algorithmic_bytes_per_launch": algo, "avg_launch_ms": round(launch_ms, 4), "launches_timed": launches}}
    eng.close()
    return rec


# The end-to-end records time FIVE calls after the first and report the median one (all five are listed): on this
# platform a pageable host-to-device copy issued right after the host process mapped or unmapped a multi-megabyte
# array can take 15-30 ms instead of 0.1 ms (the stereo class no longer provokes it, a host
# application's own allocations still can).
CLI_TIMED_CALLS = 5
CLI_TIMED_NOTE = "median of 5 calls after the first (each listed in timed_calls_s)"


def _median_call(calls):
    return sorted(calls, key=lambda t: t[0] + t[1])[len(calls) // 2]


def run_cli_defaults(n_views=16, h=3024, w=4032):
    """What the reference's CLI runs (run_reconstruction.py:131-136): PatchMatchMVS(camera, scale=0.25,
    num_iterations=3, min_views=3) -- patch 11, 3 x (2 + 8), 4 sources, the classes' default (exact) arithmetic --
    on 16 views of 4032x3024 (12 MP; processed at 1008x756), END TO END through PatchMatchMVS.reconstruct and
    utils.save_ply: 8-bit BGR uploads + image preparation on the device, depth range, source selection, sweep,
    fusion + filter on the device, PLY file.  Five calls are timed after the first (which pays the one-off
    allocations and the page-in of the library -- there is no JIT); `end_to_end_s` is the median one
    (CLI_TIMED_CALLS above); the split is measured on a further pass through the class's own steps."""
    import contextlib
    import tempfile

    import torch

    import amvs
    from amvs.core import utils as amvs_utils
    from amvs.synthetic import make_scene
    dev = torch.device("cuda", 0)
    scale = 0.25
    H, W = int(h * scale), int(w * scale)
    small = make_scene(n_views, H, W, seed=4321, device=str(dev))
    ids = sorted(small.poses)
    # 12-MP inputs: every processed pixel replicated 4 x 4 (the resize then returns the processed image itself)
    images = [{"image": np.ascontiguousarray(np.repeat(np.repeat(small.colors[i], 4, axis=0), 4, axis=1))} for i in ids]
    K = small.camera.K.copy()
    K[:2] *= 1.0 / scale
    cam = amvs.Camera(K=K, dist=np.zeros(5))
    sink = open(os.devnull, "w")
    with contextlib.redirect_stdout(sink):                     # (reconstruct prints the reference's progress lines)
        pm = amvs.PatchMatchMVS(cam, scale=scale, num_iterations=3, min_views=3, seed=1)
        tmp = tempfile.mkdtemp(prefix="amvs_bench_")
        ply = os.path.join(tmp, "cloud.ply")
        timings = []
        for _ in range(1 + CLI_TIMED_CALLS):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pts, cols = pm.reconstruct(images, small.poses)
            t1 = time.perf_counter()
            amvs_utils.save_ply(pts, cols, ply)
            timings.append((t1 - t0, time.perf_counter() - t1))
        timings = [timings[0], _median_call(timings[1:])] + timings[1:]
        # the split, through the class's own steps
        t0 = time.perf_counter()
        pm._estimate_depth_range(small.poses, None)
        proc = pm._prepare_images_device(images, ids, small.poses)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        jobs = [(r, pm._select_source_views(r, ids, small.poses, k=pm.NUM_SOURCES)) for r in ids]
        res = pm._sweep_resident(torch, jobs, proc, small.poses, ids)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        pts2, cols2, raw = pm._fuse_filter_resident(res, proc, small.poses)
        t3 = time.perf_counter()
    sink.close()
    tm = pm.last_timing
    eng = pm._engine
    vpl, S = eng.last_views_per_launch(), pm.NUM_SOURCES
    kname = f"pm_step_kernel<{pm.patch_size},{S}>"
    launch_ms = tm["sweep_ms"] / max(tm["sweep_launches"], 1)
    algo = (4 * S + 44) * vpl * H * W
    achieved = algo / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    n_hyp = n_views * H * W * pm.num_iterations * (2 + pm.num_samples)
    traffic, source = profiled_traffic_and_source(kname, f"{vpl}x{W}x{H}")
    size = os.path.getsize(ply)
    rec = {"workload": f"run_reconstruction.py defaults: PatchMatchMVS(scale=0.25, patch 11, 3 iters x (2+8), min_views 3), "
                       f"{n_views} views of {w}x{h} processed at {W}x{H}, exact arithmetic, reconstruct() + save_ply end to end",
           "end_to_end_s": round(sum(timings[1]), 4), "first_call_s": round(sum(timings[0]), 4),
           "timed_calls_s": [round(sum(t), 4) for t in timings[2:]], "timed": CLI_TIMED_NOTE,
           "reconstruct_s": round(timings[1][0], 4), "save_ply_s": round(timings[1][1], 4),
           "split_s": {"upload_and_image_preparation": round(t1 - t0, 4), "sweep": round(t2 - t1, 4),
                       "fusion_and_filter": round(t3 - t2, 4)},
           "sweep": {"metric": "Mpixel-hypotheses/s (PatchMatch sweep)",
                     "value": round(n_hyp / max(tm["sweep_ms"], 1e-9) / 1e3, 1), "unit": "Mpx-hyp/s",
                     "sweep_kernels_ms": round(tm["sweep_ms"], 3), "init_ms": round(tm["init_ms"], 3),
                     "confidence_ms": round(tm["confidence_ms"], 3), "pixel_hypotheses": n_hyp,
                     "tile_rows": eng.last_tile_rows(), "views_per_launch": vpl},
           "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": source,
                        "algorithmic_bytes_per_launch": algo, "avg_launch_ms": round(launch_ms, 4),
                        "launches_timed": int(tm["sweep_launches"])},
           "dense_points": {"raw": int(raw), "final": int(len(pts)), "points_per_s": round(len(pts) / max(sum(timings[1]), 1e-9), 1),
                            "ply_bytes": size},
           "input_bytes_uploaded": int(n_views * h * w * 3)}
    assert len(pts2) == len(pts)
    eng.close()
    # the CLI's other dense path (run_reconstruction.py:150-154): DenseStereoReconstructor(camera, scale=0.25)
    # .reconstruct(images, poses, max_pairs=30) -- 64 planes, 5x5, 6 neighbours, exact arithmetic, image preparation,
    # plane sweep, back-projection, outlier filter and voxel grid on the device -- and save_ply; median of five calls after the first
    sink = open(os.devnull, "w")
    with contextlib.redirect_stdout(sink):
        ds = amvs.DenseStereoReconstructor(cam, scale=scale)
        st = []
        for _ in range(1 + CLI_TIMED_CALLS):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            spts, scols = ds.reconstruct(images, small.poses, max_pairs=30)
            t1 = time.perf_counter()
            amvs_utils.save_ply(spts, scols, ply)
            st.append((t1 - t0, time.perf_counter() - t1))
        st = [st[0], _median_call(st[1:])] + st[1:]
        stm = ds._engine.timing() if ds._engine is not None else None
    sink.close()
    rec["stereo"] = {"workload": f"run_reconstruction.py defaults: DenseStereoReconstructor(scale=0.25) -- 64 planes, 5x5, 6 neighbours, "
                                 f"min_views 3 --, the same {n_views} views, exact arithmetic, reconstruct(max_pairs=30) + save_ply end to end",
                     "end_to_end_s": round(sum(st[1]), 4), "first_call_s": round(sum(st[0]), 4),
                     "timed_calls_s": [round(sum(t), 4) for t in st[2:]],
                     "reconstruct_s": round(st[1][0], 4), "save_ply_s": round(st[1][1], 4),
                     "sweep_kernel_ms": round(stm["sweep_ms"], 3) if stm else None,
                     "pixel_hypotheses": int(n_views * H * W * ds.num_depths),
                     "sweep_mpx_hyp_per_s": round(n_views * H * W * ds.num_depths / max(stm["sweep_ms"], 1e-9) / 1e3, 1) if stm else None,
                     "dense_points": {"final": int(len(spts)), "points_per_s": round(len(spts) / max(sum(st[1]), 1e-9), 1)}}
    if ds._engine is not None:
        ds._engine.close()
    try:
        os.remove(ply)
        os.rmdir(tmp)
    except OSError:
        pass
    return rec


def run_planesweep(args, steps, warmup, with_cpu, cpu_reps=8):
    """BASELINE config 2: plane-sweep stereo (dense_stereo.py:222-316) over all views of a
    1280x720 8-view scene, 64 inverse-depth planes, 5x5 NCC, 6 nearest neighbours.  Returns the
    record (same fields as the main line)."""
    import torch

    import amvs
    from amvs.synthetic import make_scene
    from oracle import oracle

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    H, W = (720, 1280) if (args.height, args.width) == (1080, 1920) else (args.height, args.width)
    n_views = 8 if args.views_per_gpu == 16 else args.views_per_gpu
    patch = 5 if args.patch == 7 else args.patch
    S, D = 6, args.planes
    sc = make_scene(n_views, H, W, seed=1234, device=str(dev))
    ds = amvs.DenseStereoReconstructor.__new__(amvs.DenseStereoReconstructor)
    ids = sorted(sc.poses)
    nbrs = {r: ds._find_neighbors(r, ids, sc.poses, k=S) for r in ids}
    depths = (1.0 / np.linspace(1 / sc.depth_max, 1 / sc.depth_min, D)).astype(np.float32)
    eng = amvs.Engine(H, W, n_views, sc.camera.K.astype(np.float32), device=0, mode=args.mode)
    stream = torch.cuda.Stream(device=dev)
    eng.set_stream(stream.cuda_stream)
    if getattr(args, "sweep_tile_rows", 0):
        eng.set_sweep_tuning(args.sweep_tile_rows, 0)
    lut = torch.from_numpy(np.arange(256, dtype=np.float32) / np.float32(255.0)).to(dev)
    for i in ids:
        codes = torch.round(torch.from_numpy(sc.grays[i]).to(dev) * 255.0).clamp(0, 255).long()
        g = lut[codes].contiguous()
        sc.grays[i] = g.cpu().numpy()
        eng.set_view_device(i, g.data_ptr(), sc.poses[i].R, sc.poses[i].t)
        torch.cuda.synchronize()
    kname = ("plane_sweep_fast_kernel" if args.mode == "fast" else "plane_sweep_kernel") + f"<{patch},{S}>"
    dmap = torch.empty((n_views, H, W), dtype=torch.float32, device=dev)
    conf = torch.empty((n_views, H, W), dtype=torch.float32, device=dev)
    refs = ids
    nb = [nbrs[r] for r in refs]

    def step():
        eng.plane_sweep_device(refs, nb, depths, patch, 0.8, dmap.data_ptr(), conf.data_ptr())
        eng.sync()

    # (no cyclic garbage collection inside the timed region, as timeit does: a full collection of the
    #  interpreter's ~10^6 live objects took 40 ms -- six plane-sweep steps -- when it fell into one; collected
    #  BEFORE the warm-up steps, so that the device does not idle between them and the timed ones)
    gc.collect()
    gc.disable()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    trace = []
    for _ in range(steps):
        ts = time.perf_counter()
        step()
        tm = time.perf_counter()
        kernel_ms += eng.timing()["sweep_ms"]
        trace.append((round((tm - ts) * 1e3, 2), round((time.perf_counter() - tm) * 1e3, 2)))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if os.environ.get("AMVS_BENCH_TRACE"):
        print("planesweep per step (step ms, timing() ms):", trace, file=sys.stderr)
    n_hyp = n_views * H * W * D
    value = n_hyp * steps / elapsed / 1e6
    bytes_per_hyp = 4 * S + 4 + 8.0 / D
    launch_ms = kernel_ms / steps
    achieved = bytes_per_hyp * n_hyp / (launch_ms * 1e-3) / 1e9
    out = {"metric": "Mpixel-hypotheses/s (plane sweep)", "value": round(value, 1), "unit": "Mpx-hyp/s",
           "n_gpus": 1, "steps": steps, "warmup": warmup,
           "ms_per_step": round(elapsed / steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"BASELINE config 2: {n_views}-view {W}x{H} plane-sweep stereo, {D} planes, "
                                  f"{patch}x{patch} NCC, {S} neighbours, {args.mode} arithmetic",
                      "arithmetic": args.mode, "sampling": eng.sampling_mode(),
                      "tile_rows": eng.last_tile_rows(), "pixel_hypotheses_per_step": n_hyp},
           "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": profiled_traffic(kname, f"{n_views}x{W}x{H}"),
                        "traffic_source": profiled_traffic_and_source(kname, f"{n_views}x{W}x{H}")[1],
                        "algorithmic_bytes_per_launch": int(bytes_per_hyp * n_hyp),
                        "avg_launch_ms": round(launch_ms, 4), "launches_timed": steps}}
    # the HBM ruler does not describe this kernel (traffic << algorithmic bytes): its VALU-issue roofline
    th, half = eng.last_tile_rows(), patch // 2
    tiles_x = -(-W // (64 - 2 * half))
    strip_rows = sum(min(th, H - y) + 2 * half for y in range(0, H, th))
    dpp = 3 * S * (patch - 1) * n_views * tiles_x * strip_rows * D
    out["valu_roofline"] = valu_issue_roofline(kname, f"{n_views}x{W}x{H}", dpp, launch_ms)
    if with_cpu:
        oracle.set_threads(host_cores())
        r = n_views // 2
        ctx = oracle.ViewContext(sc.camera.K.astype(np.float32), sc.grays[r], sc.poses[r].R, sc.poses[r].t,
                                 [sc.grays[i] for i in nbrs[r]], [sc.poses[i].R for i in nbrs[r]],
                                 [sc.poses[i].t for i in nbrs[r]], patch, mode=args.mode)
        reps = cpu_reps
        t0 = time.time()
        for _ in range(reps):
            ctx.plane_sweep(depths, 0.8)
        dt = time.time() - t0
        out["cpu_baseline"] = {"value": round(reps * H * W * D / dt / 1e6, 2), "unit": "Mpx-hyp/s",
                               "cores": oracle.num_threads(), "kind": "port",
                               "sample": f"1 view at {W}x{H}, all {D} planes, {reps} repetitions = "
                                         f"{reps*H*W*D/1e6:.0f} Mpx-hyp in {dt:.1f} s (oracle/amvs_oracle.c, "
                                         f"{args.mode} arithmetic, OpenMP on {oracle.num_threads()} threads)"}
    eng.close()
    return out


if __name__ == "__main__":
    main()
