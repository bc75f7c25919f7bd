#!/usr/bin/env python3
"""bench.py — stereo frames/sec through track + BA (BASELINE.json metric) on synthetic stereo + IMU.

    python bench.py --gpus N --steps K --warmup W [--mode raw|dynamic] [--config zed|euroc|kitti] [--every-second-frame] [--shard] [--sequences S]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one stereo frame through the HIP front end (FeatureTracker::TrackImage, or TrackSemanticImage + InstsFeatManager::InstsTrack in
dynamic mode) and the HIP back end (Estimator::ProcessImage: bundle adjustment + marginalization on EVERY frame = the KITTI convention; the
reference forwards only every 2nd frame on the other datasets, system/main.cpp:300-307: --every-second-frame).  Frames (and masks) are rendered
once and are resident in HBM before the timed region.  Default workload = the configuration BASELINE.json's metric is quoted on: 1280x720 ZED.

N > 1, default: every rank runs an independent sequence (replicas: config 4's "batched" axis); value = whole-job rate, weak scaling, no
data-path collective.  --shard: ONE window solve sharded by landmark over the N ranks with the reduced-system all-reduce —
a BA-only microbench whose line is reported separately.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed per launch on its own stream, a second instrumented pass) and
`cpu_baseline` (the CPU oracle = restated reference path, bounded sample, rank 0, N = 1 only).
"""
import argparse
import gc
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
T_PROCESS_START = time.perf_counter()
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
CONFIGS = {                # primary + the two secondary sizes
    "zed": dict(w=1280, h=720, max_cnt=250, min_dist=25, iters=10, use_imu=1),
    "euroc": dict(w=752, h=480, max_cnt=150, min_dist=30, iters=8, use_imu=1),
    "kitti": dict(w=1242, h=375, max_cnt=250, min_dist=25, iters=10, use_imu=0),
}


def git_head():
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=5).stdout.strip()
        if head:
            return head
    except Exception:
        pass
    try:            # the GPU box has no .git: __graft_entry__.build() leaves the commit the library was built from next to it
        return open(os.path.join(ROOT, "dynamic_vins_amd", "lib", "BUILD_HEAD")).read().strip() or None
    except OSError:
        return None


def csrc_digest():
    """sha1 over the kernel sources: the PMC file (collected in separate rocprofv3 --pmc runs) is trusted only for the kernels it was collected on"""
    import glob
    import hashlib
    h = hashlib.sha1()
    for f in sorted(sum((glob.glob(os.path.join(ROOT, "dynamic_vins_amd", "csrc", pat)) for pat in ("*.hip", "*.h", "*.inc")), [])):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def self_launch(n, argv, backend_env=None):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(backend_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", choices=["raw", "dynamic"], default="raw")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="zed")
    ap.add_argument("--every-second-frame", action="store_true", help="the reference's convention outside KITTI: only every 2nd tracked frame gets BA (system/main.cpp:300-307)")
    ap.add_argument("--shard-transport", choices=["rccl", "peer", "host"], default=None, help="with --shard: ncclAllGather (default on RCCL), the one-shot peer-write exchange over hipIpc windows, or the host call-back")
    ap.add_argument("--shard", action="store_true", help="BA-only: one window sharded by landmark over the ranks (reduced-system all-reduce)")
    ap.add_argument("--sequences", type=int, default=1, help="S independent sequences per GPU interleaved by one host thread (aggregate rate; the default single-sequence line stays the headline)")
    ap.add_argument("--batched", action="store_true", help="with --sequences: the window solves of all sequences share every launch (dv_batch), one host thread")
    ap.add_argument("--host-loop", choices=["cpp", "python"], default="cpp", help="raw mode: the per-frame host loop of the timed region in C++ inside the library (dv_runner) or in Python (pipeline.py); dynamic mode and the instrumented pass use the Python loop")
    ap.add_argument("--runner", choices=["cpp", "python"], default="cpp", help="with --sequences: the host loop in C++ inside the library (dv_runner) or the round-2 Python loop")
    ap.add_argument("--group-size", type=int, default=0, help="with --sequences --batched --runner cpp: sequences per dv_batch group (default: four groups, one host thread each; two groups below 8 sequences)")
    ap.add_argument("--no-batch-front", action="store_true", help="with --sequences --batched: one set of tracking launches per sequence instead of dv_batch_track_enqueue (A/B)")
    ap.add_argument("--runner-threads", type=int, default=0, help="with --sequences: host threads driving the groups (default and maximum: one per group)")
    ap.add_argument("--teams", action="store_true", help="(default since round 5) several host threads per dv_batch group: --runner-threads = a multiple of the group count, default two per group")
    ap.add_argument("--no-teams", action="store_true", help="with --sequences --batched: one host thread per dv_batch group (the default until round 4)")
    ap.add_argument("--sequence-threads", action="store_true", help="with --sequences: one host thread per sequence instead of one interleaving thread")
    ap.add_argument("--blocks", type=int, default=2, help="consecutive timed blocks of --steps frames (the first is `value`; all are listed in config.block_values)")
    ap.add_argument("--timing-block", type=int, default=1, help="which block the instrumented (per-kernel HIP events) pass times")
    ap.add_argument("--scene", choices=["escort", "room"], default="escort", help="--mode dynamic: escort = boxes travelling with the camera (in view in every frame); room = room-fixed boxes (round-2 scene)")
    ap.add_argument("--erode", type=int, default=5, help="--mode dynamic: mask_morphology_size (viode.yaml: 5; the ZED / KITTI dynamic configs: 20)")
    ap.add_argument("--no-dynamic-line", action="store_true", help="skip the second measurement (dynamic mode) of the default run")
    ap.add_argument("--no-extra-lines", action="store_true", help="skip config.host_frames_line (pinned host buffers, upload timed) and config.multiseq_line (16 sequences, child process) of the default run")
    ap.add_argument("--objects", type=int, default=4, help="--mode dynamic: number of moving boxes in the scene (dynsim.ring_boxes)")
    ap.add_argument("--prealloc-mb", type=int, default=0, help="debug: allocate (and keep) this much device memory before the sequence is rendered (shifts where the frames land in HBM)")
    ap.add_argument("--debug-set", default="", help="comma-separated dv_debug_set keys to switch on in every context (A/B runs of kernel variants, e.g. ldl_generic)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=16)
    ap.add_argument("--host-frames", action="store_true", help="frames handed over as host buffers: the PCIe-inclusive rate (never the headline value)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves as FRESH child processes (one per GPU, RCCL rendezvous on 127.0.0.1) — before
        # anything in this process has touched the GPU (never a re-exec of a process that initialised HIP) — and leave with the launcher's exit code.
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    import numpy as np
    if args.mode == "dynamic" or (args.sequences <= 1 and not args.no_dynamic_line):
        # dynamic mode keeps six streams busy per context (background tracker, object tracker, extra points, window solve, object solve + the renderer's): with the
        # runtime's default of 4 hardware queues the extra-point chain of frame k+1 shared a queue with the window solve of frame k and waited behind it (measured:
        # 797 -> 553 frames/s).  INTEGRATION.md lists the variable among the deployment settings.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if args.sequences > 1:
        # every sequence owns four HIP streams; the runtime maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues, and two streams that share a queue
        # serialise — with the default, two sequences run at half speed each (measured).  Must be set before the runtime initialises.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
    import torch

    from dynamic_vins_amd import dist as dv_dist

    if os.environ.get("BENCH_SELFTEST") == "1":
        # launch rehearsal without a GPU (tests/test_distributed.py): rendezvous of the ranks this process (or its self-launched children) belongs to, one
        # all-gather of the rank ids, the barrier + MAX-over-ranks the timed region uses; no compute
        rank, world, local_rank = dv_dist.init(prefer_gpu=False)
        import torch.distributed as td
        seen = [None] * world
        if world > 1:
            td.all_gather_object(seen, (rank, local_rank, os.getpid()))
        else:
            seen = [(rank, local_rank, os.getpid())]
        dv_dist.barrier()
        tmax = dv_dist.max_over_ranks(float(rank + 1))
        if rank == 0:
            print(json.dumps({"selftest": True, "n_gpus": world, "requested_gpus": args.gpus, "ranks": [s[0] for s in seen], "local_ranks": [s[1] for s in seen],
                              "distinct_processes": len({s[2] for s in seen}), "max_over_ranks": tmax, "launcher": os.environ.get("TORCHELASTIC_RUN_ID") is not None}))
        dv_dist.barrier(); dv_dist.finalize()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one process per GPU.  BENCH_DEVICE / BENCH_BACKEND exist only to rehearse the multi-rank control flow on a 1-GPU box
    forced_dev = os.environ.get("BENCH_DEVICE")
    torch.cuda.set_device(int(forced_dev) if forced_dev is not None else int(os.environ.get("LOCAL_RANK", "0")))
    rank, world, local_rank = dv_dist.init(prefer_gpu=os.environ.get("BENCH_BACKEND", "nccl") != "gloo")
    if forced_dev is not None:
        local_rank = int(forced_dev)

    from dynamic_vins_amd import sim

    if args.sequences > 1:
        from dynamic_vins_amd.multiseq_bench import run_multiseq_bench
        run_multiseq_bench(args, rank, world, local_rank)
        dv_dist.barrier(); dv_dist.finalize()
        return
    if args.shard:
        from dynamic_vins_amd.shard_bench import run_shard_bench
        run_shard_bench(args, rank, world, local_rank)
        dv_dist.barrier(); dv_dist.finalize()
        return

    # north_star's TARGET configuration (>= 500 frames/s on 1280x720 VIODE-dynamic) rides on the default command: the same workload in dynamic mode, objects in every
    # frame, timed the same way (barriers, max over ranks), reported beside the headline as config.dynamic_line
    want_dyn = args.mode == "raw" and args.config == "zed" and not args.no_dynamic_line and not args.every_second_frame and not args.host_frames
    # measured FIRST: whichever dynamic-mode pass comes second in the process runs ~12 % slower in its first block (692 against 777 frames/s, same box, same work; the raw
    # pass does not care about the order: 1000 - 1030 either way).  BENCH_DYN_FIRST=0 restores the old order.
    dyn_first = os.environ.get("BENCH_DYN_FIRST", "1") == "1"
    dyn = None
    if want_dyn and dyn_first:
        a2 = argparse.Namespace(**vars(args)); a2.mode = "dynamic"
        dyn = measure(a2, cfg, rank, world, local_rank, want_roofline=False)
    out = measure(args, cfg, rank, world, local_rank)
    extra_lines = args.mode == "raw" and args.config == "zed" and not args.no_extra_lines and not args.no_dynamic_line and not args.every_second_frame and not args.host_frames and world == 1
    if extra_lines:
        # the PCIe-inclusive rate beside the headline (never `value`): the same sequence handed over as pinned host buffers, upload inside the timed region
        a3 = argparse.Namespace(**vars(args)); a3.host_frames = True; a3.no_cpu_baseline = True
        hf = measure(a3, cfg, rank, world, local_rank, want_roofline=False)
        if rank == 0 and hf is not None:
            out["config"]["host_frames_line"] = {"value": hf["value"], "unit": "frames/s", "ms_per_step": hf["ms_per_step"], "timed_region_s": hf["timed_region_s"], "warmup": hf["warmup"], "block_values": hf["config"]["block_values"],
                                                 "frames": hf["config"]["frames"], "host_loop": hf["config"]["host_loop"], "ate_rmse_m_vs_ground_truth": hf["config"]["ate_rmse_m_vs_ground_truth"],
                                                 "bytes_uploaded_per_frame": 2 * cfg["w"] * cfg["h"]}
        # configs[3] (several sequences per GPU, window solves and front ends in shared launches) under the same clock: a CHILD process (it needs its own
        # GPU_MAX_HW_QUEUES, read when the runtime initialises) running the documented command; its line — printed only behind the per-sequence trajectory gate and the
        # bit-identity check of two members against the single-thread run (multiseq_bench.py) — is embedded as config.multiseq_line
        if rank == 0:
            cmd = [sys.executable, os.path.abspath(__file__), "--sequences", "16", "--batched", "--steps", "40"]
            t0 = time.perf_counter()
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=dict(os.environ, GPU_MAX_HW_QUEUES="12"))
                ln = [x for x in r.stdout.splitlines() if x.startswith("{")]
                if r.returncode == 0 and ln:
                    ms = json.loads(ln[-1])
                    out["config"]["multiseq_line"] = {"command": "python bench.py --sequences 16 --batched --steps 40", "value": ms["value"], "unit": "frames/s", "steps": ms["steps"],
                                                      "timed_region_s": ms["timed_region_s"], "ms_per_step": ms["ms_per_step"], "second_block_value": ms["config"]["second_block_value"],
                                                      "sequences_per_gpu": ms["config"]["sequences_per_gpu"], "group_size": ms["config"]["group_size"], "runner_threads": ms["config"]["runner_threads"],
                                                      "teams": ms["config"].get("teams"), "ate_rmse_m_vs_ground_truth_max": ms["config"]["ate_rmse_m_vs_ground_truth_max"],
                                                      "bit_identity": ms["config"].get("bit_identity"), "child_wall_s": round(time.perf_counter() - t0, 1)}
                else:
                    out["config"]["multiseq_line"] = {"error": "exit %d: %s" % (r.returncode, (r.stderr or "").strip()[-400:])}
            except subprocess.TimeoutExpired:
                out["config"]["multiseq_line"] = {"error": "timeout"}
    if want_dyn:
        if not dyn_first:
            a2 = argparse.Namespace(**vars(args)); a2.mode = "dynamic"
            dyn = measure(a2, cfg, rank, world, local_rank, want_roofline=False)
        if rank == 0 and dyn is not None:
            dc = dyn["config"]
            out["config"]["dynamic_line"] = {"value": dyn["value"], "unit": "frames/s", "ms_per_step": dyn["ms_per_step"], "timed_region_s": dyn["timed_region_s"], "warmup": dyn["warmup"], "block_values": dc["block_values"], "block_step_ms": dc.get("block_step_ms"),
                                             "ate_rmse_m_vs_oracle": dc["ate_rmse_m_vs_oracle"], "ate_rmse_m_vs_ground_truth": dc["ate_rmse_m_vs_ground_truth"],
                                             "workload": dc["workload"], "dynamic": dc.get("dynamic"), "host_loop": dc.get("host_loop"), "solver_iterations_per_frame": dc["solver_iterations_per_frame"],
                                             "cpu_baseline_value": None if dyn["cpu_baseline"] is None else dyn["cpu_baseline"]["value"], "target_frames_per_s": 500}
    if rank == 0:
        print(json.dumps(out))
    dv_dist.barrier()          # every rank stays until rank 0 has finished its extra passes
    dv_dist.finalize()


def measure(args, cfg, rank, world, local_rank, want_roofline=True):
    """one line of the bench: W warm-up + `blocks` x K timed frames of args.mode, then (rank 0) the instrumented pass and the CPU-oracle leg"""
    import numpy as np
    import torch
    from dynamic_vins_amd import dist as dv_dist
    from dynamic_vins_amd import sim
    from dynamic_vins_amd.pipeline import DynamicPipeline, DynamicSequence, Pipeline, SyntheticSequence

    w, h = cfg["w"], cfg["h"]
    stride = 2 if args.every_second_frame else 1
    warm_ba = max(args.warmup, 12)            # the estimator needs kWinSize+1 = 11 BA frames to initialise (estimator.cpp:1464-1483)
    # Rounds 4-5 warmed the dynamic line 20 frames and the PCIe-inclusive line 32 frames longer and placed a device-wide synchronisation inside the warm-up, because one-time
    # stalls of 5 - 9 ms (a copy queue the runtime brought up lazily; buffers created at the first solves) fell into the first timed block.  Round 6 removed the causes (no copy
    # engine in the per-frame path: copy.hip, DV_MEM_PINNED; be_prepare at dv_est_create) and with them the special warm-ups: every line warms up max(--warmup, 12) frames.
    # BENCH_DYN_WARM=1 / BENCH_HOST_WARM=1 bring the longer warm-ups back for A/B; `warmup_excluded` of the line says what the timed blocks leave out.
    warm_extra = 0
    if args.mode == "dynamic" and os.environ.get("BENCH_DYN_WARM", "0") == "1":
        warm_extra = 20
    if args.host_frames and os.environ.get("BENCH_HOST_WARM", "0") == "1":
        warm_extra = 32
    warm_ba += warm_extra
    n_frames = (warm_ba + max(2, args.blocks) * args.steps) * stride + 2
    cam = sim.ZED if (w, h) == (1280, 720) else sim.scaled_cam(sim.ZED, w, h, 1280, 720)
    dev = f"cuda:{local_rank}"
    _pad = torch.empty(args.prealloc_mb << 20, dtype=torch.uint8, device=dev) if args.prealloc_mb > 0 else None
    if args.mode == "dynamic":
        from dynamic_vins_amd import dynsim
        # default scene: boxes travelling with the camera (objects in EVERY frame, >= 3 detections per frame); --scene room: the round-2 scene (room-fixed boxes, seen now and then)
        boxes = ("escort", args.objects) if args.scene == "escort" else (None if args.objects == 3 else dynsim.ring_boxes(args.objects))
        seq = DynamicSequence(w, h, cam, n_frames, rate=20.0, device=dev, boxes=boxes)
    else:
        seq = SyntheticSequence(w, h, cam, n_frames, rate=20.0, phase=dv_dist.sequence_phase(rank), device=dev)

    def make_pipe():
        if args.mode == "dynamic":
            return DynamicPipeline(seq, max_cnt=cfg["max_cnt"], min_dist=cfg["min_dist"], max_iters=cfg["iters"], device=local_rank, use_imu=cfg["use_imu"], mask_morphology_size=args.erode,
                                   extra_from_disparity=os.environ.get("BENCH_DYN_PASSTHROUGH", "0") != "1",
                                   static_as_background=os.environ.get("BENCH_DYN_STATIC_BG", "1") == "1")      # para::is_static_inst_as_background: the reference's default is true (vio_parameters.h:86)      # (debug A/B: 1 = the detections' own points are handed through, no extra-point kernels)
        return Pipeline(seq, max_cnt=cfg["max_cnt"], min_dist=cfg["min_dist"], max_iters=cfg["iters"], device=local_rank, use_imu=cfg["use_imu"],
                        host_frames=args.host_frames and args.host_loop != "cpp", ba_stride=stride)      # (the C++ runner pins its own host copies: backend.Runner)

    frame_ms = []

    def run_cpp(collective, blocks):
        """the timed region on the library's C++ host loop (dv_runner): no interpreter inside it — one call per block"""
        from dynamic_vins_amd.backend import Runner
        pipe = make_pipe()
        for key in filter(None, args.debug_set.split(",")):
            assert pipe.ctx.lib.dv_debug_set(pipe.ctx.h, key.encode(), 1) == 0, key
        # --host-frames: pinned host buffers, every frame's trip over PCIe inside the timed region (BENCH_HOST_ENGINE=1: through the copy engine, hipMemcpy2DAsync, as in rounds 1-5)
        runner = Runner([pipe], host_frames=("engine" if os.environ.get("BENCH_HOST_ENGINE", "0") == "1" else True) if args.host_frames else False)
        if os.environ.get("BENCH_TRACKER_THREAD", "1") != "1":          # A/B: dynamic mode on the one-thread loop of round 4 instead of T2 beside T3
            runner.set("tracker_thread", 0)
        runner.run(warm_ba * stride)
        it0 = runner.get(0)[2]                               # window-solve iterations spent in the warm-up
        gc.collect(); gc.freeze()
        times = []
        for _b in range(blocks):
            if collective:
                dv_dist.barrier()
            torch.cuda.synchronize(); pipe.ctx.sync()
            t0 = time.perf_counter()
            runner.run(args.steps * stride)
            pipe.ctx.sync(); torch.cuda.synchronize()
            if collective:
                dv_dist.barrier()
            times.append(time.perf_counter() - t0)
            frame_ms.append({"p50": None, "p95": None, "max": None, "t_since_start_s": round(t0 - T_PROCESS_START, 2), "note": "C++ host loop: one call per block, no per-step host clock"})
        try:          # per-frame end clocks of the C++ loop (dv_runner_get_frame_clock: host clock at every frame's end, the run is not cut): where inside a block the time went
            clk = np.asarray(runner.frame_clock(0), dtype=np.float64)
            for _b in range(blocks):
                a = (warm_ba + _b * args.steps) * stride
                dd = np.diff(clk[a - 1:a + args.steps * stride]) * 1e3
                if len(dd) > 1:      # dd[0] spans two dv_runner_run calls (the host's synchronisation, gc and barriers between the blocks: not a frame); the frames are dd[1:]
                    fr = dd[1:]
                    frame_ms[len(frame_ms) - blocks + _b].update(p50=round(float(np.median(fr)), 3), p95=round(float(np.percentile(fr, 95)), 3), max=round(float(fr.max()), 3),
                                                                 first_two=[round(float(v), 3) for v in fr[:2]], inter_call_gap_ms=round(float(dd[0]), 3), frames_above_2ms=int((fr > 2.0).sum()),
                                                                 note="C++ host loop: frame-end clocks of dv_runner (one call per block); the gap across the call boundary is inter_call_gap_ms, not part of p50 / p95 / max")
        except Exception as e:      # (diagnostics only)
            frame_ms[-1]["note"] = "frame clocks unavailable: %s" % e
        st, poses, iters_all, _fr = runner.get(0)
        pipe.poses, pipe.pose_times = [q for q in poses[:, 1:8]], [float(t) for t in poses[:, 0]]
        pipe.est.state, pipe.rows = st, np.zeros(runner.last_rows)
        if args.mode == "dynamic":
            ds = runner.dynamic_stats(0)
            pipe.stat.update(frames=int(_fr), **ds)
            pipe.ifeats = np.zeros(int(round(ds["object_features"] / max(int(_fr), 1))))
        runner.close()
        return pipe, times, iters_all - it0

    def run(kernel_timing, collective=True, blocks=2):       # collective=False: rank-0-only pass, no barriers
        if args.host_loop == "cpp" and not kernel_timing:      # raw and dynamic mode both run on the library's C++ loop (dv_runner / dv_runner_set_dynamic); --host-frames too
            return run_cpp(collective, blocks)
        pipe = make_pipe()
        for key in filter(None, args.debug_set.split(",")):
            assert pipe.ctx.lib.dv_debug_set(pipe.ctx.h, key.encode(), 1) == 0, key
        for _w in range(warm_ba * stride):
            pipe.step()
        # A generation-2 collection of the interpreter's ~10^6 objects (torch, numpy) is a ~60 ms pause that lands in a random block (seen as ONE 63 ms step
        # in config.block_step_ms): collect now and move everything that exists to the permanent generation, as timeit does by disabling the collector.
        gc.collect(); gc.freeze()
        times, iters = [], 0
        for _b in range(blocks):             # two consecutive blocks of `steps` frames: a cold first block shows up instead of lowering the headline
            if kernel_timing and _b == blocks - 1:
                pipe.ctx.sync()
                pipe.ctx.timing_enable(2)   # the instrumented pass times its LAST block (--timing-block selects which one that is)
            if collective:
                dv_dist.barrier()
            torch.cuda.synchronize(); pipe.ctx.sync()
            t0 = time.perf_counter()
            marks = [t0]
            for _ in range(args.steps):
                for _s in range(stride):
                    st = pipe.step()
                iters += st.iterations
                marks.append(time.perf_counter())      # host clock only: where inside a block the time goes (uniform slow-down or bursts)
            pipe.ctx.sync(); torch.cuda.synchronize()
            if collective:
                dv_dist.barrier()
            times.append(time.perf_counter() - t0)
            d = np.diff(marks) * 1e3
            frame_ms.append({"p50": round(float(np.median(d)), 3), "p95": round(float(np.percentile(d, 95)), 3), "max": round(float(d.max()), 3),
                             "t_since_start_s": round(t0 - T_PROCESS_START, 2)})
        return pipe, times, iters

    pipe, times, iters = run(False, blocks=max(2, args.blocks))
    red_dev = dev if os.environ.get("BENCH_BACKEND", "nccl") != "gloo" else "cpu"
    times = [dv_dist.max_over_ranks(t, device=red_dev) for t in times]
    dt = times[0]                            # the contract: EXACTLY K timed steps -> the first block is the reported one
    ate = pipe.ate()
    n_lm = int(pipe.est.state.n_long)
    nfeat = len(pipe.rows)
    dyn_info = None
    if args.mode == "dynamic":
        I, S = pipe.est.instances()
        dyn_info = dict(static_inst_as_background=bool(getattr(pipe, "static_as_background", False)), objects_static_at_the_end=int(I["is_static"].sum()) if len(I) else 0,
                        objects_tracked=int(len(I)), objects_initialised=int(I["is_initial"].sum()) if len(I) else 0,
                        object_features_per_frame=int(len(pipe.ifeats)), object_solve_iterations=int(S[0]),
                        over_the_run=dict(frames=pipe.stat["frames"], frames_with_objects=pipe.stat["frames_with_objects"],
                                          detections_per_frame=round(pipe.stat["object_detections"] / max(pipe.stat["frames"], 1), 2), min_detections_in_a_frame=int(pipe.stat["min_detections"]),
                                          object_features_per_frame=round(pipe.stat["object_features"] / max(pipe.stat["frames"], 1), 1)))
    dev_poses, dev_times = np.array(pipe.poses), list(pipe.pose_times)
    pipe.ctx.close()

    # ---- roofline of the dominant kernel: second, instrumented pass over the same timed region ----
    roof, kern = None, {}
    if rank == 0 and want_roofline:
        pipe2, _, _ = run(True, collective=False, blocks=max(1, args.timing_block))
        names = ["k_be_solve", "k_be_reduce", "k_be_eval_full", "k_be_eval_cost", "k_be_accept", "k_be_marg", "obj_solve",
                 "pyr", "lk_temporal", "compact", "gftt_eig", "gftt_select", "lk_stereo", "finalize", "inst_track"]
        for nme in names:
            ms, cnt = pipe2.ctx.timing_get(nme)
            if cnt:
                kern[nme] = dict(total_ms=ms, launches=cnt, avg_us=ms / cnt * 1e3)
        dom = max((k for k in kern if k != "inst_track"), key=lambda k: kern[k]["total_ms"])
        n_state = 165 if cfg["use_imu"] else 60
        L = n_lm
        # ALGORITHMIC bytes per launch
        alg = {
            "k_be_solve": 2 * n_state * n_state * 8 + 3 * L * 69 * 8,          # Hd + Sc once, three passes over (w[66], h, g, scale) per landmark
            "k_be_reduce": L * 928 * 8 + 2 * n_state * n_state * 8,            # every packet once + Hd, Sc written
            "k_be_eval_full": L * (16 + 928 * 8) + L * 20 * 112,               # factor records read, packet written
            "k_be_marg": 97 * 97 * 8 * 2 + 150 * 20 * 112,
            "lk_temporal": nfeat * 6 * 2 * 529, "lk_stereo": nfeat * 6 * 2 * 529,
            "gftt_eig": 2 * w * h, "pyr": int(2 * 1.328 * w * h),
        }.get(dom)
        avg_s = kern[dom]["avg_us"] * 1e-6
        achieved = (alg / avg_s / 1e9) if (alg and avg_s > 0) else None
        # HBM-side bytes per launch from the committed PMC passes (a --pmc run cannot share a process with the timed run); only trusted if the
        # file was collected at this HEAD
        traffic, stale = None, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            key = {"k_be_solve": "be_solve_kernel", "k_be_reduce": "be_reduce_kernel", "k_be_eval_full": "be_eval_kernel<true>", "k_be_eval_cost": "be_eval_kernel<false>",
                   "lk_temporal": "lk_track_kernel", "lk_stereo": "lk_track_kernel", "gftt_eig": "gftt_tile_kernel", "gftt_select": "gftt_select_kernel", "pyr": "pyr_down_kernel"}.get(dom)
            hit = [v for k, v in pmc["kernels"].items() if key and k.startswith(key)]
            if hit:
                traffic = int(hit[0]["traffic_bytes"])
            stale = pmc.get("csrc_digest") != csrc_digest()
        except (OSError, KeyError, ValueError):
            traffic = None
        # end-to-end figure with SURVEY 8(d)'s algorithmic bytes: front end 4.66 P + 12.7 kB N ; back end E F 112 B (fused into the Schur accumulation)
        F_blocks = L * 12                      # ~12 residual blocks per landmark in the full window (measured: 3.0-3.4 k for 260-280 landmarks)
        e2e_bytes = 4.66 * w * h + 12700.0 * nfeat + (iters / max(2 * args.steps, 1) + 1) * F_blocks * 112
        roof = {"kernel": (dom[2:] if dom.startswith("k_") else dom),
                "bound": "hbm", "bound_note": "priced against HBM as the contract asks; the kernel is latency-bound (one workgroup, sequential pivots), see DESIGN.md 4",
                "achieved": None if achieved is None else round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 6), "traffic": None if stale else traffic, "traffic_stale": stale,
                "avg_launch_us": round(kern[dom]["avg_us"], 2), "algorithmic_bytes_per_launch": alg,
                "end_to_end": {"algorithmic_bytes_per_frame": int(e2e_bytes), "achieved_GBs": round(e2e_bytes * args.steps / dt / 1e9, 3),
                               "frac": round(e2e_bytes * args.steps / dt / 1e9 / HBM_PEAK_GBS, 6)},
                "kernels_us": {(k[2:] if k.startswith("k_") else k): round(v["avg_us"], 1) for k, v in kern.items()}}
        pipe2.ctx.close()

    # ---- CPU baseline: the oracle (restated reference path) on a bounded sample of the same workload; also gives the oracle trajectory ----
    cpu, ate_vs_oracle = None, None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        from tests import oracle_py
        flags = "-O3 -march=native -ffp-contract=off"
        try:           # timing build for THIS host; falls back to the checker's -O2 build
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "fast"], timeout=300)
            import ctypes
            o = oracle_py.Oracle(ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libdvins_oracle_fast.so")))
        except Exception:
            o, flags = oracle_py.load(), "-O2 -ffp-contract=off"
        ncores = os.cpu_count() or 1
        nthreads = min(ncores, 16)        # the per-level point loop of 250 points does not feed more (OpenCV's parallel_for_ uses a pool; the oracle spawns)
        camt = sim.cam_tuple(cam)

        def cpu_run(threads, frames_wanted):
            o.lib.dvo_set_threads(threads)
            trk = o.tracker(w, h, cfg["max_cnt"], cfg["min_dist"], 1, 1, camt, camt)
            kw = dict(use_imu=cfg["use_imu"], stereo=1, max_iters=cfg["iters"], ric=[sim.R_IC, sim.R_IC], tic=[sim.T_IC0, sim.T_IC1], **seq.noise)
            if args.mode == "dynamic":
                from dynamic_vins_amd import dynsim
                oin = o.insts(trk, 50, 5, 1)
                est = o.estimator(dynamic=1, use_det3d=1, static_inst_threshold=1.0, **kw)
            else:
                est = o.estimator(**kw)
            nf = min((frames_wanted + 12) * stride, n_frames)
            host = [seq.host_frame(k) for k in range(nf)]
            static_bg, snaps = args.mode == "dynamic" and os.environ.get("BENCH_DYN_STATIC_BG", "1") == "1", []
            from dynamic_vins_amd import _abi, viode
            k_imu, t_fe, t_be, counted, poses, ptimes = 0, 0.0, 0.0, 0, [], []
            for k in range(nf):
                t = seq.times[k]
                while k_imu < len(seq.imu_t) and seq.imu_t[k_imu] <= t + 0.006:
                    est.input_imu(seq.imu_t[k_imu], seq.imu_a[k_imu], seq.imu_g[k_imu]); k_imu += 1
                a = time.perf_counter()
                if args.mode == "dynamic":
                    mask_k = seq.inv_mask[k]
                    if static_bg:          # FeatureTrack, system/main.cpp:217-245: the estimator's static report of the newest back-end frame <= k - 2 (choice T1)
                        best = [sn for sn in snaps if sn[0] <= k - _abi.DV_STATIC_REPORT_LAG]
                        mask_k = viode.unmask_static(mask_k, seq.dets[k], best[-1][1] if best else [])
                    rows = trk.track_image(host[k][0], host[k][1], t, mask=mask_k, mode=2, erode_k=args.erode)
                    oin.set_disparity(seq.disp_host(k), seq.baseline)
                    io, fo, po = oin.track(host[k][0], host[k][1], t, seq.dets[k], seq.boxes3d[k], dynsim.INSTOBS_DTYPE, dynsim.BOX3D_DTYPE)
                else:
                    rows = trk.track_image(host[k][0], host[k][1], t)
                b = time.perf_counter()
                do_ba = (k % stride) == 0
                st = None
                if do_ba:
                    rc, st = est.process_dynamic(rows, t, io, fo, po) if args.mode == "dynamic" else est.process(rows, t)
                    if static_bg:
                        snaps = (snaps + [(k, est.static_instances())])[-4:]
                c = time.perf_counter()
                if st is not None and st.nonlinear:
                    poses.append(est.window()[10, :3].copy()); ptimes.append(t)
                if k >= 12 * stride:          # steady state only (window full, marginalization active), like the GPU timed region
                    t_fe += b - a; t_be += c - b; counted += 1
            return counted, t_fe, t_be, np.array(poses), ptimes

        n1, fe1, be1, poses_o, ptimes_o = cpu_run(1, args.cpu_frames)
        nN, feN, beN, _, _ = cpu_run(nthreads, args.cpu_frames)
        o.lib.dvo_set_threads(1)
        cpu = {"value": round(n1 / (fe1 + be1), 3), "unit": "frames/s", "cores": 1, "kind": "port",
               "value_all_cores": round(nN / (feN + beN), 3), "cores_all": nthreads, "host_cpu": cpu_model(), "host_nproc": ncores, "flags": "g++ " + flags,
               "front_end_ms": {"1_thread": round(fe1 / n1 * 1e3, 2), f"{nthreads}_threads": round(feN / nN * 1e3, 2)},
               "back_end_ms": {"1_thread": round(be1 / n1 * 1e3, 2), "4_threads_in_marginalization": round(beN / nN * 1e3, 2)},
               "sample": f"{n1} steady-state frames of the same {w}x{h} {args.mode} sequence: CPU oracle = restated reference path (LK + Shi-Tomasi + dense-Schur dogleg BA + "
                         f"marginalization{' + object branch' if args.mode == 'dynamic' else ''}); 'value' = 1 thread, 'value_all_cores' = LK over min(nproc, 16) threads + 4-thread marginalization (the reference's threading)"}
        # ATE of the HIP trajectory against the ORACLE trajectory on the frames both produced (north_star's acceptance figure; outside the timed region)
        # the reference's metric as its scripts compute it: nearest-stamp association within 0.02 s (associate.py), Horn alignment, RMSE (evaluate_ate.py)
        from dynamic_vins_amd import io_formats
        if len(ptimes_o) >= 3 and len(dev_times) >= 3:
            try:
                ate_vs_oracle, n_pairs = io_formats.evaluate_ate(ptimes_o, poses_o, dev_times, dev_poses[:, :3])
                ate_vs_oracle = float(ate_vs_oracle) if n_pairs >= 3 else None
            except ValueError:
                ate_vs_oracle = None

    if rank == 0:
        value = dv_dist.whole_job_rate(args.steps, world, dt)
        label = {"zed": "ZED intrinsics with distortion", "euroc": "EuRoC-like", "kitti": "KITTI-like, vision only"}[args.config]
        conf = {"workload": f"synthetic {w}x{h} stereo @20 Hz" + (" + IMU @200 Hz" if cfg["use_imu"] else "") + f", {label}, figure-8 in a textured box room"
                            + (f", {args.objects} textured boxes " + ("escorting the camera (in view in every frame)" if args.scene == "escort" else "moving through the room") + f" with instance masks (eroded {args.erode}x{args.erode}), 3-D detections and depth-sampled extra points (dynamic mode: TrackSemanticImage + InstsTrack + object solve)" if args.mode == "dynamic" else "")
                            + f", max_cnt {cfg['max_cnt']}, min_dist {cfg['min_dist']}, flow_back 1, {cfg['iters']} solver iterations, "
                            + ("BA + marginalization on every 2nd tracked frame (reference convention outside KITTI); a step = one BA frame = two tracked frames" if args.every_second_frame
                               else "BA + marginalization on every frame"),
                "mode": args.mode, "config": args.config,
                "parallelism": f"replicas x{world} (independent sequences, no collective)", "landmarks_in_window": n_lm, "features_per_frame": nfeat,
                "solver_iterations_per_frame": round(iters / (len(times) * args.steps), 2), "ate_rmse_m_vs_ground_truth": round(ate, 5),
                "ate_rmse_m_vs_oracle": None if ate_vs_oracle is None else float(f"{ate_vs_oracle:.3e}"),
                "second_block_ms_per_step": round(times[1] / args.steps * 1e3, 4), "second_block_value": round(dv_dist.whole_job_rate(args.steps, world, times[1]), 2),
                "block_values": [round(dv_dist.whole_job_rate(args.steps, world, t), 2) for t in times],
                "block_step_ms": frame_ms[: len(times)],
                "warmup_requested": args.warmup, "git_head": git_head(), "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                "host_loop": ("C++ (dv_runner)" if args.host_loop == "cpp" else "Python (pipeline.py)"),
                "frames": ("pinned host buffers: every frame (2 x %d bytes) crosses PCIe inside the timed region (%s)" % (w * h, "hipMemcpy2DAsync on the tracking stream" if os.environ.get("BENCH_HOST_ENGINE", "0") == "1" else "DV_MEM_PINNED: read in place by the pyramid kernel, no copy engine")) if args.host_frames else "resident in HBM before the timed region",
                "warmup_note": "at least 12 untimed BA frames: the sliding window (11 frames) must be full before a step is a steady-state step (track + BA + marginalization)"}
        if dyn_info:
            conf["dynamic"] = dyn_info
        out = {"metric": "stereo frames/sec (track+BA)", "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": warm_ba,
               "ms_per_step": round(dt / args.steps * 1e3, 4), "timed_region_s": round(dt, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic" if not args.host_frames else "synthetic (host buffers, PCIe upload inside the timed region)",
               "config": conf, "roofline": roof, "cpu_baseline": cpu}
        return out
    return None


if __name__ == "__main__":
    main()
