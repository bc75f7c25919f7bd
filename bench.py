#!/usr/bin/env python3
"""bench.py — stereo frames/sec through track + BA (BASELINE.json metric) on synthetic 1280x720 stereo + IMU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one stereo frame through the HIP front end (FeatureTracker::TrackImage) and the HIP back end
(Estimator::ProcessImage, bundle adjustment + marginalization on EVERY frame, i.e. the KITTI convention; the
reference forwards only every 2nd frame on the other datasets, system/main.cpp:300-307).  Frames are rendered once
and are resident in HBM before the timed region.  N > 1: every rank runs an independent sequence (replicas — the
single-window problem has ~300 landmarks and does not shard usefully, DESIGN.md "Multi-GPU"); value is the
whole-job rate, scaling is weak, no data-path collective.

Prints ONE JSON line on rank 0, with `roofline` (dominant kernel, HIP-event timed per launch on its own stream)
and `cpu_baseline` (the CPU oracle = restated reference path, bounded sample, rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--max-cnt", type=int, default=250)
    ap.add_argument("--min-dist", type=int, default=25)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=16)
    ap.add_argument("--host-frames", action="store_true", help="frames handed over as host buffers: the PCIe-inclusive rate (never the headline value)")
    args = ap.parse_args()

    import numpy as np
    import torch

    from dynamic_vins_amd import dist as dv_dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one process per GPU.  BENCH_DEVICE / BENCH_BACKEND exist only to rehearse the multi-rank control flow on a 1-GPU box
    # (both ranks on device 0, gloo instead of RCCL); the driver's 2/4/8-GPU runs use LOCAL_RANK and "nccl" = RCCL over xGMI.
    forced_dev = os.environ.get("BENCH_DEVICE")
    torch.cuda.set_device(int(forced_dev) if forced_dev is not None else int(os.environ.get("LOCAL_RANK", "0")))
    rank, world, local_rank = dv_dist.init(prefer_gpu=os.environ.get("BENCH_BACKEND", "nccl") != "gloo")
    if forced_dev is not None:
        local_rank = int(forced_dev)

    from dynamic_vins_amd import sim
    from dynamic_vins_amd.pipeline import Pipeline, SyntheticSequence

    warmup = max(args.warmup, 12)          # the estimator needs kWinSize+1 = 11 frames to initialise (estimator.cpp:1464-1483)
    n_frames = warmup + args.steps + 1
    cam = sim.ZED if args.width == 1280 else sim.scaled_cam(sim.ZED, args.width, args.height, 1280, 720)
    seq = SyntheticSequence(args.width, args.height, cam, n_frames, rate=20.0, phase=dv_dist.sequence_phase(rank), device=f"cuda:{local_rank}")

    def run(kernel_timing, collective=True):       # collective=False: rank-0-only pass, no barriers
        pipe = Pipeline(seq, max_cnt=args.max_cnt, min_dist=args.min_dist, max_iters=args.iters, device=local_rank, host_frames=args.host_frames)
        for _ in range(warmup):
            pipe.step()
        if kernel_timing:
            pipe.ctx.timing_enable(2)
        if collective:
            dv_dist.barrier()
        torch.cuda.synchronize()
        pipe.ctx.sync()
        t0 = time.perf_counter()
        iters = 0
        for _ in range(args.steps):
            st = pipe.step()
            iters += st.iterations
        pipe.ctx.sync()
        torch.cuda.synchronize()
        if collective:
            dv_dist.barrier()
        dt = time.perf_counter() - t0
        return pipe, dt, iters

    pipe, dt, iters = run(False)
    dt = dv_dist.max_over_ranks(dt, device=f"cuda:{local_rank}" if os.environ.get("BENCH_BACKEND", "nccl") != "gloo" else "cpu")
    ate = pipe.ate()
    n_lm = int(pipe.est.state.n_long)
    nfeat = len(pipe.rows)
    pipe.ctx.close()

    # ---- roofline of the dominant kernel: second, instrumented pass over the same timed region ----
    roof = None
    kern = {}
    if rank == 0:
        pipe2, _, _ = run(True, collective=False)
        names = ["k_be_solve", "k_be_reduce", "k_be_eval_full", "k_be_eval_cost", "k_be_accept", "k_be_marg",
                 "pyr", "lk_temporal", "compact", "gftt_eig", "gftt_select", "lk_stereo", "finalize"]
        for nme in names:
            ms, cnt = pipe2.ctx.timing_get(nme)
            if cnt:
                kern[nme] = dict(total_ms=ms, launches=cnt, avg_us=ms / cnt * 1e3)
        dom = max(kern, key=lambda k: kern[k]["total_ms"])
        n_state, L = 165, n_lm
        # algorithmic bytes per launch (DESIGN.md "Kernels and rooflines")
        alg = {
            "k_be_solve": 2 * n_state * n_state * 8 + 3 * L * 69 * 8,          # Hd + Sc once, three passes over (w[66], h, g, scale) per landmark
            "k_be_reduce": L * 928 * 8 + 2 * n_state * n_state * 8,            # every packet once + Hd, Sc written
            "k_be_eval_full": L * (16 + 928 * 8) + L * 20 * 112,               # factor records read, packet written
            "k_be_marg": 97 * 97 * 8 * 2 + 150 * 20 * 112,
            "lk_temporal": nfeat * 6 * 2 * 529, "lk_stereo": nfeat * 6 * 2 * 529,
            "gftt_eig": 2 * args.width * args.height, "pyr": int(2 * 1.328 * args.width * args.height),
        }.get(dom, 0)
        avg_s = kern[dom]["avg_us"] * 1e-6
        achieved = alg / avg_s / 1e9 if avg_s > 0 else 0.0
        # HBM-side bytes per launch from the committed PMC passes (profiles/pmc_traffic.json; a PMC run cannot share a process with the timed run)
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"]
            key = {"k_be_solve": "be_solve_kernel<1>", "k_be_reduce": "be_reduce_kernel", "k_be_eval_full": "be_eval_kernel<true>", "k_be_eval_cost": "be_eval_kernel<false>",
                   "lk_temporal": "lk_track_kernel", "lk_stereo": "lk_track_kernel", "gftt_eig": "gftt_tile_kernel", "gftt_select": "gftt_select_kernel", "pyr": "pyr_down_kernel"}.get(dom)
            if key in pmc:
                traffic = int(pmc[key]["traffic_bytes"])
        except (OSError, KeyError, ValueError):
            traffic = None
        roof = {"kernel": (dom[2:] if dom.startswith("k_") else dom), "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "avg_launch_us": round(kern[dom]["avg_us"], 2),
                "algorithmic_bytes_per_launch": int(alg),
                "kernels_us": {(k[2:] if k.startswith("k_") else k): round(v["avg_us"], 1) for k, v in kern.items()}}
        pipe2.ctx.close()

    # ---- CPU baseline: the oracle (restated reference path) on a bounded sample of the same workload ----
    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        from tests import oracle_py
        o = oracle_py.load()
        camt = sim.cam_tuple(cam)
        trk = o.tracker(args.width, args.height, args.max_cnt, args.min_dist, 1, 1, camt, camt)
        est = o.estimator(use_imu=1, stereo=1, max_iters=args.iters, ric=[sim.R_IC, sim.R_IC], tic=[sim.T_IC0, sim.T_IC1], **seq.noise)
        nf = min(args.cpu_frames + 12, n_frames)
        host = [seq.host_frame(k) for k in range(nf)]
        k_imu, t_cpu, counted = 0, 0.0, 0
        for k in range(nf):
            t = seq.times[k]
            while k_imu < len(seq.imu_t) and seq.imu_t[k_imu] <= t + 0.006:
                est.input_imu(seq.imu_t[k_imu], seq.imu_a[k_imu], seq.imu_g[k_imu])
                k_imu += 1
            a = time.perf_counter()
            rows = trk.track_image(host[k][0], host[k][1], t)
            rc, st = est.process(rows, t)
            b = time.perf_counter()
            if k >= 12:            # steady state only (window full, marginalization active), like the GPU timed region
                t_cpu += b - a
                counted += 1
        cpu = {"value": round(counted / t_cpu, 3), "unit": "frames/s", "cores": 1, "kind": "port",
               "sample": f"{counted} steady-state frames of the same {args.width}x{args.height} sequence (CPU oracle: LK + Shi-Tomasi + dense-Schur dogleg BA + marginalization, single thread, g++ -O2)"}

    if rank == 0:
        value = dv_dist.whole_job_rate(args.steps, world, dt)
        out = {
            "metric": "stereo frames/sec (track+BA)", "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic" if not args.host_frames else "synthetic (host buffers, PCIe upload inside the timed region)",
            "config": {"workload": f"synthetic {args.width}x{args.height} stereo @20 Hz + IMU @200 Hz, ZED intrinsics with distortion, figure-8 in a textured box room, "
                                   f"max_cnt {args.max_cnt}, min_dist {args.min_dist}, flow_back 1, {args.iters} solver iterations, BA + marginalization on every frame",
                       "parallelism": f"replicas x{world} (independent sequences, no collective)", "landmarks_in_window": n_lm, "features_per_frame": nfeat,
                       "solver_iterations_per_frame": round(iters / args.steps, 2), "ate_rmse_m_vs_ground_truth": round(ate, 5),
                       "warmup_requested": args.warmup,
                       "warmup_note": "at least 12 untimed frames: the sliding window (11 frames) must be full before a step is a steady-state step (track + BA + marginalization)"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    dv_dist.barrier()          # every rank stays until rank 0 has finished its extra passes
    dv_dist.finalize()


if __name__ == "__main__":
    main()
