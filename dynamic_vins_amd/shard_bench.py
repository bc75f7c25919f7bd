"""bench.py --shard: ONE sequence whose window solve is sharded by landmark over the N ranks (include/dvins.h dv_dist_*).

Every rank runs the same front end on the same frames (the tracker is not sharded: ~0.5 ms of latency-bound work) and owns 1/N of the landmarks of every
window solve; per linearisation the ranks all-gather one exchange vector (partial reduced camera system + the packet rows be_solve needs) and sum in rank
order.  `value` is the frame rate of the ONE sequence ("scaling": "strong"); the same line carries the unsharded rate measured in the same process, so the
cost of sharding a ~300-landmark window is on record: it is NOT expected to pay — the deliverable is correctness plus this curve."""
import gc
import json
import os
import time

import torch

from . import dist as dv_dist


def run_shard_bench(args, rank, world, local_rank):
    from bench import CONFIGS, git_head
    from . import sim
    from .pipeline import Pipeline, SyntheticSequence
    cfg = CONFIGS[args.config]
    w, h = cfg["w"], cfg["h"]
    warm_ba = max(args.warmup, 12)
    n_frames = warm_ba + 2 * args.steps + 2
    cam = sim.ZED if (w, h) == (1280, 720) else sim.scaled_cam(sim.ZED, w, h, 1280, 720)
    dev = f"cuda:{local_rank}"
    seq = SyntheticSequence(w, h, cam, n_frames, rate=20.0, phase=0.0, device=dev)         # the SAME sequence on every rank
    gloo = os.environ.get("BENCH_BACKEND", "nccl") == "gloo"

    def run(sharded, collective):
        pipe = Pipeline(seq, max_cnt=cfg["max_cnt"], min_dist=cfg["min_dist"], max_iters=cfg["iters"], device=local_rank, use_imu=cfg["use_imu"])
        if sharded:
            dv_dist.shard_window(pipe.ctx, rank, world, transport=getattr(args, "shard_transport", None) or ("host" if gloo else "rccl"))
        for _ in range(warm_ba):
            pipe.step()
        gc.collect(); gc.freeze()      # no generation-2 pause of the interpreter inside a timed block (bench.py)
        times = []
        for _b in range(2):
            if collective:
                dv_dist.barrier()
            torch.cuda.synchronize(); pipe.ctx.sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                pipe.step()
            pipe.ctx.sync(); torch.cuda.synchronize()
            if collective:
                dv_dist.barrier()
            times.append(time.perf_counter() - t0)
        return pipe, times

    pipe, times = run(True, True)
    red_dev = "cpu" if gloo else dev
    times = [dv_dist.max_over_ranks(t, device=red_dev) for t in times]
    info = dv_dist.dist_info(pipe.ctx)
    import numpy as np
    traj = np.array(pipe.poses) if len(pipe.poses) else np.zeros((0, 7))
    digest = float(traj[:, :3].sum()) if len(traj) else 0.0
    same = dv_dist.max_over_ranks(digest, device=red_dev) == -dv_dist.max_over_ranks(-digest, device=red_dev)      # every rank ended on the same trajectory bits
    ate = pipe.ate()
    n_lm = int(pipe.est.state.n_long)
    frames_total = warm_ba + 2 * args.steps
    pipe.ctx.close()
    plain_times, plain_digest = None, None
    if rank == 0:
        p2, plain_times = run(False, False)
        t2 = np.array(p2.poses) if len(p2.poses) else np.zeros((0, 7))
        plain_digest = float(abs(t2[:, :3] - traj[:, :3]).max()) if len(t2) == len(traj) and len(traj) else None
        p2.ctx.close()
        dt = times[0]
        cap = max(1, -(-n_lm // world))
        xb = dict(system=12424 * 8, cost=64, depth=8 * cap)      # dv_dist_exchange_bytes: per linearisation (independent of the landmark count) | per cost-only slot | once per solve
        out = {"metric": "stereo frames/sec (track+BA)", "value": round(args.steps / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": warm_ba,
               "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"synthetic {w}x{h} stereo @20 Hz" + (" + IMU @200 Hz" if cfg["use_imu"] else "") + f", max_cnt {cfg['max_cnt']}, min_dist {cfg['min_dist']}, {cfg['iters']} solver iterations, BA + marginalization on every frame",
                          "mode": "raw", "config": args.config, "git_head": git_head(),
                          "parallelism": f"ONE sequence; its window solve sharded by landmark over {world} rank(s): {info['transport']} all-gather + rank-ordered sum per linearisation; front end replicated",
                          "transport": info["transport"], "rccl_ranks": info["rccl_ranks"], "landmarks_in_window": n_lm, "landmarks_per_rank": cap, "exchanges_per_frame": round(info["exchanges"] / frames_total, 2),
                          "exchange_bytes_per_rank": xb["system"], "exchange_bytes_detail": xb, "all_ranks_same_trajectory_bits": bool(same), "ate_rmse_m_vs_ground_truth": round(ate, 5),
                          "second_block_value": round(args.steps / times[1], 2),
                          "unsharded_value_same_process": round(args.steps / plain_times[0], 2), "unsharded_second_block_value": round(args.steps / plain_times[1], 2),
                          "max_abs_position_diff_vs_unsharded_m": plain_digest},
               "roofline": None, "cpu_baseline": None}
        print(json.dumps(out))
