"""bench.py --sequences S: S independent sequences on ONE GPU, driven by ONE host thread (SURVEY 8(d) config 4's "batched" form).

The window solve of a single sequence is a chain of ~30 latency-bound launches that occupy a handful of the 256 CUs; the other sequences' chains run beside
it on their own streams.  The host interleaves the sequences through the two-phase API (Pipeline.step_begin / step_end = dv_track_stereo_enqueue / _collect +
dv_est_process_begin / _end): while sequence A's BA is in flight it prepares and enqueues B, C, ...  `value` is the AGGREGATE frame rate of the S sequences
on this GPU (x N ranks for --gpus N); the single-sequence rate stays bench.py's default line."""
import gc
import json
import os, sys
import time

import torch

from . import dist as dv_dist


def run_multiseq_bench(args, rank, world, local_rank):
    from bench import CONFIGS, git_head
    from . import sim
    from .pipeline import Pipeline, SyntheticSequence
    cfg = CONFIGS[args.config]
    S = args.sequences
    w, h = cfg["w"], cfg["h"]
    warm_ba = max(args.warmup, 12)
    n_frames = warm_ba + 2 * args.steps + 2
    cam = sim.ZED if (w, h) == (1280, 720) else sim.scaled_cam(sim.ZED, w, h, 1280, 720)
    dev = f"cuda:{local_rank}"
    seqs = [SyntheticSequence(w, h, cam, n_frames, rate=20.0, phase=dv_dist.sequence_phase(rank * S + i), device=dev) for i in range(S)]
    pipes = [Pipeline(q, max_cnt=cfg["max_cnt"], min_dist=cfg["min_dist"], max_iters=cfg["iters"], device=local_rank, use_imu=cfg["use_imu"]) for q in seqs]

    if getattr(args, "runner", "cpp") == "cpp" and not getattr(args, "sequence_threads", False):
        return run_cpp_runner(args, cfg, seqs, pipes, rank, world, local_rank, dev)
    threaded = bool(getattr(args, "sequence_threads", False))
    batch = None; groups = None
    if getattr(args, "batched", False):
        from .backend import Batch
        if threaded:     # one host thread per sequence for the host phases, meeting at a barrier around the single dv_batch_enqueue of the round
            batch = Batch([p.ctx for p in pipes])
        elif S >= 4:       # two alternating groups: while one group's batched solve runs on the GPU, the host collects / prepares the other one
            half = S // 2
            groups = [(pipes[:half], Batch([p.ctx for p in pipes[:half]])), (pipes[half:], Batch([p.ctx for p in pipes[half:]]))]
        else:
            batch = Batch([p.ctx for p in pipes])      # the window solves of all sequences share every launch (dv_batch)
    import threading

    pending = [False, False]

    def round_robin():
        if groups is not None:
            for gi, (gp, gb) in enumerate(groups):
                if pending[gi]:
                    for p in gp:
                        p.step_end()
                for p in gp:
                    p.step_begin()
                gb.enqueue()
                pending[gi] = True
            return
        for p in pipes:
            p.step_begin()
        if batch is not None:
            batch.enqueue()
        for p in pipes:
            p.step_end()

    def drain():
        if groups is not None:
            for gi, (gp, gb) in enumerate(groups):
                if pending[gi]:
                    for p in gp:
                        p.step_end()
                    pending[gi] = False

    def run_block(n_steps):
        """n_steps frames of every sequence: interleaved on this thread, or (--sequence-threads) one host thread per sequence — the C ABI releases the GIL,
        so the per-frame host work of the sequences (feature manager, problem assembly: ~0.5 ms) runs in parallel like the reference's per-process threads"""
        if not threaded:
            for _ in range(n_steps):
                round_robin()
            drain()
            return

        def worker(p):
            for _ in range(n_steps):
                if batch is None:
                    p.step()
                    continue
                p.step_begin()
                batch.arrive()          # rendezvous inside the library: the last thread to arrive enqueues the shared launches
                p.step_end()
        ths = [threading.Thread(target=worker, args=(p,)) for p in pipes]
        for t in ths:
            t.start()
        for t in ths:
            t.join()

    run_block(warm_ba)
    gc.collect(); gc.freeze()      # no generation-2 pause of the interpreter inside a timed block (bench.py)
    times = []
    for _b in range(2):
        dv_dist.barrier()
        torch.cuda.synchronize()
        for p in pipes:
            p.ctx.sync()
        t0 = time.perf_counter()
        run_block(args.steps)
        for p in pipes:
            p.ctx.sync()
        torch.cuda.synchronize()
        dv_dist.barrier()
        times.append(time.perf_counter() - t0)
    red_dev = dev if os.environ.get("BENCH_BACKEND", "nccl") != "gloo" else "cpu"
    times = [dv_dist.max_over_ranks(t, device=red_dev) for t in times]
    ates = [p.ate() for p in pipes]
    if rank == 0:
        dt = times[0]
        out = {"metric": "stereo frames/sec (track+BA)", "value": round(world * S * args.steps / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": warm_ba,
               "ms_per_step": round(dt / args.steps * 1e3, 4), "timed_region_s": round(dt, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"{S} independent synthetic {w}x{h} stereo sequences per GPU @20 Hz" + (" + IMU @200 Hz" if cfg["use_imu"] else "")
                                      + f", max_cnt {cfg['max_cnt']}, min_dist {cfg['min_dist']}, {cfg['iters']} solver iterations, BA + marginalization on every frame; a step = one frame of EVERY sequence",
                          "mode": "raw", "config": args.config, "git_head": git_head(), "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "sequences_per_gpu": S,
                          "parallelism": f"{S} sequences per GPU " + ("on one host thread each" if threaded else "interleaved by one host thread") + (", window solves batched (dv_batch: one launch per stage for all sequences)" if batch is not None else "") + f", own HIP streams, x {world} GPU(s); no collective",
                          "batch": (None if batch is None else batch.info()) if groups is None else dict(groups=2, **groups[0][1].info()),
                          "per_sequence_value": round(args.steps / dt, 2), "second_block_value": round(world * S * args.steps / times[1], 2),
                          "ate_rmse_m_vs_ground_truth_max": round(max(ates), 5)},
               "roofline": None, "cpu_baseline": None}
        print(json.dumps(out))
    if batch is not None:
        batch.close()
    if groups is not None:
        for _, gb in groups:
            gb.close()
    for p in pipes:
        p.ctx.close()


def run_cpp_runner(args, cfg, seqs, pipes, rank, world, local_rank, dev):
    """the same measurement with the host loop in C++ (dv_runner, csrc/runner.hip): one call per timed block; --batched groups the sequences into dv_batch groups of
    --group-size (default: four groups), --runner-threads host threads drive the groups (default and maximum: one per group)"""
    from bench import git_head
    from .backend import Runner
    from .pipeline import Pipeline
    S, w, h = args.sequences, cfg["w"], cfg["h"]
    warm_ba = max(args.warmup, 12)
    gsz = 0
    if getattr(args, "batched", False):
        # default: FOUR groups (one host thread each) — measured behind the per-sequence gate, interleaved on one box (round 4): 16 sequences in groups of 4 / 8 / 2:
        # 5896 - 5919 / 5058 - 5066 / 4306 frames/s; 32 in groups of 8 / 16: 7497 / 5529; fewer than 8 sequences: two groups
        gsz = args.group_size if getattr(args, "group_size", 0) > 0 else ((S + 3) // 4 if S >= 8 else ((S + 1) // 2 if S >= 4 else S))
    n_groups = (S + gsz - 1) // gsz if gsz > 0 else S
    threads = getattr(args, "runner_threads", 0)
    teams = gsz > 1 and not bool(getattr(args, "no_teams", False))
    if threads <= 0:          # default: TWO host threads per dv_batch group (a team splits its members' host phases; --no-teams: one thread per group); EIGHT for groups of 16 and more
        # (round 6, scripts/dbg/ab_threads.sh on one box, twice: 64 sequences 9346 / 9484 with 8 threads, 9906 / 9730 with 16, 10 015 / 10 141 with 32; 16 sequences in groups of 4: no difference)
        threads = ((8 if gsz >= 16 else 2) * n_groups if teams else n_groups) if gsz > 1 else 1
    if not teams:
        threads = min(threads, n_groups)
    args.runner_threads = threads
    runner = Runner(pipes, group_size=gsz, threads=max(1, threads))
    runner.set("teams", 1 if teams else 0)
    if getattr(args, "no_batch_front", False):
        runner.set("batch_front", 0)
    runner.run(warm_ba)          # (same cuts as the solo reference run of the bit-identity gate below: warm-up, block, block)
    if gsz > 1:
        runner.batch_timing(1)       # HIP events on the batch streams around one steady-state slot per round (dv_batch_timing)
    gc.collect(); gc.freeze()
    times = []
    for _b in range(2):
        dv_dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.run(args.steps)
        torch.cuda.synchronize(); dv_dist.barrier()
        times.append(time.perf_counter() - t0)
    red_dev = dev if os.environ.get("BENCH_BACKEND", "nccl") != "gloo" else "cpu"
    times = [dv_dist.max_over_ranks(t, device=red_dev) for t in times]
    from . import sim
    ates, iters = [], 0
    for i, p in enumerate(pipes):
        st, poses, it, fr = runner.get(i)
        gt = [p.seq.traj.p(t) for t in poses[:, 0]]
        ates.append(sim.align_ate(poses[:, 1:4], gt)[0]); iters += it
    n_total = S * (warm_ba + 2 * args.steps)
    # A throughput figure from a run whose trajectories are wrong measures nothing.  These synthetic sequences track to millimetres (worst phase ~ 9 mm over 100 frames); a
    # sequence more than ATE_GATE_M off its ground truth means the run is corrupted (this is how the team-path defect of round 4 showed: 0.04 - 7.3 m on single members):
    # no JSON line, non-zero exit.  BENCH_ATE_GATE_M overrides the bound (debug runs that WANT to see the corrupted figures).
    gate = float(os.environ.get("BENCH_ATE_GATE_M", "0.03"))
    bad = [(i, round(float(v), 5)) for i, v in enumerate(ates) if not (v <= gate)]
    if bad:
        sys.stderr.write(f"multiseq_bench: trajectories of {len(bad)} of {S} sequences are off their ground truth by more than {gate} m: {bad} -- the run is corrupted, no result is reported\n")
        runner.close()
        raise SystemExit(3)
    # Bit-identity of two sampled members (first and last) against the SINGLE-THREAD, UNBATCHED run of the same sequence over the same frames: the ground-truth gate above
    # passes a 1e-6 m corruption, this does not (round 4's race showed as 1e-8 .. 1e-6 m differences long before it showed as metres).  Same call pattern (warm-up, two
    # blocks) so the drains fall on the same frames.  A difference means no line (exit 3).
    bit = None
    if rank == 0 and os.environ.get("BENCH_BIT_IDENTITY", "1") == "1":
        import hashlib
        import numpy as np
        members, equal, hashes = sorted({0, S - 1}), True, []
        for i in members:
            got = runner.frames(i)
            solo_pipe = Pipeline(seqs[i], max_cnt=cfg["max_cnt"], min_dist=cfg["min_dist"], max_iters=cfg["iters"], device=local_rank, use_imu=cfg["use_imu"])
            solo = Runner([solo_pipe], group_size=0, threads=1)
            solo.run(warm_ba); solo.run(args.steps); solo.run(args.steps)
            want = solo.frames(0)
            solo.close(); solo_pipe.ctx.close()
            same = got.shape == want.shape and np.array_equal(got, want)
            equal = equal and same
            hashes.append(hashlib.sha1(np.ascontiguousarray(got).tobytes()).hexdigest()[:16])
        bit = {"members": members, "equal_to_single_thread_unbatched_run": bool(equal), "frames_compared": int(warm_ba + 2 * args.steps), "sha1_16_of_rows": hashes}
        if not equal:
            sys.stderr.write(f"multiseq_bench: members {members} differ bit-wise from the single-thread unbatched run of the same sequences -- no result is reported\n")
            runner.close()
            raise SystemExit(3)
    roof = None
    if rank == 0 and gsz > 1:
        # the batched window solve's stage launches: [be_solve_batch, be_eval_batch (full), be_reduce_batch] x `wins` windows per launch.  ALGORITHMIC bytes per
        # window as for the single-window kernels: evaluation = factor records read + one 928-double packet per landmark written; reduce = every packet
        # read once + Hd, Sc written; solve = Hd + Sc once, three passes over (w[66], h, g, scale) per landmark.
        t_ms, rounds, wins = runner.batch_timing(1)
        L = float(sum(runner.get(i)[0].n_landmarks for i in range(S))) / S
        n_state = 165 if cfg["use_imu"] else 60
        alg = {"be_solve_batch": 2 * n_state * n_state * 8 + 3 * L * 69 * 8, "be_eval_batch": L * (16 + 928 * 8) + L * 20 * 112, "be_reduce_batch": L * 928 * 8 + 2 * n_state * n_state * 8}
        names = list(alg)
        if rounds > 0 and wins > 0:
            from bench import HBM_PEAK_GBS, ROOT, csrc_digest
            dom = names[max(range(3), key=lambda k: t_ms[k])]
            stage = {n: {"avg_launch_us": round(t_ms[k] * 1e3, 2), "algorithmic_bytes_per_launch": int(alg[n] * wins),
                         "achieved_GBs": round(alg[n] * wins / (t_ms[k] * 1e-3) / 1e9, 3) if t_ms[k] > 0 else None} for k, n in enumerate(names)}
            traffic, stale = None, None
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_batched.json")))
                hit = [v for k, v in pmc["kernels"].items() if k.startswith(dom.replace("_batch", "_batch_kernel"))]
                if hit and pmc.get("windows_per_launch") == wins:
                    traffic = int(hit[0]["traffic_bytes"])
                stale = pmc.get("csrc_digest") != csrc_digest()
            except (OSError, KeyError, ValueError):
                pass
            ach = stage[dom]["achieved_GBs"]
            roof = {"kernel": dom, "bound": "hbm", "bound_note": "the batched stage launches: one workgroup (solve) or one grid slice (evaluation, reduce) per window, `windows_per_launch` windows in one launch",
                    "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None if ach is None else round(ach / HBM_PEAK_GBS, 6),
                    "traffic": None if stale else traffic, "traffic_stale": stale, "avg_launch_us": stage[dom]["avg_launch_us"],
                    "algorithmic_bytes_per_launch": stage[dom]["algorithmic_bytes_per_launch"], "windows_per_launch": wins, "landmarks_per_window": round(L, 1),
                    "rounds_timed": int(rounds), "stages": stage}
    if rank == 0:
        dt = times[0]
        out = {"metric": "stereo frames/sec (track+BA)", "value": round(world * S * args.steps / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": warm_ba,
               "ms_per_step": round(dt / args.steps * 1e3, 4), "timed_region_s": round(dt, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"{S} independent synthetic {w}x{h} stereo sequences per GPU @20 Hz" + (" + IMU @200 Hz" if cfg["use_imu"] else "")
                                      + f", max_cnt {cfg['max_cnt']}, min_dist {cfg['min_dist']}, {cfg['iters']} solver iterations, BA + marginalization on every frame; a step = one frame of EVERY sequence",
                          "mode": "raw", "config": args.config, "git_head": git_head(), "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "sequences_per_gpu": S, "host_loop": "C++ (dv_runner)",
                          "parallelism": f"{S} sequences per GPU, host loop in C++ on {max(1, getattr(args, 'runner_threads', 1))} thread(s)" + (f" ({max(1, getattr(args, 'runner_threads', 1)) // n_groups} per group)" if gsz > 1 and getattr(args, 'runner_threads', 1) > n_groups else "")
                                         + (f", window solves batched in dv_batch groups of {gsz} (one launch per stage for a group)" if gsz > 1 else ", every sequence on its own streams") + f", x {world} GPU(s); no collective",
                          "group_size": gsz, "runner_threads": max(1, getattr(args, "runner_threads", 1)), "host_threads_per_group": (max(1, threads) // n_groups if teams and gsz > 1 else 1), "teams": teams, "bit_identity": bit, "front_end_launches": dict(shared=not getattr(args, "no_batch_front", False) and gsz > 1, **runner.track_info()),
                          "per_sequence_value": round(args.steps / dt, 2), "second_block_value": round(world * S * args.steps / times[1], 2),
                          "solver_iterations_per_frame": round(iters / max(n_total, 1), 2), "ate_rmse_m_vs_ground_truth_max": round(max(ates), 5),
                          "ate_rmse_m_vs_ground_truth_per_sequence": [round(float(v), 5) for v in ates]},
               "roofline": roof, "cpu_baseline": None}
        print(json.dumps(out))
    runner.close()
    for p in pipes:
        p.ctx.close()
