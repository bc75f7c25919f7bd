"""debug: per-block frame rate AND per-block kernel averages (HIP events) of the raw pipeline on a 222-frame sequence (the default bench allocation)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamic_vins_amd import sim
from dynamic_vins_amd.pipeline import Pipeline, SyntheticSequence
steps, warm = 100, 20
seq = SyntheticSequence(1280, 720, sim.ZED, warm + 2 * steps + 2, rate=20.0, device="cuda:0")
for timing in (0, 2):
    pipe = Pipeline(seq, max_cnt=250, min_dist=25, max_iters=10)
    for _ in range(warm):
        pipe.step()
    if timing:
        pipe.ctx.timing_enable(timing)
    for b in range(2):
        torch.cuda.synchronize(); pipe.ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.step()
        pipe.ctx.sync(); torch.cuda.synchronize()
        fps = steps / (time.perf_counter() - t0)
        line = f"timing {timing} block {b}: {fps:.1f} fps"
        if timing:
            ks = {}
            for nm in ["k_be_solve", "k_be_reduce", "k_be_eval_full", "k_be_marg", "pyr", "lk_temporal", "gftt_select", "lk_stereo"]:
                ms, cnt = pipe.ctx.timing_get(nm)
                if cnt:
                    ks[nm] = round(ms / cnt * 1e3, 1)
            line += f"  {ks}"
            pipe.ctx.timing_reset()
        print(line)
    pipe.ctx.close()
