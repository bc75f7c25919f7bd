"""debug: three consecutive timed blocks of the raw pipeline, with and without the Python garbage collector"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamic_vins_amd import sim
from dynamic_vins_amd.pipeline import Pipeline, SyntheticSequence
steps, warm = 100, 20
seq = SyntheticSequence(1280, 720, sim.ZED, warm + 3 * steps + 2, rate=20.0, device="cuda:0")
for mode in ("gc on", "gc off", "gc on + short sleep between blocks"):
    pipe = Pipeline(seq, max_cnt=250, min_dist=25, max_iters=10)
    for _ in range(warm):
        pipe.step()
    if mode == "gc off":
        gc.disable()
    out = []
    for b in range(3):
        torch.cuda.synchronize(); pipe.ctx.sync()
        if mode.endswith("blocks"):
            time.sleep(0.5)
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.step()
        pipe.ctx.sync(); torch.cuda.synchronize()
        out.append(steps / (time.perf_counter() - t0))
    gc.enable()
    print(mode, [round(v, 1) for v in out])
    pipe.ctx.close()
