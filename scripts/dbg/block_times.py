"""debug: consecutive timed blocks of the raw pipeline under variations of the harness (sequence length, pose bookkeeping)"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamic_vins_amd import sim
from dynamic_vins_amd.pipeline import Pipeline, SyntheticSequence
steps, warm = 100, 20
for nblocks, extra in ((2, 2), (3, 2), (2, 200)):
    seq = SyntheticSequence(1280, 720, sim.ZED, warm + nblocks * steps + extra, rate=20.0, device="cuda:0")
    pipe = Pipeline(seq, max_cnt=250, min_dist=25, max_iters=10)
    for _ in range(warm):
        pipe.step()
    out = []
    for b in range(nblocks):
        torch.cuda.synchronize(); pipe.ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.step()
        pipe.ctx.sync(); torch.cuda.synchronize()
        out.append(steps / (time.perf_counter() - t0))
    print(f"blocks {nblocks}, frames in sequence {len(seq.frames)}:", [round(v, 1) for v in out])
    pipe.ctx.close()
    del seq, pipe
    torch.cuda.empty_cache()
