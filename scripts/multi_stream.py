"""throughput with S independent sequences on ONE GPU (one host thread + one ctx each; ctypes releases the GIL inside the
library): the per-sequence rate is latency-bound (one 1024-thread workgroup solves), so several sequences fill the idle CUs.
Reported next to, never instead of, the single-sequence headline of bench.py."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_vins_amd import sim
from dynamic_vins_amd.pipeline import Pipeline, SyntheticSequence

S_LIST = [int(x) for x in (sys.argv[1:] or ["1", "2", "4", "8"])]
STEPS, WARM = 80, 14
seqs = {}
for S in S_LIST:
    for k in range(S):
        if k not in seqs:
            seqs[k] = SyntheticSequence(1280, 720, sim.ZED, WARM + STEPS + 1, rate=20.0, phase=1.7 * k, device="cuda:0")
for S in S_LIST:
    pipes = [Pipeline(seqs[k]) for k in range(S)]
    for p in pipes:
        for _ in range(WARM):
            p.step()
    bar = threading.Barrier(S + 1)
    def run(p):
        bar.wait()
        for _ in range(STEPS):
            p.step()
        p.ctx.sync()
    th = [threading.Thread(target=run, args=(p,)) for p in pipes]
    for t in th: t.start()
    torch.cuda.synchronize()
    bar.wait(); t0 = time.perf_counter()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("sequences %d: %.1f frames/s aggregate (%.1f per sequence), ATE %s" % (S, S * STEPS / dt, STEPS / dt, ["%.4f" % p.ate() for p in pipes][:2]))
    for p in pipes: p.ctx.close()
