"""debug: the dynamic bench line cut into blocks as the driver's command cuts it (--steps 20 --warmup 5: 32 warm-up frames, two timed blocks of 20): wall time of every
dv_runner_run call and the per-frame end clocks (dv_runner_get_frame_clock) around the cuts.  usage: python scripts/dbg/dyn_block_clock.py [warm] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from dynamic_vins_amd import sim
from dynamic_vins_amd.backend import Runner
from dynamic_vins_amd.pipeline import DynamicPipeline, DynamicSequence
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N = warm + 3 * K
if os.environ.get('PRIMEQ'):          # make the runtime bring up its copy engines NOW: it creates an SDMA queue lazily (~6 ms, under a lock every other HIP call of the process waits for)
    _n = int(os.environ['PRIMEQ'])      # the first time that many copies are in flight at once
    _ss = [torch.cuda.Stream() for _ in range(_n)]
    _h = [torch.empty(8 << 20, dtype=torch.uint8).pin_memory() for _ in range(_n)]
    _d = [torch.empty(8 << 20, dtype=torch.uint8, device="cuda:0") for _ in range(_n)]
    for _k in range(3):
        for _s, _a, _b in zip(_ss, _h, _d):
            with torch.cuda.stream(_s):
                _b.copy_(_a, non_blocking=True); _a.copy_(_b, non_blocking=True)
    torch.cuda.synchronize()
seq = DynamicSequence(1280, 720, sim.ZED, N + 2, rate=20.0, device="cuda:0", boxes=("escort", 4))
pipe = DynamicPipeline(seq, max_cnt=250, min_dist=25, max_iters=10, device=0, mask_morphology_size=5)
r = Runner([pipe])
if os.environ.get('PRIME'):          # the first cut (device-wide synchronisation) inside the warm-up: r.run(warm - 2), sync, r.run(2)
    r.run(warm - 2); torch.cuda.synchronize(); pipe.ctx.sync(); r.run(2)
else:
    r.run(warm)
idle = float(os.environ.get('IDLE_MS', '0')) / 1e3      # an idle gap in front of every block (the bench's gc.collect() pause is ~50 ms)
walls = []
for b in range(3):
    if idle: time.sleep(idle)
    torch.cuda.synchronize(); pipe.ctx.sync()
    t0 = time.perf_counter(); r.run(K); pipe.ctx.sync(); torch.cuda.synchronize(); walls.append(time.perf_counter() - t0)
clk = np.asarray(r.frame_clock(0)); tc = np.asarray(r.frame_clock(0, 1))
d = np.diff(clk) * 1e3
print("blocks of %d frames behind %d warm-up frames: " % (K, warm) + " ".join("%.2f ms = %.0f frames/s" % (1e3 * w, K / w) for w in walls))
for b in range(3):
    a = warm + b * K
    print("block %d frames %d..%d end-to-end deltas (ms): " % (b, a, a + K - 1) + " ".join("%.2f" % v for v in d[a - 1:a + K - 1]))
    print("   first frame of the block: run() call start -> frame end %.2f ms" % ((clk[a] - clk[a - 1]) * 1e3))
dt = np.diff(tc) * 1e3
for i, v in enumerate(d):
    f = i + 1
    if v > 1.6 and f >= 14 and not idle:
        print("slow frame %d: estimator delta %.2f ms | tracker delivery deltas of frames %d..%d: %s | estimator end - tracker delivery of that frame: %.2f ms" % (f, v, f - 1, f + 1, " ".join("%.2f" % x for x in dt[f - 2:f + 1]), (clk[f] - tc[f]) * 1e3))
r.close(); pipe.ctx.close()
