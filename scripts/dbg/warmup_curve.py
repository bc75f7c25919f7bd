"""debug: per-frame wall time of the C++ runner from the first frame on (run(1) per frame), raw or dynamic: what the first block of a short bench pays"""
import os, sys, time
mode = sys.argv[1] if len(sys.argv) > 1 else "dynamic"
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dynamic_vins_amd import sim
from dynamic_vins_amd.backend import Runner
from dynamic_vins_amd.pipeline import DynamicPipeline, DynamicSequence, Pipeline, SyntheticSequence
N = 400
if mode == "dynamic":
    seq = DynamicSequence(1280, 720, sim.ZED, N + 1, rate=20.0, device="cuda:0", boxes=("escort", 4))
    pipe = DynamicPipeline(seq, max_cnt=250, min_dist=25, max_iters=10, device=0, mask_morphology_size=5)
else:
    seq = SyntheticSequence(1280, 720, sim.ZED, N + 1, rate=20.0, device="cuda:0")
    pipe = Pipeline(seq, max_cnt=250, min_dist=25, max_iters=10, device=0)
r = Runner([pipe])
ts = []
for k in range(N - 2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r.run(1); pipe.ctx.sync(); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
it = r.get(0)[2]
for a in range(0, N - 2, 20):
    print("frames %3d-%3d: mean %7.1f us  median %7.1f  max %8.1f" % (a, a + 19, ts[a:a + 20].mean(), np.median(ts[a:a + 20]), ts[a:a + 20].max()))
print("iterations total", it)
r.close(); pipe.ctx.close()
