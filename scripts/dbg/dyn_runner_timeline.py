"""debug: per-frame pacing of the dynamic bench line on the C++ runner WITHOUT cutting the run: frame-end clocks (dv_runner_get_frame_clock) and solver iterations
(row log) -> ms per frame and iterations per frame in windows of 20 frames.  usage: python scripts/dbg/dyn_runner_timeline.py [tracker_thread 0|1] [mode raw|dynamic]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from dynamic_vins_amd import sim
from dynamic_vins_amd.backend import Runner
from dynamic_vins_amd.pipeline import DynamicPipeline, DynamicSequence, Pipeline, SyntheticSequence
tt = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mode = sys.argv[2] if len(sys.argv) > 2 else "dynamic"
N = 12 + 200
if mode == "dynamic":
    seq = DynamicSequence(1280, 720, sim.ZED, N + 2, rate=20.0, device="cuda:0", boxes=("escort", 4))
    pipe = DynamicPipeline(seq, max_cnt=250, min_dist=25, max_iters=10, device=0, mask_morphology_size=5)
else:
    seq = SyntheticSequence(1280, 720, sim.ZED, N + 2, rate=20.0, device="cuda:0")
    pipe = Pipeline(seq, max_cnt=250, min_dist=25, max_iters=10, device=0)
r = Runner([pipe])
if mode == "dynamic":
    r.set("tracker_thread", tt)
r.run(12)
torch.cuda.synchronize()
r.run(200)
clk = r.frame_clock(0)
log = r.row_log(0) if hasattr(r, "row_log") else None
d = np.diff(clk)[12:] * 1e3
print(f"{mode} tracker_thread {tt}: frames {len(clk)}, mean {d.mean():.3f} ms -> {1e3 / d.mean():.1f} frames/s")
its = None
if log is not None and len(log):
    its = np.asarray(log)[:, 3].astype(float)
for a in range(0, len(d), 20):
    seg = d[a:a + 20]
    extra = "" if its is None else f"  iterations {its[13 + a: 13 + a + 20].mean():.2f}"
    print(f"frames {12 + a:3d}..{12 + a + len(seg) - 1:3d}: {seg.mean():.3f} ms  p95 {np.percentile(seg, 95):.3f}  max {seg.max():.3f}{extra}")
if mode == "dynamic" and tt:
    tc = r.frame_clock(0, 1)
    dt = np.diff(tc)[12:] * 1e3
    print(f"tracker deliveries: mean {dt.mean():.3f} ms; first 24: " + " ".join(f"{v:.2f}" for v in dt[:24]))
    lead = (clk[13:13 + 24] - tc[13:13 + 24]) * 1e3
    print("estimator end minus tracker delivery of the same frame (ms), first 24: " + " ".join(f"{v:.2f}" for v in lead))
print("first 24 timed frames (ms):", " ".join(f"{v:.2f}" for v in d[:24]))
r.close(); pipe.ctx.close()
