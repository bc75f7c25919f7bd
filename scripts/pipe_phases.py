"""host-side phase times of Pipeline.step (where does the frame period go?)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dynamic_vins_amd import sim
from dynamic_vins_amd.pipeline import Pipeline, SyntheticSequence
N = 80
seq = SyntheticSequence(1280, 720, sim.ZED, N + 2, rate=20.0, device="cuda:0")
pipe = Pipeline(seq)
acc = {}
def T(name, t0):
    t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
for k in range(N):
    s = pipe.seq
    if k == 30:
        acc.clear(); torch.cuda.synchronize(); tstart = time.perf_counter()
    t0 = time.perf_counter()
    pre, pipe._prefetched = pipe._prefetched, None
    if pre is None and not pipe.enqueued: pipe._enqueue(k)
    rows = pre if pre is not None else pipe.ctx.track_stereo_collect(); pipe.enqueued = False
    t0 = T("collect", t0)
    t = s.times[k]
    pipe._feed_imu(t)
    t0 = T("imu_py", t0)
    rc = pipe.est.ProcessMeasurementsBegin(rows, t)
    t0 = T("begin", t0)
    pipe._enqueue(k + 1)
    t0 = T("enqueue_fe", t0)
    pipe._feed_imu(s.times[k + 1])
    t0 = T("imu_py2", t0)
    st = pipe.est.ProcessMeasurementsEnd()
    t0 = T("end", t0)
    if st.nonlinear:
        pipe.poses.append(pipe.est.window()[10, :7]); pipe.pose_times.append(t)
    pipe.next += 1
    t0 = T("py_tail", t0)
total = time.perf_counter() - tstart
n = N - 30
print("frame period %.1f us" % (total / n * 1e6))
for k2, v in acc.items(): print("  %-12s %8.1f us" % (k2, v / n * 1e6))
pipe.ctx.timing_enable(1)
