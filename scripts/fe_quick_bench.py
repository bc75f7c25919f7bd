"""quick front-end timing at 1280x720 with device-resident frames (development aid)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dynamic_vins_amd import synth
from dynamic_vins_amd.frontend import Context, make_cam, DV_MEM_DEVICE
W, H = 1280, 720
ZED = (701.406049185687, 700.7199834541797, 663.9703743586792, 362.02045484177154, -0.17198906485492285, 0.024624053031210322, 0.0003391614313509814, -0.00045583634752113735)
ctx = Context(width=W, height=H, max_cnt=250, min_dist=25, cam0=make_cam(*ZED), cam1=make_cam(*ZED))
seq = synth.PlaneSequence(W, H, seed=1, disparity=20.3)
NF = 16
frames = []
for k in range(NF):
    l, r = seq.frame(k)
    frames.append((torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()))
order = list(range(NF)) + list(range(NF - 2, 0, -1))
torch.cuda.synchronize()
ctx.timing_enable(True)
def run(n):
    t0 = time.perf_counter()
    for i in range(n):
        l, r = frames[order[i % len(order)]]
        rows = ctx.track_stereo(l.data_ptr(), r.data_ptr(), 0.05 * i, mem=DV_MEM_DEVICE)
    return (time.perf_counter() - t0) / n, rows
run(20)
dt, rows = run(200)
print(f"sync per-frame wall: {dt*1e3:.3f} ms  ({1/dt:.0f} fps)  n={len(rows)} tracked={(rows['track_cnt']>1).sum()} stereo={rows['has_right'].sum()}")
for name in ["frame", "pyr", "lk_temporal", "compact", "gftt_eig", "gftt_select", "lk_stereo", "finalize"]:
    ms, cnt = ctx.timing_get(name)
    if cnt: print(f"  {name:12s} {ms/cnt*1e3:9.1f} us  (n={cnt})")
