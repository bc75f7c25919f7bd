"""debug: which host phase of the first frames behind a dv_runner_run cut is slow in dynamic mode: 2-frame mini-blocks with the host scopes (dv_timing_enable(ctx, -1)) reset before each"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from dynamic_vins_amd import sim
from dynamic_vins_amd.backend import Runner
from dynamic_vins_amd.pipeline import DynamicPipeline, DynamicSequence
seq = DynamicSequence(1280, 720, sim.ZED, 32 + 40 + 2, rate=20.0, device="cuda:0", boxes=("escort", 4))
pipe = DynamicPipeline(seq, max_cnt=250, min_dist=25, max_iters=10, device=0, mask_morphology_size=5)
r = Runner([pipe])
r.run(32)
pipe.ctx.timing_enable(-1)
names = "h_imu h_add_features h_triangulate h_build h_solve_begin h_solve_upload h_solve_enqueue h_solve_wait h_dynamic h_dyn_push h_dyn_propagate h_dyn_triangulate h_dyn_initial h_dyn_build h_dyn_solve_begin h_dynamic_finish h_dyn_solve_wait h_dyn_reject h_dyn_slide h_slide h_dyn_finish_frame h_process_begin h_process_end h_front_enqueue h_front_wait h_inst_enqueue h_inst_collect h_inst_wait".split()
for b in range(8):
    torch.cuda.synchronize(); pipe.ctx.sync(); pipe.ctx.timing_reset()
    t0 = time.perf_counter(); r.run(2); pipe.ctx.sync(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    row = []
    for n in names:
        ms, cnt = pipe.ctx.timing_get(n)
        if cnt and ms > 0.35: row.append("%s %.2f" % (n[2:], ms))
    print("mini-block %d: %.2f ms for 2 frames | scopes > 0.35 ms (sum over the block): %s" % (b, 1e3 * dt, ", ".join(row)))
r.close(); pipe.ctx.close()
