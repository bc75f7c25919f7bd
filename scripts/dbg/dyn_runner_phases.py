"""debug: host phase times (h_*) of the dynamic workload on the C++ runner (dv_runner_set_dynamic), dv_timing_enable(ctx, -1)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynamic_vins_amd import sim
from dynamic_vins_amd.backend import Runner
from dynamic_vins_amd.pipeline import DynamicPipeline, DynamicSequence
N = 260
seq = DynamicSequence(1280, 720, sim.ZED, N + 1, rate=20.0, device="cuda:0", boxes=("escort", 4))
pipe = DynamicPipeline(seq, max_cnt=250, min_dist=25, max_iters=10, device=0, mask_morphology_size=5)
r = Runner([pipe])
WARM = int(sys.argv[1]) if len(sys.argv) > 1 else 120
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 100
r.run(WARM)
pipe.ctx.timing_enable(-1)
torch.cuda.synchronize(); t0 = time.perf_counter(); r.run(NT); pipe.ctx.sync(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("%.1f frames/s, %.1f us per frame" % (NT / dt, dt * 1e6 / NT))
names = "h_add_features h_build h_dyn_build h_dyn_finish_frame h_dyn_initial h_dyn_propagate h_dyn_push h_dyn_reject h_dyn_slide h_dyn_solve_begin h_dyn_solve_wait h_dyn_triangulate h_dynamic h_dynamic_finish h_imu h_inst_wait h_line_only h_post h_process_begin h_process_end h_reject h_slide h_solve_begin h_solve_wait h_triangulate h_solve_upload h_solve_enqueue h_front_enqueue h_front_wait h_inst_enqueue h_inst_collect".split()
for name in names:
    ms, cnt = pipe.ctx.timing_get(name)
    if cnt: print("  %-20s %8.1f us  (n=%d)" % (name, 1e3 * ms / cnt, cnt))
r.close(); pipe.ctx.close()
