"""Frame-end clocks of a dynamic sequence (north_star's target configuration: 1280x720, TrackSemanticImage + InstsTrack + window solve + object solve every frame) on the
library's C++ loop, from a COLD process: one JSON line with the gaps between consecutive frame ends.  What a real-time estimator cares about is the worst frame, not
the mean: tests/test_frame_gaps.py asserts on this output; DVINS_COPY_ENGINE=1 in the environment puts the per-frame staging copies back on the copy engines (A/B).
usage: python scripts/dyn_cold_frames.py [frames=60] [cut=0]   (cut > 0: the run is cut into two dv_runner_run calls with a device-wide synchronisation between)"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    cut = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    import torch
    from dynamic_vins_amd import sim
    from dynamic_vins_amd.backend import Runner
    from dynamic_vins_amd.pipeline import DynamicPipeline, DynamicSequence
    w, h = 1280, 720
    seq = DynamicSequence(w, h, sim.ZED, frames + 2, rate=20.0, device="cuda:0", boxes=("escort", 3))
    pipe = DynamicPipeline(seq, max_cnt=250, min_dist=25, max_iters=10, device=0, use_imu=1, mask_morphology_size=0)
    runner = Runner([pipe])
    torch.cuda.synchronize()
    if cut > 0:
        runner.run(cut)
        torch.cuda.synchronize(); pipe.ctx.sync()
        runner.run(frames - cut)
    else:
        runner.run(frames)
    clk = np.asarray(runner.frame_clock(0), dtype=np.float64)
    gaps = np.diff(clk) * 1e3
    if cut > 0:
        gaps[cut - 1] = np.nan            # the gap across the two calls holds the host's synchronisation, not a frame
    st, poses, iters, fr = runner.get(0)
    stats = runner.dynamic_stats(0)
    g = gaps[~np.isnan(gaps)]
    print(json.dumps({"frames": int(fr), "cut": cut, "copy_engine": os.environ.get("DVINS_COPY_ENGINE", "0"), "gaps_ms": [None if np.isnan(v) else round(float(v), 3) for v in gaps],
                      "max_after_frame_2_ms": round(float(np.nanmax(gaps[2:])), 3), "p50_ms": round(float(np.median(g)), 3), "p95_ms": round(float(np.percentile(g, 95)), 3),
                      "n_above_2ms_after_frame_2": int((gaps[2:] > 2.0).sum()), "iterations": int(iters), "min_detections": int(stats["min_detections"])}))
    runner.close(); pipe.ctx.close()


if __name__ == "__main__":
    main()
