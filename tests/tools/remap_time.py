"""Frame-ingest cost with cfg::is_undistort_input: 1280x720 stereo BGR frames resident in HBM, maps installed, 60 frames.
Run under `rocprofv3 --kernel-trace --stats` to see remap_kernel<3, true> (fused remap + cvtColor into pyramid level 0)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dynamic_vins_amd.frontend import Context, DV_FMT_BGR, DV_MEM_DEVICE, DV_MODE_RAW, make_cam      # noqa: E402
from tests import oracle_py                                                                             # noqa: E402

w, h = 1280, 720
o = oracle_py.load()
cam = (700.0, 702.0, 642.0, 358.0, -0.28, 0.07, 1e-3, -7e-4)
new_k = (620.0, 622.0, 640.0, 360.0)
m1, m2 = o.init_undistort_map(cam, new_k, w, h)
ctx = Context(width=w, height=h, max_cnt=250, min_dist=30, cam0=make_cam(*new_k, 0, 0, 0, 0), cam1=make_cam(*new_k, 0, 0, 0, 0))
ctx.set_undistort_maps(0, m1, m2)
ctx.set_undistort_maps(1, m1, m2)
rng = np.random.default_rng(0)
base = (rng.integers(0, 256, (h // 8, w // 8, 3)).astype(np.uint8)).repeat(8, 0).repeat(8, 1)
l = torch.from_numpy(np.ascontiguousarray(base)).cuda()
r = torch.from_numpy(np.ascontiguousarray(np.roll(base, -6, 1))).cuda()
ctx.timing_enable(1)
for k in range(60):
    ctx.track_stereo(l.data_ptr(), r.data_ptr(), 0.05 * k, None, DV_MODE_RAW, DV_MEM_DEVICE | DV_FMT_BGR, w=w, h=h, stride=3 * w)
    if k == 9:
        ctx.timing_reset()
ms, cnt = ctx.timing_get("pyr")
print("pyr stage (remap + cvtColor + 3 x pyrDown, stereo): %.1f us / frame over %d frames" % (1e3 * ms / max(cnt, 1), cnt))
ctx.close()
