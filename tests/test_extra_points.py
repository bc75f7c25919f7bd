"""Row N4, second half: the object "extra point" pipeline of dynamic mode.
  InstFeat::DetectExtraPoints (front_end/instance_feature.cpp:413-461) and the point-cloud half of InstsFeatManager::ProcessExtraPoints
  (front_end/dynamic_tracker.cpp:268-338: pcl::RadiusOutlierRemoval(0.5, 10) + pcl::EuclideanClusterExtraction(1.0, 10, 25000), first cluster).

CPU (`-m "not gpu"`): the oracle (oracle/extra_points.cpp) against an INDEPENDENT numpy / scipy restatement (float32 arithmetic; neighbour counts by broadcasting,
clusters by scipy.sparse.csgraph.connected_components) on scenes built to hit every branch.
GPU (`-m gpu`): the HIP kernel through the C ABI (dv_extra_points) against the oracle, bit-exact on the float triples, their order and the counts; the object
tracker's device path (dv_inst_set_disparity) is covered frame by frame in tests/test_dynamic_pipeline.py / test_reference_configs.py (`po == pipe.ipts`).
PCL is not under /root/reference and the reference holds no vectors for it: parity with PCL 1.8 itself is UNPINNED (oracle/extra_points.cpp states the algorithm
it restates)."""
import numpy as np
import pytest
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components

CAM = (701.4, 700.7, 663.9, 362.0)       # fx0 fy0 cx0 cy0
BASE = 0.12
W, H = 1280, 720


def scene(seed, rect, kind="box"):
    """-> (mask [h, w] u8, (x, y), disparity [H, W] f32) with NaNs, zeros, negative values, far / near depths inside the ROI"""
    rng = np.random.default_rng(seed)
    x, y, w, h = rect
    mask = np.zeros((h, w), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == "box":
        mask[h // 8: h - h // 8, w // 10: w - w // 10] = 255
    elif kind == "ellipse":
        mask[((yy - h / 2) / (0.45 * h)) ** 2 + ((xx - w / 2) / (0.45 * w)) ** 2 < 1] = 255
    elif kind == "two":                  # two blobs at different depths -> two clusters
        mask[:, : w // 2 - 2] = 255; mask[:, w // 2 + 2:] = 200
    elif kind == "full":
        mask[:] = 1
    depth = np.full((H, W), 40.0)
    depth[y: y + h, x: x + w] = 8.0 + 0.02 * (xx - w / 2) + rng.normal(0, 0.03, (h, w))
    if kind == "two":
        depth[y: y + h, x + w // 2: x + w] = 14.0 + rng.normal(0, 0.03, (h, w - w // 2))
    disp = (np.float32(CAM[0]) * np.float32(BASE) / depth.astype(np.float32)).astype(np.float32)
    r = lambda n: (rng.integers(y, y + h, n), rng.integers(x, x + w, n))
    disp[r(40)] = np.nan
    disp[r(40)] = 0.0
    disp[r(20)] = -1.5
    disp[r(30)] = np.float32(CAM[0] * BASE / 150.0)       # depth 150 > 100: rejected
    disp[r(30)] = np.float32(CAM[0] * BASE / 0.05)        # depth 0.05 <= 0.1: rejected
    disp[r(60)] = np.float32(CAM[0] * BASE / 30.0)        # isolated far points: the radius filter's prey
    return mask, (x, y), disp


def np_detect(mask, xy, disp):
    h, w = mask.shape
    step = int(max(np.sqrt(0.8 * h * w / 1000.0), 2.0))
    f32 = np.float32
    fx, fy, cx, cy = (f32(v) for v in CAM)
    out = []
    for i in range(0, h, step):
        for j in range(0, w, step):
            if mask[i, j] == 0:
                continue
            r, c = i + xy[1], j + xy[0]
            d = disp[r, c]
            if d <= 0 or d != d:
                continue
            depth = f32(f32(fx * f32(BASE)) / d)
            if float(depth) <= 0.1 or float(depth) > 100:
                continue
            out.append((f32(f32((f32(c) - cx) * depth) / fx), f32(f32((f32(r) - cy) * depth) / fy), depth))
    return np.array(out, np.float32).reshape(-1, 3)


def np_process(p):
    p = np.asarray(p, np.float32).reshape(-1, 3)
    n = len(p)
    if n == 0:
        return p

    def d2(a, b):
        dx = a[:, None, 0] - b[None, :, 0]; dy = a[:, None, 1] - b[None, :, 1]; dz = a[:, None, 2] - b[None, :, 2]
        return ((dx * dx + dy * dy).astype(np.float32) + dz * dz).astype(np.float32)
    D = d2(p, p)
    keep = (D <= np.float32(0.25)).sum(1) >= 11
    f = p[keep]
    if len(f) < 5:
        return np.zeros((0, 3), np.float32)
    A = d2(f, f) < np.float32(1.0)
    ncomp, lab = connected_components(coo_matrix(A), directed=False)
    best, best_size = -1, 0
    for c in range(ncomp):                 # components in order of their lowest member = PCL's discovery order
        members = np.nonzero(lab == c)[0]
        if 10 <= len(members) <= 25000 and len(members) > best_size:
            best, best_size = c, len(members)
    if best < 0:
        return np.zeros((0, 3), np.float32)
    # scipy numbers components by first occurrence, i.e. by lowest member index: `>` above keeps the first of equal sizes
    return f[lab == best]


CASES = [(1, (300, 200, 180, 120), "box"), (2, (0, 0, 97, 53), "ellipse"), (3, (1100, 600, 180, 120), "two"), (4, (40, 500, 640, 24), "full"),
         (5, (500, 100, 400, 380), "ellipse"), (6, (10, 10, 30, 26), "box"), (7, (200, 300, 151, 74), "two")]


@pytest.mark.parametrize("seed,rect,kind", CASES)
def test_oracle_matches_independent_restatement(oracle, seed, rect, kind):
    mask, xy, disp = scene(seed, rect, kind)
    raw = oracle.detect_extra_points(mask, xy, disp, CAM, BASE)
    ref = np_detect(mask, xy, disp)
    assert raw.shape == ref.shape and np.array_equal(raw.view(np.uint32), ref.view(np.uint32))
    seg = oracle.process_extra_points(raw)
    want = np_process(raw)
    assert seg.shape == want.shape and np.array_equal(seg.view(np.uint32), want.view(np.uint32))
    if kind != "full" and rect[2] * rect[3] > 2000:
        assert 10 <= len(seg) < len(raw)                  # the filter and the clustering both did something


def test_oracle_cluster_rules(oracle):
    """the documented rules on hand-made clouds: largest cluster wins, equal sizes -> the one found first, clusters below 10 points dropped, fewer than 5
    filtered points -> nothing, indices come back ascending"""
    rng = np.random.default_rng(0)
    blob = lambda c, n: (np.array(c, np.float32) + rng.uniform(-0.2, 0.2, (n, 3))).astype(np.float32)
    a, b, c = blob((0, 0, 5), 30), blob((5, 0, 5), 30), blob((10, 0, 5), 40)
    inter = np.empty((60, 3), np.float32); inter[0::2], inter[1::2] = a, b          # a and b interleaved: both size 30, a's first member comes first
    out = oracle.process_extra_points(inter)
    assert len(out) == 30 and np.array_equal(out, a)
    out = oracle.process_extra_points(np.concatenate([a, c, b]))
    assert len(out) == 40 and np.array_equal(out, c)
    assert len(oracle.process_extra_points(blob((0, 0, 5), 9))) == 0                 # nobody has 10 neighbours
    sparse = (np.arange(36, dtype=np.float32)[:, None] * np.array([[0.6, 0, 0]], np.float32))      # a chain at 0.6 m spacing: <= 2 neighbours within 0.5 m
    assert len(oracle.process_extra_points(sparse)) == 0
    tight = blob((0, 0, 5), 12)                                                        # 12 points: all survive the filter, one cluster of 12
    assert len(oracle.process_extra_points(tight)) == 12
    assert len(oracle.process_extra_points(np.zeros((0, 3), np.float32))) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed,rect,kind", CASES)
def test_hip_extra_points_bit_exact(gpu_ctx_factory, oracle, seed, rect, kind):
    from dynamic_vins_amd.frontend import make_cam
    cam = make_cam(CAM[0], CAM[1], CAM[2], CAM[3], 0, 0, 0, 0)
    ctx = gpu_ctx_factory(width=W, height=H, max_cnt=50, min_dist=10, cam0=cam, cam1=cam)
    mask, xy, disp = scene(seed, rect, kind)
    raw_o = oracle.detect_extra_points(mask, xy, disp, CAM, BASE)
    raw_d = ctx.extra_points(mask, xy, disp, BASE, stage=1)
    assert raw_d.shape == raw_o.shape
    assert np.array_equal(raw_d.astype(np.float32).view(np.uint32), raw_o.view(np.uint32)) and np.array_equal(raw_d, raw_o.astype(np.float64))      # floats widened, nothing else
    seg_o = oracle.process_extra_points(raw_o)
    seg_d = ctx.extra_points(mask, xy, disp, BASE, stage=0)
    assert seg_d.shape == seg_o.shape and np.array_equal(seg_d, seg_o.astype(np.float64))


@pytest.mark.gpu
def test_hip_extra_points_empty_and_degenerate(gpu_ctx_factory, oracle):
    from dynamic_vins_amd.frontend import make_cam
    cam = make_cam(CAM[0], CAM[1], CAM[2], CAM[3], 0, 0, 0, 0)
    ctx = gpu_ctx_factory(width=W, height=H, max_cnt=50, min_dist=10, cam0=cam, cam1=cam)
    mask, xy, disp = scene(9, (100, 100, 120, 90), "box")
    assert len(ctx.extra_points(np.zeros_like(mask), xy, disp, BASE, stage=1)) == 0          # empty mask
    assert len(ctx.extra_points(mask, xy, np.zeros_like(disp), BASE, stage=0)) == 0          # no valid disparity
    nan = np.full_like(disp, np.nan)
    assert len(ctx.extra_points(mask, xy, nan, BASE, stage=1)) == 0
    # a sparse cloud: every sampled point is more than 0.5 m from the next (depth 60 m, step 2 px -> 0.17 m ... use a coarse depth ramp instead)
    far = np.full_like(disp, np.float32(CAM[0] * BASE / 95.0))
    o = oracle.process_extra_points(oracle.detect_extra_points(mask, xy, far, CAM, BASE))
    d = ctx.extra_points(mask, xy, far, BASE, stage=0)
    assert np.array_equal(d, o.astype(np.float64))
