"""Row F4: the reference's GPU tracker — cv::cuda::SparsePyrLKOpticalFlow(Size(21, 21), 3, 30[, useInitialFlow]) and FeatureTrackByLKGpu
(front_end/feature_utils.cpp:83-163, background_tracker.cpp:34-36) — restated (oracle/lk_cuda.cpp) and as a HIP kernel (csrc/lk_cuda.hip), used where the reference
uses it (naive mode: temporal + right image; semantic mode: right image; tests/test_front_parity.py runs those modes frame by frame against the oracle).

CPU: the oracle's pieces against independent numpy restatements (cuda::pyrDown's half-to-even rounding, one texture fetch with 8-bit fractions) and against
size-independent properties (known sub-pixel shift recovered, forward / backward consistency, agreement with the CPU tracker to a few hundredths of a pixel).
GPU: HIP against the oracle, bit-exact on the float32 patterns and the status bytes.
OpenCV's CUDA modules are not under /root/reference and the reference holds no vectors for them: parity with them is UNPINNED; two platform details of the original
(texture interpolation arithmetic, nvcc's contraction) are fixed by declaration — oracle/lk_cuda.cpp, DESIGN.md D4."""
import numpy as np
import pytest

from dynamic_vins_amd import synth


def np_pyr_down_cuda(img):
    h, w = img.shape
    k = np.array([1, 4, 6, 4, 1], np.int64)

    def refl(p, n):
        p = np.where(p < 0, -p, p)
        return np.where(p >= n, 2 * n - 2 - p, p)
    ys = refl(2 * np.arange((h + 1) // 2)[:, None] + np.arange(-2, 3)[None, :], h)
    xs = refl(2 * np.arange((w + 1) // 2)[:, None] + np.arange(-2, 3)[None, :], w)
    rows = (img.astype(np.int64)[:, xs] * k).sum(-1)                    # h x dw
    s = (rows[ys] * k[None, :, None]).sum(1)                            # dh x dw
    q, r = s >> 8, s & 255
    return (q + ((r > 128) | ((r == 128) & (q & 1 == 1)))).astype(np.uint8)


def np_tex(img, x, y):
    f = np.float32
    xb, yb = f(x) - f(0.5), f(y) - f(0.5)
    fx, fy = np.floor(xb), np.floor(yb)
    a = f(np.floor(f(f(xb - fx) * f(256)) + f(0.5))) * f(1 / 256); b = f(np.floor(f(f(yb - fy) * f(256)) + f(0.5))) * f(1 / 256)
    h, w = img.shape
    T = lambda i, j: f(img[min(max(int(j), 0), h - 1), min(max(int(i), 0), w - 1)]) / f(255)
    i, j = int(fx), int(fy)
    v = f(f(f(1) - a) * f(f(1) - b)) * T(i, j)
    v = f(v + f(f(a * f(f(1) - b)) * T(i + 1, j)))
    v = f(v + f(f(f(f(1) - a) * b) * T(i, j + 1)))
    v = f(v + f(f(a * b) * T(i + 1, j + 1)))
    return v


def test_oracle_pyr_down_cuda_and_texture_fetch(oracle):
    rng = np.random.default_rng(3)
    for shape in [(96, 128), (75, 101), (24, 33)]:
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        a = oracle.pyr_down_cuda(img)
        assert np.array_equal(a, np_pyr_down_cuda(img))
        b = oracle.pyr_down(img)                      # cv::pyrDown: (sum + 128) >> 8
        assert (a != b).sum() > 0 and np.abs(a.astype(int) - b.astype(int)).max() == 1      # the two roundings differ exactly at the .5 sums, by one
        for _ in range(60):
            x, y = rng.uniform(-3, shape[1] + 3), rng.uniform(-3, shape[0] + 3)
            assert np.float32(oracle.tex_read(img, x, y)).view(np.uint32) == np.float32(np_tex(img, x, y)).view(np.uint32), (x, y)
    img = rng.integers(0, 256, (20, 20), dtype=np.uint8)
    assert oracle.tex_read(img, 5.5, 7.5) == np.float32(img[7, 5]) / np.float32(255)       # texel centres are exact


def test_oracle_gpu_tracker_properties(oracle):
    seq = synth.PlaneSequence(192, 144, seed=7, disparity=3.25)
    left, right = seq.frame(0)
    pts = oracle.gftt(left, 60, 0.01, 10, None)
    p2, st = oracle.track_by_lk_gpu(left, right, pts, True)
    ok = st > 0
    assert ok.sum() >= 0.8 * len(pts)
    d = pts[ok] - p2[ok]
    assert abs(d[:, 0].mean() - 3.25) < 0.05 and np.abs(d[:, 1]).mean() < 0.05             # the known stereo shift, sub-pixel
    pc, sc = oracle.track_by_lk(left, right, pts, True, 1.0)                                 # the CPU tracker at the same threshold
    both = ok & (sc > 0)
    dd = np.abs(p2[both] - pc[both])
    assert both.sum() >= 0.75 * len(pts) and np.median(dd) < 0.02 and dd.max() < 0.5          # two different trackers, the same answer: hundredths of a pixel typically
    fwd, sf = oracle.lk_cuda(left, right, pts)
    assert np.array_equal(fwd.view(np.uint32), p2.view(np.uint32))                           # FeatureTrackByLKGpu's positions ARE the forward pass's
    bwd, sb = oracle.lk_cuda(right, left, fwd, initial=pts)                                  # backward from the previous points
    back_ok = (sf > 0) & (sb > 0) & (np.hypot(*(pts - bwd).T) <= 1.0)
    inb = (np.rint(p2[:, 0]) >= 1) & (np.rint(p2[:, 0]) < 191) & (np.rint(p2[:, 1]) >= 1) & (np.rint(p2[:, 1]) < 143)
    assert np.array_equal(back_ok & inb, ok)
    flat = np.full((144, 192), 90, np.uint8)                                                 # no texture: D < FLT_EPSILON at level 0 -> status 0 (no min-eigenvalue test elsewhere)
    _, s0 = oracle.lk_cuda(flat, flat, np.array([[50.0, 50.0]], np.float32))
    assert s0[0] == 0
    _, s1 = oracle.lk_cuda(left, right, np.array([[-4.0, 10.0], [500.0, 20.0]], np.float32))  # previous point outside the image at level 0
    assert (s1 == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,n", [(192, 144, 60), (752, 480, 150), (1280, 720, 250), (100, 75, 20)])
def test_hip_gpu_tracker_bit_exact(gpu_ctx_factory, oracle, w, h, n):
    ctx = gpu_ctx_factory(width=w, height=h, max_cnt=max(n, 8), min_dist=8)
    seq = synth.PlaneSequence(w, h, seed=11, disparity=4.5 if w > 200 else 2.0)
    left, right = seq.frame(0)
    left2, _ = seq.frame(1)
    assert np.array_equal(ctx.pyr_down_cuda(left), oracle.pyr_down_cuda(left))
    pts = oracle.gftt(left, n, 0.01, 8, None)
    rng = np.random.default_rng(1)
    extra = np.array([[0.4, 0.4], [w - 1.2, h - 1.3], [w + 5.0, 3.0], [-2.0, -2.0], [w / 2 + 0.25, 0.6]], np.float32)       # border, outside
    pts = np.concatenate([pts + rng.uniform(-0.45, 0.45, pts.shape).astype(np.float32), extra]).astype(np.float32)
    for a, b in ((left, right), (left, left2)):
        po, so = oracle.track_by_lk_gpu(a, b, pts, True)
        pd, sd = ctx.track_by_lk_gpu(a, b, pts, True)
        assert np.array_equal(so, sd) and so.sum() > 0.5 * n
        assert np.array_equal(po.view(np.uint32), pd.view(np.uint32))
        po, so = oracle.track_by_lk_gpu(a, b, pts, False)
        pd, sd = ctx.track_by_lk_gpu(a, b, pts, False)
        assert np.array_equal(so, sd) and np.array_equal(po.view(np.uint32), pd.view(np.uint32))
    for ml, iters in ((3, 30), (1, 5), (0, 30)):
        po, so = oracle.lk_cuda(left, right, pts, ml, iters)
        pd, sd = ctx.lk_cuda(left, right, pts, ml, iters)
        assert np.array_equal(so, sd) and np.array_equal(po.view(np.uint32), pd.view(np.uint32)), (ml, iters)
        init = pts + np.float32(1.5)
        po, so = oracle.lk_cuda(left, right, pts, ml, iters, initial=init)
        pd, sd = ctx.lk_cuda(left, right, pts, ml, iters, initial=init)
        assert np.array_equal(so, sd) and np.array_equal(po.view(np.uint32), pd.view(np.uint32)), (ml, iters, "initial")
    flat = np.full((h, w), 77, np.uint8)
    pd, sd = ctx.lk_cuda(flat, flat, pts[:4])
    po, so = oracle.lk_cuda(flat, flat, pts[:4])
    assert np.array_equal(sd, so) and sd.sum() == 0 and np.array_equal(po.view(np.uint32), pd.view(np.uint32))
