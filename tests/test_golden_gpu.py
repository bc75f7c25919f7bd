"""GPU parity against the COMMITTED known-answer vectors (tests/golden/*.npz) — no oracle needed at run time, so the
HIP path is also pinned on boxes where the oracle build would differ.  Bit-exact for the front end (bytes, indices,
fp32 bit patterns); stated tolerances for the fp64 back end."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def front():
    return np.load(os.path.join(G, "front_kat.npz"))


@pytest.fixture(scope="module")
def back():
    return np.load(os.path.join(G, "back_kat.npz"))


def test_front_operators_match_golden(gpu_ctx_factory, front):
    from dynamic_vins_amd.frontend import make_cam
    ctx = gpu_ctx_factory(width=128, height=96, max_cnt=30, min_dist=10)
    img0, img1 = front["left"][0], front["left"][1]
    assert np.array_equal(ctx.pyr_down(img0), front["pyr1"])
    assert np.array_equal(ctx.pyr_down(front["pyr1"]), front["pyr2"])
    assert np.array_equal(ctx.min_eigen(img0).view(np.uint32), front["min_eigen"].view(np.uint32))
    c = ctx.gftt(img0, 40, 0.01, 8, front["gftt_mask"])
    assert np.array_equal(c, front["corners"])
    assert np.array_equal(ctx.gftt(img0, 25, 0.01, 12, None), front["corners_nomask"])
    p, st = ctx.lk(img0, img1, front["corners"], 3, 30, 0.01)
    assert np.array_equal(st, front["lk_status"]) and np.array_equal(p.view(np.uint32), front["lk_pts"].view(np.uint32))
    p, st = ctx.track_by_lk(img0, img1, front["corners"], True, 0.5)
    assert np.array_equal(st, front["tbl_status"])
    assert np.array_equal(p[st > 0].view(np.uint32), front["tbl_pts"][st > 0].view(np.uint32))
    pts = np.array([[20.3, 30.7], [100.0, 5.0], [127.0, 95.0], [64.5, 48.5]], np.float32)
    assert np.array_equal(ctx.circle_mask(np.full((96, 128), 255, np.uint8), pts, 9), front["circle"])
    assert np.array_equal(ctx.erode(front["erode_in"], 5), front["erode5"])
    assert np.array_equal(ctx.lift_projective(make_cam(*front["cam"]), front["lift_in"]).view(np.uint32), front["lift_out"].view(np.uint32))


def test_gpu_detector_matches_golden(gpu_ctx_factory):
    """row F5: cv::cuda::GoodFeaturesToTrackDetector's rule (dv_gftt_cuda / dv_min_eigen_cuda) and TrackImageNaive's rows against tests/golden/gftt_cuda_kat.npz"""
    from dynamic_vins_amd.frontend import make_cam
    g = np.load(os.path.join(G, "gftt_cuda_kat.npz"))
    cam = make_cam(*g["cam"])
    ctx = gpu_ctx_factory(width=128, height=96, max_cnt=30, min_dist=10, cam0=cam, cam1=cam)
    assert np.array_equal(ctx.min_eigen(g["img"], rule="cuda").view(np.uint32), g["min_eigen"].view(np.uint32))
    assert np.array_equal(ctx.gftt(g["img"], 40, 0.01, 8, None, rule="cuda"), g["corners_nomask"])
    assert np.array_equal(ctx.gftt(g["img"], 1000, 0.01, 3, g["mask"], rule="cuda"), g["corners_mask"])
    assert np.array_equal(ctx.gftt(g["img"], 1000, 0.01, 3, g["mask"]), g["corners_mask_cpu_rule"])
    from dynamic_vins_amd.frontend import DV_MODE_NAIVE
    for k in range(len(g["left"])):
        rows = ctx.track_stereo(g["left"][k], g["right"][k], 1.0 + 0.05 * k, mask=g["track_mask"], mode=DV_MODE_NAIVE)
        n = int(g["track_n"][k])
        assert len(rows) == n and rows.tobytes() == g["track_rows"][k].tobytes()[: n * 128], f"frame {k}"


def test_track_image_sequence_matches_golden(gpu_ctx_factory, front):
    from dynamic_vins_amd.frontend import make_cam
    cam = make_cam(*front["cam"])
    ctx = gpu_ctx_factory(width=128, height=96, max_cnt=30, min_dist=10, cam0=cam, cam1=cam)
    for k in range(len(front["left"])):
        rows = ctx.track_stereo(front["left"][k], front["right"][k], 1.0 + 0.05 * k)
        n = int(front["track_n"][k])
        assert len(rows) == n
        assert rows.tobytes() == front["track_rows"][k].tobytes()[: n * 128], f"frame {k}"


def test_projection_factors_match_golden(gpu_ctx_factory, back):
    from dynamic_vins_amd.backend import FACTOR_DTYPE, proj_eval
    ctx = gpu_ctx_factory(width=64, height=48)
    n = len(back["proj_kind"])
    fac = np.zeros(n, FACTOR_DTYPE)
    obs, par = back["proj_obs"], back["proj_par"]
    for k in range(n):
        f = fac[k]
        f["pix"], f["piy"], f["pjx"], f["pjy"] = obs[k][0], obs[k][1], obs[k][3], obs[k][4]
        f["vix"], f["viy"], f["vjx"], f["vjy"] = obs[k][6:10]
        f["td_i"], f["td_j"] = obs[k][10:12]
        f["kind"] = back["proj_kind"][k]
    got = proj_eval(ctx, fac, par[:, 0:7], par[:, 7:14], par[:, 14:21], par[:, 21:28], par[:, 28], par[:, 29])
    for k in range(n):
        kind = int(back["proj_kind"][k])
        sizes = {0: [7, 7, 7, 1, 1], 1: [7, 7, 7, 7, 1, 1], 2: [7, 7, 1, 1]}[kind]
        J, off = [], 0
        for s in sizes:
            J.append(back["proj_jac"][k][off:off + 2 * s].reshape(2, s)); off += 2 * s
        g = got[k]
        assert np.allclose(g[0:2], back["proj_res"][k], rtol=1e-9, atol=1e-9)
        gJ = dict(i=g[2:14].reshape(2, 6), j=g[14:26].reshape(2, 6), e0=g[26:38].reshape(2, 6), e1=g[38:50].reshape(2, 6), l=g[50:52], td=g[52:54])
        if kind == 0:
            exp = dict(i=J[0][:, :6], j=J[1][:, :6], e0=J[2][:, :6], e1=np.zeros((2, 6)), l=J[3][:, 0], td=J[4][:, 0])
        elif kind == 1:
            exp = dict(i=J[0][:, :6], j=J[1][:, :6], e0=J[2][:, :6], e1=J[3][:, :6], l=J[4][:, 0], td=J[5][:, 0])
        else:
            exp = dict(i=np.zeros((2, 6)), j=np.zeros((2, 6)), e0=J[0][:, :6], e1=J[1][:, :6], l=J[2][:, 0], td=J[3][:, 0])
        for key in gJ:
            assert np.allclose(gJ[key], exp[key], rtol=1e-9, atol=1e-9), (k, kind, key)


def _load_window(back):
    from dynamic_vins_amd.backend import FACTOR_DTYPE, IMU_DTYPE, LM_DTYPE, WindowProblem, dv_ba_prior
    prior = dv_ba_prior.from_buffer_copy(back["win_prior"].tobytes())
    return WindowProblem(back["win_pose"], back["win_sb"], back["win_ex"], 0.0, back["win_depth"], back["win_factors"].view(FACTOR_DTYPE),
                         back["win_landmarks"].view(LM_DTYPE), back["win_imu"].view(IMU_DTYPE), use_imu=1, plane_kind=0, max_iters=6,
                         prior=prior, prior_A=back["win_priorA"], prior_b=back["win_priorb"])


def test_window_solve_matches_golden(gpu_ctx_factory, back):
    """tolerances: same iteration count and termination; costs 1e-7 relative; states 1e-6 (m, m/s, rad, 1/m) — the
    HIP path sums in a different (fixed) order and factors with LDL^T instead of LL^T"""
    from dynamic_vins_amd.backend import ba_solve
    ctx = gpu_ctx_factory(width=64, height=48)
    prob = _load_window(back)
    s = ba_solve(ctx, prob)
    exp = back["sol_summary"]
    assert s.iterations == int(exp[0]) and s.termination == int(exp[1])
    assert abs(s.initial_cost - exp[2]) <= 1e-7 * exp[2] and abs(s.final_cost - exp[3]) <= 1e-7 * exp[3]
    assert np.abs(prob.pose - back["sol_pose"]).max() < 1e-6
    assert np.abs(prob.speed_bias - back["sol_sb"]).max() < 1e-6
    assert np.abs(prob.inv_depth - back["sol_depth"]).max() < 1e-6


def test_marginalization_matches_golden(gpu_ctx_factory, back):
    from dynamic_vins_amd.backend import marginalize
    from tests import ba_gen
    ctx = gpu_ctx_factory(width=64, height=48)
    prob = _load_window(back)
    for mode in (0, 1):
        sub = ba_gen.marg_subproblem(prob, mode)
        pr, A, b = marginalize(ctx, sub, mode)[:3]
        gblocks = {(int(t), int(i)): (int(o), int(sz), None) for t, i, o, sz in back[f"marg{mode}_blocks"]}
        blocks = ba_gen.prior_to_dict(pr, A, b)
        assert set(blocks) == set(gblocks)
        A2, b2 = ba_gen.permute_prior(blocks, A, b, gblocks)
        gA, gb = back[f"marg{mode}_A"], back[f"marg{mode}_b"]
        assert np.abs(A2 - gA).max() <= 1e-8 * np.abs(gA).max()
        assert np.abs(b2 - gb).max() <= 1e-8 * max(1.0, np.abs(gb).max())
        assert abs(pr.c0 - back[f"marg{mode}_c0"][0]) <= 2e-3 * abs(pr.c0)


def test_aux_rows_match_golden(gpu_ctx_factory):
    """remap (bit-exact, incl. the fused remap + cvtColor through the tracker's level 0), object solve and line-only solve against
    tests/golden/aux_kat.npz"""
    from dynamic_vins_amd.backend import LineProblem, ObjProblem, line_solve, obj_solve
    g = np.load(os.path.join(G, "aux_kat.npz"))
    ctx = gpu_ctx_factory(width=64, height=48)
    m1, m2 = g["remap_map1"], g["remap_map2"]
    assert np.array_equal(ctx.remap(g["remap_gray"], m1, m2), g["remap_gray_out"])
    assert np.array_equal(ctx.remap(g["remap_bgr"], m1, m2), g["remap_bgr_out"])
    assert np.array_equal(ctx.bgr2gray(ctx.remap(g["remap_bgr"], m1, m2)), g["remap_fused_gray"])
    for name in ("obj_a", "obj_b"):
        p = ObjProblem(g[name + "_state"], g[name + "_dims"], g[name + "_body_pose"], g[name + "_R_bc"], g[name + "_boxes"], g[name + "_points"],
                       max_iters=int(g[name + "_opts"][0]), plane_kind=int(g[name + "_opts"][1]))
        s = obj_solve(ctx, p)
        assert [s.iterations, s.successful, s.termination] == g[name + "_summary"][:3].astype(int).tolist()
        assert np.abs(p.state - g[name + "_state_out"]).max() <= 1e-8 and np.abs(p.dims - g[name + "_dims_out"]).max() <= 1e-8
        assert abs(s.final_cost - g[name + "_summary"][4]) <= 1e-9 * max(1.0, g[name + "_summary"][3])
    p = LineProblem(g["line_orth"], g["line_pose"], g["line_ex_pose"], g["line_sqrt_info"], g["line_obs"], max_iters=int(g["line_summary"][0]))
    s = line_solve(ctx, p)
    assert [s.iterations, s.successful, s.termination] == g["line_summary"][:3].astype(int).tolist()
    d = np.abs(p.orth - g["line_orth_out"])
    assert np.median(d) <= 1e-10 and d.max() <= 1e-6 and abs(s.final_cost - g["line_summary"][4]) <= 1e-8 * g["line_summary"][3]
