"""ProjectionInstanceFactor (estimator/factor/project_instance_factor.cpp:27-172) — the "dynamic-InstanceFactor" BASELINE.json's north_star
names (dead code in the reference).  CPU: the oracle's restatement against an independent numpy residual and central differences through
PoseLocalParameterization::Plus (the reference's own check() recipe), with the inverse-depth quirk pinned.  GPU: dv_inst_proj_eval vs the oracle."""
import ctypes as C

import numpy as np
import pytest

from tests import obj_gen as G

SI = 460.0 / 1.5


def make_case(rng):
    """a geometrically sensible configuration: camera looks along +z of the camera frame at an object ~8 m away that moved between j and i"""
    def pose(p, rv):
        from scipy.spatial.transform import Rotation
        q = Rotation.from_rotvec(rv).as_quat()          # x y z w
        return np.concatenate([p, q])
    R_bc = np.array([[0.0, 0, 1], [-1, 0, 0], [0, -1, 0]])
    from scipy.spatial.transform import Rotation
    ex = np.concatenate([rng.normal(0, 0.05, 3), Rotation.from_matrix(R_bc).as_quat()])
    bj = pose(rng.normal(0, 0.5, 3), rng.normal(0, 0.1, 3))
    bi = pose(bj[:3] + rng.normal(0, 0.3, 3), rng.normal(0, 0.1, 3))
    oj = pose(np.array([8.0, 0, 0]) + rng.normal(0, 1.0, 3), rng.normal(0, 0.3, 3))
    oi = pose(oj[:3] + rng.normal(0, 0.5, 3), rng.normal(0, 0.3, 3))
    pts_j = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2), 1.0])
    pts_i = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2), 1.0])
    obs = np.concatenate([pts_j, pts_i, rng.normal(0, 0.1, 2), rng.normal(0, 0.1, 2), [0.002, 0.001]])
    lam = 1.0 / rng.uniform(5.0, 10.0)
    return obs, 0.004, [bj, bi, ex, oj, oi, np.array([lam])]


def o_eval(lib, obs, cur_td, blocks):
    lib.dvo_inst_proj_eval.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    blocks = [np.ascontiguousarray(b, np.float64) for b in blocks]
    sizes = [7, 7, 7, 7, 7, 1]
    J = [np.zeros(2 * s) for s in sizes]
    pp = (C.c_void_p * 6)(*[b.ctypes.data for b in blocks])
    Jp = (C.c_void_p * 6)(*[j.ctypes.data for j in J])
    r = np.zeros(2)
    o = np.ascontiguousarray(obs, np.float64)
    lib.dvo_inst_proj_eval(o.ctypes.data, cur_td, pp, r.ctypes.data, Jp)
    return r, [j.reshape(2, s) for j, s in zip(J, sizes)]


def np_residual(obs, cur_td, blocks):
    bj, bi, ex, oj, oi, lam = blocks
    R = lambda p: G._qR(p[3:])
    pj = obs[0:3] - (cur_td - obs[10]) * np.array([obs[6], obs[7], 0]); pi = obs[3:6] - (cur_td - obs[11]) * np.array([obs[8], obs[9], 0])
    x = pj / lam[0]
    x = R(ex) @ x + ex[:3]
    x = R(bj) @ x + bj[:3]
    x = R(oj).T @ (x - oj[:3])
    x = R(oi) @ x + oi[:3]
    x = R(bi).T @ (x - bi[:3])
    x = R(ex).T @ (x - ex[:3])
    return SI * (x[:2] / x[2] - pi[:2])


def test_oracle_restatement_residual_and_jacobians(oracle):
    rng = np.random.default_rng(5)
    for _ in range(24):
        obs, cur_td, blocks = make_case(rng)
        r, J = o_eval(oracle.lib, obs, cur_td, blocks)
        assert np.allclose(r, np_residual(obs, cur_td, blocks), rtol=1e-10, atol=1e-9)
        eps = 1e-6
        for b in range(5):                      # pose blocks: analytic == central difference through Plus
            num = np.zeros((2, 6))
            for k in range(6):
                d = np.zeros(6); d[k] = eps
                bp = list(blocks); bp[b] = G.pose_plus(blocks[b], d)
                bm = list(blocks); bm[b] = G.pose_plus(blocks[b], -d)
                num[:, k] = (np_residual(obs, cur_td, bp) - np_residual(obs, cur_td, bm)) / (2 * eps)
            assert np.allclose(J[b][:, :6], num, rtol=2e-5, atol=2e-4 * max(1.0, np.abs(num).max())), (b, J[b][:, :6], num)
            assert not J[b][:, 6].any()
        # inverse depth: NOT the derivative.  The true derivative is -reduce * T * pts_j_td / lam^2; the reference writes + and pts_j (:166).
        bj, bi, ex, oj, oi, lam_b = blocks
        lam = lam_b[0]
        R = lambda p: G._qR(p[3:])
        T = R(ex).T @ R(bi).T @ R(oi) @ R(oj).T @ R(bj) @ R(ex)
        pj_td = obs[0:3] - (cur_td - obs[10]) * np.array([obs[6], obs[7], 0])
        x = T @ (pj_td / lam)          # rotation part only: the translation terms do not depend on lam
        bp = list(blocks); bp[5] = np.array([lam + 1e-7]); bm = list(blocks); bm[5] = np.array([lam - 1e-7])
        num = (np_residual(obs, cur_td, bp) - np_residual(obs, cur_td, bm)) / 2e-7
        # camera-frame point of the residual chain (for `reduce`)
        xc = R(ex).T @ (R(bi).T @ (R(oi) @ (R(oj).T @ (R(bj) @ (R(ex) @ (pj_td / lam) + ex[:3]) + bj[:3] - oj[:3])) + oi[:3] - bi[:3]) - ex[:3])
        red = SI * np.array([[1 / xc[2], 0, -xc[0] / xc[2] ** 2], [0, 1 / xc[2], -xc[1] / xc[2] ** 2]])
        assert np.allclose(num, -red @ T @ pj_td / lam ** 2, rtol=1e-4, atol=1e-3)          # what a correct factor would return
        assert np.allclose(J[5][:, 0], red @ T @ obs[0:3] / lam ** 2, rtol=1e-10, atol=1e-9)      # what the reference returns (sic)
        assert not np.allclose(J[5][:, 0], num, rtol=1e-2)


@pytest.mark.gpu
def test_hip_inst_proj_matches_oracle(gpu_ctx_factory, oracle):
    from dynamic_vins_amd import backend as B
    ctx = gpu_ctx_factory(width=64, height=48)
    rng = np.random.default_rng(11)
    n = 200
    cases = [make_case(rng) for _ in range(n)]
    fac = np.zeros(n, B.INSTPROJ_DTYPE)
    blocks = [np.zeros((n, 7)) for _ in range(5)] + [np.zeros(n)]
    for k, (obs, cur_td, bl) in enumerate(cases):
        fac["pts_j"][k], fac["pts_i"][k], fac["vel_j"][k], fac["vel_i"][k] = obs[0:3], obs[3:6], obs[6:8], obs[8:10]
        fac["td_j"][k], fac["td_i"][k], fac["cur_td"][k] = obs[10], obs[11], cur_td
        for b in range(5):
            blocks[b][k] = bl[b]
        blocks[5][k] = bl[5][0]
    out = B.inst_proj_eval(ctx, fac, *blocks)
    assert out.shape == (n, 64)
    for k, (obs, cur_td, bl) in enumerate(cases):
        r, J = o_eval(oracle.lib, obs, cur_td, bl)
        sc = max(1.0, max(np.abs(j).max() for j in J))
        assert np.allclose(out[k, :2], r, rtol=1e-9, atol=1e-9 * sc)
        for b in range(5):
            assert np.allclose(out[k, 2 + 12 * b: 14 + 12 * b].reshape(2, 6), J[b][:, :6], rtol=1e-9, atol=1e-9 * sc), (k, b)
        assert np.allclose(out[k, 62:64], J[5][:, 0], rtol=1e-9, atol=1e-9 * sc)
    from dynamic_vins_amd._abi import DvinsError
    with pytest.raises(DvinsError):
        B.inst_proj_eval(ctx, fac[:0], *[b[:0] for b in blocks])
