"""Seeded generator of standalone sliding-window problems (test input, shared by the oracle and the product)."""
import ctypes as C

import numpy as np

from dynamic_vins_amd import sim
from dynamic_vins_amd.backend import FACTOR_DTYPE, IMU_DTYPE, LM_DTYPE, WindowProblem, dv_ba_prior


def quat_xyzw(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        v = np.zeros(3)
        v[i] = 0.25 * s
        v[j] = (R[j, i] + R[i, j]) / s
        v[k] = (R[k, i] + R[i, k]) / s
        q = np.array([v[0], v[1], v[2], (R[k, j] - R[j, k]) / s])
    return q / np.linalg.norm(q)


def small_rot(v):
    th = np.linalg.norm(v)
    K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def make_window(oracle, seed=1, nframes=11, nlm=120, use_imu=1, max_iters=8, pix_sigma=0.4, with_prior=False, dt_frame=0.1,
                pose_noise=(0.03, 0.01), depth_noise=0.08, stereo_ratio=0.8, plane_kind=0, outlier_ratio=0.0,
                feat_vel=False, td_true=0.0, ex_noise=(0.0, 0.0), free_blocks=0, prior_ex_scale=0.01):
    """feat_vel: the factors carry the features' image velocities (what the td Jacobian needs); td_true: the observations were taken td_true seconds late
    (x + v td_true: the factor undoes it with the td being estimated); ex_noise: (metres, radians) perturbation of the extrinsics handed to the solve;
    free_blocks: dv_ba_problem::free_blocks (bit 0 extrinsics, bit 1 td).  The defaults draw the same random numbers as before these options existed."""
    rng = np.random.default_rng(seed)
    traj = sim.Trajectory()
    t0 = 2.0 + 0.37 * seed
    times = t0 + dt_frame * np.arange(nframes)
    f = 460.0
    R_wb = [traj.R(t) for t in times]
    p_wb = [traj.p(t) for t in times]
    v_wb = [traj.v(t) for t in times]
    pts = sim.room_points(4000, seed=seed + 100)
    ric, tic = [sim.R_IC, sim.R_IC], [sim.T_IC0, sim.T_IC1]

    def proj(k, cam, P):
        Pc = ric[cam].T @ (R_wb[k].T @ (P - p_wb[k]) - tic[cam])
        return Pc[:2] / Pc[2], Pc[2]

    def proj_vel(k, cam, P, h=1e-4):          # image velocity of the projection (normalised plane, per second)
        def at(t):
            Pc = ric[cam].T @ (traj.R(t).T @ (P - traj.p(t)) - tic[cam])
            return Pc[:2] / Pc[2]
        return (at(times[k] + h) - at(times[k] - h)) / (2 * h)

    factors, lms, inv_depth = [], [], []
    tries = 0
    while len(lms) < nlm and tries < 20000:
        tries += 1
        P = pts[rng.integers(len(pts))]
        s = int(rng.integers(0, max(1, nframes - 3)))
        e = int(rng.integers(min(s + 3, nframes - 1), nframes))
        ok = True
        obs = []
        for k in range(s, e + 1):
            (xl, zl), (xr, zr) = proj(k, 0, P), proj(k, 1, P)
            if zl < 0.5 or zr < 0.5 or np.abs(xl).max() > 0.8 or np.abs(xr).max() > 0.8:
                ok = False
                break
            obs.append([xl + rng.normal(0, pix_sigma / f, 2), xr + rng.normal(0, pix_sigma / f, 2), rng.uniform() < stereo_ratio, np.zeros(2), np.zeros(2)])
            if feat_vel:
                obs[-1][3], obs[-1][4] = proj_vel(k, 0, P), proj_vel(k, 1, P)
                obs[-1][0] = obs[-1][0] + obs[-1][3] * td_true
                obs[-1][1] = obs[-1][1] + obs[-1][4] * td_true
        if not ok or len(obs) < 4:
            continue
        li = len(lms)
        first = len(factors)
        mask = 0
        pi, vi = obs[0][0], obs[0][3]
        if rng.uniform() < outlier_ratio:
            obs[-1][0] = obs[-1][0] + rng.normal(0, 30 / f, 2)
        for o, k in zip(obs, range(s, e + 1)):
            mask |= 1 << k
            if k != s:
                factors.append((pi[0], pi[1], o[0][0], o[0][1], vi[0], vi[1], o[3][0], o[3][1], 0, 0, 0, li, s, k, (0, 0)))
            if o[2]:
                factors.append((pi[0], pi[1], o[1][0], o[1][1], vi[0], vi[1], o[4][0], o[4][1], 0, 0, 1 if k != s else 2, li, s, k, (0, 0)))
        lms.append((first, len(factors) - first, s, mask))
        _, z = proj(s, 0, P)
        inv_depth.append(1.0 / (z * (1 + rng.normal(0, depth_noise))))
    factors = np.array(factors, FACTOR_DTYPE)
    lms = np.array(lms, LM_DTYPE)
    # states (perturbed)
    pose = np.zeros((nframes, 7))
    sb = np.zeros((nframes, 9))
    for k in range(nframes):
        Rn = R_wb[k] @ small_rot(rng.normal(0, pose_noise[1], 3)) if (use_imu or k > 0) else R_wb[k]
        pn = p_wb[k] + (rng.normal(0, pose_noise[0], 3) if (use_imu or k > 0) else 0)
        pose[k, :3] = pn
        pose[k, 3:] = quat_xyzw(Rn)
        sb[k, :3] = v_wb[k] + rng.normal(0, 0.05, 3)
        sb[k, 3:6] = rng.normal(0, 0.01, 3)
        sb[k, 6:9] = rng.normal(0, 0.002, 3)
    ex = np.zeros((2, 7))
    for c in range(2):
        ex[c, :3] = tic[c]
        ex[c, 3:] = quat_xyzw(ric[c])
        if ex_noise[0] or ex_noise[1]:
            ex[c, :3] += rng.normal(0, ex_noise[0], 3)
            ex[c, 3:] = quat_xyzw(ric[c] @ small_rot(rng.normal(0, ex_noise[1], 3)))
    # IMU pre-integration with the oracle's IntegrationBase restatement
    imu = np.zeros(nframes - 1 if use_imu else 0, IMU_DTYPE)
    if use_imu:
        noise = np.array([0.05, 0.005, 5e-4, 5e-5])
        lib = oracle.lib
        lib.dvo_preint_create.restype = C.c_void_p
        lib.dvo_preint_create.argtypes = [C.c_void_p] * 5
        lib.dvo_preint_push.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        lib.dvo_preint_get.argtypes = [C.c_void_p] * 7
        lib.dvo_preint_destroy.argtypes = [C.c_void_p]
        for k in range(nframes - 1):
            ts, acc, gyr = sim.imu_stream(traj, times[k], times[k + 1], 200.0, acc_n=noise[0] * 0.2, gyr_n=noise[1] * 0.2, seed=seed * 100 + k)
            ba, bg = np.zeros(3), np.zeros(3)
            h = lib.dvo_preint_create(acc[0].ctypes.data, gyr[0].ctypes.data, ba.ctypes.data, bg.ctypes.data, noise.ctypes.data)
            for i in range(1, len(ts)):
                a_, g_ = np.ascontiguousarray(acc[i]), np.ascontiguousarray(gyr[i])
                lib.dvo_preint_push(h, float(ts[i] - ts[i - 1]), a_.ctypes.data, g_.ctypes.data)
            sum_dt = C.c_double()
            dp, dq, dv, jac, cov = np.zeros(3), np.zeros(4), np.zeros(3), np.zeros(225), np.zeros(225)
            lib.dvo_preint_get(h, C.addressof(sum_dt), dp.ctypes.data, dq.ctypes.data, dv.ctypes.data, jac.ctypes.data, cov.ctypes.data)
            lib.dvo_preint_destroy(h)
            r = imu[k]
            r["sum_dt"], r["dp"], r["dv"] = sum_dt.value, dp, dv
            r["dq"] = [dq[3], dq[0], dq[1], dq[2]]
            r["jacobian"], r["covariance"], r["fi"], r["fj"] = jac, cov, k, k + 1
    prior = A = b = None
    if with_prior:
        prior = dv_ba_prior()
        blocks = [(0, k, 6) for k in range(0 if use_imu else 1, nframes - 1)]
        if use_imu:
            blocks.insert(1, (1, 0, 9))
        blocks += [(2, 0, 6), (2, 1, 6), (3, 0, 1)]
        off = 0
        for i, (ty, idx, sz) in enumerate(blocks):
            prior.blocks[i].type, prior.blocks[i].idx, prior.blocks[i].off, prior.blocks[i].size_local = ty, idx, off, sz
            src = pose[idx] if ty == 0 else sb[idx] if ty == 1 else ex[idx] if ty == 2 else np.array([0.0])
            x0 = np.array(src, float).copy()
            if ty == 0:
                x0[:3] += rng.normal(0, 0.01, 3)
            if ty == 1:
                x0 += rng.normal(0, 0.01, 9)
            for j in range(len(x0)):
                prior.x0[i][j] = x0[j]
            off += sz
        n = off
        M = rng.normal(0, 1, (n + 10, n)) * 30.0
        M[:, -13:] *= prior_ex_scale          # the extrinsic / td columns of the prior (weak by default: the blocks are constants of the solve)
        A = M.T @ M
        # a near-null direction, like the gauge freedom of a real prior
        v = rng.normal(0, 1, n)
        v /= np.linalg.norm(v)
        Pn = np.eye(n) - np.outer(v, v)
        A = Pn @ A @ Pn
        b = A @ rng.normal(0, 0.01, n)
        prior.valid, prior.n, prior.nblocks = 1, n, len(blocks)
        oracle.lib.dvo_prior_c0.restype = C.c_double
        oracle.lib.dvo_prior_c0.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        A = np.ascontiguousarray(A)
        b = np.ascontiguousarray(b)
        prior.c0 = oracle.lib.dvo_prior_c0(A.ctypes.data, b.ctypes.data, n)
    return WindowProblem(pose, sb, ex, 0.0, np.array(inv_depth), factors, lms, imu, use_imu=use_imu, plane_kind=plane_kind, max_iters=max_iters,
                         prior=prior, prior_A=A, prior_b=b, free_blocks=free_blocks)


def oracle_solve(oracle, prob):
    from dynamic_vins_amd.backend import dv_ba_summary
    s = dv_ba_summary()
    oracle.lib.dvo_ba_solve.argtypes = [C.c_void_p, C.c_void_p]
    oracle.lib.dvo_ba_solve(C.byref(prob.c), C.byref(s))
    return s


def marg_subproblem(prob, mode):
    """the inputs of SetMarginalizationInfo for a solved full window: mode 0 keeps only the residual blocks of the
    landmarks anchored in frame 0 and the IMU factor (0,1); mode 1 keeps only the prior"""
    if mode == 1:
        sub = WindowProblem(prob.pose, prob.speed_bias, prob.ex_pose, prob.td[0], prob.inv_depth, prob.factors[:0], prob.landmarks[:0], prob.imu[:0],
                            prob.c.use_imu, prob.c.plane_kind, prob.c.max_iters, prob.c.g_norm, prob.prior, prob.prior_A, prob.prior_b)
        return sub
    lm_keep = [l for l in range(len(prob.landmarks)) if prob.landmarks[l]["anchor"] == 0]
    facs, lms = [], []
    for l in lm_keep:
        L = prob.landmarks[l]
        first = len(facs)
        facs.extend(prob.factors[L["first"]:L["first"] + L["count"]])
        lms.append((first, L["count"], 0, L["mask"]))
    facs = np.array(facs, FACTOR_DTYPE) if facs else prob.factors[:0]
    lms = np.array(lms, LM_DTYPE)
    return WindowProblem(prob.pose, prob.speed_bias, prob.ex_pose, prob.td[0], prob.inv_depth, facs, lms, prob.imu[:1],
                         prob.c.use_imu, prob.c.plane_kind, prob.c.max_iters, prob.c.g_norm, prob.prior, prob.prior_A, prob.prior_b)


def oracle_marginalize(oracle, sub, mode):
    out = dv_ba_prior()
    A = np.zeros(192 * 192)
    b = np.zeros(192)
    oracle.lib.dvo_marginalize.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    oracle.lib.dvo_marginalize(C.byref(sub.c), mode, C.byref(out), A.ctypes.data, b.ctypes.data)
    n = out.n
    return out, A[:n * n].reshape(n, n).copy(), b[:n].copy()


def prior_to_dict(prior, A, b):
    """block-keyed view of a prior so that two priors with different block orders can be compared"""
    blocks = {}
    for i in range(prior.nblocks):
        pb = prior.blocks[i]
        gs = {0: 7, 1: 9, 2: 7, 3: 1}[pb.type]
        blocks[(pb.type, pb.idx)] = (pb.off, pb.size_local, np.array(prior.x0[i][:gs]))
    return blocks


def permute_prior(blocks_from, A, b, blocks_to):
    """reorders (A, b) given in blocks_from's layout into blocks_to's layout"""
    n = len(b)
    perm = np.zeros(n, int)
    for key, (off_to, sz, _) in blocks_to.items():
        off_from = blocks_from[key][0]
        perm[off_to:off_to + sz] = np.arange(off_from, off_from + sz)
    return A[np.ix_(perm, perm)], b[perm]
