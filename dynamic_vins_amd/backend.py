"""ctypes mirror of the bundle-adjustment part of include/dvins.h (dv_ba_*), for tests and bench.py."""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import DvinsError


class dv_ba_factor(C.Structure):
    _fields_ = [("pix", C.c_double), ("piy", C.c_double), ("pjx", C.c_double), ("pjy", C.c_double),
                ("vix", C.c_double), ("viy", C.c_double), ("vjx", C.c_double), ("vjy", C.c_double),
                ("td_i", C.c_double), ("td_j", C.c_double),
                ("kind", C.c_int32), ("lm", C.c_int32), ("fi", C.c_int32), ("fj", C.c_int32), ("pad_", C.c_double * 2)]


FACTOR_DTYPE = np.dtype([("pix", "f8"), ("piy", "f8"), ("pjx", "f8"), ("pjy", "f8"), ("vix", "f8"), ("viy", "f8"), ("vjx", "f8"), ("vjy", "f8"),
                         ("td_i", "f8"), ("td_j", "f8"), ("kind", "i4"), ("lm", "i4"), ("fi", "i4"), ("fj", "i4"), ("pad_", "f8", 2)])
LM_DTYPE = np.dtype([("first", "i4"), ("count", "i4"), ("anchor", "i4"), ("mask", "i4")])
IMU_DTYPE = np.dtype([("sum_dt", "f8"), ("dp", "f8", 3), ("dq", "f8", 4), ("dv", "f8", 3), ("lin_ba", "f8", 3), ("lin_bg", "f8", 3),
                      ("jacobian", "f8", 225), ("covariance", "f8", 225), ("fi", "i4"), ("fj", "i4"), ("pad0", "i4"), ("pad1", "i4")])
assert FACTOR_DTYPE.itemsize == C.sizeof(dv_ba_factor) == 112


class dv_ba_prior_block(C.Structure):
    _fields_ = [("type", C.c_int32), ("idx", C.c_int32), ("off", C.c_int32), ("size_local", C.c_int32)]


class dv_ba_prior(C.Structure):
    _fields_ = [("valid", C.c_int32), ("n", C.c_int32), ("nblocks", C.c_int32), ("pad", C.c_int32), ("c0", C.c_double),
                ("blocks", dv_ba_prior_block * 16), ("x0", (C.c_double * 9) * 16)]


class dv_ba_problem(C.Structure):
    _fields_ = [("nframes", C.c_int32), ("nlm", C.c_int32), ("nfac", C.c_int32), ("nimu", C.c_int32),
                ("use_imu", C.c_int32), ("plane_kind", C.c_int32), ("max_iters", C.c_int32), ("free_blocks", C.c_int32),
                ("g_norm", C.c_double),
                ("pose", C.c_void_p), ("speed_bias", C.c_void_p), ("ex_pose", C.c_void_p), ("td", C.c_void_p), ("inv_depth", C.c_void_p),
                ("factors", C.c_void_p), ("landmarks", C.c_void_p), ("imu", C.c_void_p),
                ("prior", C.c_void_p), ("prior_A", C.c_void_p), ("prior_b", C.c_void_p), ("x_norm2_extra", C.c_double)]


class dv_ba_summary(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("successful", C.c_int32), ("termination", C.c_int32), ("slots", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double)]




class WindowProblem:
    """Owns the numpy buffers of one window problem and exposes them as a dv_ba_problem.
    The struct layout is shared by the oracle's dvo_ba_problem, so the same object drives both."""

    def __init__(self, pose, speed_bias, ex_pose, td, inv_depth, factors, landmarks, imu, use_imu=1, plane_kind=0, max_iters=8,
                 g_norm=9.81, prior=None, prior_A=None, prior_b=None, free_blocks=0):
        self.pose = np.ascontiguousarray(pose, np.float64).copy()
        self.speed_bias = np.ascontiguousarray(speed_bias, np.float64).copy()
        self.ex_pose = np.ascontiguousarray(ex_pose, np.float64).copy()
        self.td = np.array([td], np.float64)
        self.inv_depth = np.ascontiguousarray(inv_depth, np.float64).copy()
        self.factors = np.ascontiguousarray(factors)
        self.landmarks = np.ascontiguousarray(landmarks)
        self.imu = np.ascontiguousarray(imu)
        self.prior, self.prior_A, self.prior_b = prior, None, None
        if prior is not None:
            self.prior_A = np.ascontiguousarray(prior_A, np.float64)
            self.prior_b = np.ascontiguousarray(prior_b, np.float64)
        p = dv_ba_problem()
        p.nframes, p.nlm, p.nfac, p.nimu = len(self.pose), len(self.landmarks), len(self.factors), len(self.imu)
        p.use_imu, p.plane_kind, p.max_iters, p.g_norm = use_imu, plane_kind, max_iters, g_norm
        p.free_blocks = free_blocks          # bit 0: para_ex_pose free (estimate_extrinsic), bit 1: para_td free (estimate_td)
        self.c = p
        self._bind()

    def _bind(self):
        p = self.c
        p.pose, p.speed_bias, p.ex_pose = self.pose.ctypes.data, self.speed_bias.ctypes.data, self.ex_pose.ctypes.data
        p.td, p.inv_depth = self.td.ctypes.data, self.inv_depth.ctypes.data
        p.factors = self.factors.ctypes.data if len(self.factors) else None
        p.landmarks = self.landmarks.ctypes.data if len(self.landmarks) else None
        p.imu = self.imu.ctypes.data if len(self.imu) else None
        if self.prior is not None:
            p.prior, p.prior_A, p.prior_b = C.addressof(self.prior), self.prior_A.ctypes.data, self.prior_b.ctypes.data
        else:
            p.prior, p.prior_A, p.prior_b = None, None, None

    def clone(self):
        q = WindowProblem(self.pose, self.speed_bias, self.ex_pose, self.td[0], self.inv_depth, self.factors, self.landmarks, self.imu,
                          self.c.use_imu, self.c.plane_kind, self.c.max_iters, self.c.g_norm, self.prior, self.prior_A, self.prior_b, self.c.free_blocks)
        return q


def ba_solve(ctx, prob: WindowProblem):
    s = dv_ba_summary()
    if ctx.lib.dv_ba_solve(ctx.h, C.byref(prob.c), C.byref(s)) != 0:
        raise DvinsError(ctx.lib.dv_last_error(ctx.h).decode())
    return s


def ba_eval(ctx, prob: WindowProblem):
    """dv_ba_eval -> (cost, S[n, n], g[n]) at the problem's current states"""
    n, cost = C.c_int(0), C.c_double(0)
    S = np.zeros(178 * 178)
    g = np.zeros(178)
    if ctx.lib.dv_ba_eval(ctx.h, C.byref(prob.c), C.byref(n), C.byref(cost), S.ctypes.data, g.ctypes.data) != 0:
        raise DvinsError(ctx.lib.dv_last_error(ctx.h).decode())
    k = n.value
    return cost.value, S[: k * k].reshape(k, k).copy(), g[:k].copy()


def proj_eval(ctx, factors, pose_i, pose_j, ex0, ex1, inv_depth, td):
    n = len(factors)
    arrs = [np.ascontiguousarray(a, np.float64) for a in (pose_i, pose_j, ex0, ex1, inv_depth, td)]
    factors = np.ascontiguousarray(factors)
    out = np.zeros((n, 54))
    if ctx.lib.dv_proj_eval(ctx.h, factors.ctypes.data, n, *[a.ctypes.data for a in arrs], out.ctypes.data) != 0:
        raise DvinsError(ctx.lib.dv_last_error(ctx.h).decode())
    return out


def imu_eval(ctx, imu_rec, g_norm, pose_i, sb_i, pose_j, sb_j):
    arrs = [np.ascontiguousarray(a, np.float64) for a in (pose_i, sb_i, pose_j, sb_j)]
    rec = np.ascontiguousarray(imu_rec)
    out = np.zeros(465)
    if ctx.lib.dv_imu_eval(ctx.h, rec.ctypes.data, float(g_norm), *[a.ctypes.data for a in arrs], out.ctypes.data) != 0:
        raise DvinsError(ctx.lib.dv_last_error(ctx.h).decode())
    return out[:15], out[15:].reshape(15, 30)


def marginalize(ctx, sub: WindowProblem, mode):
    """dv_marginalize -> (dv_ba_prior, A, b, diag)"""
    out = dv_ba_prior()
    A = np.zeros(192 * 192)
    b = np.zeros(192)
    diag = np.zeros(4)
    if ctx.lib.dv_marginalize(ctx.h, C.byref(sub.c), int(mode), C.byref(out), A.ctypes.data, b.ctypes.data, diag.ctypes.data) != 0:
        raise DvinsError(ctx.lib.dv_last_error(ctx.h).decode())
    n = out.n
    return out, A[:n * n].reshape(n, n).copy(), b[:n].copy(), diag


LINE_DTYPE = np.dtype([("obs", "f8", 4), ("sqrt_info", "f8", 4)])
BOXPT_DTYPE = np.dtype([("pts_w", "f8", 3), ("dims", "f8", 3)])


def _chk(ctx, rc):
    if rc != 0:
        raise DvinsError(ctx.lib.dv_last_error(ctx.h).decode())


def line_eval(ctx, factors, pose, ex_pose, orth):
    """lineProjectionFactor::Evaluate for n blocks -> (r[n,2], J_pose[n,2,6], J_ex[n,2,6], J_orth[n,2,4])"""
    n = len(factors)
    f = np.ascontiguousarray(factors, LINE_DTYPE)
    a = [np.ascontiguousarray(x, np.float64) for x in (pose, ex_pose, orth)]
    out = np.zeros((n, 34))
    _chk(ctx, ctx.lib.dv_line_eval(ctx.h, f.ctypes.data, n, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, out.ctypes.data))
    return out[:, :2], out[:, 2:14].reshape(n, 2, 6), out[:, 14:26].reshape(n, 2, 6), out[:, 26:34].reshape(n, 2, 4)


def line_plus(ctx, orth, delta):
    a, d = np.ascontiguousarray(orth, np.float64), np.ascontiguousarray(delta, np.float64)
    out = np.zeros_like(a)
    _chk(ctx, ctx.lib.dv_line_plus(ctx.h, a.ctypes.data, d.ctypes.data, len(a), out.ctypes.data))
    return out


def box_enclose_eval(ctx, points, pose_obj):
    n = len(points)
    f, p = np.ascontiguousarray(points, BOXPT_DTYPE), np.ascontiguousarray(pose_obj, np.float64)
    out = np.zeros((n, 21))
    _chk(ctx, ctx.lib.dv_box_enclose_eval(ctx.h, f.ctypes.data, n, p.ctypes.data, out.ctypes.data))
    return out[:, :3], out[:, 3:].reshape(n, 3, 6)


INSTPROJ_DTYPE = np.dtype([("pts_j", "f8", 3), ("pts_i", "f8", 3), ("vel_j", "f8", 2), ("vel_i", "f8", 2), ("td_j", "f8"), ("td_i", "f8"), ("cur_td", "f8"), ("pad_", "f8")])
assert INSTPROJ_DTYPE.itemsize == 112


def inst_proj_eval(ctx, factors, pose_bj, pose_bi, ex_pose, pose_oj, pose_oi, inv_dep_j):
    """ProjectionInstanceFactor::Evaluate for n blocks -> out[n, 64] = r[2] | five 2x6 pose Jacobians | J_inv_dep[2]"""
    n = len(factors)
    f = np.ascontiguousarray(factors, INSTPROJ_DTYPE)
    a = [np.ascontiguousarray(x, np.float64) for x in (pose_bj, pose_bi, ex_pose, pose_oj, pose_oi, inv_dep_j)]
    out = np.zeros((n, 64))
    _chk(ctx, ctx.lib.dv_inst_proj_eval(ctx.h, f.ctypes.data, n, *[x.ctypes.data for x in a], out.ctypes.data))
    return out


def box_dims_eval(ctx, dims, box):
    d, b = np.ascontiguousarray(dims, np.float64), np.ascontiguousarray(box, np.float64)
    out = np.zeros((len(d), 4))
    _chk(ctx, ctx.lib.dv_box_dims_eval(ctx.h, d.ctypes.data, b.ctypes.data, len(d), out.ctypes.data))
    return out[:, 0], out[:, 1:]


def box_orientation_eval(ctx, R_cioi, R_bc, pose_body, pose_obj):
    a = [np.ascontiguousarray(x, np.float64) for x in (R_cioi, R_bc, pose_body, pose_obj)]
    n = len(a[2])
    out = np.zeros((n, 39))
    _chk(ctx, ctx.lib.dv_box_orientation_eval(ctx.h, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, n, out.ctypes.data))
    return out[:, :3], out[:, 3:21].reshape(n, 3, 6), out[:, 21:].reshape(n, 3, 6)


OBJBOX_DTYPE = np.dtype([("obj", "i4"), ("frame", "i4"), ("dims", "f8", 3), ("R_cioi", "f8", 9)])
OBJPT_DTYPE = np.dtype([("obj", "i4"), ("frame", "i4"), ("p_w", "f8", 3)])


class dv_obj_problem(C.Structure):
    _fields_ = [("n_obj", C.c_int32), ("n_boxes", C.c_int32), ("n_points", C.c_int32), ("max_iters", C.c_int32),
                ("plane_kind", C.c_int32), ("reserved", C.c_int32),
                ("state", C.c_void_p), ("dims", C.c_void_p), ("body_pose", C.c_void_p), ("R_bc", C.c_double * 9),
                ("boxes", C.c_void_p), ("points", C.c_void_p)]


assert OBJBOX_DTYPE.itemsize == 104 and OBJPT_DTYPE.itemsize == 32


class ObjProblem:
    """Owns the numpy buffers of one InstanceManager::Optimization problem (estimator_insts.cpp:772-807) and exposes them
    as a dv_obj_problem.  The oracle's dvo_obj_problem has the same layout, so the same object drives both."""

    def __init__(self, state, dims, body_pose, R_bc, boxes, points, max_iters=10, plane_kind=0):
        self.state = np.ascontiguousarray(state, np.float64).copy().reshape(-1, 11, 7)
        self.dims = np.ascontiguousarray(dims, np.float64).copy().reshape(-1, 3)
        self.body_pose = np.ascontiguousarray(body_pose, np.float64).copy().reshape(11, 7)
        self.R_bc = np.ascontiguousarray(R_bc, np.float64).reshape(9).copy()
        self.boxes = np.ascontiguousarray(boxes, OBJBOX_DTYPE).copy()
        self.points = np.ascontiguousarray(points, OBJPT_DTYPE).copy()
        self.max_iters, self.plane_kind = max_iters, plane_kind
        assert len(self.state) == len(self.dims)

    def struct(self):
        p = dv_obj_problem()
        p.n_obj, p.n_boxes, p.n_points, p.max_iters, p.plane_kind = len(self.dims), len(self.boxes), len(self.points), self.max_iters, self.plane_kind
        p.state, p.dims, p.body_pose = self.state.ctypes.data, self.dims.ctypes.data, self.body_pose.ctypes.data
        p.R_bc[:] = self.R_bc.tolist()
        p.boxes = self.boxes.ctypes.data if len(self.boxes) else None
        p.points = self.points.ctypes.data if len(self.points) else None
        return p

    def clone(self):
        return ObjProblem(self.state, self.dims, self.body_pose, self.R_bc, self.boxes, self.points, self.max_iters, self.plane_kind)


def obj_solve(ctx, prob):
    """ceres::Solve of InstanceManager::Optimization on the device; prob.state / prob.dims are updated in place."""
    p, s = prob.struct(), dv_ba_summary()
    _chk(ctx, ctx.lib.dv_obj_solve(ctx.h, C.byref(p), C.byref(s)))
    return s


LINEOBS_DTYPE = np.dtype([("line", "i4"), ("frame", "i4"), ("obs", "f8", 4)])


class dv_line_problem(C.Structure):
    _fields_ = [("n_lines", C.c_int32), ("n_obs", C.c_int32), ("max_iters", C.c_int32), ("reserved", C.c_int32),
                ("orth", C.c_void_p), ("pose", C.c_void_p), ("ex_pose", C.c_void_p), ("sqrt_info", C.c_double * 4), ("obs", C.c_void_p)]


class LineProblem:
    """Owns the buffers of one Estimator::OptimizationWithOnlyLine problem (estimator.cpp:345-395); same layout as the oracle's dvo_line_problem."""

    def __init__(self, orth, pose, ex_pose, sqrt_info, obs, max_iters=10):
        self.orth = np.ascontiguousarray(orth, np.float64).copy().reshape(-1, 4)
        self.pose = np.ascontiguousarray(pose, np.float64).copy().reshape(11, 7)
        self.ex_pose = np.ascontiguousarray(ex_pose, np.float64).copy().reshape(7)
        self.sqrt_info = np.ascontiguousarray(sqrt_info, np.float64).reshape(4).copy()
        self.obs = np.ascontiguousarray(obs, LINEOBS_DTYPE).copy()
        self.max_iters = max_iters

    def struct(self):
        p = dv_line_problem()
        p.n_lines, p.n_obs, p.max_iters = len(self.orth), len(self.obs), self.max_iters
        p.orth, p.pose, p.ex_pose = self.orth.ctypes.data, self.pose.ctypes.data, self.ex_pose.ctypes.data
        p.sqrt_info[:] = self.sqrt_info.tolist()
        p.obs = self.obs.ctypes.data if len(self.obs) else None
        return p

    def clone(self):
        return LineProblem(self.orth, self.pose, self.ex_pose, self.sqrt_info, self.obs, self.max_iters)


def line_solve(ctx, prob):
    """ceres::Solve of Estimator::OptimizationWithOnlyLine on the device; prob.orth is updated in place."""
    p, s = prob.struct(), dv_ba_summary()
    _chk(ctx, ctx.lib.dv_line_solve(ctx.h, C.byref(p), C.byref(s)))
    return s


LINEROW_DTYPE = np.dtype([("id", "u4"), ("has_right", "i4"), ("left", "f8", 4), ("right", "f8", 4)])
LINELM_DTYPE = np.dtype([("id", "i4"), ("start_frame", "i4"), ("n_obs", "i4"), ("is_triangulation", "i4"), ("plucker", "f8", 6), ("ptw1", "f8", 3), ("ptw2", "f8", 3)])


class dv_est_config(C.Structure):
    _fields_ = [("use_imu", C.c_int32), ("stereo", C.c_int32), ("plane_constraint", C.c_int32), ("max_iters", C.c_int32),
                ("keyframe_parallax", C.c_double), ("init_depth", C.c_double), ("g_norm", C.c_double), ("td", C.c_double),
                ("acc_n", C.c_double), ("gyr_n", C.c_double), ("acc_w", C.c_double), ("gyr_w", C.c_double),
                ("ric", (C.c_double * 9) * 2), ("tic", (C.c_double * 3) * 2),
                ("dynamic", C.c_int32), ("use_det3d", C.c_int32), ("instance_init_min_num", C.c_int32), ("estimate", C.c_int32), ("static_inst_threshold", C.c_double),
                ("use_line", C.c_int32), ("line_min_obs", C.c_int32), ("line_sqrt_info", C.c_double * 4)]


class dv_est_state(C.Structure):
    _fields_ = [("frame", C.c_int32), ("nonlinear", C.c_int32), ("margin_old", C.c_int32), ("n_landmarks", C.c_int32),
                ("n_long", C.c_int32), ("iterations", C.c_int32), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("window", (C.c_double * 16) * 11)]


class Estimator:
    """Mirror of dynamic_vins::Estimator (estimator/estimator.h:55-164) on top of dv_est_*.
    InputIMU / ProcessMeasurements keep the reference's names; camelCase aliases provided."""

    def __init__(self, ctx, use_imu=1, stereo=1, plane_constraint=0, max_iters=8, keyframe_parallax=10.0, init_depth=5.0, g_norm=9.81, td=0.0,
                 acc_n=0.1, gyr_n=0.01, acc_w=0.001, gyr_w=1e-4, ric=None, tic=None, dynamic=0, use_det3d=0, instance_init_min_num=4, static_inst_threshold=10.0,
                 use_line=0, line_min_obs=5, line_sqrt_info=(0.0, 0.0, 0.0, 0.0), estimate=0):
        self.ctx = ctx
        c = dv_est_config()
        c.estimate = estimate          # bit 0: estimate_extrinsic 1, bit 1: estimate_td 1
        c.use_imu, c.stereo, c.plane_constraint, c.max_iters = use_imu, stereo, plane_constraint, max_iters
        c.keyframe_parallax, c.init_depth, c.g_norm, c.td = keyframe_parallax, init_depth, g_norm, td
        c.acc_n, c.gyr_n, c.acc_w, c.gyr_w = acc_n, gyr_n, acc_w, gyr_w
        c.dynamic, c.use_det3d, c.instance_init_min_num, c.static_inst_threshold = dynamic, use_det3d, instance_init_min_num, static_inst_threshold
        c.use_line, c.line_min_obs = use_line, line_min_obs
        for i in range(4):
            c.line_sqrt_info[i] = float(line_sqrt_info[i])
        for k in range(2):
            for i in range(9):
                c.ric[k][i] = float(np.asarray(ric[k]).reshape(-1)[i])
            for i in range(3):
                c.tic[k][i] = float(tic[k][i])
        self.cfg = c
        self.state = dv_est_state()
        self._check(ctx.lib.dv_est_create(ctx.h, C.byref(c)))

    def _check(self, rc):
        if rc < 0:
            raise DvinsError(self.ctx.lib.dv_last_error(self.ctx.h).decode())
        return rc

    def extrinsics(self):
        """(ric[2, 3, 3], tic[2, 3], td) as the last solve left them"""
        ric, tic, td = np.zeros(18), np.zeros(6), C.c_double(0)
        self._check(self.ctx.lib.dv_est_get_extrinsics(self.ctx.h, ric.ctypes.data, tic.ctypes.data, C.addressof(td)))
        return ric.reshape(2, 3, 3), tic.reshape(2, 3), td.value

    def InputIMU(self, t, acc, gyr):
        a = np.ascontiguousarray(acc, np.float64)
        g = np.ascontiguousarray(gyr, np.float64)
        self._check(self.ctx.lib.dv_est_input_imu(self.ctx.h, float(t), a.ctypes.data, g.ctypes.data))

    def ProcessMeasurements(self, rows, t):
        rows = np.ascontiguousarray(rows)
        rc = self._check(self.ctx.lib.dv_est_process(self.ctx.h, rows.ctypes.data, len(rows), float(t), C.byref(self.state)))
        return rc, self.state

    def ProcessMeasurementsBegin(self, rows, t):
        """host bookkeeping + enqueue of the window solve / marginalization; 0 = started, 1 = IMU data missing"""
        self._rows = np.ascontiguousarray(rows)          # kept alive until End
        return self._check(self.ctx.lib.dv_est_process_begin(self.ctx.h, self._rows.ctypes.data, len(self._rows), float(t)))

    def ProcessMeasurementsEnd(self):
        self._check(self.ctx.lib.dv_est_process_end(self.ctx.h, C.byref(self.state)))
        return self.state

    # ---- dynamic mode (cfg::slam == kDynamic): frame.instances travel with the background features ----
    def _dyn_args(self, insts, inst_feats, points):
        from .dynsim import INSTOBS_DTYPE
        self._insts = np.ascontiguousarray(insts, INSTOBS_DTYPE)
        self._ifeats = np.ascontiguousarray(inst_feats)
        self._pts = np.ascontiguousarray(points, np.float64)
        return (self._insts.ctypes.data if len(self._insts) else None, len(self._insts), self._ifeats.ctypes.data if len(self._ifeats) else None,
                self._pts.ctypes.data if len(self._pts) else None)

    def ProcessMeasurementsDynamic(self, rows, t, insts, inst_feats, points):
        rows = np.ascontiguousarray(rows)
        rc = self._check(self.ctx.lib.dv_est_process_dynamic(self.ctx.h, rows.ctypes.data, len(rows), float(t), *self._dyn_args(insts, inst_feats, points), C.byref(self.state)))
        return rc, self.state

    def ProcessMeasurementsDynamicBegin(self, rows, t, insts, inst_feats, points):
        self._rows = np.ascontiguousarray(rows)
        return self._check(self.ctx.lib.dv_est_process_dynamic_begin(self.ctx.h, self._rows.ctypes.data, len(self._rows), float(t), *self._dyn_args(insts, inst_feats, points)))

    def ProcessMeasurementsDynamicBeginEgo(self, rows, t):
        """three-phase form: the window solve goes to the GPU with the background rows alone; AttachInstances follows while it is in flight"""
        self._rows = np.ascontiguousarray(rows)
        return self._check(self.ctx.lib.dv_est_process_dynamic_begin_ego(self.ctx.h, self._rows.ctypes.data, len(self._rows), float(t)))

    def AttachInstances(self, insts, inst_feats, points):
        return self._check(self.ctx.lib.dv_est_process_dynamic_attach(self.ctx.h, *self._dyn_args(insts, inst_feats, points)))

    def static_instances(self, cap=256):
        """InstanceManager::GetOutputInstInfo as FeatureTrack reads it (system/main.cpp:194,217-245): ids reported static at the last dynamic frame's snapshot"""
        ids = np.zeros(cap, np.uint32); n = C.c_int(0)
        self._check(self.ctx.lib.dv_est_get_static_instances(self.ctx.h, ids.ctypes.data, cap, C.byref(n)))
        return ids[: n.value].copy()

    def instances(self, cap=64):
        """Estimator::im.instances (ascending id) -> (INSTSTATE_DTYPE array, [iterations, termination, initial_cost, final_cost] of the last object solve)"""
        from .dynsim import INSTSTATE_DTYPE
        out = np.zeros(cap, INSTSTATE_DTYPE); n = C.c_int(0); summ = np.zeros(4)
        self._check(self.ctx.lib.dv_est_get_instances(self.ctx.h, out.ctypes.data, cap, C.byref(n), summ.ctypes.data))
        return out[: n.value].copy(), summ

    # ---- line mode (cfg::use_line): frame.features.lines of the next frame, line landmarks read-out ----
    def SetLines(self, rows):
        self._lines = np.ascontiguousarray(rows, LINEROW_DTYPE)
        self._check(self.ctx.lib.dv_est_set_lines(self.ctx.h, self._lines.ctypes.data if len(self._lines) else None, len(self._lines)))

    def lines(self, cap=4096):
        out = np.zeros(cap, LINELM_DTYPE); n = C.c_int(0)
        self._check(self.ctx.lib.dv_est_get_lines(self.ctx.h, out.ctypes.data, cap, C.byref(n)))
        return out[: n.value].copy()

    def ClearState(self):
        self._check(self.ctx.lib.dv_est_reset(self.ctx.h))

    def window(self):
        return np.ctypeslib.as_array(self.state.window).copy()

    inputIMU = InputIMU
    processMeasurements = ProcessMeasurements


class dv_track_job(C.Structure):
    _fields_ = [("member", C.c_int32), ("mem", C.c_int32), ("gray0", C.c_void_p), ("gray1", C.c_void_p), ("stride", C.c_int32), ("mode", C.c_int32), ("t", C.c_double), ("mask", C.c_void_p)]


class Batch:
    """dv_batch: several contexts (one estimator each, same GPU) whose window solves share every launch.  Protocol per round: ProcessMeasurements*Begin
    on every member that has a frame, enqueue(), ProcessMeasurementsEnd on each."""

    def __init__(self, ctxs):
        self.lib = ctxs[0].lib
        arr = (C.c_void_p * len(ctxs))(*[c.h for c in ctxs])
        self.h = self.lib.dv_batch_create(arr, len(ctxs))
        if not self.h:
            raise DvinsError(self.lib.dv_last_error(None).decode())
        self.ctxs = list(ctxs)

    def enqueue(self):
        if self.lib.dv_batch_enqueue(self.h) != 0:
            raise DvinsError(self.lib.dv_last_error(self.ctxs[0].h).decode())

    def arrive(self):
        """rendezvous of one host thread per member (the ctypes call releases the GIL while it waits)"""
        if self.lib.dv_batch_arrive(self.h) != 0:
            raise DvinsError((self.lib.dv_last_error(self.ctxs[0].h) or self.lib.dv_last_error(None) or b"dv_batch_arrive failed").decode())

    def abort(self):
        """a worker that failed before arriving calls this so that the other members' threads do not wait for ever"""
        if getattr(self, "h", None):
            self.lib.dv_batch_abort(self.h)

    def info(self):
        a, b = C.c_longlong(0), C.c_longlong(0)
        self.lib.dv_batch_info(self.h, C.byref(a), C.byref(b))
        return dict(batched_rounds=a.value, single_rounds=b.value)

    def track_enqueue(self, jobs):
        """dv_batch_track_enqueue: the members' front ends in shared launches.  jobs: list of dict(member, gray0, gray1, t[, stride, mode, mask, mem]) with device
        pointers (ints) or host uint8 arrays for the images; collect every member with its ctx.track_stereo_collect()"""
        arr = (dv_track_job * max(len(jobs), 1))()
        self._track_keep = []
        for k, j in enumerate(jobs):
            a = arr[k]
            g0, g1 = j["gray0"], j.get("gray1")
            host = isinstance(g0, np.ndarray)
            if host:
                g0 = np.ascontiguousarray(g0); g1 = np.ascontiguousarray(g1) if g1 is not None else None
                self._track_keep += [g0, g1]
            a.member, a.mem = int(j["member"]), int(j.get("mem", 0 if host else 1))
            a.gray0 = g0.ctypes.data if host else int(g0)
            a.gray1 = (g1.ctypes.data if host else int(g1)) if g1 is not None else None
            a.stride, a.mode, a.t = int(j.get("stride", 0)), int(j.get("mode", 0)), float(j["t"])
            m = j.get("mask")
            if isinstance(m, np.ndarray):
                m = np.ascontiguousarray(m); self._track_keep.append(m); m = m.ctypes.data
            a.mask = m
        if self.lib.dv_batch_track_enqueue(self.h, C.cast(arr, C.c_void_p), len(jobs)) != 0:
            raise DvinsError((self.lib.dv_last_error(self.ctxs[0].h) or self.lib.dv_last_error(None) or b"dv_batch_track_enqueue failed").decode())

    def track_info(self):
        a, b, c = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
        self.lib.dv_batch_track_info(self.h, C.byref(a), C.byref(b), C.byref(c))
        return dict(rounds=a.value, members_batched=b.value, members_single=c.value)

    def close(self):
        if getattr(self, "h", None):
            self.lib.dv_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class dv_seq_input(C.Structure):
    _fields_ = [("left", C.c_void_p), ("right", C.c_void_p), ("times", C.c_void_p), ("n_frames", C.c_int32), ("mem", C.c_int32), ("stride", C.c_int32), ("ba_stride", C.c_int32),
                ("imu_t", C.c_void_p), ("imu_acc", C.c_void_p), ("imu_gyr", C.c_void_p), ("n_imu", C.c_int32), ("reserved2", C.c_int32)]


class dv_seq_dynamic(C.Structure):
    _fields_ = [("inv_mask", C.c_void_p), ("mask_mem", C.c_int32), ("mode", C.c_int32), ("dets", C.c_void_p), ("n_dets", C.c_void_p), ("boxes3d", C.c_void_p), ("n_boxes3d", C.c_void_p),
                ("disp", C.c_void_p), ("disp_mem", C.c_int32), ("disp_stride", C.c_int32), ("baseline", C.c_double),
                ("right_keys", C.c_void_p), ("right_keys_mem", C.c_int32), ("static_as_background", C.c_int32)]


class Runner:
    """dv_runner: the per-frame host loop of pipeline.Pipeline in C++ inside the library, for one or many sequences (include/dvins.h).  `pipes` are Pipeline
    objects (each owns its Context + Estimator and a SyntheticSequence whose frames are resident in HBM); the runner takes over driving them."""

    def __init__(self, pipes, group_size=0, threads=1, first_frame=0, host_frames=False):
        """host_frames: the frames are handed over as PINNED HOST buffers: every frame's trip over PCIe then lies inside whatever region times dv_runner_run — the
        PCIe-inclusive rate of bench.py's host_frames_line.  True / "pinned": dv_seq_input::mem = DV_MEM_PINNED, the pyramid kernel reads the frames in place (no copy
        engine in the per-frame path); "engine": DV_MEM_HOST, one hipMemcpy2DAsync per image on the tracking stream (rounds 1-5)"""
        from .frontend import DV_MEM_DEVICE, DV_MEM_HOST, DV_MEM_PINNED
        self.lib = pipes[0].ctx.lib
        self.pipes = list(pipes)
        n = len(pipes)
        self._keep = []
        arr = (dv_seq_input * n)()
        for i, p in enumerate(pipes):
            q = p.seq
            frames = q.frames[first_frame:]
            if host_frames:
                import torch
                frames = [(f[0].cpu().pin_memory(), f[1].cpu().pin_memory()) for f in frames]
                assert all(a.is_contiguous() and b.is_contiguous() for a, b in frames)
                # one throw-away DMA out of every pinned buffer: the FIRST transfer from a freshly pinned page range pays its mapping (seen as an intermittent 40 - 80 ms stall in the
                # first timed block when only the warm-up's frames had been uploaded before); the per-frame upload itself stays inside whatever times dv_runner_run
                scratch = torch.empty_like(q.frames[0][0])
                for a, b in frames:
                    scratch.copy_(a, non_blocking=True); scratch.copy_(b, non_blocking=True)
                torch.cuda.synchronize()
                self._keep.append(frames)
            L = (C.c_void_p * len(frames))(*[f[0].data_ptr() for f in frames]); R = (C.c_void_p * len(frames))(*[f[1].data_ptr() for f in frames])
            t = np.ascontiguousarray(q.times[first_frame:], np.float64)
            it, ia, ig = np.ascontiguousarray(q.imu_t, np.float64), np.ascontiguousarray(q.imu_a, np.float64), np.ascontiguousarray(q.imu_g, np.float64)
            k0 = getattr(p, "k_imu", 0)                 # samples the Python pipeline has already fed
            self._keep += [L, R, t, it, ia, ig]
            a = arr[i]
            a.left, a.right, a.times = C.cast(L, C.c_void_p), C.cast(R, C.c_void_p), t.ctypes.data
            a.n_frames, a.mem, a.stride, a.ba_stride = len(frames), (DV_MEM_DEVICE if not host_frames else DV_MEM_HOST if host_frames == "engine" else DV_MEM_PINNED), 0, getattr(p, "ba_stride", 1)
            a.imu_t, a.imu_acc, a.imu_gyr, a.n_imu = it.ctypes.data + 8 * k0, ia.ctypes.data + 24 * k0, ig.ctypes.data + 24 * k0, len(it) - k0
        self._arr = arr
        ctxs = (C.c_void_p * n)(*[p.ctx.h for p in pipes])
        self.h = self.lib.dv_runner_create(ctxs, C.cast(arr, C.c_void_p), n, int(group_size), int(threads))
        if not self.h:
            raise DvinsError((self.lib.dv_last_error(None) or b"dv_runner_create failed").decode())
        for i, p in enumerate(pipes):
            if getattr(p, "mode", 0) != 0:          # a DynamicPipeline: hand the per-frame perception outputs of its sequence to the runner's dynamic loop
                self._set_dynamic(i, p, first_frame)

    def _set_dynamic(self, i, p, first_frame):
        from .dynsim import BOX3D_DTYPE
        from .frontend import DV_MEM_DEVICE, dv_inst_det
        q = p.seq
        nf = len(q.frames) - first_frame
        masks = (C.c_void_p * nf)(*[m.data_ptr() for m in q.inv_mask_dev[first_frame:]])
        det_ptrs, n_dets, box_ptrs, n_boxes = (C.c_void_p * nf)(), np.zeros(nf, np.int32), (C.c_void_p * nf)(), np.zeros(nf, np.int32)
        for k in range(nf):
            dets = q.dets[first_frame + k]
            arr = (dv_inst_det * max(len(dets), 1))()
            for j, d in enumerate(dets):
                m = np.ascontiguousarray(d["mask"], np.uint8)
                pts = None if d.get("points") is None else np.ascontiguousarray(d["points"], np.float64)
                x, y, w, h = [int(v) for v in d["rect"]]
                arr[j].track_id, arr[j].class_id, arr[j].x, arr[j].y, arr[j].w, arr[j].h = int(d["track_id"]), int(d.get("class_id", 0)), x, y, w, h
                arr[j].mask = m.ctypes.data
                arr[j].points = pts.ctypes.data if pts is not None and len(pts) else None
                arr[j].n_points = 0 if pts is None else len(pts)
                self._keep += [m, pts]
            det_ptrs[k] = C.cast(arr, C.c_void_p); n_dets[k] = len(dets)
            b3 = np.ascontiguousarray(q.boxes3d[first_frame + k], BOX3D_DTYPE) if p.use_det3d else np.zeros(0, BOX3D_DTYPE)
            box_ptrs[k] = b3.ctypes.data if len(b3) else None; n_boxes[k] = len(b3)
            self._keep += [arr, b3]
        use_disp = getattr(p, "extra_from_disparity", False)
        disps = (C.c_void_p * nf)(*[d.data_ptr() for d in q.disp_dev[first_frame:]]) if use_disp else None
        dyn = dv_seq_dynamic()
        dyn.inv_mask, dyn.mask_mem, dyn.mode = C.cast(masks, C.c_void_p), DV_MEM_DEVICE, int(p.mode)
        dyn.dets, dyn.n_dets = C.cast(det_ptrs, C.c_void_p), n_dets.ctypes.data
        dyn.boxes3d, dyn.n_boxes3d = C.cast(box_ptrs, C.c_void_p), n_boxes.ctypes.data
        dyn.disp, dyn.disp_mem, dyn.disp_stride, dyn.baseline = (C.cast(disps, C.c_void_p) if use_disp else None), DV_MEM_DEVICE, 0, float(q.baseline)
        dyn.static_as_background = 1 if getattr(p, "static_as_background", False) else 0
        rk = getattr(q, "right_keys", None)          # VIODE-style sequences: per frame the uint32 key image of seg1 (host arrays)
        if rk is not None:
            rk = [np.ascontiguousarray(a, np.uint32) for a in rk[first_frame:]]
            rkp = (C.c_void_p * nf)(*[a.ctypes.data for a in rk])
            dyn.right_keys, dyn.right_keys_mem = C.cast(rkp, C.c_void_p), 0
            self._keep += [rk, rkp]
        self._keep += [masks, det_ptrs, n_dets, box_ptrs, n_boxes, disps, dyn]
        if self.lib.dv_runner_set_dynamic(self.h, i, C.byref(dyn)) != 0:
            raise DvinsError(self.lib.dv_runner_error(self.h).decode())

    def dynamic_stats(self, i):
        a, b, c, d = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0), C.c_int(0)
        self.lib.dv_runner_dynamic_stats(self.h, i, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return dict(object_detections=a.value, object_features=b.value, frames_with_objects=c.value, min_detections=d.value)

    def run(self, rounds):
        """rounds frames of every sequence -> wall seconds of the call (all streams drained)"""
        w = C.c_double(0)
        if self.lib.dv_runner_run(self.h, int(rounds), C.byref(w)) != 0:
            raise DvinsError(self.lib.dv_runner_error(self.h).decode())
        return w.value

    def get(self, i, cap=100000):
        st = dv_est_state(); poses = np.zeros((cap, 8)); n = C.c_int(0); it = C.c_longlong(0); fr = C.c_longlong(0); nr = C.c_int(0)
        self.lib.dv_runner_get(self.h, i, C.byref(st), poses.ctypes.data, cap, C.byref(n), C.byref(it), C.byref(fr), C.byref(nr))
        self.last_rows = nr.value
        return st, poses[: n.value].copy(), it.value, fr.value

    def set(self, key, value):
        if self.lib.dv_runner_set(self.h, key.encode(), int(value)) != 0:
            raise DvinsError(self.lib.dv_runner_error(self.h).decode())

    def track_info(self):
        a, b, c = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
        self.lib.dv_runner_track_info(self.h, C.byref(a), C.byref(b), C.byref(c))
        return dict(rounds=a.value, members_batched=b.value, members_single=c.value)

    def batch_timing(self, on=1):
        """-> (avg ms per launch of [be_solve_batch, be_eval_batch, be_reduce_batch], rounds timed, windows per launch)"""
        o = np.zeros(3); r = C.c_longlong(0); w = C.c_int(0)
        self.lib.dv_runner_batch_timing(self.h, int(on), o.ctypes.data, C.byref(r), C.byref(w))
        return o, r.value, w.value

    def row_log(self, i, cap=100000):
        """diagnostics, per frame handed to the back end: [frame index, rows collected, hash of the rows, solver iterations] (uint64)"""
        rows = np.zeros((cap, 4), dtype=np.uint64); n = C.c_int(0)
        self.lib.dv_runner_get_row_log(self.h, i, rows.ctypes.data, cap, C.byref(n))
        return rows[: n.value].copy()

    def batch_rounds(self):
        """(batched_rounds, single_rounds) summed over the groups' dv_batch objects: single = rounds that fell back to every member's own launches"""
        a, b = C.c_longlong(0), C.c_longlong(0)
        if hasattr(self.lib, "dv_runner_batch_rounds") and self.lib.dv_runner_batch_rounds(self.h, C.byref(a), C.byref(b)) == 0:
            return a.value, b.value
        return None

    def frame_clock(self, i, which=0, cap=100000):
        """host steady clock (seconds) at the end of every frame of sequence i so far (which = 1: at which its tracker thread delivered every frame); diagnostics"""
        t = np.zeros(cap, np.float64); n = C.c_int(0)
        self.lib.dv_runner_get_frame_clock(self.h, i, int(which), t.ctypes.data, cap, C.byref(n))
        return t[: n.value].copy()

    def frames(self, i, cap=100000):
        """every frame handed to the back end: rows [t, px py pz qx qy qz qw, nonlinear]"""
        rows = np.zeros((cap, 9)); n = C.c_int(0)
        self.lib.dv_runner_get_frames(self.h, i, rows.ctypes.data, cap, C.byref(n))
        return rows[: n.value].copy()

    def close(self):
        if getattr(self, "h", None):
            self.lib.dv_runner_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
