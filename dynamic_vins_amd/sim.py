"""Synthetic stereo + IMU sequences (seeded).  Input generation only — neither the
product path nor the oracle.

  * Trajectory: smooth figure-8 at ~1 m/s with gentle roll/pitch, analytic derivatives -> exact IMU.
  * Scene: points on the walls/floor/ceiling of a box room, so depths span 2..12 m.
  * FeatureSim: feature-level front end (projects the 3-D points, keeps ids while visible, tops up to
    max_cnt) producing dv_feat rows directly — used to drive the back end without images.
  * RoomRenderer: renders the same room as textured planes through the (distorted) pinhole cameras, so the
    real front end (LK + Shi-Tomasi) can be run on images that are consistent with the IMU stream.

Conventions (VINS): world z up, gravity g = (0,0,+g_norm) subtracted from rotated accelerometer
readings; body x forward, y left, z up; camera z forward, x right, y down.
"""
import numpy as np

FEAT_DTYPE = np.dtype([("id", np.uint32), ("track_cnt", np.int32), ("has_right", np.int32), ("pad_", np.int32),
                       ("left", np.float64, 7), ("right", np.float64, 7)])

TEX_SEED = 0xD1CE
R_IC = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])      # body_T_cam rotation
T_IC0 = np.array([0.0, 0.0, 0.0])
T_IC1 = np.array([0.0, -0.12, 0.0])                                          # 0.12 m baseline (ZED), camera 1 to the right

ZED = dict(fx=701.406049185687, fy=700.7199834541797, cx=663.9703743586792, cy=362.02045484177154,
           k1=-0.17198906485492285, k2=0.024624053031210322, p1=0.0003391614313509814, p2=-0.00045583634752113735)
VIODE = dict(fx=376.0, fy=376.0, cx=376.0, cy=240.0, k1=0.0, k2=0.0, p1=0.0, p2=0.0)          # config/viode/cam0_pinhole.yaml (752x480, no distortion)
EUROC = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375, k1=-0.28340811, k2=0.07395907, p1=0.00019359, p2=1.76187114e-05)


def rig(baseline=0.12, body_is_camera=False):
    """Stereo rig of a synthetic sequence.  Rendering always places camera 0 at the trajectory's body origin looking along body x (R_IC) and camera 1
    `baseline` metres to its right.  What the ESTIMATOR is told (body_T_cam0 / body_T_cam1 of the YAML) is either that same rig (VIO configs: the body is
    the IMU) or — the vision-only ZED configs, config/custom/zed_1280x720_vision_only/dynamic.yaml:38-53 — identity extrinsics with camera 1 at
    (+baseline, 0, 0): the body IS camera 0 and the trajectory comes out in the first camera's frame (ATE aligns it)."""
    t1 = np.array([0.0, -baseline, 0.0])
    if body_is_camera:
        return dict(t_ic1=t1, est_ric=[np.eye(3), np.eye(3)], est_tic=[np.zeros(3), np.array([baseline, 0.0, 0.0])])
    return dict(t_ic1=t1, est_ric=[R_IC, R_IC], est_tic=[T_IC0, t1])


def scaled_cam(cam, w, h, w0, h0):
    s = dict(cam)
    s["fx"] *= w / w0; s["cx"] *= w / w0; s["fy"] *= h / h0; s["cy"] *= h / h0
    return s


def cam_tuple(c):
    return (c["fx"], c["fy"], c["cx"], c["cy"], c["k1"], c["k2"], c["p1"], c["p2"])


def rot_zyx(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx


class Trajectory:
    """Figure-8 in the x-y plane; orientation = yaw along the velocity + small roll/pitch oscillation."""

    def __init__(self, a=4.0, b=2.5, period=30.0, z_amp=0.25):
        self.a, self.b, self.T, self.z_amp = a, b, period, z_amp

    def p(self, t):
        w = 2 * np.pi / self.T
        return np.array([self.a * np.sin(w * t), self.b * np.sin(2 * w * t), self.z_amp * np.sin(3 * w * t)])

    def _d(self, f, t, h=1e-4):
        return (f(t + h) - f(t - h)) / (2 * h)

    def v(self, t):
        w = 2 * np.pi / self.T
        return np.array([self.a * w * np.cos(w * t), 2 * self.b * w * np.cos(2 * w * t), 3 * self.z_amp * w * np.cos(3 * w * t)])

    def acc(self, t):
        w = 2 * np.pi / self.T
        return np.array([-self.a * w * w * np.sin(w * t), -4 * self.b * w * w * np.sin(2 * w * t), -9 * self.z_amp * w * w * np.sin(3 * w * t)])

    def ypr(self, t):
        v = self.v(t)
        yaw = np.arctan2(v[1], v[0])
        return np.array([yaw, 0.05 * np.sin(0.7 * t), 0.04 * np.sin(0.9 * t + 1.0)])

    def R(self, t):
        y, p, r = self.ypr(t)
        return rot_zyx(y, p, r)

    def omega_body(self, t, h=1e-5):
        R0, R1 = self.R(t - h), self.R(t + h)
        dR = R0.T @ R1
        w = np.array([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]]) / 2.0
        return w / (2 * h)


def imu_stream(traj, t0, t1, rate=200.0, g_norm=9.81, acc_n=0.0, gyr_n=0.0, acc_w=0.0, gyr_w=0.0, seed=0xBEEF,
               ba0=(0.0, 0.0, 0.0), bg0=(0.0, 0.0, 0.0)):
    """returns (t, acc, gyr) arrays.  Noise follows the *discrete* interpretation IntegrationBase uses
    (estimator/imu/integration_base.h:37-43,118-131): acc_n / gyr_n are per-sample standard deviations,
    acc_w / gyr_w are bias increments per second (per-sample increment std = w * dt)."""
    rng = np.random.default_rng(seed)
    n = int(round((t1 - t0) * rate)) + 1
    ts = t0 + np.arange(n) / rate
    g = np.array([0.0, 0.0, g_norm])
    ba, bg = np.array(ba0, float), np.array(bg0, float)
    dt = 1.0 / rate
    acc, gyr = np.zeros((n, 3)), np.zeros((n, 3))
    for i, t in enumerate(ts):
        R = traj.R(t)
        acc[i] = R.T @ (traj.acc(t) + g) + ba + rng.normal(0, 1, 3) * acc_n
        gyr[i] = traj.omega_body(t) + bg + rng.normal(0, 1, 3) * gyr_n
        ba = ba + rng.normal(0, 1, 3) * acc_w * dt
        bg = bg + rng.normal(0, 1, 3) * gyr_w * dt
    return ts, acc, gyr


def room_points(n, half=(9.0, 7.0, 3.0), seed=7):
    """points on the six faces of a box centred at the origin"""
    rng = np.random.default_rng(seed)
    hx, hy, hz = half
    pts = []
    for _ in range(n):
        f = rng.integers(0, 6)
        u, v = rng.uniform(-1, 1, 2)
        if f == 0: pts.append([hx, u * hy, v * hz])
        elif f == 1: pts.append([-hx, u * hy, v * hz])
        elif f == 2: pts.append([u * hx, hy, v * hz])
        elif f == 3: pts.append([u * hx, -hy, v * hz])
        elif f == 4: pts.append([u * hx, v * hy, hz])
        else: pts.append([u * hx, v * hy, -hz])
    return np.array(pts)


def distort(cam, xn, yn):
    k1, k2, p1, p2 = cam["k1"], cam["k2"], cam["p1"], cam["p2"]
    r2 = xn * xn + yn * yn
    rad = k1 * r2 + k2 * r2 * r2
    xd = xn + xn * rad + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)
    yd = yn + yn * rad + 2 * p2 * xn * yn + p1 * (r2 + 2 * yn * yn)
    return cam["fx"] * xd + cam["cx"], cam["fy"] * yd + cam["cy"]


class FeatureSim:
    """Feature-level front end: deterministic ids, max_cnt top-up, pixel noise, velocities."""

    def __init__(self, traj, cam, w, h, points, max_cnt=150, pix_sigma=0.3, seed=3, stereo=True, min_depth=0.5):
        self.traj, self.cam, self.w, self.h, self.pts = traj, cam, w, h, points
        self.max_cnt, self.sig, self.stereo, self.min_depth = max_cnt, pix_sigma, stereo, min_depth
        self.rng = np.random.default_rng(seed)
        self.next_id = 1
        self.tracked = {}          # point index -> (id, track_cnt)
        self.prev_un, self.prev_run, self.prev_t = {}, {}, None

    def _project(self, t, t_ic):
        R, p = self.traj.R(t), self.traj.p(t)
        Pc = (R_IC.T @ (R.T @ (self.pts - p).T - t_ic[:, None])).T
        z = Pc[:, 2]
        ok = z > self.min_depth
        xn = np.where(ok, Pc[:, 0] / np.where(ok, z, 1), 0)
        yn = np.where(ok, Pc[:, 1] / np.where(ok, z, 1), 0)
        u, v = distort(self.cam, xn, yn)
        ok &= (u > 2) & (u < self.w - 3) & (v > 2) & (v < self.h - 3) & (np.abs(xn) < 1.2) & (np.abs(yn) < 0.9)
        return xn, yn, u, v, ok

    def frame(self, t):
        xl, yl, ul, vl, okl = self._project(t, T_IC0)
        xr, yr, ur, vr, okr = self._project(t, T_IC1)
        self.tracked = {k: (i, c + 1) for k, (i, c) in self.tracked.items() if okl[k]}
        if len(self.tracked) < self.max_cnt:
            cand = [k for k in np.flatnonzero(okl) if k not in self.tracked]
            self.rng.shuffle(cand)
            for k in cand[: self.max_cnt - len(self.tracked)]:
                self.tracked[k] = (self.next_id, 1)
                self.next_id += 1
        rows = np.zeros(len(self.tracked), FEAT_DTYPE)
        dt = (t - self.prev_t) if self.prev_t is not None else 1.0
        new_un, new_run = {}, {}
        f = self.cam["fx"]
        for r, (k, (fid, cnt)) in zip(rows, sorted(self.tracked.items(), key=lambda kv: kv[1][0])):
            n = self.rng.normal(0, self.sig, 4)
            x, y = np.float32(xl[k] + n[0] / f), np.float32(yl[k] + n[1] / f)
            vx, vy = (0.0, 0.0)
            if fid in self.prev_un:
                vx, vy = np.float32((x - self.prev_un[fid][0]) / dt), np.float32((y - self.prev_un[fid][1]) / dt)
            new_un[fid] = (x, y)
            r["id"], r["track_cnt"] = fid, cnt
            r["left"] = [x, y, 1.0, np.float32(ul[k] + n[0]), np.float32(vl[k] + n[1]), vx, vy]
            if self.stereo and okr[k]:
                x2, y2 = np.float32(xr[k] + n[2] / f), np.float32(yr[k] + n[3] / f)
                vx2, vy2 = (0.0, 0.0)
                if fid in self.prev_run:
                    vx2, vy2 = np.float32((x2 - self.prev_run[fid][0]) / dt), np.float32((y2 - self.prev_run[fid][1]) / dt)
                new_run[fid] = (x2, y2)
                r["has_right"] = 1
                r["right"] = [x2, y2, 1.0, np.float32(ur[k] + n[2]), np.float32(vr[k] + n[3]), vx2, vy2]
        self.prev_un, self.prev_run, self.prev_t = new_un, new_run, t
        return rows


class LineSim:
    """Line-level front end (what TrackImageLine hands over as FeatureBackground::lines): 3-D segments on the room's walls, ids fixed per segment, a row
    per frame in which both end points project inside the image; end points carry pixel noise and, like the LSD detector, are not the same physical points
    from frame to frame (each end slides along the segment by up to `slide` of its length)."""

    def __init__(self, traj, w, h, n=60, half=(9.0, 7.0, 3.0), pix_sigma=0.3, focal=460.0, slide=0.05, seed=11):
        rng = np.random.default_rng(seed)
        a = room_points(n, half, seed=seed + 1)
        d = rng.normal(0, 1, (n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        self.a, self.b = a, a + d * rng.uniform(0.8, 2.5, (n, 1))
        self.traj, self.w, self.h, self.sig, self.f, self.slide = traj, w, h, pix_sigma, focal, slide
        self.rng = rng

    def _norm(self, t, P, t_ic):
        R, p = self.traj.R(t), self.traj.p(t)
        Pc = (R_IC.T @ (R.T @ (P - p).T - t_ic[:, None])).T
        ok = Pc[:, 2] > 0.5
        z = np.where(ok, Pc[:, 2], 1.0)
        x, y = Pc[:, 0] / z, Pc[:, 1] / z
        return x, y, ok & (np.abs(x) < 1.2) & (np.abs(y) < 0.75)

    def frame(self, t):
        from .backend import LINEROW_DTYPE
        n = len(self.a)
        s = self.rng.uniform(0, self.slide, (n, 2))
        A = self.a + (self.b - self.a) * s[:, :1]
        B = self.b - (self.b - self.a) * s[:, 1:]
        rows = []
        xa, ya, oka = self._norm(t, A, T_IC0); xb, yb, okb = self._norm(t, B, T_IC0)
        xar, yar, okar = self._norm(t, A, T_IC1); xbr, ybr, okbr = self._norm(t, B, T_IC1)
        noise = self.rng.normal(0, self.sig / self.f, (n, 8))
        for k in range(n):
            if not (oka[k] and okb[k]):
                continue
            r = np.zeros((), LINEROW_DTYPE)
            r["id"] = k + 1
            r["left"] = np.array([xa[k], ya[k], xb[k], yb[k]]) + noise[k, :4]
            if okar[k] and okbr[k]:
                r["has_right"] = 1
                r["right"] = np.array([xar[k], yar[k], xbr[k], ybr[k]]) + noise[k, 4:]
            rows.append(r)
        return np.array(rows, LINEROW_DTYPE) if rows else np.zeros(0, LINEROW_DTYPE)


class SegmentSim:
    """Pixel-level stand-in for the LSD + LBD line detector / matcher upstream of TrackImageLine (line_detector/line_detector.cpp — OpenCV line_descriptor, CPU,
    out of scope): 3-D segments on the room's walls projected through the DISTORTED pinhole cameras of the rig; per frame the matched segments of the left
    image (id, x1 y1 x2 y2 in pixels, float32 like cv::line_descriptor::KeyLine) and of the right image.  The end points slide along the segment from frame to
    frame like a detector's do.  What follows it on the path — FrameLines::UndistortedLineEndPoints, the `lines` map of SetOutputFeats — is the product's."""

    def __init__(self, traj, cam, w, h, n=80, half=(9.0, 7.0, 3.0), pix_sigma=0.3, slide=0.05, seed=11, t_ic1=T_IC1):
        rng = np.random.default_rng(seed)
        a = room_points(n, half, seed=seed + 1)
        d = rng.normal(0, 1, (n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        self.a, self.b = a, a + d * rng.uniform(0.8, 2.5, (n, 1))
        self.traj, self.cam, self.w, self.h, self.sig, self.slide, self.t_ic1, self.rng = traj, cam, w, h, pix_sigma, slide, np.asarray(t_ic1, float), rng

    def _pix(self, t, P, t_ic):
        R, p = self.traj.R(t), self.traj.p(t)
        Pc = (R_IC.T @ (R.T @ (P - p).T - t_ic[:, None])).T
        ok = Pc[:, 2] > 0.5
        z = np.where(ok, Pc[:, 2], 1.0)
        x, y = Pc[:, 0] / z, Pc[:, 1] / z
        u, v = distort(self.cam, x, y)
        return u, v, ok & (np.abs(x) < 1.1) & (np.abs(y) < 0.7) & (u > 2) & (u < self.w - 3) & (v > 2) & (v < self.h - 3)

    def frame(self, t):
        """-> (ids_left u32 [m], segs_left f32 [m, 4], ids_right u32 [k], segs_right f32 [k, 4]); every right id also appears on the left"""
        n = len(self.a)
        s = self.rng.uniform(0, self.slide, (n, 2))
        A = self.a + (self.b - self.a) * s[:, :1]
        B = self.b - (self.b - self.a) * s[:, 1:]
        ua, va, oka = self._pix(t, A, T_IC0); ub, vb, okb = self._pix(t, B, T_IC0)
        uar, var, okar = self._pix(t, A, self.t_ic1); ubr, vbr, okbr = self._pix(t, B, self.t_ic1)
        noise = self.rng.normal(0, self.sig, (n, 8))
        left = oka & okb
        right = left & okar & okbr
        L = (np.stack([ua, va, ub, vb], 1) + noise[:, :4]).astype(np.float32)
        Rr = (np.stack([uar, var, ubr, vbr], 1) + noise[:, 4:]).astype(np.float32)
        ids = np.arange(1, n + 1, dtype=np.uint32)
        return ids[left], np.ascontiguousarray(L[left]), ids[right], np.ascontiguousarray(Rr[right])


def line_rows(ids_l, un_l, ids_r, un_r):
    """FeatureBackground::lines as dv_line_row records (SetOutputFeats, background_tracker.cpp:373-392): left entries in detector order, the right
    observation attached to the same id"""
    from .backend import LINEROW_DTYPE
    rows = np.zeros(len(ids_l), LINEROW_DTYPE)
    rows["id"], rows["left"] = ids_l, un_l
    pos = {int(i): k for k, i in enumerate(ids_l)}
    for i, u in zip(ids_r, un_r):
        k = pos[int(i)]
        rows["has_right"][k] = 1; rows["right"][k] = u
    return rows


def align_ate(est, gt):
    """Horn alignment (rotation + translation, no scale) and translational RMSE — the metric of
    dynamic_vins/scripts/tum_tools/evaluate_ate.py:47-79,155 (align() + rmse)."""
    est, gt = np.asarray(est, float).T, np.asarray(gt, float).T          # 3 x n
    mc, dc = est.mean(1, keepdims=True), gt.mean(1, keepdims=True)
    W = (est - mc) @ (gt - dc).T
    U, _, Vh = np.linalg.svd(W.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0:
        S[2, 2] = -1
    R = U @ S @ Vh
    t = dc - R @ mc
    err = R @ est + t - gt
    return float(np.sqrt((err * err).sum(0).mean())), R, t
