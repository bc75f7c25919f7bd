"""Synthetic dynamic-mode inputs (seeded; SURVEY.md 8(d) "dynamic variant"): rigid boxes moving through the box room of sim.py.
Input generation only — neither the product path nor the oracle.

  * MovingBox: a box with dims (x, y, z) in its own frame (y pointing down, like the camera-style object frame of Box3D::R_cioi,
    basic/box3d.h:79-83) whose centre oscillates along a horizontal line while it yaws.
  * InstSim: feature-level stand-in for InstsFeatManager::Output() (front_end/dynamic_tracker.cpp:521-577): per visible box its tracked
    surface points as dv_feat rows (left + right observation, velocities), a 3-D detection (class, centre in the camera frame, dims, yaw)
    and "extra" 3-D points in the camera frame (what DetectExtraPoints + the PCL clustering deliver from the disparity map).
"""
import ctypes as C

import numpy as np

from . import sim

BOX3D_DTYPE = np.dtype([("class_id", "i4"), ("pad_", "i4"), ("score", "f8"), ("center", "f8", 3), ("dims", "f8", 3), ("yaw", "f8"),
                        ("rect_min", "f4", 2), ("rect_max", "f4", 2)])
INSTOBS_DTYPE = np.dtype([("id", "u4"), ("has_box3d", "i4"), ("first_feat", "i4"), ("n_feats", "i4"), ("first_point", "i4"), ("n_points", "i4"),
                          ("rect", "f4", 4), ("box3d", BOX3D_DTYPE)])
INSTSTATE_DTYPE = np.dtype([("id", "u4"), ("is_initial", "i4"), ("is_tracking", "i4"), ("is_curr_visible", "i4"), ("is_static", "i4"), ("is_init_velocity", "i4"),
                            ("age", "i4"), ("lost_number", "i4"), ("static_frame", "i4"), ("n_landmarks", "i4"), ("n_valid", "i4"), ("triangle_num", "i4"),
                            ("dims", "f8", 3), ("vel_v", "f8", 3), ("vel_a", "f8", 3), ("window", "f8", (11, 7)), ("time", "f8", 11)])
assert BOX3D_DTYPE.itemsize == 88 and INSTOBS_DTYPE.itemsize == 128 and INSTSTATE_DTYPE.itemsize == 48 + 72 + 616 + 88

# object frame -> world for yaw 0: object x = world x, object y = world -z (down), object z = world y
R_WO0 = np.array([[1.0, 0, 0], [0, 0, 1.0], [0, -1.0, 0]])


class MovingBox:
    def __init__(self, inst_id, center, direction, amplitude, period, dims=(1.2, 1.0, 2.0), yaw0=0.3, yaw_rate=0.1, phase=0.0, class_id=2):
        self.id, self.c0, self.dir = inst_id, np.asarray(center, float), np.asarray(direction, float) / np.linalg.norm(direction)
        self.amp, self.T, self.dims, self.yaw0, self.yaw_rate, self.phase, self.class_id = amplitude, period, np.asarray(dims, float), yaw0, yaw_rate, phase, class_id

    def p(self, t):
        return self.c0 + self.dir * self.amp * np.sin(2 * np.pi * t / self.T + self.phase)

    def R(self, t):
        return sim.rot_zyx(self.yaw0 + self.yaw_rate * t, 0.0, 0.0) @ R_WO0

    def surface_points(self, n, rng):
        """n points on the six faces, object frame"""
        hx, hy, hz = self.dims / 2
        pts = np.zeros((n, 3))
        for k in range(n):
            f = rng.integers(0, 6); u, v = rng.uniform(-1, 1, 2)
            pts[k] = [(hx, u * hy, v * hz), (-hx, u * hy, v * hz), (u * hx, hy, v * hz), (u * hx, -hy, v * hz), (u * hx, v * hy, hz), (u * hx, v * hy, -hz)][f]
        return pts


class EscortBox(MovingBox):
    """A box that travels WITH the camera (a vehicle ahead of the ego vehicle — the KITTI / VIODE situation): its centre is a point `ahead` metres in front
    of the trajectory's heading, `left` metres to the side and `up` metres above, each slowly oscillating, its yaw follows the heading plus a slow swing.  It moves
    through the world at roughly the camera's speed and stays in view in every frame, which the room-fixed MovingBoxes do not."""

    def __init__(self, inst_id, traj, ahead, left, up, dims=(0.9, 0.8, 1.4), swing=(0.5, 0.35, 0.1), rate=(0.55, 0.4, 0.3), phase=0.0, yaw_off=0.3, yaw_swing=0.4, class_id=2):
        self.id, self.traj, self.off, self.dims = inst_id, traj, np.array([ahead, left, up], float), np.asarray(dims, float)
        self.swing, self.rate, self.phase, self.yaw_off, self.yaw_swing, self.class_id = np.asarray(swing, float), np.asarray(rate, float), phase, yaw_off, yaw_swing, class_id

    def _yaw(self, t):
        return float(self.traj.ypr(t)[0])

    def p(self, t):
        o = self.off + self.swing * np.sin(self.rate * t + self.phase + np.array([0.0, 1.3, 2.1]))
        return self.traj.p(t) + sim.rot_zyx(self._yaw(t), 0.0, 0.0) @ o

    def R(self, t):
        return sim.rot_zyx(self._yaw(t) + self.yaw_off + self.yaw_swing * np.sin(0.45 * t + self.phase), 0.0, 0.0) @ R_WO0


def escort_boxes(traj, n=4):
    """n boxes escorting the camera at 2.6 - 5 m, spread over the field of view so that they do not hide each other for long: the bench's and the
    reference-parameter tests' dynamic scene (objects in EVERY frame, >= 3 detections per frame for n >= 4)"""
    spec = [(3.0, 1.3, -0.25, (0.9, 0.8, 1.4), 0.0, 0.35), (3.6, -1.5, -0.1, (1.0, 0.9, 1.6), 1.1, -0.5), (4.6, 0.1, 0.35, (1.2, 0.8, 1.8), 2.3, 0.9),
            (2.7, -0.4, -0.75, (0.7, 0.6, 1.0), 3.1, 1.4), (5.2, 2.4, 0.0, (1.1, 1.0, 1.7), 4.0, -1.1), (5.0, -2.6, 0.3, (1.0, 0.9, 1.5), 5.2, 0.2)]
    return [EscortBox(k + 1, traj, a, l, u, dims=d, phase=ph, yaw_off=yo) for k, (a, l, u, d, ph, yo) in enumerate(spec[:n])]


def default_boxes():
    """three boxes: two moving at 2-3 m/s peak, one nearly at rest"""
    return [MovingBox(1, (5.0, 1.5, -1.0), (0, 1, 0), 2.5, 6.0, yaw0=0.4, yaw_rate=0.15),
            MovingBox(2, (-4.0, -2.0, -0.5), (1, 0.3, 0), 2.0, 5.0, dims=(1.0, 0.8, 1.6), yaw0=-0.8, yaw_rate=-0.1, phase=1.0),
            MovingBox(3, (1.0, 4.5, -1.2), (1, 0, 0), 0.15, 9.0, dims=(1.4, 1.2, 2.4), yaw0=1.2, yaw_rate=0.0, phase=0.5)]


def ring_boxes(n):
    """n boxes spread over the room (deterministic): the default three plus boxes on two rings, moving along tangents with different periods"""
    out = default_boxes()
    for k in range(max(0, n - 3)):
        ang = 0.7 + 2.399963 * k                       # golden-angle spacing
        r = 4.0 + 1.5 * (k % 3)
        c = (r * np.cos(ang), r * np.sin(ang), -1.0 + 0.2 * (k % 4))
        d = (-np.sin(ang), np.cos(ang), 0.0)
        out.append(MovingBox(4 + k, c, d, 0.5 + 0.4 * (k % 5), 5.0 + k % 4, dims=(1.0 + 0.1 * (k % 4), 0.9, 1.6 + 0.2 * (k % 3)), yaw0=ang, yaw_rate=0.05 * ((k % 3) - 1), phase=0.3 * k))
    return out[:n]


class InstSim:
    def __init__(self, traj, cam, w, h, boxes=None, max_cnt=50, n_surface=220, n_extra=60, pix_sigma=0.3, seed=11, first_id=100000, with_det3d=True):
        self.traj, self.cam, self.w, self.h = traj, cam, w, h
        self.boxes = boxes if boxes is not None else default_boxes()
        self.max_cnt, self.n_extra, self.sig, self.with_det3d = max_cnt, n_extra, pix_sigma, with_det3d
        self.rng = np.random.default_rng(seed)
        self.surf = {b.id: b.surface_points(n_surface, self.rng) for b in self.boxes}
        self.tracked = {b.id: {} for b in self.boxes}          # point index -> feature id
        self.prev_un = {b.id: {} for b in self.boxes}; self.prev_run = {b.id: {} for b in self.boxes}
        self.next_id, self.prev_t = first_id, None

    def _project(self, Pw, t, t_ic):
        R, p = self.traj.R(t), self.traj.p(t)
        Pc = (sim.R_IC.T @ (R.T @ (Pw - p).T - t_ic[:, None])).T
        z = Pc[:, 2]
        ok = z > 0.5
        zz = np.where(ok, z, 1.0)
        xn, yn = Pc[:, 0] / zz, Pc[:, 1] / zz
        u, v = sim.distort(self.cam, xn, yn)
        ok &= (u > 2) & (u < self.w - 3) & (v > 2) & (v < self.h - 3) & (np.abs(xn) < 1.2) & (np.abs(yn) < 0.9)
        return Pc, xn, yn, u, v, ok

    def frame(self, t, visible=None):
        """-> (insts [INSTOBS_DTYPE], feats [sim.FEAT_DTYPE], points [m, 3] camera frame); visible: optional set of ids allowed this frame"""
        insts, rows_all, pts_all = [], [], []
        dt = (t - self.prev_t) if self.prev_t is not None else 1.0
        f = self.cam["fx"]
        Rwc, pwc = self.traj.R(t) @ sim.R_IC, self.traj.p(t) + self.traj.R(t) @ sim.T_IC0
        for b in self.boxes:
            if visible is not None and b.id not in visible:
                self.tracked[b.id], self.prev_un[b.id], self.prev_run[b.id] = {}, {}, {}
                continue
            Rwo, Pwo = b.R(t), b.p(t)
            Pw = (Rwo @ self.surf[b.id].T).T + Pwo
            Pc, xl, yl, ul, vl, okl = self._project(Pw, t, sim.T_IC0)
            _, xr, yr, ur, vr, okr = self._project(Pw, t, sim.T_IC1)
            # only faces turned to the camera: the outward normal of a surface point ~ its object-frame direction of largest relative extent
            rel = self.surf[b.id] / (b.dims / 2)
            nrm = np.zeros_like(rel); ax = np.abs(rel).argmax(1); nrm[np.arange(len(rel)), ax] = np.sign(rel[np.arange(len(rel)), ax])
            facing = (((Rwo @ nrm.T).T) * (Pw - pwc)).sum(1) < 0
            okl &= facing
            if okl.sum() < 8:
                self.tracked[b.id], self.prev_un[b.id], self.prev_run[b.id] = {}, {}, {}
                continue
            tr = {k: i for k, i in self.tracked[b.id].items() if okl[k]}
            if len(tr) < self.max_cnt:
                cand = [k for k in np.flatnonzero(okl) if k not in tr]
                self.rng.shuffle(cand)
                for k in cand[: self.max_cnt - len(tr)]:
                    tr[k] = self.next_id; self.next_id += 1
            self.tracked[b.id] = tr
            rows = np.zeros(len(tr), sim.FEAT_DTYPE)
            new_un, new_run = {}, {}
            for r, (k, fid) in zip(rows, sorted(tr.items(), key=lambda kv: kv[1])):
                n = self.rng.normal(0, self.sig, 4)
                x, y = np.float32(xl[k] + n[0] / f), np.float32(yl[k] + n[1] / f)
                vx = vy = 0.0
                if fid in self.prev_un[b.id]:
                    vx, vy = np.float32((x - self.prev_un[b.id][fid][0]) / dt), np.float32((y - self.prev_un[b.id][fid][1]) / dt)
                new_un[fid] = (x, y)
                r["id"], r["track_cnt"] = fid, 1
                r["left"] = [x, y, 1.0, np.float32(ul[k] + n[0]), np.float32(vl[k] + n[1]), vx, vy]
                if okr[k]:
                    x2, y2 = np.float32(xr[k] + n[2] / f), np.float32(yr[k] + n[3] / f)
                    vx2 = vy2 = 0.0
                    if fid in self.prev_run[b.id]:
                        vx2, vy2 = np.float32((x2 - self.prev_run[b.id][fid][0]) / dt), np.float32((y2 - self.prev_run[b.id][fid][1]) / dt)
                    new_run[fid] = (x2, y2)
                    r["has_right"] = 1
                    r["right"] = [x2, y2, 1.0, np.float32(ur[k] + n[2]), np.float32(vr[k] + n[3]), vx2, vy2]
            self.prev_un[b.id], self.prev_run[b.id] = new_un, new_run
            # extra points: a subset of the visible surface in the camera frame, with depth noise
            vis = np.flatnonzero(okl)
            pick = vis[self.rng.permutation(len(vis))[: self.n_extra]]
            ep = Pc[pick] * (1.0 + self.rng.normal(0, 0.004, (len(pick), 1)))
            io = np.zeros(1, INSTOBS_DTYPE)[0]
            io["id"], io["first_feat"], io["n_feats"] = b.id, sum(len(r) for r in rows_all), len(rows)
            io["first_point"], io["n_points"] = sum(len(p) for p in pts_all), len(ep)
            umin, umax, vmin, vmax = ul[okl].min(), ul[okl].max(), vl[okl].min(), vl[okl].max()
            io["rect"] = [umin, vmin, umax - umin, vmax - vmin]
            if self.with_det3d:
                Rco = Rwc.T @ Rwo
                io["has_box3d"] = 1
                bx = io["box3d"]
                bx["class_id"], bx["score"] = b.class_id, 0.9
                bx["center"] = Rwc.T @ (Pwo - pwc)
                bx["dims"] = b.dims
                bx["yaw"] = np.arctan2(-Rco[2, 0], Rco[0, 0])          # R_cioi = [c 0 s; 0 1 0; -s 0 c]
                bx["rect_min"], bx["rect_max"] = [umin, vmin], [umax, vmax]
            insts.append(io); rows_all.append(rows); pts_all.append(ep)
        self.prev_t = t
        if not insts:
            return np.zeros(0, INSTOBS_DTYPE), np.zeros(0, sim.FEAT_DTYPE), np.zeros((0, 3))
        return np.array(insts, INSTOBS_DTYPE), np.concatenate(rows_all), np.ascontiguousarray(np.concatenate(pts_all))
