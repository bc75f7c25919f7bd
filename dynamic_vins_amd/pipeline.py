"""Track + BA pipeline on one GPU (what the reference runs as threads T2 + T3, system/main.cpp:178,394-404):
the front end of frame k+1 is enqueued on the ctx's tracking stream before the back end of frame k is solved on
the BA stream, so the two overlap on the device like the reference's two threads overlap on the CPU."""
import numpy as np

from . import sim
from .backend import Estimator
from .frontend import Context, DV_MEM_DEVICE, DV_MEM_HOST, DV_MODE_RAW, make_cam


class SyntheticSequence:
    """rendered stereo frames resident in HBM + IMU stream for one trajectory (SURVEY 8(d) primary input)"""

    def __init__(self, w, h, cam, n_frames, rate=20.0, t0=1.0, phase=0.0, noise=None, device=None, seed=sim.TEX_SEED, baseline=0.12, body_is_camera=False, cam1=None, traj=None):
        import torch
        from .render import RoomRenderer
        self.w, self.h, self.cam, self.dt, self.t0 = w, h, cam, 1.0 / rate, t0 + phase
        self.cam1 = cam1 if cam1 is not None else cam
        self.noise = noise or dict(acc_n=0.02, gyr_n=0.002, acc_w=2e-4, gyr_w=2e-5)
        self.rig = sim.rig(baseline, body_is_camera)
        self.traj = traj if traj is not None else sim.Trajectory()
        rr = RoomRenderer(cam, w, h, device=device, seed=seed, cam1=cam1)
        self.frames = [rr.stereo(self.traj, self.t0 + k * self.dt, self.rig["t_ic1"]) for k in range(n_frames)]
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.times = [self.t0 + k * self.dt for k in range(n_frames)]
        self.imu_t, self.imu_a, self.imu_g = sim.imu_stream(self.traj, self.t0 - 0.05, self.times[-1] + 0.1, 200.0, seed=0xBEEF, **self.noise)

    def host_frame(self, k):
        return self.frames[k][0].cpu().numpy(), self.frames[k][1].cpu().numpy()


class Pipeline:
    def __init__(self, seq: SyntheticSequence, max_cnt=250, min_dist=25, max_iters=10, device=0, use_imu=1, host_frames=False, ba_stride=1, est_kw=None):
        self.seq = seq
        self.host = [seq.host_frame(k) for k in range(len(seq.frames))] if host_frames else None
        c = make_cam(*sim.cam_tuple(seq.cam))
        self.ctx = Context(width=seq.w, height=seq.h, max_cnt=max_cnt, min_dist=min_dist, cam0=c, cam1=make_cam(*sim.cam_tuple(seq.cam1)), device=device)
        self.est_kw = dict(use_imu=use_imu, stereo=1, max_iters=max_iters, ric=seq.rig["est_ric"], tic=seq.rig["est_tic"], **seq.noise)
        self.est_kw.update(est_kw or {})          # keyframe_parallax, g_norm, ... of a shipped YAML (ref_configs.py)
        self.est = Estimator(self.ctx, **self.est_kw)
        self.k_imu = 0
        self.next = 0
        self.enqueued = False
        self._prefetched = None
        self.poses, self.pose_times = [], []
        self.ba_stride = ba_stride          # 2: only every 2nd tracked frame is forwarded to the back end (system/main.cpp:300-307)
        self.last_state = None

    def _enqueue(self, k):
        if self.host is not None:                      # host buffers: the upload rides on the tracking stream
            l, r = self.host[k]
            self.ctx.track_stereo_enqueue(l, r, self.seq.times[k], None, DV_MODE_RAW, DV_MEM_HOST)
        else:
            l, r = self.seq.frames[k]
            self.ctx.track_stereo_enqueue(l.data_ptr(), r.data_ptr(), self.seq.times[k], None, DV_MODE_RAW, DV_MEM_DEVICE)
        self.enqueued = True

    def _feed_imu(self, t):
        s = self.seq
        hi = self.k_imu
        while hi < len(s.imu_t) and s.imu_t[hi] <= t + 0.006:
            hi += 1
        for j in range(self.k_imu, hi):
            self.est.InputIMU(s.imu_t[j], s.imu_a[j], s.imu_g[j])
        self.k_imu = hi

    def step(self, defer_end=False):
        """processes frame self.next through track + BA; returns the estimator state.
        Order (the reference's T2 / T3 overlap on one host thread): collect tracking of k -> begin BA of k (host prep +
        enqueue on the BA stream) -> enqueue tracking of k+1 and feed k+1's IMU samples while the GPU solves -> end BA of k."""
        k = self.next
        s = self.seq
        rows, self._prefetched = self._prefetched, None
        if rows is None:
            if not self.enqueued:
                self._enqueue(k)
            rows = self.ctx.track_stereo_collect()
        self.enqueued = False
        t = s.times[k]
        if self.ba_stride > 1 and (k % self.ba_stride) != 0:      # tracked only: the reference pushes frames 0, 2, 4, ... (cnt % 2 == 0, cnt from 0: system/main.cpp:181,300-312)
            if k + 1 < len(s.frames):
                self._enqueue(k + 1)
            self.next += 1
            self.rows = rows
            return self.last_state if self.last_state is not None else self.est.state
        self._feed_imu(t)
        rc = self.est.ProcessMeasurementsBegin(rows, t)
        if rc != 0:
            raise RuntimeError("IMU stream does not cover the frame")
        if k + 1 < len(s.frames):
            self._enqueue(k + 1)                     # overlaps with the BA of frame k
            self._feed_imu(s.times[k + 1])
            if not defer_end:
                # the tracker finishes frame k+1 long before the BA of frame k does: its rows are collected NOW, not between the end of this frame's BA and the
                # begin of the next (where the collect sat on the critical path of the BA stream: ~11 us per frame)
                self._prefetched = self.ctx.track_stereo_collect()
        self.rows = rows
        if defer_end:
            self._pending_t = t
            return None
        return self._finish(t)

    def _finish(self, t):
        st = self.est.ProcessMeasurementsEnd()
        if getattr(self, "keep_frame_rows", False):          # every frame handed to the back end, as the runner's dv_runner_get_frames keeps them (what SaveBodyTrajectory writes)
            self.frame_rows.append((t, self.est.window()[10, :7].copy(), int(st.nonlinear)))
        if st.nonlinear:
            self.poses.append(self.est.window()[10, :7])
            self.pose_times.append(t)
        self.next += 1
        self.last_state = st
        return st

    def step_begin(self):
        """first half of step(): everything up to and including the enqueue of frame k's BA and of frame k+1's tracking.  With step_end() it lets ONE host
        thread interleave several independent sequences on one GPU (each Pipeline owns its streams): while the BA of sequence A runs, the host prepares B."""
        self._pending_t = None
        st = self.step(defer_end=True)
        self._early = st                              # tracked-only frame (ba_stride): already complete
        return st

    def step_end(self):
        if self._pending_t is None:
            return self._early
        t, self._pending_t = self._pending_t, None
        return self._finish(t)

    def ate(self):
        """ATE RMSE against the synthetic ground truth, computed the way the reference's scripts do: associate.py's nearest-stamp pairing (0.02 s) of the two
        stamped trajectories, then evaluate_ate.py's alignment (io_formats.evaluate_ate)"""
        from . import io_formats
        gt = [self.seq.traj.p(t) for t in self.pose_times]
        return io_formats.evaluate_ate(self.pose_times, gt, self.pose_times, np.array(self.poses)[:, :3])[0]


class DynamicSequence(SyntheticSequence):
    """SyntheticSequence with moving boxes (dynsim.default_boxes): per frame the stereo pair (resident in HBM), and — as host arrays — what the perception
    front end of the reference delivers: the inverse merged instance mask, one detection per visible object (track id, rectangle, ROI mask, 3-D box,
    extra 3-D points sampled from the depth map like InstFeat::DetectExtraPoints, front_end/instance_feature.cpp:413-461) (dynamic variant)."""

    def __init__(self, w, h, cam, n_frames, rate=20.0, t0=1.0, noise=None, device=None, boxes=None, min_pixels=400, seed=sim.TEX_SEED, baseline=0.12, body_is_camera=False, cam1=None, traj=None):
        import torch
        from . import dynsim
        from .render import DynRoomRenderer
        self.w, self.h, self.cam, self.dt, self.t0 = w, h, cam, 1.0 / rate, t0
        self.cam1 = cam1 if cam1 is not None else cam
        self.noise = noise or dict(acc_n=0.02, gyr_n=0.002, acc_w=2e-4, gyr_w=2e-5)
        self.rig = sim.rig(baseline, body_is_camera)
        self.traj = traj if traj is not None else sim.Trajectory()
        if boxes == "escort" or (isinstance(boxes, tuple) and boxes[0] == "escort"):      # boxes travelling with the camera: objects in every frame
            boxes = dynsim.escort_boxes(self.traj, boxes[1] if isinstance(boxes, tuple) else 4)
        self.boxes = boxes if boxes is not None else dynsim.default_boxes()
        rr = DynRoomRenderer(cam, w, h, device=device, seed=seed, cam1=cam1)
        self.times = [self.t0 + k * self.dt for k in range(n_frames)]
        self.frames, self.inv_mask, self.inv_mask_dev, self.dets, self.boxes3d = [], [], [], [], []
        self.disp_dev, self.baseline = [], float(baseline)      # SemanticImage::disp per frame (CV_32F, resident): what the stereo network of the reference delivers
        rays = rr.rays.cpu().numpy().reshape(h, w, 3)
        for t in self.times:
            left, right, ident, depth = rr.stereo_dynamic(self.traj, t, self.boxes, self.rig["t_ic1"])
            self.frames.append((left, right))
            idm, dep = ident.cpu().numpy(), depth.cpu().numpy()
            self.inv_mask.append(np.ascontiguousarray(np.where(idm == 0, 255, 0).astype(np.uint8)))
            self.inv_mask_dev.append(torch.where(ident == 0, 255, 0).to(torch.uint8).contiguous())      # resident next to the frames
            # disparity of the left image: fx0 * baseline / depth in float (0 where the ray hits nothing), the map InstFeat::DetectExtraPoints samples
            fxb = torch.tensor(np.float32(np.float32(cam["fx"]) * np.float32(baseline)), dtype=torch.float32, device=depth.device)
            d32 = depth.to(torch.float32)
            self.disp_dev.append(torch.where(torch.isfinite(d32) & (d32 > 0), fxb / d32, torch.zeros_like(d32)).contiguous())
            dets, b3 = [], np.zeros(0, dynsim.BOX3D_DTYPE)
            Rwc, pwc = self.traj.R(t) @ sim.R_IC, self.traj.p(t) + self.traj.R(t) @ sim.T_IC0
            for b in self.boxes:
                ys, xs = np.nonzero(idm == b.id)
                if len(ys) < min_pixels:
                    continue
                x0, x1, y0, y1 = int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max())      # VIODE::SetViodeMaskAndRoi: rect = (min_pt, max_pt), i.e. max EXCLUSIVE (cv::Rect(pt1, pt2))
                rw, rh = x1 - x0, y1 - y0
                if rw < 24 or rh < 24:
                    continue
                mask = np.ascontiguousarray(np.where(idm[y0:y0 + rh, x0:x0 + rw] == b.id, 255, 0).astype(np.uint8))
                step = int(max(np.sqrt(0.8 * rh * rw / 1000.0), 2.0))                              # DetectExtraPoints: N_max 1000, step >= 2
                ii, jj = np.mgrid[0:rh:step, 0:rw:step]
                sel = mask[ii, jj] > 0
                r, c = ii[sel] + y0, jj[sel] + x0
                z = dep[r, c]
                ok = (z > 0.1) & (z <= 100)
                pts = rays[r[ok], c[ok]] * z[ok, None]
                Rco = Rwc.T @ b.R(t)
                bx = np.zeros(1, dynsim.BOX3D_DTYPE)
                bx["class_id"], bx["score"], bx["center"], bx["dims"] = b.class_id, 0.9, Rwc.T @ (b.p(t) - pwc), b.dims
                bx["yaw"] = np.arctan2(-Rco[2, 0], Rco[0, 0])
                bx["rect_min"], bx["rect_max"] = [x0, y0], [x1, y1]
                b3 = np.concatenate([b3, bx])
                dets.append(dict(track_id=b.id, class_id=b.class_id, rect=(x0, y0, rw, rh), mask=mask, points=np.ascontiguousarray(pts)))
            self.dets.append(dets); self.boxes3d.append(b3)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.imu_t, self.imu_a, self.imu_g = sim.imu_stream(self.traj, self.t0 - 0.05, self.times[-1] + 0.1, 200.0, seed=0xBEEF, **self.noise)


    def disp_host(self, k):
        """the frame's disparity map as a host float32 array (for the CPU oracle)"""
        return self.disp_dev[k].cpu().numpy()


class DynamicPipeline(Pipeline):
    """Pipeline in dynamic mode (cfg::slam == kDynamic): TrackSemanticImage for the background + InstsFeatManager::InstsTrack for the objects on the tracking
    stream, Estimator::ProcessImage with the object branch (window solve on the BA stream, object solve on a third one).  `segments` (a sim.SegmentSim):
    use_line — the detector's matched segments of every frame go through FrameLines::UndistortedLineEndPoints (dv_undistort_lines) into frame.features.lines,
    as TrackSemanticImage's line thread delivers them (background_tracker.cpp:774-780, 809-817)."""

    def __init__(self, seq: DynamicSequence, max_cnt=250, min_dist=25, max_iters=10, device=0, use_imu=1, max_dynamic_cnt=50, min_dynamic_dist=5, use_det3d=1,
                 static_inst_threshold=1.0, mask_morphology_size=0, segments=None, est_kw=None, extra_from_disparity=True, ba_stride=1,
                 static_as_background=False):
        from .frontend import DV_MODE_SEMANTIC
        self.extra_from_disparity = extra_from_disparity      # False: the detections' own `points` are handed through (the caller ran the extra-point pipeline)
        self.seq, self.host = seq, None
        c = make_cam(*sim.cam_tuple(seq.cam))
        self.cam_c, self.cam1_c = c, make_cam(*sim.cam_tuple(seq.cam1))
        self.ctx = Context(width=seq.w, height=seq.h, max_cnt=max_cnt, min_dist=min_dist, cam0=c, cam1=self.cam1_c, device=device, mask_morphology_size=mask_morphology_size)
        self.ctx.inst_config(max_dynamic_cnt, min_dynamic_dist, use_det3d)
        self.est_kw = dict(use_imu=use_imu, stereo=1, max_iters=max_iters, ric=seq.rig["est_ric"], tic=seq.rig["est_tic"], dynamic=1, use_det3d=use_det3d,
                           static_inst_threshold=static_inst_threshold, use_line=int(segments is not None), **seq.noise)
        self.est_kw.update(est_kw or {})
        self.est = Estimator(self.ctx, **self.est_kw)
        self.mode, self.use_det3d, self.segments = DV_MODE_SEMANTIC, use_det3d, segments
        self.k_imu = self.next = 0
        self.enqueued = False
        self._prefetched = None
        self.ba_stride = ba_stride          # 2: every tracked frame goes through both trackers, every 2nd one to the back end (system/main.cpp:300-307: every data set but KITTI)
        self.last_state = None
        self.poses, self.pose_times = [], []
        # para::is_static_inst_as_background (reference default: true): before tracking frame f the pixels of the instances the estimator reported static leave the merged mask
        # (system/main.cpp:194,217-245).  The reference reads that report across threads without an order; here: the snapshot of the newest back-end frame <= f - 2 (runner.hip)
        self.static_as_background, self.static_snaps = static_as_background, []
        self.frame_rows, self.keep_frame_rows = [], False      # (t, [px py pz qx qy qz qw], nonlinear) of every frame handed to the back end when keep_frame_rows is set
        self.stat = dict(frames=0, frames_with_objects=0, object_detections=0, object_features=0, min_detections=10 ** 9)      # what the object branch was fed over the run

    def static_ids_for(self, k):
        from ._abi import DV_STATIC_REPORT_LAG
        best = [s for s in self.static_snaps if s[0] <= k - DV_STATIC_REPORT_LAG]
        return best[-1][1] if best else np.zeros(0, np.uint32)

    def _enqueue(self, k):
        l, r = self.seq.frames[k]
        if self.static_as_background and len(self.seq.dets[k]):
            self.ctx.track_unmask_static(self.seq.dets[k], self.static_ids_for(k))
        self.ctx.track_stereo_enqueue(l.data_ptr(), r.data_ptr(), self.seq.times[k], self.seq.inv_mask_dev[k].data_ptr(), self.mode, DV_MEM_DEVICE)
        if self.extra_from_disparity:      # the extra points of the objects: DetectExtraPoints + ProcessExtraPoints on the device from the frame's disparity map
            self.ctx.inst_set_disparity(self.seq.disp_dev[k].data_ptr(), self.seq.baseline, DV_MEM_DEVICE)
        if getattr(self.seq, "right_keys", None) is not None:      # VIODE: seg1's key image -> TrackRightByPad's segmentation-key test
            self.ctx.inst_set_right_keys(self.seq.right_keys[k])
        self.ctx.inst_track_enqueue(self.seq.times[k], self.seq.dets[k], self.seq.boxes3d[k] if self.use_det3d else None)
        self.enqueued = True

    def _collect(self):
        return (self.ctx.track_stereo_collect(),) + tuple(self.ctx.inst_track_collect())

    def line_rows(self, t):
        """frame.features.lines of the frame at time t: the detector's pixel segments through UndistortedLineEndPoints (cam0 / cam1)"""
        il, sl, ir, sr = self.segments.frame(t)
        self.seg_px = (il, sl, ir, sr)
        return sim.line_rows(il, self.ctx.undistort_lines(self.cam_c, sl), ir, self.ctx.undistort_lines(self.cam1_c, sr))

    def step(self, defer_end=False):
        k, s = self.next, self.seq
        pre, self._prefetched = self._prefetched, None
        if pre is None:
            if not self.enqueued:
                self._enqueue(k)
            pre = self._collect()
        rows, insts, ifeats, pts = pre
        self.enqueued = False
        t = s.times[k]
        if self.ba_stride > 1 and (k % self.ba_stride) != 0:      # tracked only (both trackers keep their state up to date; system/main.cpp:300-312)
            if k + 1 < len(s.frames):
                self._enqueue(k + 1)
                self._prefetched = self._collect()
            self.next += 1
            self.rows, self.insts, self.ifeats, self.ipts = rows, insts, ifeats, pts
            return self.last_state if self.last_state is not None else self.est.state
        self._feed_imu(t)
        if self.segments is not None:
            self.lrows = self.line_rows(t)
            self.est.SetLines(self.lrows)
        # Three-phase back end (dv_est_process_dynamic_begin_ego / _attach): the window solve of frame k goes to the GPU first, with the background rows alone (~0.17 ms
        # of host work in front of it); then the tracking of frame k+1 (background + objects: ~60 launches, 0.25 ms of host time) is enqueued — thread T2 of the
        # reference, independent of T3 —; then the object branch of ProcessImage (~0.4 ms of host work + the object solve on the third stream).  All of it runs
        # beside the window solve (the chain of ~30 dependent launches that IS the frame) instead of in front of it.
        if self.est.ProcessMeasurementsDynamicBeginEgo(rows, t) != 0:
            raise RuntimeError("IMU stream does not cover the frame")
        if k + 1 < len(s.frames):
            self._enqueue(k + 1)
        self.est.AttachInstances(insts, ifeats, pts)
        if self.static_as_background:
            self.static_snaps = (self.static_snaps + [(k, self.est.static_instances())])[-4:]
        if k + 1 < len(s.frames):
            self._feed_imu(s.times[k + 1])
            if not defer_end:
                self._prefetched = self._collect()      # while the BA of frame k runs (see Pipeline.step)
        self.rows, self.insts, self.ifeats, self.ipts = rows, insts, ifeats, pts
        self.stat["frames"] += 1; self.stat["frames_with_objects"] += int(len(insts) > 0); self.stat["object_detections"] += len(insts); self.stat["object_features"] += len(ifeats)
        self.stat["min_detections"] = min(self.stat["min_detections"], len(insts))
        if defer_end:
            self._pending_t = t
            return None
        return self._finish(t)
