"""Track + BA pipeline on one GPU (what the reference runs as threads T2 + T3, system/main.cpp:178,394-404):
the front end of frame k+1 is enqueued on the ctx's tracking stream before the back end of frame k is solved on
the BA stream, so the two overlap on the device like the reference's two threads overlap on the CPU."""
import numpy as np

from . import sim
from .backend import Estimator
from .frontend import Context, DV_MEM_DEVICE, DV_MEM_HOST, DV_MODE_RAW, make_cam


class SyntheticSequence:
    """rendered stereo frames resident in HBM + IMU stream for one trajectory (SURVEY 8(d) primary input)"""

    def __init__(self, w, h, cam, n_frames, rate=20.0, t0=1.0, phase=0.0, noise=None, device=None, seed=sim.TEX_SEED):
        import torch
        from .render import RoomRenderer
        self.w, self.h, self.cam, self.dt, self.t0 = w, h, cam, 1.0 / rate, t0 + phase
        self.noise = noise or dict(acc_n=0.02, gyr_n=0.002, acc_w=2e-4, gyr_w=2e-5)
        self.traj = sim.Trajectory()
        rr = RoomRenderer(cam, w, h, device=device, seed=seed)
        self.frames = [rr.stereo(self.traj, self.t0 + k * self.dt) for k in range(n_frames)]
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.times = [self.t0 + k * self.dt for k in range(n_frames)]
        self.imu_t, self.imu_a, self.imu_g = sim.imu_stream(self.traj, self.t0 - 0.05, self.times[-1] + 0.1, 200.0, seed=0xBEEF, **self.noise)

    def host_frame(self, k):
        return self.frames[k][0].cpu().numpy(), self.frames[k][1].cpu().numpy()


class Pipeline:
    def __init__(self, seq: SyntheticSequence, max_cnt=250, min_dist=25, max_iters=10, device=0, use_imu=1, host_frames=False):
        self.seq = seq
        self.host = [seq.host_frame(k) for k in range(len(seq.frames))] if host_frames else None
        c = make_cam(*sim.cam_tuple(seq.cam))
        self.ctx = Context(width=seq.w, height=seq.h, max_cnt=max_cnt, min_dist=min_dist, cam0=c, cam1=c, device=device)
        self.est = Estimator(self.ctx, use_imu=use_imu, stereo=1, max_iters=max_iters, ric=[sim.R_IC, sim.R_IC], tic=[sim.T_IC0, sim.T_IC1], **seq.noise)
        self.k_imu = 0
        self.next = 0
        self.enqueued = False
        self.poses, self.pose_times = [], []

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

    def step(self):
        """processes frame self.next through track + BA; returns the estimator state.
        Order (the reference's T2 / T3 overlap on one host thread): collect tracking of k -> begin BA of k (host prep +
        enqueue on the BA stream) -> enqueue tracking of k+1 and feed k+1's IMU samples while the GPU solves -> end BA of k."""
        k = self.next
        s = self.seq
        if not self.enqueued:
            self._enqueue(k)
        rows = self.ctx.track_stereo_collect()
        self.enqueued = False
        t = s.times[k]
        self._feed_imu(t)
        rc = self.est.ProcessMeasurementsBegin(rows, t)
        if rc != 0:
            raise RuntimeError("IMU stream does not cover the frame")
        if k + 1 < len(s.frames):
            self._enqueue(k + 1)                     # overlaps with the BA of frame k
            self._feed_imu(s.times[k + 1])
        st = self.est.ProcessMeasurementsEnd()
        if st.nonlinear:
            self.poses.append(self.est.window()[10, :7])
            self.pose_times.append(t)
        self.next += 1
        self.rows = rows
        return st

    def ate(self):
        gt = [self.seq.traj.p(t) for t in self.pose_times]
        return sim.align_ate(np.array(self.poses)[:, :3], gt)[0]
