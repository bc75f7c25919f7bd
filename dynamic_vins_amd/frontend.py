"""Python mirror of the reference's front-end interface, on top of the C ABI (for tests and bench.py;
the C++ drop-in shim with the reference's class signatures is dynamic_vins_amd/host/feature_tracker.h).

Names follow dynamic_vins/src/front_end/background_tracker.h:41-88 and feature_utils.h.
"""
import ctypes as C
import numpy as np

from . import _abi
from ._abi import DvinsError, dv_cam, dv_config, dv_feat, DV_MEM_HOST, DV_MEM_DEVICE, DV_MEM_PINNED, DV_FMT_BGR, DV_MODE_RAW, DV_MODE_NAIVE, DV_MODE_SEMANTIC

class dv_inst_det(C.Structure):
    _fields_ = [("track_id", C.c_uint32), ("class_id", C.c_int32), ("x", C.c_int32), ("y", C.c_int32), ("w", C.c_int32), ("h", C.c_int32),
                ("mask", C.c_void_p), ("points", C.c_void_p), ("n_points", C.c_int32), ("pad_", C.c_int32)]


FEAT_DTYPE = np.dtype([("id", np.uint32), ("track_cnt", np.int32), ("has_right", np.int32), ("pad_", np.int32),
                       ("left", np.float64, 7), ("right", np.float64, 7)])
assert FEAT_DTYPE.itemsize == C.sizeof(dv_feat) == 128


def make_cam(fx, fy, cx, cy, k1=0.0, k2=0.0, p1=0.0, p2=0.0):
    return dv_cam(fx, fy, cx, cy, k1, k2, p1, p2)


def _ptr(a):
    """pointer of a host ndarray, or pass through a raw device pointer (int)."""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return C.c_void_p(int(a))
    return C.c_void_p(a.ctypes.data)


class Context:
    """Owns a dv_ctx (device memory + stream).  One per thread, like the reference's FeatureTracker."""

    def __init__(self, width, height, max_cnt=150, min_dist=30, flow_back=1, stereo=1, cam0=None, cam1=None, device=0, mask_morphology_size=0):
        self.lib = _abi.load()
        cfg = dv_config()
        cfg.width, cfg.height, cfg.max_cnt, cfg.min_dist = width, height, max_cnt, min_dist
        cfg.flow_back, cfg.stereo, cfg.device = flow_back, stereo, device
        cfg.mask_morphology_size = mask_morphology_size
        cfg.cam0 = cam0 if cam0 is not None else make_cam(1, 1, 0, 0)
        cfg.cam1 = cam1 if cam1 is not None else cfg.cam0
        self.cfg = cfg
        self.h = self.lib.dv_create(C.byref(cfg))
        if not self.h:
            raise DvinsError(self.lib.dv_last_error(None).decode())
        self._out = np.zeros(_abi.DV_MAX_FEATS, FEAT_DTYPE)

    def close(self):
        if getattr(self, "h", None):
            self.lib.dv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise DvinsError(self.lib.dv_last_error(self.h).decode())

    # ---- whole-frame entries ----
    def track_stereo(self, gray0, gray1, t, mask=None, mode=DV_MODE_RAW, mem=DV_MEM_HOST, w=None, h=None, stride=None):
        w = w or self.cfg.width
        h = h or self.cfg.height
        stride = stride or w
        n = C.c_int(0)
        self._check(self.lib.dv_track_stereo(self.h, _ptr(gray0), _ptr(gray1), w, h, stride, float(t), _ptr(mask), mode, mem,
                                             self._out.ctypes.data_as(C.POINTER(dv_feat)), C.byref(n)))
        return self._out[: n.value].copy()

    def track_stereo_enqueue(self, gray0, gray1, t, mask=None, mode=DV_MODE_RAW, mem=DV_MEM_HOST, stride=None):
        w, h = self.cfg.width, self.cfg.height
        self._check(self.lib.dv_track_stereo_enqueue(self.h, _ptr(gray0), _ptr(gray1), w, h, stride or w, float(t), _ptr(mask), mode, mem))

    # ---- dynamic mode: the per-object tracker (InstsFeatManager) ----
    def inst_config(self, max_dynamic_cnt=50, min_dynamic_dist=5, use_det3d=0):
        self._check(self.lib.dv_inst_config(self.h, int(max_dynamic_cnt), int(min_dynamic_dist), int(use_det3d)))
        from .dynsim import INSTOBS_DTYPE
        self._inst_out = (np.zeros(64, INSTOBS_DTYPE), np.zeros(64 * 256, FEAT_DTYPE), np.zeros((1 << 16, 3), np.float64))

    def inst_track_enqueue(self, t, dets, boxes3d=None):
        """dets: list of dict(track_id, class_id, rect=(x, y, w, h), mask=uint8[h, w], points=float64[n, 3] or None); boxes3d: BOX3D_DTYPE array or None.
        Call right after track_stereo_enqueue of the same frame."""
        from .dynsim import BOX3D_DTYPE
        arr = (dv_inst_det * max(len(dets), 1))()
        self._inst_keep = []
        for k, d in enumerate(dets):
            m = np.ascontiguousarray(d["mask"], np.uint8)
            x, y, w, h = [int(v) for v in d["rect"]]
            assert m.shape == (h, w)
            pts = None if d.get("points") is None else np.ascontiguousarray(d["points"], np.float64)
            self._inst_keep += [m, pts]
            arr[k].track_id, arr[k].class_id, arr[k].x, arr[k].y, arr[k].w, arr[k].h = int(d["track_id"]), int(d.get("class_id", 0)), x, y, w, h
            arr[k].mask = m.ctypes.data
            arr[k].points = pts.ctypes.data if pts is not None and len(pts) else None
            arr[k].n_points = 0 if pts is None else len(pts)
        b3 = np.ascontiguousarray(boxes3d, BOX3D_DTYPE) if boxes3d is not None else np.zeros(0, BOX3D_DTYPE)
        self._inst_keep.append(b3)
        self._check(self.lib.dv_inst_track_enqueue(self.h, float(t), C.addressof(arr) if len(dets) else None, len(dets), b3.ctypes.data if len(b3) else None, len(b3)))

    def inst_set_disparity(self, disp, baseline, mem=DV_MEM_HOST, stride_bytes=0):
        """SemanticImage::disp of the frame the next inst_track_enqueue processes: float32 [h, w] numpy array (host) or a device pointer (int, mem=DV_MEM_DEVICE, kept alive
        by the caller until inst_track_collect).  The extra points of every visible object are then computed on the device (DetectExtraPoints + ProcessExtraPoints)."""
        if disp is None:
            self._check(self.lib.dv_inst_set_disparity(self.h, None, 0, 0, 0.0))
            return
        if isinstance(disp, np.ndarray):
            self._disp_keep = np.ascontiguousarray(disp, np.float32)
            ptr, mem, stride_bytes = self._disp_keep.ctypes.data, DV_MEM_HOST, self._disp_keep.strides[0]
        else:
            ptr = int(disp)
        self._check(self.lib.dv_inst_set_disparity(self.h, ptr, int(stride_bytes), int(mem), float(baseline)))

    def track_unmask_static(self, dets, static_ids):
        """FeatureTrack, system/main.cpp:217-245 (para::is_static_inst_as_background): for the NEXT track_stereo_enqueue (which must carry a mask) the pixels of the frame's
        detections whose track_id the estimator reported static leave the merged instance mask.  dets: the frame's detection dicts (rect, mask)."""
        ids = np.ascontiguousarray(static_ids, np.uint32)
        if not len(ids) or not len(dets):
            self._check(self.lib.dv_track_unmask_static(self.h, None, 0, None, 0))
            return
        arr = (dv_inst_det * len(dets))()
        keep = []
        for k, d in enumerate(dets):
            m = np.ascontiguousarray(d["mask"], np.uint8); keep.append(m)
            x, y, w, h = [int(v) for v in d["rect"]]
            arr[k].track_id, arr[k].x, arr[k].y, arr[k].w, arr[k].h, arr[k].mask = int(d["track_id"]), x, y, w, h, m.ctypes.data
        self._check(self.lib.dv_track_unmask_static(self.h, C.addressof(arr), len(dets), ids.ctypes.data, len(ids)))

    def inst_set_right_keys(self, key_img, mem=DV_MEM_HOST):
        """VIODE: the key image of seg1 (viode_mask(...)[2]: uint32 [h, w] numpy array, or a device pointer with mem=DV_MEM_DEVICE) of the frame the next inst_track_enqueue
        processes: TrackRightByPad's segmentation-key test (front_end/instance_feature.cpp:263-268)"""
        if key_img is None:
            self._check(self.lib.dv_inst_set_right_keys(self.h, None, 0, 0))
            return
        if isinstance(key_img, np.ndarray):
            self._keys_keep = np.ascontiguousarray(key_img, np.uint32)
            self._check(self.lib.dv_inst_set_right_keys(self.h, self._keys_keep.ctypes.data, self._keys_keep.strides[0], DV_MEM_HOST))
        else:
            self._check(self.lib.dv_inst_set_right_keys(self.h, int(key_img), 0, int(mem)))

    def extra_points(self, mask, box_xy, disp, baseline, stage=0):
        """operator form: one object through InstFeat::DetectExtraPoints (stage 1) or the whole extra-point pipeline (stage 0) -> float64 [n, 3]"""
        mask = np.ascontiguousarray(mask, np.uint8); disp = np.ascontiguousarray(disp, np.float32)
        h, w = mask.shape
        out = np.zeros((4096, 3), np.float64)
        n = C.c_int(0)
        self._check(self.lib.dv_extra_points(self.h, mask.ctypes.data, int(box_xy[0]), int(box_xy[1]), w, h, disp.ctypes.data, disp.strides[0], DV_MEM_HOST, float(baseline), int(stage),
                                             out.ctypes.data, len(out), C.byref(n)))
        return out[: n.value].copy()

    def inst_track_collect(self):
        """-> (insts [INSTOBS_DTYPE], feats [FEAT_DTYPE], points [n, 3]) laid out as Estimator.ProcessMeasurementsDynamic takes them"""
        oi, of, op = self._inst_out
        ni, nf, npt = C.c_int(0), C.c_int(0), C.c_int(0)
        self._check(self.lib.dv_inst_track_collect(self.h, oi.ctypes.data, len(oi), C.byref(ni), of.ctypes.data, len(of), C.byref(nf), op.ctypes.data, len(op), C.byref(npt)))
        return oi[: ni.value].copy(), of[: nf.value].copy(), op[: npt.value].copy()

    def track_stereo_collect(self):
        n = C.c_int(0)
        self._check(self.lib.dv_track_stereo_collect(self.h, self._out.ctypes.data_as(C.POINTER(dv_feat)), C.byref(n)))
        return self._out[: n.value].copy()

    def reset(self):
        self._check(self.lib.dv_reset(self.h))

    def sync(self):
        self._check(self.lib.dv_sync(self.h))

    # ---- operator-level entries (host numpy in / out) ----
    def lk(self, img_a, img_b, pts_a, max_level=3, iters=30, eps=0.01, initial=None):
        h, w = img_a.shape
        pts_a = np.ascontiguousarray(pts_a, np.float32)
        n = len(pts_a)
        pts_b = np.ascontiguousarray(initial, np.float32).copy() if initial is not None else np.zeros((n, 2), np.float32)
        st = np.zeros(n, np.uint8)
        self._check(self.lib.dv_lk(self.h, _ptr(img_a), _ptr(img_b), w, h, img_a.strides[0], _ptr(pts_a), n, max_level, iters, float(eps),
                                   1 if initial is not None else 0, _ptr(pts_b), _ptr(st), DV_MEM_HOST))
        return pts_b, st

    def track_by_lk(self, img1, img2, pts1, flow_back=True, dist_thresh=0.5):
        """FeatureTrackByLK (front_end/feature_utils.cpp:35-69)."""
        h, w = img1.shape
        pts1 = np.ascontiguousarray(pts1, np.float32)
        n = len(pts1)
        pts2 = np.zeros((n, 2), np.float32)
        st = np.zeros(n, np.uint8)
        self._check(self.lib.dv_track_by_lk(self.h, _ptr(img1), _ptr(img2), w, h, img1.strides[0], _ptr(pts1), n, int(flow_back),
                                            float(dist_thresh), _ptr(pts2), _ptr(st), DV_MEM_HOST))
        return pts2, st

    def lk_cuda(self, img_a, img_b, pts_a, max_level=3, iters=30, initial=None):
        """cv::cuda::SparsePyrLKOpticalFlow(Size(21, 21), max_level, iters, useInitialFlow)->calc"""
        h, w = img_a.shape
        pts_a = np.ascontiguousarray(pts_a, np.float32); n = len(pts_a)
        pts_b = np.ascontiguousarray(initial, np.float32).copy() if initial is not None else np.zeros((n, 2), np.float32)
        st = np.zeros(n, np.uint8)
        self._check(self.lib.dv_lk_cuda(self.h, _ptr(img_a), _ptr(img_b), w, h, img_a.strides[0], _ptr(pts_a), n, int(max_level), int(iters), 1 if initial is not None else 0,
                                        _ptr(pts_b), _ptr(st), DV_MEM_HOST))
        return pts_b, st

    def track_by_lk_gpu(self, img1, img2, pts1, flow_back=True):
        """FeatureTrackByLKGpu (front_end/feature_utils.cpp:83-163)"""
        h, w = img1.shape
        pts1 = np.ascontiguousarray(pts1, np.float32); n = len(pts1)
        pts2 = np.zeros((n, 2), np.float32); st = np.zeros(n, np.uint8)
        self._check(self.lib.dv_track_by_lk_gpu(self.h, _ptr(img1), _ptr(img2), w, h, img1.strides[0], _ptr(pts1), n, int(flow_back), _ptr(pts2), _ptr(st), DV_MEM_HOST))
        return pts2, st

    def pyr_down_cuda(self, img):
        h, w = img.shape
        dst = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
        self._check(self.lib.dv_pyr_down_cuda(self.h, _ptr(img), w, h, img.strides[0], _ptr(dst), DV_MEM_HOST))
        return dst

    def gftt(self, img, max_n, quality, min_dist, mask=None, rule="cpu"):
        """rule "cpu": cv::goodFeaturesToTrack (dv_gftt); "cuda": cv::cuda::GoodFeaturesToTrackDetector, TrackImageNaive's detector (dv_gftt_cuda)"""
        h, w = img.shape
        out = np.zeros((_abi.DV_MAX_FEATS, 2), np.float32)
        n = C.c_int(0)
        self._check((self.lib.dv_gftt_cuda if rule == "cuda" else self.lib.dv_gftt)(self.h, _ptr(img), _ptr(mask), w, h, img.strides[0], int(max_n), float(quality), float(min_dist),
                                     _ptr(out), C.byref(n), DV_MEM_HOST))
        return out[: n.value].copy()

    def min_eigen(self, img, rule="cpu"):
        h, w = img.shape
        eig = np.zeros((h, w), np.float32)
        self._check((self.lib.dv_min_eigen_cuda if rule == "cuda" else self.lib.dv_min_eigen)(self.h, _ptr(img), w, h, img.strides[0], _ptr(eig), DV_MEM_HOST))
        return eig

    def viode_mask(self, seg_bgr, dyn_keys, want_keys=True):
        """VIODE::SetViodeMaskSimple / BuildViodeMask -> (merge_mask, inv_merge_mask, key_image or None, boxes[nkeys, 4])"""
        seg = np.ascontiguousarray(seg_bgr)
        h, w, _ = seg.shape
        keys = np.ascontiguousarray(dyn_keys, np.uint32)
        merge, inv = np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8)
        kimg = np.zeros((h, w), np.uint32) if want_keys else None
        boxes = np.zeros((len(keys), 4), np.int32)
        self._check(self.lib.dv_viode_mask(self.h, _ptr(seg), w, h, seg.strides[0], _ptr(keys), len(keys), _ptr(merge), _ptr(inv), _ptr(kimg), _ptr(boxes)))
        return merge, inv, kimg, boxes

    def bgr2gray(self, bgr):
        """cv::cvtColor(BGR2GRAY) of an (h, w, 3) uint8 image"""
        bgr = np.ascontiguousarray(bgr)
        h, w, _ = bgr.shape
        out = np.zeros((h, w), np.uint8)
        self._check(self.lib.dv_bgr2gray(self.h, _ptr(bgr), w, h, bgr.strides[0], _ptr(out), DV_MEM_HOST))
        return out

    def remap(self, src, map1, map2):
        """cv::remap(src, map1 (h, w, 2) int16, map2 (h, w) uint16, INTER_LINEAR) of an (h, w) or (h, w, 3) uint8 image"""
        src = np.ascontiguousarray(src)
        h, w = src.shape[:2]
        cn = 1 if src.ndim == 2 else src.shape[2]
        m1, m2 = np.ascontiguousarray(map1, np.int16), np.ascontiguousarray(map2, np.uint16)
        assert m1.shape == (h, w, 2) and m2.shape == (h, w)
        out = np.zeros_like(src)
        self._check(self.lib.dv_remap(self.h, _ptr(src), w, h, src.strides[0], cn, m1.ctypes.data, m2.ctypes.data, _ptr(out), DV_MEM_HOST))
        return out

    def set_undistort_maps(self, cam, map1=None, map2=None):
        """cfg::is_undistort_input: install (or, with map1=None, remove) the fixed-point undistortion maps of camera 0 / 1"""
        if map1 is None:
            self._check(self.lib.dv_set_undistort_maps(self.h, int(cam), None, None, 0, 0))
            return
        m1, m2 = np.ascontiguousarray(map1, np.int16), np.ascontiguousarray(map2, np.uint16)
        h, w = m2.shape
        assert m1.shape == (h, w, 2)
        self._check(self.lib.dv_set_undistort_maps(self.h, int(cam), m1.ctypes.data, m2.ctypes.data, w, h))

    def pyr_down(self, img):
        h, w = img.shape
        dst = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
        self._check(self.lib.dv_pyr_down(self.h, _ptr(img), w, h, img.strides[0], _ptr(dst), DV_MEM_HOST))
        return dst

    def circle_mask(self, mask, pts, radius):
        h, w = mask.shape
        pts = np.ascontiguousarray(pts, np.float32)
        out = np.ascontiguousarray(mask).copy()
        self._check(self.lib.dv_circle_mask(self.h, _ptr(out), w, h, out.strides[0], _ptr(pts), len(pts), int(radius), DV_MEM_HOST))
        return out

    def erode(self, mask, k):
        h, w = mask.shape
        out = np.zeros_like(mask)
        self._check(self.lib.dv_erode(self.h, _ptr(mask), w, h, mask.strides[0], int(k), _ptr(out), DV_MEM_HOST))
        return out

    def lift_projective(self, cam, pts, offset=(0.0, 0.0)):
        pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
        out = np.zeros_like(pts)
        if len(pts):
            self._check(self.lib.dv_lift_projective_offset(self.h, C.byref(cam), _ptr(pts), len(pts), float(offset[0]), float(offset[1]), _ptr(out), DV_MEM_HOST))
        return out

    # ---- measurement ----
    def undistort_lines(self, cam, segs_px):
        """FrameLines::UndistortedLineEndPoints: [n, 4] pixel end points (x1 y1 x2 y2, float32) -> [n, 4] normalised undistorted, float64"""
        segs = np.ascontiguousarray(segs_px, np.float32).reshape(-1, 4)
        out = np.zeros((len(segs), 4), np.float64)
        if len(segs):
            self._check(self.lib.dv_undistort_lines(self.h, C.byref(cam), segs.ctypes.data, len(segs), out.ctypes.data))
        return out

    def timing_enable(self, on=1):
        self._check(self.lib.dv_timing_enable(self.h, int(on)))

    def timing_reset(self):
        self._check(self.lib.dv_timing_reset(self.h))

    def timing_get(self, name):
        ms, cnt = C.c_double(0), C.c_longlong(0)
        self._check(self.lib.dv_timing_get(self.h, name.encode(), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value


class FeatureTracker:
    """Mirror of dynamic_vins::FeatureTracker (front_end/background_tracker.h:41-88).

    TrackImage / TrackImageNaive return the FeatureBackground.points map:
        {id: [(0, [x,y,1,u,v,vx,vy]), (1, [...right...])]}
    camelCase aliases (VINS-Fusion spelling used by BASELINE.json) are provided.
    """

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.cur_time = 0.0

    @staticmethod
    def _to_map(rows):
        pts = {}
        for r in rows:
            obs = [(0, np.array(r["left"]))]
            if r["has_right"]:
                obs.append((1, np.array(r["right"])))
            pts[int(r["id"])] = obs
        return pts

    def TrackImage(self, gray0, gray1, time0):
        self.cur_time = time0
        self.rows = self.ctx.track_stereo(gray0, gray1, time0, None, DV_MODE_RAW)
        return self._to_map(self.rows)

    def TrackImageNaive(self, gray0, gray1, time0, inv_merge_mask=None):
        self.cur_time = time0
        self.rows = self.ctx.track_stereo(gray0, gray1, time0, inv_merge_mask, DV_MODE_NAIVE)
        return self._to_map(self.rows)

    def TrackSemanticImage(self, gray0, gray1, time0, inv_merge_mask=None):
        """background half of dynamic mode (background_tracker.cpp:757-837)"""
        self.cur_time = time0
        self.rows = self.ctx.track_stereo(gray0, gray1, time0, inv_merge_mask, DV_MODE_SEMANTIC)
        return self._to_map(self.rows)

    trackImage = TrackImage
    trackImageNaive = TrackImageNaive


class InstFeat:
    """Mirror of dynamic_vins::InstFeat for ONE object instance as InstsFeatManager::InstsTrack drives it
    (front_end/dynamic_tracker.cpp:348-470, front_end/instance_feature.cpp): ROI-local LK on zero-padded crops,
    Shi-Tomasi top-up inside the eroded instance mask, undistortion with the 2-D box offset, right-image LK on the
    full frames.  Host composition of the C-ABI operators (dv_track_by_lk, dv_erode, dv_circle_mask, dv_gftt,
    dv_lift_projective_offset); every pixel / index result is bit-exact against the oracle's dvo_inst_track.
    The VIODE segmentation-key test of TrackRightByPad is applied when the key image of the right frame (Context.viode_mask) is given."""

    global_id_count = 1        # InstFeat::global_id_count (static, shared with the background tracker in the reference)

    def __init__(self, ctx: Context, cam0, cam1, max_cnt=50, min_dist=4, flow_back=True):
        self.ctx, self.cam0, self.cam1 = ctx, cam0, cam1
        self.max_cnt, self.min_dist, self.flow_back = max_cnt, min_dist, flow_back      # max_dynamic_cnt, min_dynamic_dist, flow_back
        self.last_points = np.zeros((0, 2), np.float32)
        self.ids = np.zeros(0, np.uint32)
        self.track_cnt = np.zeros(0, np.int32)
        self.prev_roi_gray = None

    def Track(self, roi_gray, roi_mask, box_tl, gray0, gray1=None, seg1_keys=None, key=None):
        """one frame; returns dict(curr_points, ids, track_cnt, curr_un_points, right_points, right_ids, right_un_points)"""
        ctx = self.ctx
        roi_gray = np.ascontiguousarray(roi_gray)
        curr, ids, cnt = np.zeros((0, 2), np.float32), np.zeros(0, np.uint32), np.zeros(0, np.int32)
        if self.prev_roi_gray is not None and len(self.last_points):
            h = max(self.prev_roi_gray.shape[0], roi_gray.shape[0]); w = max(self.prev_roi_gray.shape[1], roi_gray.shape[1])
            a = np.zeros((h, w), np.uint8); b = np.zeros((h, w), np.uint8)          # InstanceImagePadding (feature_utils.cpp:406-413)
            a[: self.prev_roi_gray.shape[0], : self.prev_roi_gray.shape[1]] = self.prev_roi_gray
            b[: roi_gray.shape[0], : roi_gray.shape[1]] = roi_gray
            pts, st = ctx.track_by_lk(a, b, self.last_points, self.flow_back, 0.5)   # InstFeat::TrackLeft, no mask
            keep = st > 0
            curr, ids, cnt = pts[keep], self.ids[keep], self.track_cnt[keep] + 1
        if len(curr) < self.max_cnt:
            m = ctx.erode(np.ascontiguousarray(roi_mask), 5) if roi_mask is not None else np.full(roi_gray.shape, 255, np.uint8)
            if len(curr):
                m = ctx.circle_mask(m, curr, self.min_dist)
            new = ctx.gftt(roi_gray, self.max_cnt - len(curr), 0.01, self.min_dist, m)
            if len(new):
                nid = np.arange(InstFeat.global_id_count, InstFeat.global_id_count + len(new), dtype=np.uint32)
                InstFeat.global_id_count += len(new)
                curr = np.concatenate([curr, new]); ids = np.concatenate([ids, nid]); cnt = np.concatenate([cnt, np.ones(len(new), np.int32)])
        un = ctx.lift_projective(self.cam0, curr, offset=box_tl)                   # UndistortedPointsWithAddOffset
        out = dict(curr_points=curr, ids=ids, track_cnt=cnt, curr_un_points=un, right_points=np.zeros((0, 2), np.float32),
                   right_ids=np.zeros(0, np.uint32), right_un_points=np.zeros((0, 2), np.float32))
        if gray1 is not None and len(curr):                                         # TrackRightByPad
            padded = (curr + np.array(box_tl, np.float32)).astype(np.float32)
            rp, st = ctx.track_by_lk(gray0, gray1, padded, self.flow_back, 0.5)
            if seg1_keys is not None:                                              # VIODE::PixelToKey(right_points[i], img.seg1) != id -> lost
                xi = np.rint(rp[:, 0]).astype(int).clip(0, seg1_keys.shape[1] - 1); yi = np.rint(rp[:, 1]).astype(int).clip(0, seg1_keys.shape[0] - 1)
                st = np.where((st > 0) & (seg1_keys[yi, xi] == key), 1, 0).astype(np.uint8)
            keep = st > 0
            out["right_points"], out["right_ids"] = rp[keep], ids[keep]
            out["right_un_points"] = ctx.lift_projective(self.cam1, rp[keep])
        self.last_points, self.ids, self.track_cnt, self.prev_roi_gray = curr, ids, cnt, roi_gray      # PostProcess
        return out
