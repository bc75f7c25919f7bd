"""ctypes wrapper of the CPU oracle (oracle/_build/libdvins_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
OLIB = os.environ.get("DVO_LIB", os.path.join(ODIR, "_build", "libdvins_oracle.so"))


class dvo_cam(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2")]


class dvo_fe_config(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("max_cnt", C.c_int), ("min_dist", C.c_int),
                ("flow_back", C.c_int), ("stereo", C.c_int), ("cam0", dvo_cam), ("cam1", dvo_cam)]


FEAT_DTYPE = np.dtype([("id", np.uint32), ("track_cnt", np.int32), ("has_right", np.int32), ("pad_", np.int32),
                       ("left", np.float64, 7), ("right", np.float64, 7)])


def build():
    subprocess.check_call(["make", "-s", "-C", ODIR])


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.dvo_tracker_create.restype = C.c_void_p
        lib.dvo_tracker_create.argtypes = [C.POINTER(dvo_fe_config)]
        lib.dvo_tracker_destroy.argtypes = [C.c_void_p]
        lib.dvo_tracker_track_image.restype = C.c_int
        lib.dvo_tracker_track_image.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        lib.dvo_tracker_track_image_naive.restype = C.c_int
        lib.dvo_tracker_track_image_naive.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        lib.dvo_tracker_track_image_mode.restype = C.c_int
        lib.dvo_tracker_track_image_mode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
        lib.dvo_lk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                               C.c_void_p, C.c_void_p]
        lib.dvo_track_by_lk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
        lib.dvo_gftt.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.POINTER(C.c_int)]
        lib.dvo_min_eigen.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.dvo_gftt_cuda.argtypes = lib.dvo_gftt.argtypes
        lib.dvo_min_eigen_cuda.argtypes = lib.dvo_min_eigen.argtypes
        lib.dvo_pyr_down.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.dvo_scharr.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.dvo_circle_mask.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        lib.dvo_erode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.dvo_lift_projective.argtypes = [C.POINTER(dvo_cam), C.c_void_p, C.c_int, C.c_void_p]

    def bgr2gray(self, bgr):
        bgr = np.ascontiguousarray(bgr)
        h, w, _ = bgr.shape
        out = np.zeros((h, w), np.uint8)
        self.lib.dvo_bgr2gray.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        self.lib.dvo_bgr2gray(_p(bgr), w, h, bgr.strides[0], _p(out))
        return out

    def remap(self, src, map1, map2):
        """cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) with CV_16SC2 + CV_16UC1 maps; src (h, w) or (h, w, 3) uint8"""
        src = np.ascontiguousarray(src)
        h, w = src.shape[:2]
        cn = 1 if src.ndim == 2 else src.shape[2]
        m1, m2 = np.ascontiguousarray(map1, np.int16), np.ascontiguousarray(map2, np.uint16)
        out = np.zeros_like(src)
        self.lib.dvo_remap.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.dvo_remap(_p(src), w, h, src.strides[0], cn, _p(m1), _p(m2), _p(out))
        return out

    def init_undistort_map(self, cam, new_k, w, h):
        """cv::initUndistortRectifyMap(K, D, I, newK, (w, h), CV_16SC2) -> (map1 (h, w, 2) int16, map2 (h, w) uint16)"""
        c = dvo_cam(*cam)
        nk = np.ascontiguousarray(new_k, np.float64)
        m1, m2 = np.zeros((h, w, 2), np.int16), np.zeros((h, w), np.uint16)
        self.lib.dvo_init_undistort_map.argtypes = [C.POINTER(dvo_cam), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self.lib.dvo_init_undistort_map(C.byref(c), _p(nk), w, h, _p(m1), _p(m2))
        return m1, m2

    def pyr_down(self, img):
        h, w = img.shape
        img = np.ascontiguousarray(img)
        dst = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
        self.lib.dvo_pyr_down(_p(img), w, h, _p(dst))
        return dst

    def scharr(self, img):
        h, w = img.shape
        img = np.ascontiguousarray(img)
        out = np.zeros((h, w, 2), np.int16)
        self.lib.dvo_scharr(_p(img), w, h, _p(out))
        return out

    def lk(self, a, b, pts_a, max_level=3, iters=30, eps=0.01, initial=None):
        h, w = a.shape
        a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
        pts_a = np.ascontiguousarray(pts_a, np.float32)
        n = len(pts_a)
        pts_b = np.ascontiguousarray(initial, np.float32).copy() if initial is not None else np.zeros((n, 2), np.float32)
        st = np.zeros(n, np.uint8)
        self.lib.dvo_lk(_p(a), _p(b), w, h, _p(pts_a), n, max_level, iters, eps, 1 if initial is not None else 0, _p(pts_b), _p(st))
        return pts_b, st

    def track_by_lk(self, a, b, pts1, flow_back=True, dist_thresh=0.5):
        h, w = a.shape
        a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
        pts1 = np.ascontiguousarray(pts1, np.float32)
        n = len(pts1)
        pts2 = np.zeros((n, 2), np.float32)
        st = np.zeros(n, np.uint8)
        self.lib.dvo_track_by_lk(_p(a), _p(b), w, h, _p(pts1), n, int(flow_back), dist_thresh, _p(pts2), _p(st))
        return pts2, st

    def viode_mask(self, seg_bgr, dyn_keys):
        """VIODE::SetViodeMaskSimple / BuildViodeMask (utils/dataset/viode_utils.cpp:21-170) -> (merge_mask, inv_merge_mask, key image, boxes[nkeys, 4]): frontend.Context.viode_mask's twin"""
        seg = np.ascontiguousarray(seg_bgr, np.uint8)
        h, w, _ = seg.shape
        keys = np.ascontiguousarray(dyn_keys, np.uint32)
        merge, inv, kimg, boxes = np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint32), np.zeros((len(keys), 4), np.int32)
        self.lib.dvo_viode_mask.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.dvo_viode_mask(_p(seg), w, h, seg.strides[0], _p(keys), len(keys), _p(merge), _p(inv), _p(kimg), _p(boxes))
        return merge, inv, kimg, boxes

    def gftt(self, img, max_n, quality, min_dist, mask=None, rule="cpu"):
        """rule "cpu": cv::goodFeaturesToTrack; "cuda": cv::cuda::GoodFeaturesToTrackDetector (oracle/gftt_cuda.cpp)"""
        h, w = img.shape
        img = np.ascontiguousarray(img)
        if mask is not None:
            mask = np.ascontiguousarray(mask)
        out = np.zeros((max(h * w // 4, 16), 2), np.float32)
        n = C.c_int(0)
        (self.lib.dvo_gftt_cuda if rule == "cuda" else self.lib.dvo_gftt)(_p(img), _p(mask), w, h, int(max_n), float(quality), float(min_dist), _p(out), C.byref(n))
        return out[: n.value].copy()

    def min_eigen(self, img, rule="cpu"):
        h, w = img.shape
        img = np.ascontiguousarray(img)
        eig = np.zeros((h, w), np.float32)
        (self.lib.dvo_min_eigen_cuda if rule == "cuda" else self.lib.dvo_min_eigen)(_p(img), w, h, _p(eig))
        return eig

    def circle_mask(self, mask, pts, radius):
        h, w = mask.shape
        out = np.ascontiguousarray(mask).copy()
        pts = np.ascontiguousarray(pts, np.float32)
        self.lib.dvo_circle_mask(_p(out), w, h, _p(pts), len(pts), int(radius))
        return out

    def erode(self, mask, k):
        h, w = mask.shape
        mask = np.ascontiguousarray(mask)
        out = np.zeros_like(mask)
        self.lib.dvo_erode(_p(mask), w, h, int(k), _p(out))
        return out

    def lift_projective(self, cam, pts):
        pts = np.ascontiguousarray(pts, np.float32)
        out = np.zeros_like(pts)
        c = dvo_cam(*cam)
        self.lib.dvo_lift_projective(C.byref(c), _p(pts), len(pts), _p(out))
        return out

    def inst_track(self, prev_roi, cur_roi, cur_mask, box_tl, gray0, gray1, cam0, cam1, last_pts, ids, track_cnt, max_cnt, min_dist, flow_back, global_id):
        """dvo_inst_track -> (dict like frontend.InstFeat.Track, new global id)"""
        lib = self.lib
        lib.dvo_inst_track.restype = C.c_int
        cap = max_cnt + len(last_pts) + 8
        cur_pts, cur_ids, cur_cnt, cur_un = np.zeros((cap, 2), np.float32), np.zeros(cap, np.uint32), np.zeros(cap, np.int32), np.zeros((cap, 2), np.float32)
        r_pts, r_ids, r_un = np.zeros((cap, 2), np.float32), np.zeros(cap, np.uint32), np.zeros((cap, 2), np.float32)
        n_cur, n_right, gid = C.c_int(0), C.c_int(0), C.c_uint32(global_id)
        c0, c1 = dvo_cam(*cam0), dvo_cam(*cam1)
        cur_roi = np.ascontiguousarray(cur_roi)
        prev = np.ascontiguousarray(prev_roi) if prev_roi is not None else None
        mask = np.ascontiguousarray(cur_mask) if cur_mask is not None else None
        g0 = np.ascontiguousarray(gray0); g1 = np.ascontiguousarray(gray1) if gray1 is not None else None
        last = np.ascontiguousarray(last_pts, np.float32); idv = np.ascontiguousarray(ids, np.uint32); cv = np.ascontiguousarray(track_cnt, np.int32)
        V = C.c_void_p
        lib.dvo_inst_track.argtypes = [V, C.c_int, C.c_int, V, C.c_int, C.c_int, V, C.c_int, C.c_int, V, V, C.c_int, C.c_int, C.POINTER(dvo_cam), C.POINTER(dvo_cam),
                                       C.c_int, V, V, V, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int), V, V, V, V, C.POINTER(C.c_int), V, V, V]
        lib.dvo_inst_track(_p(prev), prev.shape[1] if prev is not None else 0, prev.shape[0] if prev is not None else 0, _p(cur_roi), cur_roi.shape[1], cur_roi.shape[0],
                           _p(mask), int(box_tl[0]), int(box_tl[1]), _p(g0), _p(g1), g0.shape[1], g0.shape[0], C.byref(c0), C.byref(c1),
                           len(last), _p(last), _p(idv), _p(cv), int(max_cnt), int(min_dist), int(flow_back), C.byref(gid), C.byref(n_cur), _p(cur_pts), _p(cur_ids), _p(cur_cnt),
                           _p(cur_un), C.byref(n_right), _p(r_pts), _p(r_ids), _p(r_un))
        n, m = n_cur.value, n_right.value
        return dict(curr_points=cur_pts[:n], ids=cur_ids[:n], track_cnt=cur_cnt[:n], curr_un_points=cur_un[:n], right_points=r_pts[:m], right_ids=r_ids[:m],
                    right_un_points=r_un[:m]), gid.value

    def pyr_down_cuda(self, img):
        img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
        out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
        self.lib.dvo_pyr_down_cuda.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        self.lib.dvo_pyr_down_cuda(_p(img), w, h, _p(out))
        return out

    def tex_read(self, img, x, y):
        img = np.ascontiguousarray(img, np.uint8); h, w = img.shape
        f = self.lib.dvo_tex_read
        f.restype = C.c_float; f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
        return f(_p(img), w, h, float(x), float(y))

    def lk_cuda(self, a, b, pts_a, max_level=3, iters=30, initial=None):
        """cv::cuda::SparsePyrLKOpticalFlow(Size(21, 21), max_level, iters, useInitialFlow = initial is not None)->calc"""
        a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8); h, w = a.shape
        pa = np.ascontiguousarray(pts_a, np.float32).reshape(-1, 2)
        pb = np.ascontiguousarray(initial, np.float32).reshape(-1, 2).copy() if initial is not None else np.zeros_like(pa)
        st = np.zeros(len(pa), np.uint8)
        self.lib.dvo_lk_cuda.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self.lib.dvo_lk_cuda(_p(a), _p(b), w, h, _p(pa), len(pa), int(max_level), int(iters), int(initial is not None), _p(pb), _p(st))
        return pb, st

    def track_by_lk_gpu(self, a, b, pts1, flow_back=True):
        """FeatureTrackByLKGpu (front_end/feature_utils.cpp:83-163)"""
        a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8); h, w = a.shape
        p1 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2)
        p2 = np.zeros_like(p1); st = np.zeros(len(p1), np.uint8)
        self.lib.dvo_track_by_lk_gpu.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self.lib.dvo_track_by_lk_gpu(_p(a), _p(b), w, h, _p(p1), len(p1), int(flow_back), _p(p2), _p(st))
        return p2, st

    def detect_extra_points(self, mask, box_xy, disp, cam4, baseline):
        """InstFeat::DetectExtraPoints -> float32 [n, 3] (x, y, depth), row-major scan order"""
        mask = np.ascontiguousarray(mask, np.uint8); disp = np.ascontiguousarray(disp, np.float32)
        out = np.zeros((mask.size, 3), np.float32)
        f = self.lib.dvo_detect_extra_points
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int]
        n = f(_p(mask), mask.shape[1], mask.shape[0], int(box_xy[0]), int(box_xy[1]), _p(disp), disp.shape[1], disp.shape[0], float(cam4[0]), float(cam4[1]), float(cam4[2]), float(cam4[3]),
              float(baseline), _p(out), len(out))
        return out[:n].copy()

    def process_extra_points(self, xyz):
        """RadiusOutlierRemoval(0.5, 10) + EuclideanClusterExtraction(1.0, 10, 25000), first cluster -> float32 [m, 3]"""
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        out = np.zeros((max(len(xyz), 1), 3), np.float32)
        f = self.lib.dvo_process_extra_points
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        m = f(_p(xyz) if len(xyz) else None, len(xyz), _p(out))
        return out[:m].copy()

    def insts(self, tracker, max_dynamic_cnt=50, min_dynamic_dist=5, use_det3d=0):
        return OracleInsts(self, tracker, max_dynamic_cnt, min_dynamic_dist, use_det3d)

    def tracker(self, width, height, max_cnt, min_dist, flow_back, stereo, cam0, cam1):
        return OracleTracker(self, width, height, max_cnt, min_dist, flow_back, stereo, cam0, cam1)

    def estimator(self, **kw):
        return OracleEstimator(self, **kw)


class OracleTracker:
    def __init__(self, o, width, height, max_cnt, min_dist, flow_back, stereo, cam0, cam1):
        self.o = o
        cfg = dvo_fe_config(width, height, max_cnt, min_dist, flow_back, stereo, dvo_cam(*cam0), dvo_cam(*cam1))
        self.h = o.lib.dvo_tracker_create(C.byref(cfg))
        self.out = np.zeros(max_cnt + 8, FEAT_DTYPE)

    def track_image(self, g0, g1, t, mask=None, naive=False, mode=None, erode_k=0):
        g0 = np.ascontiguousarray(g0)
        g1 = np.ascontiguousarray(g1) if g1 is not None else None
        if mode is not None:
            m = np.ascontiguousarray(mask) if mask is not None else None
            n = self.o.lib.dvo_tracker_track_image_mode(self.h, _p(g0), _p(g1), _p(m), int(mode), int(erode_k), t, _p(self.out))
        elif naive:
            n = self.o.lib.dvo_tracker_track_image_naive(self.h, _p(g0), _p(g1), _p(mask), t, _p(self.out))
        else:
            n = self.o.lib.dvo_tracker_track_image(self.h, _p(g0), _p(g1), t, _p(self.out))
        return self.out[:n].copy()

    def close(self):
        if self.h:
            self.o.lib.dvo_tracker_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class dvo_inst_det(C.Structure):
    _fields_ = [("track_id", C.c_uint32), ("class_id", C.c_int32), ("x", C.c_int32), ("y", C.c_int32), ("w", C.c_int32), ("h", C.c_int32),
                ("mask", C.c_void_p), ("points", C.c_void_p), ("n_points", C.c_int32), ("pad_", C.c_int32)]


class OracleInsts:
    """dvo_insts: InstsFeatManager of the oracle, sharing the id counter of an OracleTracker (the background tracker must run first each frame)"""

    def __init__(self, o, tracker, max_dynamic_cnt, min_dynamic_dist, use_det3d):
        self.lib, self.tracker = o.lib, tracker
        L = self.lib
        L.dvo_insts_create.restype = C.c_void_p
        L.dvo_insts_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.dvo_insts_destroy.argtypes = [C.c_void_p]
        L.dvo_insts_track.restype = C.c_int
        L.dvo_insts_track.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.dvo_insts_output.restype = C.c_int
        L.dvo_insts_output.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        self.h = L.dvo_insts_create(tracker.h, max_dynamic_cnt, min_dynamic_dist, use_det3d)

    def set_disparity(self, disp, baseline):
        """SemanticImage::disp of the next track() call: the extra points then come from InstFeat::DetectExtraPoints + ProcessExtraPoints (oracle/extra_points.cpp)"""
        d = None if disp is None else np.ascontiguousarray(disp, np.float32)
        self.lib.dvo_insts_set_disparity.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        self.lib.dvo_insts_set_disparity(self.h, _p(d), float(baseline))

    def set_right_keys(self, key_img):
        """VIODE: the key image (oracle.viode_mask(...)[2]) of seg1 of the next track() call: TrackRightByPad's segmentation-key test (instance_feature.cpp:263-268)"""
        k = None if key_img is None else np.ascontiguousarray(key_img, np.uint32)
        self.lib.dvo_insts_set_right_keys.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.dvo_insts_set_right_keys(self.h, _p(k))

    def track(self, g0, g1, t, dets, boxes3d, inst_dtype, box_dtype):
        g0 = np.ascontiguousarray(g0); g1 = np.ascontiguousarray(g1) if g1 is not None else None
        arr = (dvo_inst_det * max(len(dets), 1))()
        keep = []
        for k, d in enumerate(dets):
            m = np.ascontiguousarray(d["mask"], np.uint8)
            pts = None if d.get("points") is None else np.ascontiguousarray(d["points"], np.float64)
            keep += [m, pts]
            x, y, w, h = [int(v) for v in d["rect"]]
            arr[k].track_id, arr[k].class_id, arr[k].x, arr[k].y, arr[k].w, arr[k].h = int(d["track_id"]), int(d.get("class_id", 0)), x, y, w, h
            arr[k].mask = m.ctypes.data
            arr[k].points = pts.ctypes.data if pts is not None and len(pts) else None
            arr[k].n_points = 0 if pts is None else len(pts)
        b3 = np.ascontiguousarray(boxes3d, box_dtype) if boxes3d is not None else np.zeros(0, box_dtype)
        self.lib.dvo_insts_track(self.h, _p(g0), _p(g1), float(t), C.addressof(arr) if len(dets) else None, len(dets), _p(b3) if len(b3) else None, len(b3))
        oi, of, op = np.zeros(64, inst_dtype), np.zeros(64 * 256, FEAT_DTYPE), np.zeros((1 << 16, 3))
        ni, nf, npt = C.c_int(0), C.c_int(0), C.c_int(0)
        rc = self.lib.dvo_insts_output(self.h, _p(oi), len(oi), C.byref(ni), _p(of), len(of), C.byref(nf), _p(op), len(op), C.byref(npt))
        assert rc == 0
        return oi[: ni.value].copy(), of[: nf.value].copy(), op[: npt.value].copy()

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.dvo_insts_destroy(self.h); self.h = None


class dvo_be_config(C.Structure):
    _fields_ = [("use_imu", C.c_int), ("stereo", C.c_int), ("plane_constraint", C.c_int), ("max_iters", C.c_int),
                ("keyframe_parallax", C.c_double), ("init_depth", C.c_double), ("g_norm", C.c_double), ("td", C.c_double),
                ("acc_n", C.c_double), ("gyr_n", C.c_double), ("acc_w", C.c_double), ("gyr_w", C.c_double),
                ("ric", (C.c_double * 9) * 2), ("tic", (C.c_double * 3) * 2),
                ("dynamic", C.c_int), ("use_det3d", C.c_int), ("instance_init_min_num", C.c_int), ("estimate", C.c_int), ("static_inst_threshold", C.c_double),
                ("use_line", C.c_int), ("line_min_obs", C.c_int), ("line_sqrt_info", C.c_double * 4)]


class dvo_be_state(C.Structure):
    _fields_ = [("frame", C.c_int), ("nonlinear", C.c_int), ("margin_old", C.c_int), ("n_landmarks", C.c_int),
                ("n_long", C.c_int), ("iterations", C.c_int), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("window", (C.c_double * 16) * 11)]


def make_be_config(cls, use_imu=1, stereo=1, plane_constraint=0, max_iters=8, keyframe_parallax=10.0, init_depth=5.0,
                   g_norm=9.81, td=0.0, acc_n=0.1, gyr_n=0.01, acc_w=0.001, gyr_w=1e-4, ric=None, tic=None,
                   dynamic=0, use_det3d=0, instance_init_min_num=4, static_inst_threshold=10.0, use_line=0, line_min_obs=5, line_sqrt_info=(0.0, 0.0, 0.0, 0.0), estimate=0):
    c = cls()
    c.estimate = estimate
    c.use_line, c.line_min_obs = use_line, line_min_obs
    for i in range(4):
        c.line_sqrt_info[i] = float(line_sqrt_info[i])
    c.dynamic, c.use_det3d, c.instance_init_min_num, c.static_inst_threshold = dynamic, use_det3d, instance_init_min_num, static_inst_threshold
    c.use_imu, c.stereo, c.plane_constraint, c.max_iters = use_imu, stereo, plane_constraint, max_iters
    c.keyframe_parallax, c.init_depth, c.g_norm, c.td = keyframe_parallax, init_depth, g_norm, td
    c.acc_n, c.gyr_n, c.acc_w, c.gyr_w = acc_n, gyr_n, acc_w, gyr_w
    for k in range(2):
        for i in range(9):
            c.ric[k][i] = float(np.asarray(ric[k]).reshape(-1)[i])
        for i in range(3):
            c.tic[k][i] = float(tic[k][i])
    return c


class OracleEstimator:
    def __init__(self, o, **kw):
        self.lib = o.lib
        L = self.lib
        L.dvo_estimator_create.restype = C.c_void_p
        L.dvo_estimator_create.argtypes = [C.POINTER(dvo_be_config)]
        L.dvo_estimator_destroy.argtypes = [C.c_void_p]
        L.dvo_estimator_input_imu.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        L.dvo_estimator_process.restype = C.c_int
        L.dvo_estimator_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.POINTER(dvo_be_state)]
        self.cfg = make_be_config(dvo_be_config, **kw)
        self.h = L.dvo_estimator_create(C.byref(self.cfg))
        self.state = dvo_be_state()

    def input_imu(self, t, acc, gyr):
        a = np.ascontiguousarray(acc, np.float64)
        g = np.ascontiguousarray(gyr, np.float64)
        self.lib.dvo_estimator_input_imu(self.h, float(t), _p(a), _p(g))

    def process(self, rows, t):
        rows = np.ascontiguousarray(rows)
        rc = self.lib.dvo_estimator_process(self.h, _p(rows), len(rows), float(t), C.byref(self.state))
        return rc, self.state

    def window(self):
        return np.array([list(r) for r in self.state.window])

    def extrinsics(self):
        ric, tic, td = np.zeros(18), np.zeros(6), C.c_double(0)
        self.lib.dvo_estimator_get_extrinsics.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.dvo_estimator_get_extrinsics(self.h, _p(ric), _p(tic), C.addressof(td))
        return ric.reshape(2, 3, 3), tic.reshape(2, 3), td.value

    def process_dynamic(self, rows, t, insts, inst_feats, points):
        L = self.lib
        L.dvo_estimator_process_dynamic.restype = C.c_int
        L.dvo_estimator_process_dynamic.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(dvo_be_state)]
        rows, insts, inst_feats = np.ascontiguousarray(rows), np.ascontiguousarray(insts), np.ascontiguousarray(inst_feats)
        points = np.ascontiguousarray(points, np.float64)
        rc = L.dvo_estimator_process_dynamic(self.h, _p(rows), len(rows), float(t), _p(insts) if len(insts) else None, len(insts), _p(inst_feats) if len(inst_feats) else None,
                                             _p(points) if len(points) else None, C.byref(self.state))
        return rc, self.state

    def static_instances(self, cap=256):
        """ids the estimator reported static at its last dynamic frame (InstanceManager::GetOutputInstInfo, is_static; system/main.cpp:194,217-245)"""
        ids = np.zeros(cap, np.uint32); n = C.c_int(0)
        self.lib.dvo_estimator_get_static_instances.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        self.lib.dvo_estimator_get_static_instances(self.h, _p(ids), cap, C.byref(n))
        return ids[: n.value].copy()

    def instances(self, dtype, cap=64):
        L = self.lib
        L.dvo_estimator_get_instances.restype = C.c_int
        L.dvo_estimator_get_instances.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p]
        out = np.zeros(cap, dtype); n = C.c_int(0); summ = np.zeros(4)
        L.dvo_estimator_get_instances(self.h, _p(out), cap, C.byref(n), _p(summ))
        return out[: n.value].copy(), summ

    def set_lines(self, rows):
        L = self.lib
        L.dvo_estimator_set_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        rows = np.ascontiguousarray(rows)
        return L.dvo_estimator_set_lines(self.h, _p(rows) if len(rows) else None, len(rows))

    def lines(self, dtype, cap=1024):
        L = self.lib
        L.dvo_estimator_get_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        out = np.zeros(cap, dtype); n = C.c_int(0)
        L.dvo_estimator_get_lines(self.h, _p(out), cap, C.byref(n))
        return out[: n.value].copy()

    def close(self):
        if self.h:
            self.lib.dvo_estimator_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


_inst = None


def load():
    global _inst
    if _inst is None:
        if not os.path.exists(OLIB):
            build()
        _inst = Oracle(C.CDLL(OLIB))
    return _inst
