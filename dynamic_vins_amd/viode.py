"""VIODE input side of dynamic / naive mode: the segmentation images that stand in for the detector on this data set.

Reference: utils/dataset/viode_utils.{h,cpp} — PixelToKey (r * 1000000 + g * 1000 * b, sic), ReadViodeRgbIds (rgb_ids.txt: id,r,g,b per line, first line a header),
SetParameters (dynamic_label_id, rgb_to_label_file), IsDynamic (the key's label id is one of dynamic_label_id), SetViodeMaskAndRoi (one Box2D per key present in seg0:
id = track_id = key, rect = cv::Rect(min_pt, max_pt) — max EXCLUSIVE —, ROI mask = the key's pixels inside the rect), image_process/image_process.cpp:161-178 (which
of the two is called), front_end/dynamic_tracker.cpp:585-605 AddViodeInstances (the boxes become the frame's instances, key = track id) and
front_end/instance_feature.cpp:263-268 (TrackRightByPad keeps a right-image point only where seg1 carries the object's key).

This module is the HOST logic of that path in Python (dvins_node.cpp holds the same logic in C++): which keys are dynamic, detections from dv_viode_mask's outputs, and a
writer of VIODE-layout sequence directories from the synthetic renderer, so that config 3 can be run end to end from files.  The per-pixel work is dv_viode_mask."""
import os
import zlib

import numpy as np

MIN_INST_SIZE = 8          # a detection's rectangle must be at least this many pixels on both sides (see detections())


def pixel_to_key(r, g, b):
    """VIODE::PixelToKey (viode_utils.h:23-26): r * 1000000 + g * 1000 * b — the product is the reference's formula, not a typo here"""
    return (np.asarray(r, np.uint32) * np.uint32(1000000) + np.asarray(g, np.uint32) * np.uint32(1000) * np.asarray(b, np.uint32)).astype(np.uint32)


def read_rgb_ids(path):
    """ReadViodeRgbIds (viode_utils.cpp:223-248): header line skipped, `id,r,g,b` per line, key -> id with unordered_map::insert semantics (the FIRST line of a key wins)"""
    key_to_id = {}
    with open(path) as f:
        f.readline()
        for line in f:
            parts = line.strip().split(",")
            if len(parts) < 4:
                continue
            v = [int(float(x)) if x.strip() else 0 for x in parts[:4]]          # atoi
            key_to_id.setdefault(int(pixel_to_key(v[1], v[2], v[3])), v[0])
    return key_to_id


def dynamic_keys(key_to_id, dynamic_label_ids):
    """the keys VIODE::IsDynamic accepts, ascending (ViodeKeyToIndex[key] in ViodeDynamicIndex; an unknown key maps to label 0)"""
    dyn = set(int(i) for i in dynamic_label_ids)
    return np.array(sorted(k for k, i in key_to_id.items() if i in dyn), np.uint32)


def detections(key_img, boxes, dyn_keys, min_size=MIN_INST_SIZE):
    """SetViodeMaskAndRoi's Box2D list from dv_viode_mask's key image and per-key bounding boxes (row_min, row_max, col_min, col_max; row_min > row_max = key absent):
    ascending key (the reference walks an unordered_map; the object tracker visits its instances in ascending id either way).  rect = (col_min, row_min, col_max - col_min,
    row_max - row_min): cv::Rect(min_pt, max_pt) excludes the max row / column.  A rectangle under min_size pixels on a side is not handed on: the reference would go
    on with an ROI that cannot hold a feature (min_dynamic_dist 5, 5 x 5 erosion) and, at 0 pixels, with an empty cv::Mat — declared deviation."""
    dets = []
    for k, key in enumerate(np.asarray(dyn_keys, np.uint32)):
        r0, r1, c0, c1 = [int(v) for v in boxes[k]]
        if r1 < r0 or c1 < c0:
            continue
        w, h = c1 - c0, r1 - r0
        if w < min_size or h < min_size:
            continue
        mask = np.ascontiguousarray(np.where(key_img[r0:r0 + h, c0:c0 + w] == key, 255, 0).astype(np.uint8))
        dets.append(dict(track_id=int(key), class_id=0, rect=(c0, r0, w, h), mask=mask, points=None))
    return dets


def unmask_static(inv_mask, dets, static_ids):
    """FeatureTrack, system/main.cpp:217-245, on the host (the oracle side of the parity tests; the product does this on the device: dv_track_unmask_static): the ROI-mask pixels
    of every detection whose track_id is in static_ids become background (inv_merge_mask 255)"""
    out = np.array(inv_mask, np.uint8, copy=True)
    ids = set(int(i) for i in static_ids)
    for d in dets:
        if int(d["track_id"]) in ids:
            x, y, w, h = [int(v) for v in d["rect"]]
            roi = out[y:y + h, x:x + w]
            roi[np.asarray(d["mask"])[: roi.shape[0], : roi.shape[1]] >= 1] = 255
    return out


# ---- a VIODE-layout sequence directory from the synthetic renderer ----

def object_colour(obj_id):
    """a colour (r, g, b) whose key is unique per object id (and never the background's)"""
    return (10 + 7 * int(obj_id), 3 + int(obj_id), 5 + 2 * int(obj_id))


BACKGROUND_COLOUR = (90, 120, 60)


def label_image(ident):
    """id map (0 = background) -> the segmentation image as cv::imread delivers it: H x W x 3, B G R"""
    ident = np.asarray(ident)
    out = np.empty(ident.shape + (3,), np.uint8)
    out[...] = BACKGROUND_COLOUR[::-1]
    for i in np.unique(ident):
        if i:
            out[ident == i] = object_colour(i)[::-1]
    return out


def write_png(path, img):
    """8-bit gray (H x W) or RGB (H x W x 3) PNG, filter 0, one IDAT"""
    import struct
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    ctype = 0 if img.ndim == 2 else 2
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 1)) + chunk(b"IEND", b""))


class ViodeSequence:
    """pipeline.DynamicSequence's scene as a VIODE bag would deliver it: stereo frames, the two segmentation images per frame, IMU — and NO detector outputs: the
    detections, masks and the right key image come out of the segmentation images (through `masker`: a frontend.Context or the oracle, both have viode_mask).
    Duck-types what DynamicPipeline / backend.Runner read of a DynamicSequence."""

    def __init__(self, w, h, cam, n_frames, masker, rate=20.0, t0=1.0, device=None, boxes=("escort", 3), baseline=0.12, dynamic_label_ids=(241, 242, 243), noise=None):
        import torch
        from . import dynsim, sim
        from .render import DynRoomRenderer
        self.w, self.h, self.cam, self.cam1, self.dt, self.t0 = w, h, cam, cam, 1.0 / rate, t0
        self.noise = noise or dict(acc_n=0.02, gyr_n=0.002, acc_w=2e-4, gyr_w=2e-5)
        self.rig = sim.rig(baseline, False)
        self.traj = sim.Trajectory()
        self.boxes = dynsim.escort_boxes(self.traj, boxes[1]) if isinstance(boxes, tuple) and boxes[0] == "escort" else (boxes if boxes is not None else dynsim.default_boxes())
        rr = DynRoomRenderer(cam, w, h, device=device, seed=sim.TEX_SEED)
        self.times = [t0 + k * self.dt for k in range(n_frames)]
        # rgb_ids.txt: every object gets a label id of dynamic_label_ids (cycled), the background label 1; one extra static label that is NOT dynamic
        self.dynamic_label_ids = list(dynamic_label_ids)
        self.rgb_rows = [(1,) + BACKGROUND_COLOUR] + [(self.dynamic_label_ids[i % len(self.dynamic_label_ids)],) + object_colour(b.id) for i, b in enumerate(self.boxes)] + [(7, 1, 2, 3)]
        key_to_id = {}
        for i, r, g, b in self.rgb_rows:
            key_to_id.setdefault(int(pixel_to_key(r, g, b)), i)
        self.dyn_keys = dynamic_keys(key_to_id, self.dynamic_label_ids)
        self.frames, self.seg0, self.seg1, self.inv_mask, self.inv_mask_dev, self.dets, self.boxes3d, self.right_keys = [], [], [], [], [], [], [], []
        self.disp_dev, self.baseline = [], float(baseline)
        for t in self.times:
            R, p = self.traj.R(t), self.traj.p(t)
            left, id0, _ = rr._render_dyn(R @ sim.R_IC, p + R @ sim.T_IC0, self.boxes, t, True)
            right, id1, _ = rr._render_dyn(R @ sim.R_IC, p + R @ self.rig["t_ic1"], self.boxes, t, True, rr.rays1)
            self.frames.append((left, right))
            s0, s1 = label_image(id0.cpu().numpy()), label_image(id1.cpu().numpy())
            self.seg0.append(s0); self.seg1.append(s1)
            _, inv, kimg, bx = masker.viode_mask(s0, self.dyn_keys)
            self.inv_mask.append(inv)
            self.inv_mask_dev.append(torch.from_numpy(inv).to(left.device) if left.is_cuda else torch.from_numpy(inv))
            self.dets.append(detections(kimg, bx, self.dyn_keys))
            self.boxes3d.append(np.zeros(0, dynsim.BOX3D_DTYPE))
            self.right_keys.append(masker.viode_mask(s1, self.dyn_keys)[2])
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.imu_t, self.imu_a, self.imu_g = sim.imu_stream(self.traj, t0 - 0.05, self.times[-1] + 0.1, 200.0, seed=0xBEEF, **self.noise)

    def host_frame(self, k):
        return self.frames[k][0].cpu().numpy(), self.frames[k][1].cpu().numpy()

    def write(self, root, cfg_name="viode_like.yaml", est=None):
        """<root>/left, right, segmentation0, segmentation1 (PNG, %06d), times.txt, imu.csv (EuRoC layout), rgb_ids.txt, cam0/1_pinhole.yaml and a config in the
        reference's YAML dialect with the VIODE keys (dataset_type viode, slam_type dynamic, dynamic_label_id, rgb_to_label_file) -> the config's path"""
        from . import sim
        est = est or {}
        for d in ("left", "right", "segmentation0", "segmentation1"):
            os.makedirs(os.path.join(root, d), exist_ok=True)
        for k in range(len(self.frames)):
            l, r = self.host_frame(k)
            write_png(os.path.join(root, "left", "%06d.png" % k), l); write_png(os.path.join(root, "right", "%06d.png" % k), r)
            write_png(os.path.join(root, "segmentation0", "%06d.png" % k), self.seg0[k][..., ::-1]); write_png(os.path.join(root, "segmentation1", "%06d.png" % k), self.seg1[k][..., ::-1])
        with open(os.path.join(root, "times.txt"), "w") as f:
            f.write("".join("%.17g\n" % t for t in self.times))
        with open(os.path.join(root, "imu.csv"), "w") as f:
            f.write("#timestamp [s],w_x,w_y,w_z,a_x,a_y,a_z\n")
            for t, a, g in zip(self.imu_t, self.imu_a, self.imu_g):
                f.write("%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g\n" % (t, g[0], g[1], g[2], a[0], a[1], a[2]))
        with open(os.path.join(root, "rgb_ids.txt"), "w") as f:
            f.write("id,r,g,b\n" + "".join("%d,%d,%d,%d\n" % row for row in self.rgb_rows))
        c = sim.cam_tuple(self.cam)
        for name in ("cam0_pinhole.yaml", "cam1_pinhole.yaml"):
            with open(os.path.join(root, name), "w") as f:
                f.write("%%YAML:1.0\n---\nmodel_type: PINHOLE\ncamera_name: camera\nimage_width: %d\nimage_height: %d\ndistortion_parameters:\n   k1: %.17g\n   k2: %.17g\n   p1: %.17g\n   p2: %.17g\n"
                        "projection_parameters:\n   fx: %.17g\n   fy: %.17g\n   cx: %.17g\n   cy: %.17g\n" % (self.w, self.h, c[4], c[5], c[6], c[7], c[0], c[1], c[2], c[3]))

        def mat(T):
            return "!!opencv-matrix\n   rows: 4\n   cols: 4\n   dt: d\n   data: [" + ", ".join("%.17g" % v for v in np.asarray(T).ravel()) + "]"
        T0, T1 = np.eye(4), np.eye(4)
        T0[:3, :3], T0[:3, 3] = self.rig["est_ric"][0], self.rig["est_tic"][0]
        T1[:3, :3], T1[:3, 3] = self.rig["est_ric"][1], self.rig["est_tic"][1]
        path = os.path.join(root, cfg_name)
        with open(path, "w") as f:
            f.write("%%YAML:1.0\n\nimu: 1\nnum_of_cam: 2\ndataset_type: \"viode\"\nslam_type: \"dynamic\"\nuse_line: 0\nundistort_input: 0\nplane_constraint: 0\n"
                    "image_width: %d\nimage_height: %d\ncam0_calib: \"cam0_pinhole.yaml\"\ncam1_calib: \"cam1_pinhole.yaml\"\nestimate_extrinsic: 0\n" % (self.w, self.h))
            f.write("body_T_cam0: " + mat(T0) + "\nbody_T_cam1: " + mat(T1) + "\n")
            f.write("max_cnt: %d\nmin_dist: %d\nflow_back: 1\nmin_dynamic_dist: 5\nmax_dynamic_cnt: 50\nuse_mask_morphology: %d\nmask_morphology_size: %d\n"
                    % (est.get("max_cnt", 150), est.get("min_dist", 20), 1 if est.get("morph", 0) else 0, est.get("morph", 5) or 5))
            f.write("max_solver_time: 0.04\nmax_num_iterations: %d\nkeyframe_parallax: 10.0\nacc_n: %.17g\ngyr_n: %.17g\nacc_w: %.17g\ngyr_w: %.17g\ng_norm: %.17g\nestimate_td: 0\ntd: 0.0\n"
                    % (est.get("iters", 8), self.noise["acc_n"], self.noise["gyr_n"], self.noise["acc_w"], self.noise["gyr_w"], sim.G_NORM if hasattr(sim, "G_NORM") else 9.81007))
            f.write("instance_init_min_num: 4\ninstance_static_err_threshold: %.17g\nuse_det3d: 0\n" % est.get("static_inst_threshold", 10.0))
            f.write("rgb_to_label_file: \"rgb_ids.txt\"\ndynamic_label_id: [" + ",".join(str(i) for i in self.dynamic_label_ids) + "]\n")
        return path
