"""dvins_node (dynamic_vins_amd/host/dvins_node.cpp): the ROS-free node on the library's C++ runner.
CPU: its image readers (binary PGM / PPM, 8-bit non-interlaced PNG of colour types 0 / 2 / 6 with every filter type) decode what was written.
GPU: a rendered stereo + IMU sequence written to disk (PGM left, PNG right, EuRoC imu.csv, a config in the reference's YAML dialect) goes through the node; the
`<seq>_<mode>_Odometry.txt` it writes equals, byte for byte, the lines the Python pipeline produces on the same frames with the same frame flow (every pair tracked,
every 2nd pair to the back end: system/main.cpp:300-307)."""
import os
import struct
import subprocess
import zlib

import numpy as np

from dynamic_vins_amd import _abi
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = os.path.join(ROOT, "dynamic_vins_amd", "bin", "dvins_node")


def write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n# a comment line\n%d %d\n255\n" % (img.shape[1], img.shape[0])); f.write(np.ascontiguousarray(img, np.uint8).tobytes())


def write_png(path, img, filters=(0, 1, 2, 3, 4)):
    """minimal PNG writer: 8-bit gray [h, w] or RGB / RGBA [h, w, c]; row y is stored with filter type filters[y % len]"""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    ctype = {1: 0, 3: 2, 4: 6}[ch]
    rows = img.reshape(h, w * ch).astype(np.int32)
    raw = bytearray()
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        f = filters[y % len(filters)]
        cur = rows[y]
        a = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]]); b = prev; c = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
        if f == 0: out = cur
        elif f == 1: out = cur - a
        elif f == 2: out = cur - b
        elif f == 3: out = cur - ((a + b) >> 1)
        else:
            p = a + b - c; pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c)); out = cur - pred
        raw.append(f); raw += (out & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    z = zlib.compress(bytes(raw), 6)
    half = len(z) // 2          # two IDAT chunks: the reader must concatenate them
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) + chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b""))


def checksum(img):
    d = np.ascontiguousarray(img, np.uint8).reshape(-1).astype(np.uint64)
    k = (np.arange(len(d), dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
    return int(d.sum()), int((d * k).sum())


@pytest.mark.skipif(not os.path.exists(NODE), reason="build first: python -c 'import __graft_entry__ as g; g.build()'")
def test_node_decodes_pgm_and_png(tmp_path):
    rng = np.random.default_rng(5)
    gray = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    rgb = rng.integers(0, 256, (29, 41, 3), dtype=np.uint8)
    rgba = rng.integers(0, 256, (16, 23, 4), dtype=np.uint8)
    write_pgm(tmp_path / "a.pgm", gray); write_png(tmp_path / "b.png", gray); write_png(tmp_path / "c.png", rgb); write_png(tmp_path / "d.png", rgba, filters=(4, 3))
    want_gray = lambda im: ((im[..., 2].astype(np.int64) * 1868 + im[..., 1].astype(np.int64) * 9617 + im[..., 0].astype(np.int64) * 4899 + 8192) >> 14).astype(np.uint8)      # cvtColor BGR2GRAY
    out = subprocess.run([NODE, "--decode", str(tmp_path / "a.pgm"), str(tmp_path / "b.png"), str(tmp_path / "c.png"), str(tmp_path / "d.png")], capture_output=True, text=True, check=True).stdout.split("\n")
    for line, img in zip(out, [gray, gray, want_gray(rgb), want_gray(rgba[..., :3])]):
        w, h, s, ws = (int(v) for v in line.split())
        assert (w, h) == (img.shape[1], img.shape[0]) and (s, ws) == checksum(img)


CFG = """%YAML:1.0
imu: 1
num_of_cam: 2
dataset_type: "custom"
slam_type: "raw"
use_line: 0
undistort_input: 0
plane_constraint: 0
image_width: {w}
image_height: {h}
cam0_calib: "cam.yaml"
cam1_calib: "cam.yaml"
estimate_extrinsic: 0
body_T_cam0: !!opencv-matrix
  rows: 4
  cols: 4
  dt: d
  data: [0.0, 0.0, 1.0, 0.0,
         -1.0, 0.0, 0.0, 0.0,
         0.0, -1.0, 0.0, 0.0,
         0.0, 0.0, 0.0, 1.0]
body_T_cam1: !!opencv-matrix
  rows: 4
  cols: 4
  dt: d
  data: [0.0, 0.0, 1.0, 0.0,
         -1.0, 0.0, 0.0, -0.12,
         0.0, -1.0, 0.0, 0.0,
         0.0, 0.0, 0.0, 1.0]
max_cnt: 150
min_dist: 20
flow_back: 1
use_mask_morphology: 0
max_solver_time: 0.04
max_num_iterations: 8
keyframe_parallax: 10.0
acc_n: 0.02
gyr_n: 0.002
acc_w: 2.0e-4
gyr_w: 2.0e-5
g_norm: 9.81
estimate_td: 0
td: 0.0
INIT_DEPTH: 5.0
"""
CAM = """%YAML:1.0
---
model_type: PINHOLE
camera_name: camera
image_width: {w}
image_height: {h}
distortion_parameters:
   k1: {k1!r}
   k2: {k2!r}
   p1: {p1!r}
   p2: {p2!r}
projection_parameters:
   fx: {fx!r}
   fy: {fy!r}
   cx: {cx!r}
   cy: {cy!r}
"""


@pytest.mark.gpu
def test_node_trajectory_file_equals_the_python_pipeline(tmp_path):
    from dynamic_vins_amd import io_formats, sim
    from dynamic_vins_amd.pipeline import Pipeline, SyntheticSequence
    w, h, frames = 640, 360, 50
    cam = sim.scaled_cam(sim.ZED, w, h, 1280, 720)
    seq = SyntheticSequence(w, h, cam, frames, rate=20.0, t0=0.0)
    sd = tmp_path / "seq07"
    (sd / "left").mkdir(parents=True); (sd / "right").mkdir()
    for k in range(frames):
        l, r = seq.host_frame(k)
        write_pgm(sd / "left" / f"{k:06d}.pgm", l); write_png(sd / "right" / f"{k:06d}.png", r)
    with open(sd / "imu.csv", "w") as f:
        f.write("#timestamp [ns],w_RS_S_x [rad s^-1],w_RS_S_y,w_RS_S_z,a_RS_S_x [m s^-2],a_RS_S_y,a_RS_S_z\n")
        for t, a, g in zip(seq.imu_t, seq.imu_a, seq.imu_g):
            f.write("%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g\n" % (t, g[0], g[1], g[2], a[0], a[1], a[2]))      # (stamps in seconds: the reader accepts ns or s)
    open(sd / "times.txt", "w").write("".join("%.17g\n" % t for t in seq.times))      # exact stamps (without the file the node counts 0.05 s per pair like Dataloader::LoadStereo)
    (tmp_path / "cfg").mkdir()
    open(tmp_path / "cfg" / "node.yaml", "w").write(CFG.format(w=w, h=h))
    open(tmp_path / "cfg" / "cam.yaml", "w").write(CAM.format(w=w, h=h, **cam))
    out = subprocess.run([NODE, str(tmp_path / "cfg" / "node.yaml"), str(sd), str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    got = open(tmp_path / "seq07_VIO_raw_PointOnly_Odometry.txt").read().splitlines()
    # the same frames through the Python pipeline: host frames, every 2nd tracked pair to the back end, the YAML's estimator parameters
    pipe = Pipeline(seq, max_cnt=150, min_dist=20, max_iters=8, host_frames=True, ba_stride=2, est_kw=dict(keyframe_parallax=10.0, g_norm=9.81))
    want = []
    for k in range(frames):
        pipe.step()
        if k % 2 == 0:          # frames 0, 2, 4, ... reach the back end (system/main.cpp:181,300-312: cnt % 2 == 0, cnt from 0)
            want.append(io_formats.trajectory_line(seq.times[k], pipe.est.window()[10, :7]))
    pipe.ctx.close()
    assert len(got) == len(want) == frames // 2
    assert got == want, [i for i, (a, b) in enumerate(zip(got, want)) if a != b][:5]
    assert sum(1 for ln in got if not ln.endswith("0.000000 0.000000 0.000000 0.000000 0.000000 0.000000 1.000000")) >= frames // 2 - 12      # the solved part is not the identity


def _viode_setup(tmp_path, frames, w=640, h=360):
    """a VIODE-layout directory written from the synthetic renderer (dynamic_vins_amd/viode.py) + the context that produced its masks"""
    from dynamic_vins_amd import sim
    from dynamic_vins_amd.frontend import Context, make_cam
    from dynamic_vins_amd.viode import ViodeSequence
    cam = sim.scaled_cam(sim.ZED, w, h, 1280, 720)
    c = make_cam(*sim.cam_tuple(cam))
    masker = Context(width=w, height=h, max_cnt=150, min_dist=20, cam0=c, cam1=c)
    seq = ViodeSequence(w, h, cam, frames, masker, rate=20.0, t0=0.0)
    masker.close()
    sd = tmp_path / "city_day_3_high"
    cfg = seq.write(str(sd), est=dict(max_cnt=150, min_dist=20, iters=8, morph=5, static_inst_threshold=10.0))
    return seq, sd, cfg


@pytest.mark.gpu
def test_node_viode_dynamic_mode_from_files_equals_the_python_pipeline_and_the_oracle(tmp_path, oracle):
    """config 3 (VIODE, slam_type dynamic) end to end FROM FILES: left / right / segmentation0 / segmentation1 PNGs + imu.csv + the VIODE keys of the YAML (rgb_to_label_file,
    dynamic_label_id) -> dvins_node: dv_viode_mask (masks, key images, boxes; track_id = key) -> TrackSemanticImage + InstsTrack with the segmentation-key test of the right
    image -> the dynamic back end on every 2nd pair -> <seq>_VIO_dynamic_PointOnly_Odometry.txt.  The file must equal the Python dynamic pipeline's lines byte for byte, and
    the oracle (tracker mode 2 + instance tracker + dynamic estimator fed from the oracle's own viode_mask) must agree to 1e-5 m.
    Reference: system/main.cpp:196-215, image_process/image_process.cpp:161-178, utils/dataset/viode_utils.cpp:21-218, front_end/dynamic_tracker.cpp:585-605."""
    from dynamic_vins_amd import dynsim, io_formats, sim, viode
    from dynamic_vins_amd.pipeline import DynamicPipeline
    frames = 64
    seq, sd, cfg = _viode_setup(tmp_path, frames)
    assert min(len(d) for d in seq.dets) >= 2 and len(seq.dyn_keys) == 3
    out = subprocess.run([NODE, cfg, str(sd), str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "dynamic mode from the segmentation images" in out.stdout
    got = open(tmp_path / "city_day_3_high_VIO_dynamic_PointOnly_Odometry.txt").read().splitlines()
    # ---- the same through the Python dynamic pipeline (ctypes on the same C ABI), every 2nd pair to the back end ----
    ekw = dict(keyframe_parallax=10.0, g_norm=9.81007, instance_init_min_num=4)
    pipe = DynamicPipeline(seq, max_cnt=150, min_dist=20, max_iters=8, use_det3d=0, static_inst_threshold=10.0, mask_morphology_size=5, extra_from_disparity=False, ba_stride=2, est_kw=ekw,
                           static_as_background=True)          # (the YAML does not switch it off: the reference's default)
    want = []
    for k in range(frames):
        pipe.step()
        if k % 2 == 0:
            want.append(io_formats.trajectory_line(seq.times[k], pipe.est.window()[10, :7]))
    Id, _ = pipe.est.instances()
    assert pipe.stat["object_features"] > 10 * frames // 2 and len(Id) >= 2, pipe.stat
    hip_traj = np.array([p[:3] for p in pipe.poses])
    pipe.ctx.close()
    assert len(got) == len(want) == frames // 2
    assert got == want, [i for i, (a, b) in enumerate(zip(got, want)) if a != b][:5]
    # ---- the oracle from the same files' content: its own viode_mask, its own trackers and estimator ----
    camt = sim.cam_tuple(seq.cam)
    trk = oracle.tracker(seq.w, seq.h, 150, 20, 1, 1, camt, camt)
    oin = oracle.insts(trk, 50, 5, 0)
    est = oracle.estimator(use_imu=1, stereo=1, max_iters=8, ric=seq.rig["est_ric"], tic=seq.rig["est_tic"], dynamic=1, use_det3d=0, static_inst_threshold=10.0, instance_init_min_num=4,
                           keyframe_parallax=10.0, g_norm=9.81007, **seq.noise)
    k_imu, o_traj, snaps = 0, [], []
    for k in range(frames):
        t = seq.times[k]
        left, right = seq.host_frame(k)
        _, inv, kimg, bx = oracle.viode_mask(seq.seg0[k], seq.dyn_keys)
        assert np.array_equal(inv, seq.inv_mask[k])
        dets = viode.detections(kimg, bx, seq.dyn_keys)
        best = [sn for sn in snaps if sn[0] <= k - _abi.DV_STATIC_REPORT_LAG]
        inv = viode.unmask_static(inv, dets, best[-1][1] if best else [])          # FeatureTrack, system/main.cpp:217-245
        rows = trk.track_image(left, right, t, mask=inv, mode=2, erode_k=5)
        oin.set_right_keys(oracle.viode_mask(seq.seg1[k], seq.dyn_keys)[2])
        io, fo, po = oin.track(left, right, t, dets, None, dynsim.INSTOBS_DTYPE, dynsim.BOX3D_DTYPE)
        if k % 2:
            continue
        while k_imu < len(seq.imu_t) and seq.imu_t[k_imu] <= t + 0.006:
            est.input_imu(seq.imu_t[k_imu], seq.imu_a[k_imu], seq.imu_g[k_imu]); k_imu += 1
        rc, so = est.process_dynamic(rows, t, io, fo, po)
        assert rc == 0
        snaps = (snaps + [(k, est.static_instances())])[-4:]
        if so.nonlinear:
            o_traj.append(est.window()[10, :3].copy())
    o_traj = np.array(o_traj)
    assert len(o_traj) == len(hip_traj) >= frames // 2 - 12
    assert np.abs(o_traj - hip_traj).max() < 1e-5, np.abs(o_traj - hip_traj).max()


@pytest.mark.gpu
def test_node_viode_naive_mode_from_files(tmp_path):
    """viode.yaml as SHIPPED says slam_type naive: segmentation0 -> VIODE::SetViodeMaskSimple -> TrackImageNaive (GPU tracker's and GPU detector's rules) with the inverse mask
    -> the raw back end.  The node's file must equal the Python pipeline's lines (dv_runner_set_mask vs per-frame ctypes calls)."""
    from dynamic_vins_amd import io_formats
    from dynamic_vins_amd.backend import Estimator
    from dynamic_vins_amd.frontend import DV_MODE_NAIVE
    frames = 40
    seq, sd, cfg = _viode_setup(tmp_path, frames)
    text = open(cfg).read().replace('slam_type: "dynamic"', 'slam_type: "naive"')
    open(cfg, "w").write(text)
    out = subprocess.run([NODE, cfg, str(sd), str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    got = open(tmp_path / "city_day_3_high_VIO_naive_PointOnly_Odometry.txt").read().splitlines()
    from dynamic_vins_amd.frontend import Context, make_cam
    from dynamic_vins_amd import sim
    c = make_cam(*sim.cam_tuple(seq.cam))
    ctx = Context(width=seq.w, height=seq.h, max_cnt=150, min_dist=20, cam0=c, cam1=c, mask_morphology_size=5)
    est = Estimator(ctx, use_imu=1, stereo=1, max_iters=8, ric=seq.rig["est_ric"], tic=seq.rig["est_tic"], keyframe_parallax=10.0, g_norm=9.81007, **seq.noise)
    k_imu, want = 0, []
    for k in range(frames):
        t = seq.times[k]
        l, r = seq.host_frame(k)
        rows = ctx.track_stereo(l, r, t, seq.inv_mask[k], DV_MODE_NAIVE)
        if k % 2:
            continue
        while k_imu < len(seq.imu_t) and seq.imu_t[k_imu] <= t + 0.006:
            est.InputIMU(seq.imu_t[k_imu], seq.imu_a[k_imu], seq.imu_g[k_imu]); k_imu += 1
        rc, st = est.ProcessMeasurements(rows, t)
        assert rc == 0
        want.append(io_formats.trajectory_line(t, est.window()[10, :7]))
    ctx.close()
    assert got == want, [i for i, (a, b) in enumerate(zip(got, want)) if a != b][:5]


REF_SRC = "/root/reference/dynamic_vins/src"


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference sources are not on this machine")
def test_topic_table_matches_the_reference(tmp_path):
    """N1's ROS surface as data (host/dvins_topics.hpp): every subscription of SystemCallBack's constructor and every topic the publishers use appear in the table
    with the reference's names, message types and queue sizes; nothing in the table is absent from the reference."""
    import re
    src = '#include <cstdio>\n#include "dvins_topics.hpp"\nint main() { for (auto& t : dynamic_vins::kSubscriptions) std::printf("S %s %d %s %d %s\\n", t.topic_or_config_key, (int)t.from_config, t.msg_type, t.queue, t.callback);' \
          ' for (auto& t : dynamic_vins::kPublications) std::printf("P %s %s\\n", t.topic, t.msg_type); }\n'
    open(tmp_path / "t.cpp", "w").write(src)
    exe = str(tmp_path / "t")
    subprocess.run(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "dynamic_vins_amd", "host"), str(tmp_path / "t.cpp"), "-o", exe], check=True)
    rows = [ln.split() for ln in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines()]
    subs = {r[1]: r for r in rows if r[0] == "S"}; pubs = {r[1]: r[2] for r in rows if r[0] == "P"}
    cb = open(os.path.join(REF_SRC, "utils/io/system_call_back.cpp")).read()
    hdr = open(os.path.join(REF_SRC, "utils/io/system_call_back.h")).read()
    key_of = {"kImage0Topic": "image0_topic", "kImage1Topic": "image1_topic", "kImage0SegTopic": "image0_segmentation_topic", "kImage1SegTopic": "image1_segmentation_topic", "kImuTopic": "imu_topic"}
    found = re.findall(r"nh\.subscribe\(\s*(io_para::(\w+)|\"([^\"]+)\")\s*,\s*(\d+)\s*,\s*&SystemCallBack::(\w+)", cb)
    assert len(found) == len(subs) == 9
    for _, var, lit, queue, callback in found:
        name = key_of[var] if var else lit
        assert name in subs, name
        _, _, from_cfg, mtype, q, cbk = subs[name]
        assert int(from_cfg) == int(bool(var)) and int(q) == int(queue) and cbk == callback
        arg = re.search(r"void\s+%s\(const\s+(\w+)::(\w+)ConstPtr" % callback, hdr)
        assert arg and mtype == f"{arg.group(1)}/{arg.group(2)}", (callback, mtype)
    vis = "".join(open(os.path.join(REF_SRC, f)).read() for f in ("utils/io/visualization.cpp", "system/main.cpp", "estimator/estimator.cpp"))
    ref_pubs = {}
    for t, name in re.findall(r"PublisherMap::Pub<\s*([\w:]+)\s*>\([^,]+,\s*\"(\w+)\"", vis): ref_pubs[name] = {"Marker": "visualization_msgs/Marker"}.get(t, t.replace("::", "/"))
    for t, name in re.findall(r"GetPublisher<\s*([\w:]+)\s*>\(\s*\"(\w+)\"", vis): ref_pubs[name] = {"MarkerArray": "visualization_msgs/MarkerArray"}.get(t, t.replace("::", "/"))
    for name in re.findall(r"PubMarkers\([^,]+,\s*\"(\w+)\"", vis): ref_pubs[name] = "visualization_msgs/MarkerArray"
    for name in re.findall(r"PubImage\([^,]+,\s*\"(\w+)\"", vis): ref_pubs[name] = "sensor_msgs/Image"
    pc = re.findall(r"PubPointCloud\(\s*\*?(\w+)\s*,\s*\"(\w+)\"", vis)
    for var, name in pc: ref_pubs[name] = "sensor_msgs/PointCloud2" if name in ("instance_point_cloud", "stereo_point_cloud") else "sensor_msgs/PointCloud"      # pcl clouds go out as PointCloud2 (publisher_map.cpp)
    assert set(ref_pubs) == set(pubs), (set(ref_pubs) ^ set(pubs))
    for name, t in ref_pubs.items():
        assert pubs[name] == t, (name, pubs[name], t)
