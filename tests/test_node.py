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
