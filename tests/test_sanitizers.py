"""Sanitizer builds of the CPU-side code (SURVEY 5 "sanitizers on the host shim").  CPU only — GPU AddressSanitizer is not available on this pool and
nothing here loads the HIP runtime under a sanitizer.
  * AddressSanitizer + UndefinedBehaviorSanitizer: the CPU oracle (`make -C oracle asan`, loaded into an interpreter started with LD_PRELOAD=libasan.so) under the
    oracle's own checks, and the g++ builds of tests/host/*.cpp (the product's host-side geometry headers, the header shim's CPU commands) through their usual tests;
  * ThreadSanitizer: the C++ runner's host machinery (runner.hip compiled as plain C++: spin barriers, teams, thread-per-group, the tracker thread + ring of a dynamic
    sequence, the failure path) and the shim's concurrent surface (FeatureQueue, Estimator::InputIMU beside ProcessMeasurements and the getters, StereoSync) against a
    stand-in C ABI that sleeps instead of launching (tests/host/stub_abi.cpp).
Findings of the first run (round 6), fixed: group_track read a teammate's frame counter while that teammate advanced it (runner.hip; the value was unused);
StereoSync::pending_left / pending_right / dropped_* read the deques unguarded (dvins_shim.hpp).  The whole CPU suite under ASan + UBSan: clean."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")


def _lib(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_lib("libasan.so") is None, reason="no libasan in this toolchain")
def test_oracle_and_host_cpp_under_asan_ubsan():
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ)
    env.update(LD_PRELOAD=_lib("libasan.so"), ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               DVO_LIB=os.path.join(ROOT, "oracle", "_build", "libdvins_oracle_asan.so"),
               DVINS_CXX_SANITIZE="-g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined")
    files = ["tests/test_oracle_checks.py", "tests/test_host_logic.py", "tests/test_host_shim.py", "tests/test_io_formats.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + files, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    assert " passed" in r.stdout


@pytest.mark.skipif(_lib("libtsan.so") is None, reason="no libtsan in this toolchain")
def test_runner_and_shim_under_tsan():
    r = subprocess.run(["make", "-s", "-C", HOST, "tsan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ)
    env["TSAN_OPTIONS"] = "halt_on_error=1:exitcode=66:second_deadlock_stack=1"
    cfg = os.path.join(ROOT, "tests", "golden", "config", "zed_like.yaml")
    runs = [([os.path.join(HOST, "_build", "runner_tsan"), "raw"], {}), ([os.path.join(HOST, "_build", "runner_tsan"), "dynamic"], {}),
            ([os.path.join(HOST, "_build", "runner_tsan"), "fail"], {"DVSTUB_FAIL": "1:12"}), ([os.path.join(HOST, "_build", "shim_tsan"), cfg], {})]
    for cmd, extra in runs:
        e = dict(env); e.update(extra)
        r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=600)
        if r.returncode != 0 and "unexpected memory mapping" in r.stderr:
            pytest.skip("ThreadSanitizer cannot map its shadow memory on this kernel")
        assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, (cmd, r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    assert "DIFFERENT" not in r.stdout


@pytest.mark.skipif(_lib("libasan.so") is None, reason="no libasan in this toolchain")
def test_node_image_decoders_on_malformed_files_under_asan(tmp_path):
    """dvins_node's PGM / PNG readers face files from outside.  A mutation run under ASan + UBSan (2169 truncated / bit-flipped / header-edited images; 3974 YAML / calibration
    files and 974 feature files through the shim's parsers: clean) found two defects in round 6, both fixed: a PGM cut right behind its header read past the buffer
    (`buf.size() - p` wrapped), and a PNG whose IHDR claims a size its compressed stream cannot fill allocated (and cleared) that size — tens of gigabytes from a 500-byte file.
    The deterministic cases below are the regression: each must end in the node's own error (exit 1, "dvins_node: ..."), never in a sanitizer report, and quickly."""
    import struct
    import numpy as np
    from tests.test_node import write_pgm, write_png
    r = subprocess.run(["make", "-s", "-C", HOST, "_build/dvins_node_asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    exe = os.path.join(HOST, "_build", "dvins_node_asan")
    rng = np.random.default_rng(7)
    write_png(tmp_path / "g.png", rng.integers(0, 256, (19, 23), dtype=np.uint8)); write_png(tmp_path / "a.png", rng.integers(0, 256, (9, 13, 4), dtype=np.uint8), filters=(4, 3))
    write_pgm(tmp_path / "p.pgm", rng.integers(0, 256, (7, 9), dtype=np.uint8))
    png, pgm = open(tmp_path / "g.png", "rb").read(), open(tmp_path / "p.pgm", "rb").read()
    hdr = pgm.index(b"255") + 3
    cases = {"pgm_cut_behind_header.pgm": pgm[:hdr], "pgm_cut_in_header.pgm": pgm[:5], "pgm_half.pgm": pgm[: hdr + 20], "pgm_negative.pgm": b"P5\n-1 -1\n255\n" + pgm[hdr + 1:],
             "pgm_huge.pgm": b"P5\n70000 70000\n255\n" + pgm[hdr + 1:], "png_short_ihdr.png": png[:8] + struct.pack(">I", 3) + png[12:], "png_cut.png": png[: len(png) // 2],
             "png_ihdr_at_the_end.png": png + struct.pack(">I", 0) + b"IHDR"}
    for w, h in [(0, 5), (5, 0), (0x7FFFFFFF, 1), (1, 0x7FFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF), (65536, 65536), (65535, 65535), (23, 1000000)]:
        cases["png_%x_%x.png" % (w, h)] = png[:16] + struct.pack(">II", w, h) + png[24:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    ok = subprocess.run([exe, "--decode", str(tmp_path / "g.png"), str(tmp_path / "a.png"), str(tmp_path / "p.pgm")], env=env, capture_output=True, text=True, timeout=60)
    assert ok.returncode == 0 and len(ok.stdout.splitlines()) == 3, ok.stdout + ok.stderr[-2000:]
    for name, data in cases.items():
        open(tmp_path / name, "wb").write(data)
        r = subprocess.run([exe, "--decode", str(tmp_path / name)], env=env, capture_output=True, text=True, errors="replace", timeout=20)
        assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (name, r.stderr[-2500:])
        if name == "png_ihdr_at_the_end.png":          # bytes behind IEND are not read: decodes like the original
            assert r.returncode == 0 and r.stdout == ok.stdout.splitlines()[0] + "\n", (name, r.returncode, r.stdout)
            continue
        assert r.returncode == 1 and "dvins_node" in (r.stdout + r.stderr), (name, r.returncode, r.stdout[-300:], r.stderr[-300:])


@pytest.mark.skipif(_lib("libasan.so") is None, reason="no libasan in this toolchain")
def test_node_parse_path_under_asan_up_to_the_missing_device(tmp_path):
    """the node's whole input side — YAML config + camera file, directory listing, every image of the pairs (PGM left, PNG right), imu.csv with comment / short / ns-stamped lines,
    times.txt — under ASan + UBSan on a CPU box: everything is read and checked BEFORE dv_create, which then fails loudly (no device).  With a GPU the run goes through."""
    import numpy as np
    import torch
    from tests.test_node import CAM, CFG, write_pgm, write_png
    r = subprocess.run(["make", "-s", "-C", HOST, "_build/dvins_node_asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    w, h, n = 64, 48, 5
    rng = np.random.default_rng(3)
    sd = tmp_path / "seq"
    (sd / "left").mkdir(parents=True); (sd / "right").mkdir()
    for k in range(n):
        write_pgm(sd / "left" / f"{k:06d}.pgm", rng.integers(0, 256, (h, w), dtype=np.uint8)); write_png(sd / "right" / f"{k:06d}.png", rng.integers(0, 256, (h, w), dtype=np.uint8))
    with open(sd / "imu.csv", "w") as f:
        f.write("#timestamp [ns],w_x,w_y,w_z,a_x,a_y,a_z\n\n1403636579758555392,0.1,0.2,0.3,9.8,0.0,0.1\n0.05,0,0\nnot,a,number,at,all,x,y\n")
        for i in range(40):
            f.write("%.6f,0.0,0.0,0.0,0.0,0.0,9.81\n" % (0.005 * i))
    open(sd / "times.txt", "w").write("".join("%.3f\n" % (0.05 * k) for k in range(n)))
    (tmp_path / "cfg").mkdir()
    open(tmp_path / "cfg" / "node.yaml", "w").write(CFG.format(w=w, h=h))
    open(tmp_path / "cfg" / "cam.yaml", "w").write(CAM.format(w=w, h=h, k1=-0.1, k2=0.01, p1=0.0, p2=0.0, fx=35.0, fy=35.0, cx=32.0, cy=24.0))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    exe = os.path.join(HOST, "_build", "dvins_node_asan")
    r = subprocess.run([exe, str(tmp_path / "cfg" / "node.yaml"), str(sd), str(tmp_path)], env=env, capture_output=True, text=True, errors="replace", timeout=300)
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stdout[-500:] + r.stderr[-500:]
    else:
        assert r.returncode == 1 and "dvins_node:" in (r.stdout + r.stderr), (r.returncode, r.stdout[-500:], r.stderr[-500:])
    # a right image of another size is refused by name, a missing imu.csv likewise
    write_png(sd / "right" / "000002.png", rng.integers(0, 256, (h, w + 2), dtype=np.uint8))
    r = subprocess.run([exe, str(tmp_path / "cfg" / "node.yaml"), str(sd), str(tmp_path)], env=env, capture_output=True, text=True, errors="replace", timeout=300)
    assert r.returncode == 1 and "000002.png" in (r.stdout + r.stderr) and "AddressSanitizer" not in r.stderr, (r.returncode, r.stdout[-300:], r.stderr[-800:])

