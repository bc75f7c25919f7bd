"""CPU test of the product's host-side geometry (csrc/line_host.h, csrc/inst_host.h — plain C++ headers inside the .so): compiled on their own with g++
(tests/host/host_logic_test.cpp, no HIP, no GPU) and compared with the oracle's restatements of the same reference functions on random inputs:
Plücker <-> orthonormal conversion, LineTrimming, TriangulateOneLine (through LineMgr::add / triangulate), FitBox3DWithRANSAC (seeded) and
FitBox3DFromCameraFrame."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from dynamic_vins_amd import sim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("hl") / "host_logic_test"
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-ffp-contract=off"] + os.environ.get("DVINS_CXX_SANITIZE", "").split()          # (tests/test_sanitizers.py: ASan + UBSan)
                   + ["-I", os.path.join(ROOT, "dynamic_vins_amd", "csrc"), "-o", str(out), os.path.join(ROOT, "tests", "host", "host_logic_test.cpp")], check=True)
    return str(out)


def run(exe, mode, text):
    r = subprocess.run([exe, mode], input=text, capture_output=True, text=True, check=True)
    return [np.array(ln.split(), float) for ln in r.stdout.splitlines()]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_line_geometry_matches_oracle(exe, oracle):
    rng = np.random.default_rng(5)
    L = oracle.lib
    rows, want = [], []
    for _ in range(40):
        p1, p2 = rng.uniform(-2, 2, 3) + [0, 0, 4], rng.uniform(-2, 2, 3) + [0, 0, 4]
        v = p2 - p1
        plk = np.concatenate([np.cross(p1, p2), v])
        obs = np.array([p1[0] / p1[2], p1[1] / p1[2], p2[0] / p2[2], p2[1] / p2[2]]) + rng.normal(0, 1e-3, 4)
        rows.append(" ".join(repr(float(x)) for x in np.concatenate([plk, obs])))
        orth, back, e1, e2 = np.zeros(4), np.zeros(6), np.zeros(3), np.zeros(3)
        L.dvo_plk_to_orth(_p(plk), _p(orth)); L.dvo_orth_to_plk(_p(orth), _p(back))
        L.dvo_line_trimming.restype = C.c_int
        ok = L.dvo_line_trimming(_p(plk), _p(obs), _p(e1), _p(e2))
        want.append(np.concatenate([orth, back, [ok], e1 if ok else np.zeros(3), e2 if ok else np.zeros(3)]))
    got = run(exe, "line", "\n".join(rows) + "\n")
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g[10] == w[10]
        assert np.allclose(g[:10], w[:10], rtol=0, atol=1e-12)
        if w[10]:
            assert np.allclose(g[11:], w[11:], rtol=1e-12, atol=1e-12)


def test_line_triangulation_matches_oracle(exe, oracle):
    rng = np.random.default_rng(6)
    L = oracle.lib
    L.dvo_triangulate_line.restype = C.c_int
    traj = sim.Trajectory()
    ric, tic = sim.R_IC, sim.T_IC0
    text, want = [], []
    for case in range(12):
        t0 = 2.0 + 0.7 * case
        Rs = np.array([traj.R(t0 + 0.4 * f) for f in range(11)]); Ps = np.array([traj.p(t0 + 0.4 * f) for f in range(11)])
        start, nobs = int(rng.integers(0, 4)), int(rng.integers(3, 7))
        # a segment 3-5 m in front of the start camera
        Rwc, twc = Rs[start] @ ric, Ps[start] + Rs[start] @ tic
        a = Rwc @ (rng.uniform(-1, 1, 3) + [0, 0, 4]) + twc
        b = a + Rwc @ rng.uniform(-1.5, 1.5, 3)
        obs = []
        for k in range(nobs):
            R, t = Rs[start + k] @ ric, Ps[start + k] + Rs[start + k] @ tic
            pa, pb = R.T @ (a - t), R.T @ (b - t)
            obs.append([pa[0] / pa[2], pa[1] / pa[2], pb[0] / pb[2], pb[1] / pb[2]])
        obs = np.array(obs) + rng.normal(0, 2e-4, (nobs, 4))
        text.append(f"{nobs} {start} " + " ".join(repr(float(x)) for x in np.concatenate([Rs.ravel(), Ps.ravel(), ric.ravel(), tic, obs.ravel()])))
        plk, w1, w2 = np.zeros(6), np.zeros(3), np.zeros(3)
        Rs_c, Ps_c, ric_c, tic_c, obs_c = (np.ascontiguousarray(x, np.float64) for x in (Rs, Ps, ric, tic, obs))
        ok = L.dvo_triangulate_line(_p(obs_c), nobs, start, _p(Rs_c), _p(Ps_c), _p(ric_c), _p(tic_c), _p(plk), _p(w1), _p(w2))
        want.append((ok, plk, w1, w2))
    got = run(exe, "tri", "\n".join(text) + "\n")
    assert len(got) == len(want) and sum(w[0] for w in want) >= 6
    for g, (ok, plk, w1, w2) in zip(got, want):
        assert int(g[0]) == ok
        if ok:
            assert np.allclose(g[1:7], plk, rtol=1e-10, atol=1e-12) and np.allclose(g[7:10], w1, atol=1e-9) and np.allclose(g[10:13], w2, atol=1e-9)


def test_box_fits_match_oracle(exe, oracle):
    rng = np.random.default_rng(7)
    L = oracle.lib
    L.dvo_fit_box_ransac.restype = C.c_int
    L.dvo_fit_box_ransac.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_ulonglong, C.c_void_p]
    L.dvo_fit_box_camera.restype = C.c_int
    L.dvo_fit_box_camera.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    text, want = [], []
    for case in range(16):
        n = int(rng.integers(1, 60))
        dims = rng.uniform(0.5, 3.0, 3)
        c = rng.uniform(-3, 3, 3) + [0, 0, 8]
        pts = c + rng.uniform(-0.5, 0.5, (n, 3)) * dims
        pts[rng.random(n) < 0.2] += rng.normal(0, 4, 3)                    # outliers
        seed = int(rng.integers(1, 2**62))
        text.append(" ".join([str(n)] + [repr(float(x)) for x in dims] + [str(seed)] + [repr(float(x)) for x in pts.ravel()]))
        o1, o2 = np.zeros(3), np.zeros(3)
        pc = np.ascontiguousarray(pts)
        L.dvo_fit_box_ransac(_p(pc), n, _p(dims), seed, _p(o1))
        ok = L.dvo_fit_box_camera(_p(pc), n, _p(dims), _p(o2))
        want.append((o1, ok, o2))
    got = run(exe, "box", "\n".join(text) + "\n")
    assert len(got) == len(want)
    for g, (o1, ok, o2) in zip(got, want):
        assert np.array_equal(g[:3], o1)                                   # same seeded draws, same sums: identical bits
        assert int(g[3]) == ok and np.array_equal(g[4:7], o2)
