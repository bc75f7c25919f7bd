"""seeded inputs + oracle drivers for the line / dynamic-object factors (rows L1, I1-I3)"""
import ctypes as C

import numpy as np


def rand_pose(rng, scale=2.0):
    q = rng.normal(0, 1, 4)
    return np.concatenate([rng.normal(0, scale, 3), q / np.linalg.norm(q)])


def pose_plus(x, d):
    """PoseLocalParameterization::Plus (factor/pose_local_parameterization.cpp:26-45)"""
    x1, y1, z1, w1 = x[3:7]
    x2, y2, z2, w2 = d[3] / 2, d[4] / 2, d[5] / 2, 1.0
    qq = np.array([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
                   w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2])
    return np.concatenate([x[:3] + d[:3], qq / np.linalg.norm(qq)])


def _call(fn, consts, blocks, nres, jsizes):
    blocks = [np.ascontiguousarray(b, np.float64) for b in blocks]
    J = [np.zeros(nres * s) for s in jsizes]
    pp = (C.c_void_p * len(blocks))(*[b.ctypes.data for b in blocks])
    Jp = (C.c_void_p * len(blocks))(*[j.ctypes.data for j in J])
    r = np.zeros(nres)
    consts = [np.ascontiguousarray(c, np.float64) for c in consts]
    fn(*[c.ctypes.data for c in consts], pp, r.ctypes.data, Jp)
    return r, [j.reshape(nres, s) for j, s in zip(J, jsizes)]


def o_line(lib, obs, sqrt_info, pose, ex, orth):
    lib.dvo_line_eval.argtypes = [C.c_void_p] * 5
    return _call(lib.dvo_line_eval, [obs, sqrt_info], [pose, ex, orth], 2, [7, 7, 4])


def o_line_plus(lib, orth, delta):
    lib.dvo_line_plus.argtypes = [C.c_void_p] * 3
    a, d, out = np.ascontiguousarray(orth, np.float64), np.ascontiguousarray(delta, np.float64), np.zeros(4)
    lib.dvo_line_plus(a.ctypes.data, d.ctypes.data, out.ctypes.data)
    return out


def o_box_enclose(lib, pts_w, dims, pose_obj):
    lib.dvo_box_enclose_eval.argtypes = [C.c_void_p] * 5
    return _call(lib.dvo_box_enclose_eval, [pts_w, dims], [pose_obj], 3, [7])


def o_box_dims(lib, dims, box):
    lib.dvo_box_dims_eval.argtypes = [C.c_void_p] * 4
    return _call(lib.dvo_box_dims_eval, [dims], [box], 1, [3])


def o_box_orientation(lib, R_cioi, R_bc, pose_b, pose_o):
    lib.dvo_box_orientation_eval.argtypes = [C.c_void_p] * 5
    return _call(lib.dvo_box_orientation_eval, [R_cioi, R_bc], [pose_b, pose_o], 3, [7, 7])


def rot(rng, angle=None):
    a = rng.normal(0, 1, 3)
    a /= np.linalg.norm(a)
    th = rng.uniform(0.05, 2.5) if angle is None else angle
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def make_batch(seed, n):
    rng = np.random.default_rng(seed)
    d = {}
    d["obs"] = rng.uniform(-0.6, 0.6, (n, 4))
    d["sqrt_info"] = np.tile(np.array([460 / 1.5, 0, 0, 460 / 1.5]), (n, 1))
    d["sqrt_info"][::5] = rng.normal(0, 100, (len(d["sqrt_info"][::5]), 4))
    d["sqrt_info"][1::7] = 0.0                      # what the reference actually runs with
    d["pose"] = np.array([rand_pose(rng) for _ in range(n)])
    d["ex"] = np.array([np.concatenate([rng.normal(0, 0.05, 3), np.array([0.5, -0.5, 0.5, -0.5]) + rng.normal(0, 0.02, 4)]) for _ in range(n)])
    d["ex"][:, 3:] /= np.linalg.norm(d["ex"][:, 3:], axis=1, keepdims=True)
    d["orth"] = np.stack([rng.uniform(-3, 3, n), rng.uniform(-1.4, 1.4, n), rng.uniform(-3, 3, n), rng.uniform(0.1, 1.4, n)], 1)
    d["delta"] = rng.normal(0, 0.05, (n, 4))
    d["pose_obj"] = np.array([rand_pose(rng, 5.0) for _ in range(n)])
    d["dims"] = rng.uniform(1.0, 4.5, (n, 3))
    local = rng.normal(0, 1.5, (n, 3))
    d["pts_w"] = np.array([d["pose_obj"][k][:3] + _qR(d["pose_obj"][k][3:]) @ local[k] for k in range(n)])
    d["box"] = d["dims"] + rng.normal(0, 0.3, (n, 3))
    d["R_cioi"] = np.array([rot(rng).ravel() for _ in range(n)])
    d["R_bc"] = np.array([rot(rng).ravel() for _ in range(n)])
    d["pose_body"] = np.array([rand_pose(rng) for _ in range(n)])
    return d


def _qR(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
