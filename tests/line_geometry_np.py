"""Line geometry of the path's line mode: Plücker / orthonormal representations, rigid transforms,
endpoint trimming and two-view line triangulation — host-side O(1) work per line, numpy mirror of
line_detector/line_geometry.cpp:75-296 and estimator/vio_util.cpp:447-561 (TriangulateOneLine).
A Plücker line is (n, v): moment and direction; an observation is (x1, y1, x2, y2) on the normalised plane."""
import numpy as np


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def orth_to_plk(orth):
    t1, t2, t3, phi = orth
    s1, c1, s2, c2, s3, c3 = np.sin(t1), np.cos(t1), np.sin(t2), np.cos(t2), np.sin(t3), np.cos(t3)
    R = np.array([[c2 * c3, s1 * s2 * c3 - c1 * s3, c1 * s2 * c3 + s1 * s3], [c2 * s3, s1 * s2 * s3 + c1 * c3, c1 * s2 * s3 - s1 * c3], [-s2, s1 * c2, c1 * c2]])
    return np.concatenate([np.cos(phi) * R[:, 0], np.sin(phi) * R[:, 1]])


def plk_to_orth(plk):
    n, v = plk[:3], plk[3:]
    u1, u2 = n / np.linalg.norm(n), v / np.linalg.norm(v)
    u3 = np.cross(u1, u2)
    w = np.array([np.linalg.norm(n), np.linalg.norm(v)])
    w = w / np.linalg.norm(w)
    return np.array([np.arctan2(u2[2], u3[2]), np.arcsin(-u1[2]), np.arctan2(u1[1], u1[0]), np.arcsin(w[1])])


def plk_to_pose(plk_w, Rcw, tcw):
    n, v = plk_w[:3], plk_w[3:]
    return np.concatenate([Rcw @ n + skew(tcw) @ (Rcw @ v), Rcw @ v])


def plk_from_pose(plk_c, Rcw, tcw):
    Rwc = Rcw.T
    return plk_to_pose(plk_c, Rwc, -Rwc @ tcw)


def pi_from_ppp(x1, x2, x3):
    return np.concatenate([np.cross(x1 - x3, x2 - x3), [-x3 @ np.cross(x1, x2)]])


def pipi_plk(pi1, pi2):
    dp = np.outer(pi1, pi2) - np.outer(pi2, pi1)
    return np.array([dp[0, 3], dp[1, 3], dp[2, 3], -dp[1, 2], dp[0, 2], -dp[0, 1]])


def line_trimming(plk, obs):
    """LineTrimming: 3-D end points (camera frame) of the line under the two observed image end points -> (valid, p1, p2)"""
    nc, vc = plk[:3], plk[3:]
    Lc = np.zeros((4, 4)); Lc[:3, :3] = skew(nc); Lc[:3, 3] = vc; Lc[3, :3] = -vc
    p11, p21 = np.array([obs[0], obs[1], 1.0]), np.array([obs[2], obs[3], 1.0])
    ln = np.cross(p11, p21)[:2]
    ln = ln / np.linalg.norm(ln)
    p12, p22 = np.array([p11[0] + ln[0], p11[1] + ln[1], 1.0]), np.array([p21[0] + ln[0], p21[1] + ln[1], 1.0])
    cam = np.zeros(3)
    e1, e2 = Lc @ pi_from_ppp(cam, p11, p12), Lc @ pi_from_ppp(cam, p21, p22)
    e1, e2 = e1 / e1[3], e2 / e2[3]
    return bool(e1[2] >= 0 and e2[2] >= 0), e1[:3], e2[:3]


def line_reprojection_error(obs, Rwc, twc, line_w):
    nc = plk_from_pose(line_w, Rwc, twc)[:3]
    nc = nc / np.hypot(nc[0], nc[1])
    return (abs(nc @ [obs[0], obs[1], 1.0]) + abs(nc @ [obs[2], obs[3], 1.0])) / 2.0


def triangulate_one_line(obs_list, start_frame, Rs, Ps, ric, tic):
    """TriangulateOneLine: obs_list[k] observed in frame start_frame + k; Rs/Ps body poses; ric/tic camera-0 extrinsics.
    -> None (parallax below the threshold / invalid / longer than 10 m) or dict(plk (camera frame of start_frame), ptw1, ptw2)"""
    i = start_frame
    t0, R0 = Ps[i] + Rs[i] @ tic, Rs[i] @ ric
    min_cos, best = 1.0, None
    pii = ni = None
    for k, obs in enumerate(obs_list):
        j = i + k
        if k == 0:
            pii = pi_from_ppp(np.array([obs[0], obs[1], 1.0]), np.array([obs[2], obs[3], 1.0]), np.zeros(3))
            ni = pii[:3] / np.linalg.norm(pii[:3])
            continue
        t1, R1 = Ps[j] + Rs[j] @ tic, Rs[j] @ ric
        t, R = R0.T @ (t1 - t0), R0.T @ R1
        p3, p4 = R @ np.array([obs[0], obs[1], 1.0]) + t, R @ np.array([obs[2], obs[3], 1.0]) + t
        nj = pi_from_ppp(p3, p4, t)[:3]
        nj = nj / np.linalg.norm(nj)
        c = ni @ nj
        if c < min_cos:
            min_cos, best = c, (t, R, obs)
    if min_cos > 0.998 or best is None:
        return None
    t, R, obs = best
    p3, p4 = R @ np.array([obs[0], obs[1], 1.0]) + t, R @ np.array([obs[2], obs[3], 1.0]) + t
    plk = pipi_plk(pii, pi_from_ppp(p3, p4, t))
    valid, e1, e2 = line_trimming(plk, obs_list[0])
    if not valid or np.linalg.norm(e1 - e2) > 10.0:
        return None
    to_w = lambda p: Rs[i] @ (ric @ p + tic) + Ps[i]
    return dict(plk=plk, ptw1=to_w(e1), ptw2=to_w(e2))
