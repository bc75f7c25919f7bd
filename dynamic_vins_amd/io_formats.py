"""On-disk formats on either side of the path, so that fixtures can be replayed across machines:

* point-feature frames, one line per feature id (utils/io/feature_serialization.cpp:26-75):
      "<0|1> id x y z u v vx vy [x y z u v vx vy]"      leading flag 1 = stereo observation present
  numbers are written in shortest round-trip form (fmt's "{}"), read back with float();
* trajectory, one line per processed frame (utils/io/output.cpp:189-227, SaveBodyTrajectory):
      "<sec>.<nsec, 9 digits> px py pz qx qy qz qw"       fixed notation, 6 decimals  (TUM format -> evaluate_ate.py)
"""
import numpy as np


def _num(v):
    s = repr(float(v))
    return s[:-2] if s.endswith(".0") else s          # fmt prints 1.0 as "1"


def point_feature_lines(points):
    """points: {id: [(0, vec7), (1, vec7)?]} (FeatureBackground.points) -> list of text lines, ids ascending (std::map order)"""
    out = []
    for fid in sorted(points):
        obs = points[fid]
        body = " ".join(_num(x) for x in obs[0][1])
        if len(obs) == 1:
            out.append(f"0 {fid} {body}")
        else:
            out.append(f"1 {fid} {body} " + " ".join(_num(x) for x in obs[1][1]))
    return out


def serialize_point_features(path, points):
    with open(path, "w") as f:
        for ln in point_feature_lines(points):
            f.write(ln + "\n")


def deserialize_point_features(path):
    pts = {}
    with open(path) as f:
        for line in f:
            t = line.split(" ")
            if len(t) < 9:
                continue
            fid = int(t[1])
            pts.setdefault(fid, []).append((0, np.array([float(x) for x in t[2:9]])))
            if t[0] == "1":
                pts[fid].append((1, np.array([float(x) for x in t[9:16]])))
    return pts


def rows_to_points(rows):
    """dv_feat rows (frontend.FEAT_DTYPE) -> FeatureBackground.points"""
    pts = {}
    for r in rows:
        obs = [(0, np.array(r["left"]))]
        if r["has_right"]:
            obs.append((1, np.array(r["right"])))
        pts[int(r["id"])] = obs
    return pts


def points_to_rows(points, dtype):
    rows = np.zeros(len(points), dtype)
    for k, fid in enumerate(sorted(points)):
        rows[k]["id"], rows[k]["track_cnt"] = fid, 1
        for cam, v in points[fid]:
            if cam == 0:
                rows[k]["left"] = v
            else:
                rows[k]["right"], rows[k]["has_right"] = v, 1
    return rows


def trajectory_line(stamp, pose7):
    """stamp in seconds; pose7 = px py pz qx qy qz qw (body.Ps / Rs[kWinSize])"""
    sec = int(np.floor(stamp))
    nsec = int(round((stamp - sec) * 1e9))
    if nsec >= 1000000000:
        sec, nsec = sec + 1, nsec - 1000000000
    return f"{sec}.{nsec:09d} " + " ".join(f"{float(v):.6f}" for v in pose7)


def read_trajectory(path):
    """-> (stamps[n], poses[n, 7]) of a TUM-format file"""
    a = np.loadtxt(path, ndmin=2)
    return a[:, 0], a[:, 1:8]
