"""On-disk formats on either side of the path, so that fixtures can be replayed across machines:

* point-feature frames, one line per feature id (utils/io/feature_serialization.cpp:26-75):
      "<0|1> id x y z u v vx vy [x y z u v vx vy]"      leading flag 1 = stereo observation present
  numbers are written in shortest round-trip form (fmt's "{}"), read back with float();
* trajectory, one line per processed frame (utils/io/output.cpp:189-227, SaveBodyTrajectory):
      "<sec>.<nsec, 9 digits> px py pz qx qy qz qw"       fixed notation, 6 decimals  (TUM format -> evaluate_ate.py)
* the reading side of the ATE metric (scripts/tum_tools/associate.py:49-100, evaluate_ate.py:130-155): `read_file_list` (stamp -> data tokens, comments / commas /
  tabs tolerated), `associate` (greedy nearest-stamp pairing within max_difference) and `evaluate_ate` (association -> Horn alignment -> RMSE).  Pinned against the
  reference's own functions by tests/golden/ate_associate.npz.
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


def read_file_list(path_or_text, is_text=False):
    """associate.py:49-68: {stamp: [tokens]} of a "stamp d1 d2 ..." file; commas and tabs count as blanks, lines starting with '#' and lines with a single token
    are dropped, a repeated stamp keeps its LAST line (dict construction)."""
    data = path_or_text if is_text else open(path_or_text).read()
    lines = data.replace(",", " ").replace("\t", " ").split("\n")
    rows = [[v.strip() for v in line.split(" ") if v.strip() != ""] for line in lines if len(line) > 0 and line[0] != "#"]
    return dict((float(r[0]), r[1:]) for r in rows if len(r) > 1)


def associate(first_stamps, second_stamps, offset=0.0, max_difference=0.02):
    """associate.py:70-100: every pair (a, b) with |a - (b + offset)| < max_difference is a candidate; candidates are taken in ascending (difference, a, b) order,
    each stamp at most once; the matches come back sorted by (a, b).  -> list of (a, b).  (The reference's O(n1 n2) candidate list is built the same way:
    the tie-breaking IS the tuple order, so nothing smarter is substituted.)"""
    first = [float(a) for a in first_stamps]
    second = [float(b) for b in second_stamps]
    cand = [(abs(a - (b + offset)), a, b) for a in first for b in second if abs(a - (b + offset)) < max_difference]
    cand.sort()
    fa, sb = set(first), set(second)
    matches = []
    for _, a, b in cand:
        if a in fa and b in sb:
            fa.remove(a); sb.remove(b)
            matches.append((a, b))
    matches.sort()
    return matches


def evaluate_ate(gt_stamps, gt_xyz, est_stamps, est_xyz, offset=0.0, max_difference=0.02, scale=1.0):
    """evaluate_ate.py:130-155 on arrays: associate(ground truth, estimate) -> align(estimate, ground truth) (Horn, no scale) -> RMSE of the translational error.
    -> (rmse, n_pairs).  Raises like the script exits when fewer than two stamps pair up."""
    from .sim import align_ate
    gi = {float(t): i for i, t in enumerate(gt_stamps)}
    ei = {float(t): i for i, t in enumerate(est_stamps)}
    m = associate(list(gi), list(ei), offset, max_difference)
    if len(m) < 2:
        raise ValueError("Couldn't find matching timestamp pairs between groundtruth and estimated trajectory!")
    g = np.asarray(gt_xyz, float)[[gi[a] for a, _ in m], :3]
    e = np.asarray(est_xyz, float)[[ei[b] for _, b in m], :3] * float(scale)
    return align_ate(e, g)[0], len(m)


def evaluate_ate_files(gt_path, est_path, offset=0.0, max_difference=0.02, scale=1.0):
    """the script's command line: two TUM files -> (rmse, n_pairs)"""
    g, e = read_file_list(gt_path), read_file_list(est_path)
    gs, es = sorted(g), sorted(e)
    return evaluate_ate(gs, [[float(v) for v in g[t][0:3]] for t in gs], es, [[float(v) for v in e[t][0:3]] for t in es], offset, max_difference, scale)
