"""Row N3: the feature / trajectory text formats on either side of the path (round trips, the reference's line layout)."""
import numpy as np

from dynamic_vins_amd import io_formats as F
from dynamic_vins_amd.frontend import FEAT_DTYPE


def test_point_feature_lines_layout_and_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    pts = {}
    for fid in (7, 3, 1000021):
        l = np.concatenate([rng.normal(0, 0.3, 2), [1.0], rng.uniform(0, 1280, 2), rng.normal(0, 0.1, 2)])
        pts[fid] = [(0, l)]
        if fid != 3:
            pts[fid].append((1, l + 1e-3))
    pts[3][0][1][5:] = 0.0                                 # a new feature: zero velocity
    lines = F.point_feature_lines(pts)
    assert [ln.split(" ")[1] for ln in lines] == ["3", "7", "1000021"]            # std::map order
    assert lines[0].startswith("0 3 ") and len(lines[0].split(" ")) == 9 and lines[0].split(" ")[4] == "1" and lines[0].endswith(" 0 0")
    assert lines[1].startswith("1 7 ") and len(lines[1].split(" ")) == 16
    p = tmp_path / "feat.txt"
    F.serialize_point_features(p, pts)
    back = F.deserialize_point_features(p)
    assert sorted(back) == sorted(pts)
    for fid in pts:
        assert len(back[fid]) == len(pts[fid])
        for (c0, v0), (c1, v1) in zip(pts[fid], back[fid]):
            assert c0 == c1 and np.array_equal(v0, v1)     # shortest round-trip digits: bit-exact


def test_rows_points_round_trip():
    rows = np.zeros(3, FEAT_DTYPE)
    rows["id"] = [5, 2, 9]; rows["has_right"] = [1, 0, 1]
    rows["left"] = np.arange(21).reshape(3, 7) * 0.5; rows["right"] = np.arange(21).reshape(3, 7) * 0.25
    pts = F.rows_to_points(rows)
    assert len(pts[2]) == 1 and len(pts[5]) == 2
    back = F.points_to_rows(pts, FEAT_DTYPE)
    assert list(back["id"]) == [2, 5, 9] and np.array_equal(back["left"][1], rows["left"][0]) and np.array_equal(back["right"][2], rows["right"][2])


def test_trajectory_line_is_tum_format(tmp_path):
    ln = F.trajectory_line(1403636579.763555992, [1.0, -2.5, 0.125, 0.0, 0.0, 0.70710678, 0.70710678])
    t = ln.split(" ")
    assert t[0].startswith("1403636579.7635") and len(t[0].split(".")[1]) == 9 and t[1:] == ["1.000000", "-2.500000", "0.125000", "0.000000", "0.000000", "0.707107", "0.707107"]
    assert F.trajectory_line(2.9999999996, [0] * 7).startswith("3.000000000 ")           # nsec carry
    p = tmp_path / "traj.txt"
    p.write_text("\n".join(F.trajectory_line(1.0 + 0.05 * k, [k, 0, 0, 0, 0, 0, 1]) for k in range(4)) + "\n")
    st, poses = F.read_trajectory(p)
    assert np.allclose(st, [1.0, 1.05, 1.1, 1.15]) and np.array_equal(poses[:, 0], [0, 1, 2, 3])


def test_associate_and_read_file_list_match_the_reference_functions():
    """REFERENCE-PINNED (tests/golden/ate_associate.npz, written by gen_golden.py from scripts/tum_tools/associate.py:49-100): the nearest-stamp association the
    ATE metric starts with — jittered subsets, an independent clock with unmatched stamps, dense lists with ties, offsets and several search radii"""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ate_associate.npz"))
    total = 0
    for k in range(int(g["n_cases"])):
        off, md = g[f"par{k}"]
        m = F.associate(g[f"a{k}"], g[f"b{k}"], off, md)
        want = g[f"m{k}"]
        assert np.array_equal(np.array(m, np.float64).reshape(-1, 2), want), k
        total += len(m)
    assert total > 400
    lst = F.read_file_list(bytes(g["rfl_text"]).decode(), is_text=True)
    keys = sorted(lst)
    assert np.array_equal(np.array(keys), g["rfl_stamps"])
    assert np.array_equal(np.array([[float(v) for v in lst[t][0:3]] for t in keys]), g["rfl_xyz"])


def test_evaluate_ate_files_is_association_then_the_pinned_alignment(tmp_path):
    """the whole metric on two TUM files written by the trajectory writer: an estimate that is a rigidly moved, jittered, partly missing copy of the ground truth"""
    from dynamic_vins_amd import sim
    rng = np.random.default_rng(5)
    t = 1403636579.0 + 0.05 * np.arange(120)
    gt = np.stack([4 * np.sin(0.05 * np.arange(120)), 2.5 * np.sin(0.1 * np.arange(120)), 0.25 * np.sin(0.07 * np.arange(120))], 1)
    th = 0.3
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    keep = np.sort(rng.choice(120, 90, replace=False))
    est = (gt[keep] @ R.T + [1.0, -2.0, 0.5]) + rng.normal(0, 0.01, (90, 3))
    te = t[keep] + rng.normal(0, 0.004, 90)
    a, b = tmp_path / "gt.txt", tmp_path / "est.txt"
    a.write_text("# ground truth\n" + "\n".join(F.trajectory_line(tt, list(p) + [0, 0, 0, 1]) for tt, p in zip(t, gt)) + "\n")
    b.write_text("\n".join(F.trajectory_line(tt, list(p) + [0, 0, 0, 1]) for tt, p in zip(te, est)) + "\n")
    rmse, n = F.evaluate_ate_files(a, b)
    assert n == 90 and 0.005 < rmse < 0.03
    # same pairs by hand -> the alignment pinned by ate_align.npz
    gl, el = F.read_file_list(a), F.read_file_list(b)
    m = F.associate(list(gl), list(el))
    g3 = np.array([[float(v) for v in gl[x][:3]] for x, _ in m]); e3 = np.array([[float(v) for v in el[y][:3]] for _, y in m])
    assert abs(sim.align_ate(e3, g3)[0] - rmse) < 1e-15
