#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run in the authoring container, where /root/reference exists):

    python tests/golden/gen_golden.py

1. ate_align.npz   — REFERENCE-PINNED: outputs of the reference's own `align()` (Horn alignment; the ATE metric
                     of BASELINE.json) imported from dynamic_vins/scripts/tum_tools/evaluate_ate.py:47-79 and run
                     here on seeded trajectories.  (The file as a whole is Python 2 — print statements — so the
                     function is loaded from its source text at generation time; nothing of it is stored here.)
2. front_kat.npz   — known-answer vectors of the front-end path produced by the CPU oracle (oracle/) on seeded
                     128x96 images: pyramid, Scharr, min-eigenvalue map, corners, LK tracks, masks, lifted points,
                     and a 6-frame stereo TrackImage sequence.  The third-party arithmetic (OpenCV 3.4.16) is not
                     under /root/reference and the reference has no tests that pin it: **parity unpinned** — these
                     vectors pin the ORACLE (against compiler/flag drift) and let the HIP path be checked on the
                     GPU box against committed data; tests/test_oracle_checks.py adds independent numpy/scipy
                     restatements and the size-independent properties.
3. back_kat.npz    — known-answer vectors of the back-end path from the oracle: factor residuals/Jacobians,
                     pre-integration of a fixed IMU sequence, one standalone window solve, one marginalization.
1b. ate_associate.npz — REFERENCE-PINNED: outputs of the reference's own `associate()` (nearest-stamp matching of two TUM trajectories within
                     max_difference, dynamic_vins/scripts/tum_tools/associate.py:70-100) and `read_file_list()` (:49-68) on seeded stamp lists / a small TUM
                     text; the functions are Python-2 era (they .remove() from dict.keys()), so they are run from their source text on dict objects whose
                     keys() returns a list, as it did when the file was written.
4. aux_kat.npz     — the same for cv::remap with fixed-point maps, the object solve and the line-only solve (oracle outputs; parity unpinned).
Only data is written: inputs and expected outputs.
"""
import ctypes as C
import os
import re
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

REF_ATE = "/root/reference/dynamic_vins/scripts/tum_tools/evaluate_ate.py"


def gen_ate():
    src = open(REF_ATE).read()
    fn = re.search(r"^def align\(.*?(?=^def )", src, flags=re.S | re.M).group(0)
    # the function text comes from the untrusted reference tree: it is compiled only after its AST has been checked to be one function definition that
    # calls nothing but numpy attributes and a few arithmetic builtins, and it runs with an empty __builtins__ (no import, open, eval ...)
    import ast
    tree = ast.parse(fn)
    assert len(tree.body) == 1 and isinstance(tree.body[0], ast.FunctionDef) and tree.body[0].name == "align"
    for node in ast.walk(tree):
        assert not isinstance(node, (ast.Import, ast.ImportFrom, ast.Global, ast.Nonlocal, ast.Lambda, ast.ClassDef, ast.With, ast.Try)), type(node).__name__
        if isinstance(node, ast.Attribute):
            assert not node.attr.startswith("_"), node.attr
        if isinstance(node, ast.Name):
            assert node.id in {"numpy", "align", "model", "data", "model_zerocentered", "data_zerocentered", "W", "U", "d", "Vh", "S", "rot", "trans", "model_aligned",
                               "alignment_error", "trans_error", "column", "range", "sum"} or not node.id.startswith("_"), node.id
    ns = {"numpy": np, "__builtins__": {"range": range, "sum": sum}}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exec(compile(tree, "<reference align()>", "exec"), ns)
        rng = np.random.default_rng(0xA7E)
        model, data, rot, trans, rmse = [], [], [], [], []
        for case in range(8):
            n = 40 + 10 * case
            t = np.linspace(0, 6.0, n)
            gt = np.stack([4 * np.sin(0.5 * t), 2.5 * np.sin(t), 0.25 * np.sin(0.7 * t)])
            a = rng.normal(0, 1, 3)
            th = np.linalg.norm(a)
            K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]]) / th
            R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
            if case == 5:
                gt[2] = 0.0            # planar trajectory: exercises the det(U)det(V) < 0 branch candidates
            est = R @ gt + rng.normal(0, 1, (3, 1)) + rng.normal(0, 0.01 * (1 + case), gt.shape)
            r, tr, err = ns["align"](np.matrix(est), np.matrix(gt))
            model.append(np.pad(est, ((0, 0), (0, 120 - n))))
            data.append(np.pad(gt, ((0, 0), (0, 120 - n))))
            rot.append(np.asarray(r)); trans.append(np.asarray(tr)[:, 0])
            rmse.append(float(np.sqrt(np.dot(err, err) / len(err))))
    np.savez_compressed(os.path.join(HERE, "ate_align.npz"), model=np.array(model), data=np.array(data), n=np.array([40 + 10 * c for c in range(8)]),
                        rot=np.array(rot), trans=np.array(trans), rmse=np.array(rmse))
    print("ate_align.npz: rmse", np.round(rmse, 5))


REF_ASSOC = "/root/reference/dynamic_vins/scripts/tum_tools/associate.py"


def gen_associate():
    import ast
    src = open(REF_ASSOC).read()
    ns = {"__builtins__": {"abs": abs, "float": float, "dict": dict, "len": len, "open": None}}
    for name in ("read_file_list", "associate"):
        fn = re.search(r"^def %s\(.*?(?=^def |^if __name__)" % name, src, flags=re.S | re.M).group(0)
        tree = ast.parse(fn)
        assert len(tree.body) == 1 and isinstance(tree.body[0], ast.FunctionDef) and tree.body[0].name == name
        for node in ast.walk(tree):      # untrusted text: one plain function, no imports / dunder access / nested definitions
            assert not isinstance(node, (ast.Import, ast.ImportFrom, ast.Global, ast.Nonlocal, ast.Lambda, ast.ClassDef, ast.With, ast.Try)), type(node).__name__
            if isinstance(node, ast.Attribute):
                assert not node.attr.startswith("_"), node.attr
            if isinstance(node, ast.Name):
                assert not node.id.startswith("_"), node.id
        exec(compile(tree, "<reference %s()>" % name, "exec"), ns)

    class Py2Dict(dict):                   # dict.keys() of the interpreter the file was written for: a list (the function removes matched stamps from it)
        def keys(self):
            return list(dict.keys(self))
    rng = np.random.default_rng(0xA550C)
    out = {}
    cases = []
    for case in range(10):
        n1, n2 = 30 + 7 * case, 25 + 9 * case
        a = np.cumsum(rng.uniform(0.03, 0.07, n1)) + 1403636579.0
        if case % 3 == 0:                  # second list = jittered subset of the first (the usual ground truth / estimate pairing)
            b = np.sort(rng.choice(a, size=min(n2, n1), replace=False)) + rng.normal(0, 0.006, min(n2, n1))
        elif case % 3 == 1:                # independent clock: many stamps without a partner, some with two candidates
            b = np.cumsum(rng.uniform(0.02, 0.09, n2)) + 1403636579.0 + 0.011
        else:                              # dense second list: several candidates per stamp, ties in |diff| broken by the tuple order (diff, a, b)
            b = np.round(np.cumsum(rng.uniform(0.004, 0.012, 4 * n2)) + 1403636579.0, 3)
            a = np.round(a, 3)
        offset = [0.0, 0.013, -0.02][case % 3]
        maxd = [0.02, 0.02, 0.01, 0.05][case % 4]
        first = Py2Dict((float(t), ["%d" % i]) for i, t in enumerate(a))
        second = Py2Dict((float(t), ["%d" % i]) for i, t in enumerate(b))
        m = ns["associate"](first, second, offset, maxd)
        cases.append((np.array(sorted(first.keys())), np.array(sorted(second.keys())), offset, maxd, np.array(m, np.float64).reshape(-1, 2)))
    for k, (a, b, off, md, m) in enumerate(cases):
        out[f"a{k}"], out[f"b{k}"], out[f"par{k}"], out[f"m{k}"] = a, b, np.array([off, md]), m
    out["n_cases"] = np.array(len(cases))
    # read_file_list on a TUM text with comments, commas, tabs, blank lines and a one-token line (dropped): stamps and the first three data columns
    text = ("# timestamp tx ty tz qx qy qz qw\n1403636579.763555992 1.0 -2.5 0.125 0 0 0.7 0.7\n\n1403636579.813555992,4.0,5.5,6.25,0,0,0,1\n"
            "1403636579.863555992\t7 8 9 0 0 0 1\n12345\n  1403636579.913555992   10 11 12   0 0 0 1\n")
    import io as _io
    ns["__builtins__"]["open"] = lambda fn: _io.StringIO(text)
    lst = ns["read_file_list"]("mem")
    keys = sorted(lst.keys())
    out["rfl_text"] = np.frombuffer(text.encode(), np.uint8)
    out["rfl_stamps"] = np.array(keys)
    out["rfl_xyz"] = np.array([[float(v) for v in lst[k][0:3]] for k in keys])
    np.savez_compressed(os.path.join(HERE, "ate_associate.npz"), **out)
    print("ate_associate.npz: matches per case", [len(c[4]) for c in cases], "read_file_list rows", len(keys))


def gen_front(o):
    from dynamic_vins_amd import sim, synth
    W, H = 128, 96
    seq = synth.PlaneSequence(W, H, seed=0xD1CE, disparity=3.25, margin=48)
    frames = [seq.frame(k) for k in range(6)]
    img0, img1 = frames[0][0], frames[1][0]
    out = dict(left=np.array([f[0] for f in frames]), right=np.array([f[1] for f in frames]))
    out["pyr1"] = o.pyr_down(img0)
    out["pyr2"] = o.pyr_down(out["pyr1"])
    out["scharr"] = o.scharr(img0)
    out["min_eigen"] = o.min_eigen(img0)
    mask = np.full((H, W), 255, np.uint8)
    mask[:, :12] = 0
    out["gftt_mask"] = mask
    out["corners"] = o.gftt(img0, 40, 0.01, 8, mask)
    out["corners_nomask"] = o.gftt(img0, 25, 0.01, 12, None)
    p2, st = o.lk(img0, img1, out["corners"], 3, 30, 0.01)
    out["lk_pts"], out["lk_status"] = p2, st
    p3, st3 = o.track_by_lk(img0, img1, out["corners"], True, 0.5)
    out["tbl_pts"], out["tbl_status"] = p3, st3
    out["circle"] = o.circle_mask(np.full((H, W), 255, np.uint8), np.array([[20.3, 30.7], [100.0, 5.0], [127.0, 95.0], [64.5, 48.5]], np.float32), 9)
    m = np.full((H, W), 255, np.uint8)
    m[30:50, 40:70] = 0
    m[80, 100] = 0
    out["erode_in"], out["erode5"] = m, o.erode(m, 5)
    cam = sim.scaled_cam(sim.ZED, W, H, 1280, 720)
    out["cam"] = np.array(sim.cam_tuple(cam))
    pts = np.array([[0, 0], [127, 95], [64, 48], [10.5, 80.25], [100.75, 3.5]], np.float32)
    out["lift_in"], out["lift_out"] = pts, o.lift_projective(sim.cam_tuple(cam), pts)
    trk = o.tracker(W, H, 30, 10, 1, 1, sim.cam_tuple(cam), sim.cam_tuple(cam))
    rows = []
    for k, (l, r) in enumerate(frames):
        rw = trk.track_image(l, r, 1.0 + 0.05 * k)
        pad = np.zeros(64, rw.dtype)
        pad[: len(rw)] = rw
        rows.append(pad)
        out.setdefault("track_n", []).append(len(rw))
    trk.close()
    out["track_rows"] = np.array([r.view(np.uint8).reshape(64, 128) for r in rows])
    out["track_n"] = np.array(out["track_n"])
    np.savez_compressed(os.path.join(HERE, "front_kat.npz"), **out)
    print("front_kat.npz: corners", len(out["corners"]), "lk ok", int(out["lk_status"].sum()), "tracked rows", out["track_n"])


def gen_back(o):
    import ba_gen
    lib = o.lib
    rng = np.random.default_rng(0xBA)
    out = {}
    # --- projection factors (B3-B5) ---
    n = 24
    lib.dvo_proj_eval.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]

    def rand_pose():
        q = rng.normal(0, 1, 4)
        return np.concatenate([rng.normal(0, 2, 3), q / np.linalg.norm(q)])
    obs = np.zeros((n, 12)); kinds = np.arange(n) % 3
    par = np.zeros((n, 30)); res = np.zeros((n, 2)); jac = np.zeros((n, 2 * 30))
    for k in range(n):
        pi = rand_pose(); pj = pi.copy(); pj[:3] += rng.normal(0, 0.3, 3)
        e0 = np.concatenate([rng.normal(0, 0.05, 3), np.array([0.5, -0.5, 0.5, -0.5]) + rng.normal(0, 0.01, 4)]); e0[3:] /= np.linalg.norm(e0[3:])
        e1 = e0.copy(); e1[:3] += [0, -0.12, 0]
        lam, td = rng.uniform(0.05, 0.8), rng.normal(0, 0.01)
        obs[k] = [*rng.uniform(-0.5, 0.5, 2), 1.0, *rng.uniform(-0.5, 0.5, 2), 1.0, *rng.normal(0, 0.2, 4), *rng.normal(0, 0.01, 2)]
        par[k] = np.concatenate([pi, pj, e0, e1, [lam, td]])
        blocks = {0: [pi, pj, e0, [lam], [td]], 1: [pi, pj, e0, e1, [lam], [td]], 2: [e0, e1, [lam], [td]]}[int(kinds[k])]
        blocks = [np.ascontiguousarray(b, np.float64) for b in blocks]
        J = [np.zeros(2 * len(b)) for b in blocks]
        pp = (C.c_void_p * len(blocks))(*[b.ctypes.data for b in blocks]); Jp = (C.c_void_p * len(blocks))(*[j.ctypes.data for j in J])
        r = np.zeros(2)
        lib.dvo_proj_eval(int(kinds[k]), obs[k].ctypes.data, pp, r.ctypes.data, Jp)
        res[k] = r
        flat = np.concatenate(J)
        jac[k, : len(flat)] = flat
    out.update(proj_obs=obs, proj_kind=kinds, proj_par=par, proj_res=res, proj_jac=jac)
    # --- one window: solve + marginalization (B2, B6-B8) ---
    prob = ba_gen.make_window(o, seed=21, nlm=60, with_prior=True, max_iters=6)
    out.update(win_pose=prob.pose.copy(), win_sb=prob.speed_bias.copy(), win_depth=prob.inv_depth.copy(), win_factors=prob.factors.view(np.uint8).copy(),
               win_landmarks=prob.landmarks.view(np.uint8).copy(), win_imu=prob.imu.view(np.uint8).copy(), win_ex=prob.ex_pose.copy(),
               win_prior=np.frombuffer(bytes(prob.prior), np.uint8).copy(), win_priorA=prob.prior_A.copy(), win_priorb=prob.prior_b.copy())
    ref = prob.clone()
    s = ba_gen.oracle_solve(o, ref)
    out.update(sol_pose=ref.pose.copy(), sol_sb=ref.speed_bias.copy(), sol_depth=ref.inv_depth.copy(),
               sol_summary=np.array([s.iterations, s.termination, s.initial_cost, s.final_cost]))
    for mode in (0, 1):
        sub = ba_gen.marg_subproblem(prob, mode)
        pr, A, b = ba_gen.oracle_marginalize(o, sub, mode)
        out[f"marg{mode}_A"], out[f"marg{mode}_b"], out[f"marg{mode}_c0"] = A, b, np.array([pr.c0])
        out[f"marg{mode}_blocks"] = np.array([[pr.blocks[i].type, pr.blocks[i].idx, pr.blocks[i].off, pr.blocks[i].size_local] for i in range(pr.nblocks)])
        out[f"marg{mode}_x0"] = np.array([[pr.x0[i][j] for j in range(9)] for i in range(pr.nblocks)])
    np.savez_compressed(os.path.join(HERE, "back_kat.npz"), **out)
    print("back_kat.npz: solve", out["sol_summary"], "marg n", out["marg0_A"].shape, out["marg1_A"].shape)


def gen_aux(o):
    """4. aux_kat.npz — known-answer vectors of the rows added after the core path: cv::remap with fixed-point undistortion maps (1 and 3
    channels, incl. out-of-range coordinates), the per-frame object solve and the line-only solve (inputs, solved parameters, summaries)."""
    import obj_gen as G
    rng = np.random.default_rng(11)
    w, h = 96, 64
    cam = (70.0, 71.0, 48.2, 30.4, -0.3, 0.08, 1e-3, -1e-3)
    m1, m2 = o.init_undistort_map(cam, (52.0, 53.0, 47.0, 33.0), w, h)          # a zoom-out: the border rows / columns leave the source
    gray = rng.integers(0, 256, (h, w), dtype=np.uint8)
    bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    out = dict(remap_map1=m1, remap_map2=m2, remap_gray=gray, remap_bgr=bgr, remap_gray_out=o.remap(gray, m1, m2), remap_bgr_out=o.remap(bgr, m1, m2),
               remap_fused_gray=o.bgr2gray(o.remap(bgr, m1, m2)))
    for name, kw in (("obj_a", dict(seed=2, n_obj=3, pts_per_obj=0, max_iters=10)), ("obj_b", dict(seed=5, n_obj=6, outside=0.0, pose_noise=(0.02, 0.3), max_iters=12))):
        p = G.make_obj_scene(**kw)
        for k in ("state", "dims", "body_pose", "R_bc", "boxes", "points"):
            out[name + "_" + k] = getattr(p, k).copy()
        out[name + "_opts"] = np.array([p.max_iters, p.plane_kind])
        s = G.o_obj_solve(o.lib, p)
        out[name + "_state_out"], out[name + "_dims_out"] = p.state.copy(), p.dims.copy()
        out[name + "_summary"] = np.array([s.iterations, s.successful, s.termination, s.initial_cost, s.final_cost])
    p = G.make_line_scene(seed=1, max_iters=5)
    for k in ("orth", "pose", "ex_pose", "sqrt_info", "obs"):
        out["line_" + k] = getattr(p, k).copy()
    s = G.o_line_solve(o.lib, p)
    out["line_orth_out"] = p.orth.copy()
    out["line_summary"] = np.array([s.iterations, s.successful, s.termination, s.initial_cost, s.final_cost])
    np.savez_compressed(os.path.join(HERE, "aux_kat.npz"), **out)
    print("aux_kat.npz: obj", out["obj_a_summary"], out["obj_b_summary"], "line", out["line_summary"])


def gen_gftt_cuda(o):
    """5. gftt_cuda_kat.npz — the reference's GPU corner detector (row F5: cv::cuda::GoodFeaturesToTrackDetector as TrackImageNaive calls it, oracle/gftt_cuda.cpp):
    response map, corners without / with a mask whose excluded region holds the strongest response (the threshold is NOT taken under the mask), and six frames of
    TrackImageNaive rows (GPU tracker + GPU detector)."""
    from dynamic_vins_amd import sim, synth
    W, H = 128, 96
    seq = synth.PlaneSequence(W, H, seed=0xD1CE, disparity=3.25, margin=48)
    frames = [seq.frame(k) for k in range(6)]
    img = frames[0][0].copy()
    img[60:76, 90:106] = 0
    img[68:76, 98:106] = 255                                     # a full-contrast corner: the strongest response of the frame by a wide margin
    mask = np.full((H, W), 255, np.uint8)
    mask[52:84, 82:114] = 0                                       # ... excluded by the mask
    out = dict(img=img, mask=mask, left=np.array([f[0] for f in frames]), right=np.array([f[1] for f in frames]))
    out["min_eigen"] = o.min_eigen(img, rule="cuda")
    out["corners_nomask"] = o.gftt(img, 40, 0.01, 8, None, rule="cuda")
    out["corners_mask"] = o.gftt(img, 1000, 0.01, 3, mask, rule="cuda")          # 383 corners: 1 % of the response of the excluded corner
    out["corners_mask_cpu_rule"] = o.gftt(img, 1000, 0.01, 3, mask)                  # 459: 1 % of the strongest response under the mask
    cam = sim.scaled_cam(sim.ZED, W, H, 1280, 720)
    out["cam"] = np.array(sim.cam_tuple(cam))
    trk = o.tracker(W, H, 30, 10, 1, 1, sim.cam_tuple(cam), sim.cam_tuple(cam))
    tmask = np.full((H, W), 255, np.uint8)
    tmask[30:60, 50:90] = 0
    out["track_mask"] = tmask
    rows = []
    for k, (l, r) in enumerate(frames):
        rw = trk.track_image(l, r, 1.0 + 0.05 * k, mask=tmask, mode=1)
        pad = np.zeros(64, rw.dtype)
        pad[: len(rw)] = rw
        rows.append(pad)
        out.setdefault("track_n", []).append(len(rw))
    trk.close()
    out["track_rows"] = np.array([r.view(np.uint8).reshape(64, 128) for r in rows])
    out["track_n"] = np.array(out["track_n"])
    np.savez_compressed(os.path.join(HERE, "gftt_cuda_kat.npz"), **out)
    print("gftt_cuda_kat.npz: corners", len(out["corners_nomask"]), len(out["corners_mask"]), "cpu rule under the same mask", len(out["corners_mask_cpu_rule"]), "naive rows", out["track_n"])


if __name__ == "__main__":
    import oracle_py
    which = sys.argv[1:] or ["ate", "front", "back", "aux", "gftt_cuda"]
    if "ate" in which:
        gen_ate()
        gen_associate()
    o = oracle_py.load()
    if "front" in which:
        gen_front(o)
    if "back" in which:
        gen_back(o)
    if "aux" in which:
        gen_aux(o)
    if "gftt_cuda" in which:
        gen_gftt_cuda(o)
