"""debug (open defect of round 4): where does a multi-threaded multi-sequence run first leave the single-thread run of the SAME sequences?
Reference: one host thread, no groups.  Candidate: --groups G (dv_batch groups) on --threads T.  Per member: the first frame whose record [t, pose, flag] differs, and at
that frame whether the rows handed over by the tracker already differ (front end) or only the solve does (back end); plus the batch's fallback-round count.
usage: multiseq_first_diff.py [S=16] [group_size=8] [threads=2] [frames=75] [repeats=4]      DBG_TEAMS=1: several host threads per group (dv_runner_set "teams"), threads = a multiple of the group count"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np, torch
from dynamic_vins_amd import sim, dist as dv_dist
from dynamic_vins_amd.backend import Runner
from dynamic_vins_amd.pipeline import Pipeline, SyntheticSequence
S, gsz, T, frames, reps = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 16), (2, 8), (3, 2), (4, 75), (5, 4)))
seqs = [SyntheticSequence(1280, 720, sim.ZED, frames + 2, rate=20.0, phase=dv_dist.sequence_phase(i), device="cuda:0") for i in range(S)]
KEYS = [k for k in os.environ.get("DBG_KEYS", "").split(",") if k]
def run(group_size, threads):
    pipes = [Pipeline(q, max_cnt=250, min_dist=25, max_iters=10, device=0) for q in seqs]
    for p in pipes:
        assert p.ctx.lib.dv_debug_set(p.ctx.h, b"hash_log", 1) == 0
        assert p.ctx.lib.dv_debug_set(p.ctx.h, b"hash_light", int(os.environ.get("DBG_LIGHT", "1"))) == 0
        for k in (KEYS if group_size else []):
            assert p.ctx.lib.dv_debug_set(p.ctx.h, k.encode(), 1) == 0, k
    r = Runner(pipes, group_size=group_size, threads=threads)
    if group_size and os.environ.get("DBG_TEAMS", "0") == "1":
        r.set("teams", 1)
    r.run(frames)
    def hlog(p):
        rows = np.zeros((4096, 6), dtype=np.uint64); n = C.c_int(0)
        p.ctx.lib.dv_est_debug_hash_log(p.ctx.h, rows.ctypes.data, 4096, C.byref(n))
        return rows[: n.value].copy()
    def dlog(p):
        rows = np.zeros((4096, 7), dtype=np.uint64); n = C.c_int(0)
        p.ctx.lib.dv_ba_debug_dev_log(p.ctx.h, rows.ctypes.data, 4096, C.byref(n))
        return rows[: n.value].copy()
    def slog(p):
        n = C.c_longlong(0); rl = C.c_int(0)
        p.ctx.lib.dv_ba_debug_slot_log(p.ctx.h, None, 0, C.byref(n), C.byref(rl))
        v = np.zeros(max(n.value, 1), dtype=np.uint64)
        p.ctx.lib.dv_ba_debug_slot_log(p.ctx.h, v.ctypes.data, n.value, C.byref(n), C.byref(rl))
        return v[: n.value].reshape(-1, 16, 5, 16)          # [fused solve][slot][launch kind][buffer]
    out = [(r.frames(i), r.row_log(i), hlog(pipes[i]), dlog(pipes[i]), slog(pipes[i])) for i in range(S)]
    info = r.batch_rounds()
    r.close()
    for p in pipes: p.ctx.close()
    return out, info
ref, _ = run(0, 1)
ref2, _ = run(0, 1)
print("single thread twice: identical" if all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) for a, b in zip(ref, ref2)) else "single thread twice: DIFFERENT (!)")
KIND = ["head_eval", "head_reduce", "solve", "cand_eval", "cand_reduce"]
BUF = ["packets0", "packets1", "imu_out0", "imu_out1", "prior_out0", "prior_out1", "cand_cost", "Hd0", "Hd1", "Sc0", "Sc1", "gvec0", "gvec1", "x", "cand", "ctl"]
DCOLS = ["counter", "DEV_uploaded_block", "DEV_priorA", "DEV_priorb", "DEV_x_after", "DEV_cand_after", "DEV_ctl_after"]
COLS = ["counter", "states_uploaded", "tables_uploaded", "states+flags_downloaded", "PRIOR_ON_DEVICE_at_start", "iterations"]
for rep in range(reps):
    got, info = run(gsz, T)
    bad = []
    for i in range(S):
        fa, la, ha, da, sa = ref[i]; fb, lb, hb, db, sbb = got[i]
        sd = None
        for q in range(min(len(sa), len(sbb))):
            if not np.array_equal(sa[q], sbb[q]):
                w = np.argwhere(sa[q] != sbb[q])
                it, kd = int(w[0][0]), int(w[0][1])
                sd = dict(fused_solve=q, slot=it, launch=KIND[kd], buffers=[BUF[int(b)] for (a_, k_, b) in w if a_ == it and k_ == kd])
                break
        nd = min(len(da), len(db))
        dd = next(((q, [DCOLS[c] for c in range(7) if da[q][c] != db[q][c]]) for q in range(nd) if not np.array_equal(da[q], db[q])), None)
        nh = min(len(ha), len(hb))
        hd = next(((q, [COLS[c] for c in range(6) if ha[q][c] != hb[q][c]]) for q in range(nh) if not np.array_equal(ha[q], hb[q])), None)
        n = min(len(fa), len(fb))
        d = [k for k in range(n) if not np.array_equal(fa[k], fb[k])]
        if d or len(fa) != len(fb):
            k = d[0] if d else n
            rows_same = k < min(len(la), len(lb)) and la[k][0] == lb[k][0] and la[k][1] == lb[k][1] and la[k][2] == lb[k][2]
            first_row_diff = next((q for q in range(min(len(la), len(lb))) if not np.array_equal(la[q][:3], lb[q][:3])), None)
            qd = dd[0] if dd else None
            align = None if qd is None or qd >= min(len(sa), len(sbb)) else dict(rows=(len(da), len(db), len(sa), len(sbb)), slot_rows_of_that_solve_identical=bool(np.array_equal(sa[qd], sbb[qd])),
                                                                                 launches_logged_in_it=int((sa[qd] != 0).any(axis=2).sum()))
            bad.append(dict(member=i, group=i // gsz, ALIGN=align, FIRST_LAUNCH_DIFF=sd, first_device_diff=None if dd is None else dict(fused_solve=dd[0], of=nd, what=dd[1]), first_solve_diff=None if hd is None else dict(solve=hd[0], of=nh, what=hd[1]), first_frame_record_diff=k, of=n, rows_identical_there=bool(rows_same), first_rows_diff_at=first_row_diff,
                            ref=dict(k=int(la[k][0]), rows=int(la[k][1]), iters=int(la[k][3])) if k < len(la) else None, got=dict(k=int(lb[k][0]), rows=int(lb[k][1]), iters=int(lb[k][3])) if k < len(lb) else None,
                            pos_err_last=float(np.abs(fa[n - 1][1:4] - fb[n - 1][1:4]).max())))
    print(f"run {rep} [{','.join(KEYS) or 'default'}]: groups of {gsz} on {T} threads, batch rounds (shared, fallback) = {info}: {len(bad)} of {S} members differ from the single-thread run")
    for b in bad: print("   ", b)
