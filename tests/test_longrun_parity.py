"""Long-run parity: what the 30 - 44-frame parity tests cannot say.  tests/tools/longrun_parity.py runs images -> tracker -> estimator on the HIP path
and on the oracle for hundreds of frames and counts, per frame: bit-identity of the rows handed from the front end to the back end, equality of the estimator's flags /
counts, and how often a window solve ended a different number of iterations apart (tests/conftest.py::iterations_agree's +-1 allowance: the prior's constant c0 carries
rounding noise that moves only Ceres' RELATIVE function-tolerance test).
Recorded on one MI355X box in round 6 (gpurun_out -> profiles/r06_longrun_*.json): 1000 raw frames at 1280x720: rows bit-identical on 1000 / 1000 frames, 987 of 990
solves with the oracle's iteration count, 3 one apart, trajectories 4.7e-5 m ATE apart (max 1.2e-4 m); 1000 raw frames at 640x360: 988 / 990, 3.5e-5 m; 500 dynamic
frames at 640x360: 490 / 490 equal, object rows bit-identical on every frame, 5.9e-6 m.  Once ONE solve ends an iteration apart the two estimators carry on from states
~1e-5 m apart (priors, linearisation points) and drift slowly: the short tests' 1e-5 m window bar holds for runs without such a frame, not for a thousand frames;
north_star's bar is 1e-3 m ATE.  3000 raw frames at 640x360 (profiles/r06_longrun_raw_640x360_3000.json): rows bit-identical on 3000 / 3000 frames, 2982 of 2990 solves with the oracle's iteration count,
ATE 8.7e-5 m, largest window deviation 3.2e-4 m; the deviation passes 1e-6 m at frame 24, 1e-5 m at frame 173 — BEFORE the first iteration mismatch (frame 746) — and 1e-4 m at frame
1020: it is the two marginalization forms (information form on the device, eigen-decomposed square-root form in the oracle, as in the reference) accumulating rounding differences of
an ill-conditioned prior, about 1e-4 m per 1000 frames.  Default: 300 raw + 160 dynamic frames at 640x360 (~40 s); DVINS_LONGRUN=1: the recorded lengths."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
FULL = os.environ.get("DVINS_LONGRUN", "0") == "1"


@pytest.mark.parametrize("mode,frames", [("raw", 1000 if FULL else 300), ("dynamic", 500 if FULL else 160), ("dynamic_static", 500 if FULL else 120)])
def test_long_run_against_the_oracle(mode, frames):
    import longrun_parity
    st = longrun_parity.run(mode, frames, 640, 360)
    solved = st["solved"]
    assert solved >= frames - 12
    assert st["rows_bit_identical"] == frames, st["rows_differ_first"]              # front end -> back end hand-over: every frame, bit for bit
    assert st["obj_rows_differ"] == 0 and st["flags_differ"] == 0, st
    if mode == "dynamic_static":          # para::is_static_inst_as_background on both sides (choice T1): the estimator's static report, identical every frame, did unmask object pixels
        assert st["static_reports_differ"] == 0 and st["unmasked_frames"] >= 8, st            # recorded: 124 of 500 frames, 1.57 M pixels
    mismatches = st["iter_plus_minus_one"] + st["iter_other"]
    assert mismatches <= max(2, solved // 150), st["iteration_mismatches"]           # recorded: 2 in 990 (640x360), 3 in 990 (1280x720), 0 in 490 (dynamic)
    assert all(abs(m["hip"] - m["oracle"]) <= 3 for m in st["iteration_mismatches"]), st["iteration_mismatches"]      # recorded: one solve in 3000 frames three apart (5 vs 8: the function-tolerance test a hair's breadth from its threshold on both sides), the others one apart
    assert st["ate_hip_vs_oracle_m"] < 2e-4 and st["max_abs_traj_diff_m"] < 5e-4, st   # north_star: 1e-3 m ATE; recorded 3.5e-5 / 9e-5 m after 1000 frames
    if mismatches == 0:
        assert st["max_dp_m"] < 3e-5, st                                              # without such a frame the windows stay together (recorded 1.6e-5 m after 500 dynamic frames)
