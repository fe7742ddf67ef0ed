"""Runs the oracle's own golden tests against the AddressSanitizer + UBSan build of oracle/bg_oracle.c (BG_ORACLE_LIB), in a
process started with libasan preloaded (tests/test_sanitizers_cpu.py).  No pytest session: its conftest builds the HIP library."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import test_oracle_golden as T  # noqa: E402

assert "asan" in T.O._LIB_PATH, T.O._LIB_PATH
golden = os.path.join(ROOT, "tests", "golden")
w = np.fromfile(os.path.join(golden, "tdgammonNEW100k.f32"), dtype=np.float32)
for fn in (T.test_valid_origin_start, T.test_valid_destination_start, T.test_capture_and_errors, T.test_try_move_bar_entry_and_plain,
           T.test_game_over_and_freeing, T.test_legal_moves_known_answers, T.test_turn_sequence_known_answers,
           T.test_overrun_asymmetry_q1, T.test_no_move_asymmetry_q4, T.test_philox_known_answers):
    fn()
for fn in (T.test_g2_start_counts, T.test_g1_edge_calls, T.test_g3_random_trajectories, T.test_g4_encoder):
    fn(golden)
T.test_g5_values(golden, w)
T.test_g5_greedy_trajectories(golden, w)
# the whole-lane driver the GPU tests compare with: greedy and random play with auto-reset
snap, fin, ct, _ = T.O.lane_run(20240603, 5, 64, 400, 1, weights=w)
snap2, fin2, ct2, _ = T.O.lane_run(20240603, 5, 64, 400, 0)
assert fin >= 1 and fin2 >= 1 and ct > 0 and ct2 > 0
print("OK")
