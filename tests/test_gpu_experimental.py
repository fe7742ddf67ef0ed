"""Tests of the kernels that lost their same-box A/B and compile only with -DBGAMD_EXPERIMENTAL (round 4):
  * eval_rows_mdelta_kernel (csrc/bg_eval_mfma.h, BGAMD_MFMA_DELTA=1): fixtures G7 / G8, agreement with the VALU kernel, values that do not
    depend on the piece a row falls into;
  * eval_rows_d16_kernel (csrc/bg_eval_dense16.h, BGAMD_F16X2_RESIDENT=1) against the LDS-staged f16 x 2 kernel;
  * the LDS-staged root pass (BGAMD_ROOT_RESIDENT=0) against the resident one, bit for bit; the f32-MFMA root pass (BGAMD_ROOT_F32=1);
  * the learner's unfused matrix-pipe forward (BGAMD_TD_FUSED=0).
  * (round 5) the launch structures of rounds 1-4 that lost their A/B, as bit-identity references of the default step: the root pass forced in / out of
    the boundary launch or forked onto a second stream (BGAMD_ROOT_IN_BOUNDARY, BGAMD_OVERLAP), the expansion as two launches (BGAMD_EXPAND_MERGED=0).
Marker gpu_experimental: `pytest -m gpu_experimental` (tools/round_check.sh runs it) builds libbgamd_experimental.so and loads it through
BGAMD_LIB (tests/conftest.py); `-m gpu` and `-m "not gpu"` do not select these.  No parity test was dropped: what moved here still runs."""
import os

import numpy as np
import pytest

from test_gpu_parity import _np
from test_gpu_parity import test_incremental_value_net_equals_dense_chain as _dense_chain      # its third env: BGAMD_ROOT_F32=1
from test_gpu_round2 import _greedy_65536_sampled_lanes, _stream_modes_run_and_graph_capture, _streamed_replay_variant
from test_gpu_round3 import _fixture_step
from test_gpu_round3 import test_bar_positions_rows_vs_reference_values as _bar_rows
from test_gpu_round4 import (_expansion_in_one_launch_odd_env_sizes, _expansion_in_one_launch_plays_the_same_games,
                             _root_pass_inside_the_boundary_launch)

pytestmark = pytest.mark.gpu_experimental

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def bg():
    import backgammon_env
    assert backgammon_env._capi.load().bgamd_build_flags() == b"experimental", "these tests need libbgamd_experimental.so (BGAMD_LIB)"
    return backgammon_env


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def test_bar_positions_rows_vs_reference_values_mfma_delta(bg, golden_dir, weights, monkeypatch):
    monkeypatch.setenv("BGAMD_MFMA_DELTA", "1")
    _bar_rows(bg, golden_dir, weights, "mfma_delta")
    assert bg.VecGame(8).kernel_choice()["experimental_build"]


def test_incremental_value_net_equals_dense_chain_with_the_f32_root_pass(bg, O, weights):
    _dense_chain(bg, O, weights)


def test_streamed_replay_matches_host_closed_form_direct_unfused(bg, weights):
    _streamed_replay_variant(bg, weights, "direct_unfused")


def test_mfma_delta_kernel_g7_and_agreement_with_the_valu_kernel(bg, golden_dir, weights, monkeypatch):
    """The opt-in MFMA delta kernel on fixture G7 (the reference model's own values, 2 390 rows), against the VALU kernel on
    the same rows (the fixed-point W table costs up to ~1.4e-6), and through the stand-alone operator bit for bit."""
    g = np.load(os.path.join(golden_dir, "g7_candidate_values.npz"))
    monkeypatch.setenv("BGAMD_MFMA_DELTA", "1")
    env, info, st, val, n_rows, e32, e64 = _fixture_step(bg, weights, g, "mfma_delta")
    print("eval_rows_mdelta_kernel, %d rows: max |gpu - reference fp32| = %.3g, fp64 %.3g" % (n_rows, e32, e64))
    assert e32 < 1e-5 and e64 < 1e-5
    monkeypatch.setenv("BGAMD_MFMA_DELTA", "0")
    env2, info2, st2, val2, _, e32v, _ = _fixture_step(bg, weights, g, "valu_delta")
    key = {(int(gi),) + tuple(int(x) for x in s): v for s, v, gi in zip(st2, val2, info2[:, 0])}
    gap = max(abs(float(v) - float(key[(int(gi),) + tuple(int(x) for x in s)])) for s, v, gi in zip(st, val, info[:, 0]))
    print("MFMA delta kernel vs VALU delta kernel on the same rows: max |dv| = %.3g (VALU kernel vs reference %.3g)" % (gap, e32v))
    assert gap < 5e-6
    roots, off = g["roots"], g["off"]
    ridx = np.repeat(np.arange(len(roots)), np.diff(off)).astype(np.int32)
    v2 = _np(env.evaluate_incremental(roots[:, :28], roots[:, 28], g["states"].astype(np.int32), ridx))
    by_key = {(int(gi),) + tuple(int(x) for x in s): v for s, v, gi in zip(st, val, info[:, 0])}
    assert all(by_key[(int(r),) + tuple(int(x) for x in s)] == v for s, v, r in zip(g["states"], v2, ridx))


def test_mfma_delta_values_do_not_depend_on_the_piece(bg, weights, monkeypatch):
    """Why bg_eval_mfma.h quantises its W table: the matrix pipe rounds its running sum, so with a floating hi + lo split a
    row's last bits depended on which other rows shared its K-compacted product (on where the leaf stage put the game in the
    arena).  With every term a multiple of the unit's quantum all sums are exact: the SAME rows evaluated in arena order,
    in a random order (every row mostly alone in its piece) and in reversed order give bit-identical values -- and so do
    duplicates of an afterstate, shards of an env and two runs of one seed."""
    monkeypatch.setenv("BGAMD_MFMA_DELTA", "1")
    n = 2048
    env = bg.VecGame(n, seed=515)
    env.load_weights(weights)
    env.run_greedy(30)
    s0, t0 = env.states().clone(), env.turns().clone()
    env.step_greedy(auto_reset=False, precision=bg.F32)
    info, st, val = env.unique_rows()
    assert env.stats()["error_flags"] == 0 and st.shape[0] > 10 * n
    gidx = info[:, 0].to(torch.int32)
    base = _np(env.evaluate_incremental(s0, t0, st, gidx))
    assert np.array_equal(base, _np(val))                     # the stand-alone operator == the step
    rng = np.random.RandomState(7)
    for perm in (rng.permutation(st.shape[0]), np.arange(st.shape[0])[::-1].copy()):
        p = torch.from_numpy(perm).to(st.device)
        out = _np(env.evaluate_incremental(s0, t0, st[p].contiguous(), gidx[p].contiguous()))
        assert np.array_equal(out, base[perm])
    # (the same experiment on the VALU kernel, whose per-row fp32 FMA chain never depended on its neighbours)
    monkeypatch.setenv("BGAMD_MFMA_DELTA", "0")
    env2 = bg.VecGame(n, seed=515)
    env2.load_weights(weights)
    b2 = _np(env2.evaluate_incremental(s0, t0, st, gidx))
    p = torch.from_numpy(rng.permutation(st.shape[0])).to(st.device)
    assert np.array_equal(_np(env2.evaluate_incremental(s0, t0, st[p].contiguous(), gidx[p].contiguous())), b2[_np(p)])
    print("MFMA vs VALU delta kernel on %d mid-game rows: max |dv| = %.3g" % (st.shape[0], np.abs(b2 - base).max()))
    assert np.abs(b2 - base).max() < 5e-6


def test_resident_and_lds_staged_root_pass_are_bit_identical(bg, weights, monkeypatch):
    """The root pass of the incremental value net exists twice: with a wave's weight planes resident in registers (the default since
    round 3; since round 4 W1 as f16 hi + lo, two MFMAs per K-step) and staged through LDS in two K phases as bf16 hi + mid + lo
    (BGAMD_ROOT_RESIDENT=0, rounds 1-2).  Until the default took the f16 planes the two were the same bits; now they are two roundings of
    the same product: the chosen values within 1e-6 of each other from identical positions, and the same move on > 99.9 % of the lanes over 40 steps."""
    n = 8192
    monkeypatch.delenv("BGAMD_ROOT_RESIDENT", raising=False)
    a = bg.VecGame(n, seed=4711)
    monkeypatch.setenv("BGAMD_ROOT_RESIDENT", "0")
    b = bg.VecGame(n, seed=4711)
    monkeypatch.delenv("BGAMD_ROOT_RESIDENT", raising=False)
    a.load_weights(weights); b.load_weights(weights)
    assert b.kernel_choice()["experimental_build"]
    a.run_greedy(40); b.run_greedy(40)
    same = (a.states() == b.states()).all(1) & (a.turns() == b.turns())
    assert same.float().mean().item() > 0.999
    idx = torch.nonzero(same).flatten()
    a.step_greedy(auto_reset=False); b.step_greedy(auto_reset=False)
    assert b.kernel_choice()["root"] == "root_hidden_bf16x3_kernel" and a.kernel_choice()["root"] == "root_hidden_resident_kernel"
    la, lb = a.last_choice(), b.last_choice()
    moved = la["count"][idx] > 0
    d = (la["value"][idx] - lb["value"][idx]).abs()[moved].max().item()
    print(f"root pass f16 x 2 (resident) vs bf16 x 3 (LDS-staged): max |chosen value difference| = {d:.2e} on {int(moved.sum())} lanes")
    assert d < 1e-6
    assert a.stats()["error_flags"] == 0 and b.stats()["error_flags"] == 0


def test_f16x2_with_resident_weights_equals_the_lds_staged_f16x2_kernel(bg, weights, monkeypatch):
    """BGAMD_F16X2_RESIDENT=1 (csrc/bg_eval_dense16.h: the dense f16 hi + lo value net with a wave's weight planes resident in registers,
    -log2 e and b1 folded into the table) against round 1's LDS-staged f16 x 2 kernel: values within 1e-6 of each other and of the fp32
    path's, and the same games for 40 steps (a near-tie may resolve differently: > 99.9 % of the lanes identical is asserted)."""
    n = 4096
    monkeypatch.delenv("BGAMD_F16X2_RESIDENT", raising=False)
    a = bg.VecGame(n, seed=1357)
    monkeypatch.setenv("BGAMD_F16X2_RESIDENT", "1")
    b = bg.VecGame(n, seed=1357)
    monkeypatch.delenv("BGAMD_F16X2_RESIDENT", raising=False)
    c = bg.VecGame(n, seed=1357)
    for e in (a, b, c):
        e.load_weights(weights)
    a.run_greedy(10, precision=bg.F16X2); b.run_greedy(10, precision=bg.F16X2); c.run_greedy(10)
    same = (a.states() == b.states()).all(1)
    assert same.float().mean().item() > 0.999
    # one more step from IDENTICAL positions: chosen values side by side
    idx = torch.nonzero(same & (a.states() == c.states()).all(1)).flatten()
    a.step_greedy(precision=bg.F16X2, auto_reset=False); b.step_greedy(precision=bg.F16X2, auto_reset=False); c.step_greedy(auto_reset=False)
    va, vb, vc = (e.last_choice()["value"][idx] for e in (a, b, c))
    moved = (a.last_choice()["count"][idx] > 0)
    d_ab, d_bc = (va - vb).abs()[moved].max().item(), (vb - vc).abs()[moved].max().item()
    print("f16x2 resident vs LDS-staged: max |dv| = %.3g; vs the fp32 incremental path: %.3g (%d lanes)" % (d_ab, d_bc, int(moved.sum())))
    assert d_ab < 1e-6 and d_bc < 2e-6
    for e in (a, b):
        e.run_greedy(30, precision=bg.F16X2)
    assert ((a.states() == b.states()).all(1)).float().mean().item() > 0.995
    assert a.stats()["error_flags"] == 0 and b.stats()["error_flags"] == 0


# ---- round 5: the launch structures that lost their A/B are honoured by this build only ------------------------------------------------------

def test_greedy_65536_sampled_lanes_vs_oracle_against_rounds_1_to_3_launch_structure(bg, O, weights):
    a, b = _greedy_65536_sampled_lanes(bg, O, weights, twin_switches=("BGAMD_ROOT_IN_BOUNDARY=0", "BGAMD_OVERLAP=1"))
    assert b.kernel_choice()["root_on_second_stream"] and not a.kernel_choice()["root_on_second_stream"]


@pytest.mark.parametrize("mode", ["BGAMD_OVERLAP", "BGAMD_NO_OVERLAP"])
def test_stream_modes_run_and_graph_capture(bg, weights, mode):
    _stream_modes_run_and_graph_capture(bg, weights, mode)


@pytest.mark.parametrize("n", [1000, 33000])
def test_root_pass_forced_in_and_out_of_the_boundary_launch_is_bit_identical(bg, weights, monkeypatch, n):
    _root_pass_inside_the_boundary_launch(bg, weights, monkeypatch, n, force=True)


@pytest.mark.parametrize("n", [700, 33000])
def test_expansion_in_one_launch_plays_the_same_games(bg, weights, monkeypatch, n):
    _expansion_in_one_launch_plays_the_same_games(bg, weights, monkeypatch, n)


@pytest.mark.parametrize("n", [1, 63, 65, 1000, 4097, 24576])
def test_expansion_in_one_launch_odd_env_sizes(bg, weights, monkeypatch, n):
    _expansion_in_one_launch_odd_env_sizes(bg, weights, monkeypatch, n)
