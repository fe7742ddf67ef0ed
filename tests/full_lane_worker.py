"""Worker of tests/test_gpu_full_lanes.py: runs in spawned processes that never touch the GPU -- numpy + the oracle (test
infrastructure) only.  For a slice of lanes: the state the HIP step produced must be one of the oracle's afterstates of the lane's
pre-move state and dice, with a value (fp64 restatement of the reference model) within 1e-5 of the arg-max / arg-min."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_W = None
_O = None


def init(wpath):
    global _W, _O
    from oracle import oracle as O
    _O = O
    _W = np.fromfile(wpath, dtype=np.float32)


def check(args):
    pre, pt, dice, post = args[:4]
    tol = args[4] if len(args) > 4 else 1e-5
    O = _O
    worst, not_first, stuck, nrows = 0.0, 0, 0, 0
    for i in range(len(pre)):
        s = O.State.from28(pre[i], pt[i])
        _, _, cand = O.evaluate_turn_sequences(s, int(pt[i]), int(dice[i, 0]), int(dice[i, 1]))
        if len(cand) == 0:
            if not (post[i] == pre[i]).all():
                return ("FAIL", "a stuck lane moved", pre[i].tolist())
            stuck += 1
            continue
        nrows += len(cand)
        v = O.forward_f64(_W, O.encode(cand, int(pt[i])))
        k = [j for j in range(len(cand)) if (cand[j] == post[i]).all()]
        if not k:
            return ("FAIL", "the applied state is not among the oracle's afterstates", pre[i].tolist(), post[i].tolist())
        best = v.max() if pt[i] == 0 else v.min()
        gap = abs(float(v[k[0]]) - float(best))
        if gap >= tol:
            return ("FAIL", "value %g vs best %g" % (v[k[0]], best), pre[i].tolist())
        worst = max(worst, gap)
        first = int(np.argmax(v) if pt[i] == 0 else np.argmin(v))
        if not (cand[first] == post[i]).all():
            not_first += 1
    return ("OK", worst, not_first, stuck, nrows)


def run_random(args):
    """Config 2: the oracle's whole env for a slice of lanes (Philox dice, reference-order enumeration, k = u32 C >> 32, apply, terminal
    check, auto-reset) against the per-step snapshots of the HIP env: every state, turn and flag, every step."""
    seed, lanes, n, steps, snaps, flags = args
    O = _O
    fin = ctot = 0
    for j, lane in enumerate(lanes):
        ref, f, c, _ = O.lane_run(seed, int(lane), n, steps, 0)
        if not (ref[:, :29] == snaps[:, j].astype(np.int32)).all():
            t = int(np.nonzero((ref[:, :29] != snaps[:, j].astype(np.int32)).any(1))[0][0])
            return ("FAIL", int(lane), t)
        if not (ref[:, 29] == flags[:, j]).all():
            return ("FAIL", int(lane), "flags")
        fin += f
        ctot += c
    return ("OK", fin, ctot)


def check_eps(args):
    """epsilon-greedy self-play (model.py:205-206 with the Philox TURN stream): a lane explores iff (x3 >> 8) 2^-24 < eps and then plays
    reference-order candidate k = (x2 C) >> 32; every other lane plays the value-optimal afterstate."""
    seed, n, eps, lanes, pre, pt, ply, epi, post, chosen = args
    O = _O
    n_explore, worst = 0, 0.0
    for i, lane in enumerate(lanes):
        gid = int(lane) + int(epi[i]) * n
        d1, d2, cu, eu = O.turn_randoms(seed, gid, int(ply[i]))
        _, _, cand = O.evaluate_turn_sequences(O.State.from28(pre[i], pt[i]), int(pt[i]), d1, d2)
        if len(cand) == 0:
            if not (post[i] == pre[i]).all():
                return ("FAIL", int(lane), "a stuck lane moved")
            continue
        if np.float32(eu >> 8) * np.float32(1.0 / 16777216.0) < np.float32(eps):
            k = (cu * len(cand)) >> 32
            if chosen[i] != k or not (post[i] == cand[k]).all():
                return ("FAIL", int(lane), "exploring lane: not candidate k")
            n_explore += 1
            continue
        v = O.forward_f64(_W, O.encode(cand, int(pt[i])))
        k = [j for j in range(len(cand)) if (cand[j] == post[i]).all()]
        if not k:
            return ("FAIL", int(lane), "not an afterstate")
        best = v.max() if pt[i] == 0 else v.min()
        gap = abs(float(v[k[0]]) - float(best))
        if gap >= 1e-5:
            return ("FAIL", int(lane), "value gap %g" % gap)
        worst = max(worst, gap)
    return ("OK", n_explore, worst)
