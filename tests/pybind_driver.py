#!/usr/bin/env python3
"""Drives the compiled drop-in module (backgammon-engine_amd/pybind/backgammon_env*.so) in a process of its own: the module
carries the reference's name, `backgammon_env`, and CPython hands out ONE extension module per name and process -- in the
test process that name may already be taken by the Python package or by the reference's own module (oracle/_ref).

    python tests/pybind_driver.py surface         -> checks the exported surface (no GPU needed), prints OK
    python tests/pybind_driver.py games <n>       -> plays fixture G3's first n games through it, prints 'OK turns seconds'
"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "backgammon-engine_amd", "pybind"))       # where a reference user points sys.path (model.py:26)
import backgammon_env as bg  # noqa: E402  (the compiled module: this process never imports the package of the same name)

START = [2, 0, 0, 0, 0, -5, 0, -3, 0, 0, 0, 5, -5, 0, 0, 0, 3, 0, 5, 0, 0, 0, 0, -2]


def surface():
    assert bg.__file__.endswith(".so") and os.sep + "pybind" + os.sep in bg.__file__
    game_methods = {"setPlayers", "getPlayers", "getTurn", "setTurn", "getGameBoard", "getPieces", "legalMoves", "legalTurnSequences",
                    "evaluateTurnSequences", "tryMove", "is_game_over", "clone", "getJailedCount", "setBorneOffPieces",
                    "getBornOffCount", "setGameBoard", "setDice", "printGameBoard", "reset", "populateBoard", "roll_dice", "get_last_dice"}
    assert game_methods <= set(dir(bg.Game)), game_methods - set(dir(bg.Game))
    assert {"getName", "getNum"} <= set(dir(bg.Player)) and {"numJailed", "numFreed"} <= set(dir(bg.Pieces))
    assert bg.PlayerType.PLAYER1 == 0 and bg.PlayerType.PLAYER2 == 1 and int(bg.PlayerType.PLAYER2) == 1     # unscoped enum: equals ints
    p = bg.Player("White", bg.PlayerType.PLAYER1)
    assert p.getName() == "White" and p.getNum() == 0
    try:
        bg.Player("x", 0)                                     # the constructor rejects a plain int, as the reference's does
        raise SystemExit("Player accepted an int")
    except TypeError:
        pass
    print("OK", bg.source_hash())
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "backgammon-engine_amd", "libbgamd.so"))
    lib.bgamd_device_count.restype = ctypes.c_int
    if lib.bgamd_device_count() == 0:
        try:
            bg.Game(0)
            raise SystemExit("Game() worked without a device")
        except RuntimeError as e:
            assert "no usable gfx950 device" in str(e), e
            print("NODEVICE")


def games(n):
    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "g3_random_trajectories.npz"))
    rows = g["rows"]
    p1, p2 = bg.Player("White", bg.PlayerType.PLAYER1), bg.Player("Black", bg.PlayerType.PLAYER2)
    warm = bg.Game(0)                                         # HIP start-up and the first launches are not the surface's throughput
    warm.legalTurnSequences(0, 3, 1)
    del warm
    turns, t0 = 0, time.time()
    for lane in range(n):
        r = rows[rows[:, 0] == lane]
        game = bg.Game(0)
        game.setPlayers(p1, p2)
        game.setTurn(int(r[0, 30]))
        for t in range(len(r)):
            turn, d1, d2, C, k = int(r[t, 30]), int(r[t, 31]), int(r[t, 32]), int(r[t, 33]), int(r[t, 34])
            assert game.getTurn() == turn
            assert list(game.getGameBoard()) == list(r[t, 2:26])
            assert [game.getJailedCount(0), game.getJailedCount(1), game.getBornOffCount(0), game.getBornOffCount(1)] == list(r[t, 26:30])
            game.setDice(d1, d2)
            assert list(game.get_last_dice()) == [d1, d2]
            seqs, states = game.evaluateTurnSequences(turn, d1, d2)
            assert len(seqs) == C and states.shape == (C, 28) and states.dtype == np.int32
            if C:
                pl = game.getPlayers(turn)
                for o, d in seqs[k]:
                    ok, msg = game.tryMove(pl, abs(o - d), o, d)
                    assert ok and msg == ""
                assert list(game.getGameBoard()) + [game.getJailedCount(0), game.getJailedCount(1), game.getBornOffCount(0),
                                                    game.getBornOffCount(1)] == list(states[k])
            over, winner = game.is_game_over()
            assert int(over) == r[t, 35] and (not over or winner == r[t, 36])
            turns += 1
            if over:
                assert t == len(r) - 1
                break
            game.setTurn(1 - turn)
    dt = time.time() - t0
    # known answers of cppsrc/tests.cpp through the compiled module, clone(), error strings
    game = bg.Game(0)
    game.setPlayers(p1, p2)
    assert game.legalMoves(bg.PlayerType.PLAYER1, 1) == [(1, 2), (17, 18), (19, 20)]                  # tests.cpp:287
    assert len(game.legalTurnSequences(0, 1, 2)) == 30 and len(game.legalTurnSequences(0, 1, 1)) == 245   # :346, :367
    assert game.tryMove(p1, 5, 1, 6) == (False, "Invalid destination.")                               # :116-121
    c = game.clone()
    assert list(c.getGameBoard()) == START and list(c.get_last_dice()) == [1, 1] and c.getPieces().numFreed(0) == 0
    assert c.tryMove(p1, 1, 1, 2)[0] and list(game.getGameBoard()) == START and list(c.getGameBoard()) != START
    r1 = game.roll_dice()
    assert len(r1) == 2 and all(1 <= v <= 6 for v in r1) and list(game.get_last_dice()) == list(r1)
    try:
        game.setGameBoard([0] * 25)
        raise SystemExit("setGameBoard accepted 25 entries")
    except ValueError:
        pass
    print("OK", turns, "%.3f" % dt)


if __name__ == "__main__":
    surface() if sys.argv[1] == "surface" else games(int(sys.argv[2]))
