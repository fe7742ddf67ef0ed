"""How evenly does the strided tile assignment of the value-net kernel spread the heavy (doubles) tiles over the 256
workgroups?  Rows in arena order -> tiles of 64 -> workgroup = tile mod 256; cost model: a tile costs its longest list
~ 4 + 2 hits entries for a plain turn, twice that for doubles -> 1 unit / 2 units.  python tools/tile_balance.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "backgammon-engine_amd")]
import backgammon_env as bg
w = np.fromfile(os.path.join(ROOT, "tests/golden/tdgammonNEW100k.f32"), dtype=np.float32)
n, G = 65536, 256
env = bg.VecGame(n, seed=5); env.load_weights(w); env.run_greedy(160)
for rep in range(3):
    env.step_greedy(auto_reset=False) if False else None
    # one step, then look at the arena it left behind (dice of the step are still in meta until the next roll)
    env.step_greedy(roll=True, auto_reset=True)
    info = env.unique_rows_info().cpu().numpy()
    # the dice the rows were generated with: key bit pattern does not carry them; doubles turns have up to 4 moves
    length = info[:, 1] & 7
    key_len4 = length >= 3                                  # only doubles sequences reach 3 or 4 moves
    game = info[:, 0]
    dbl_game = np.zeros(n, bool); np.logical_or.at(dbl_game, game[key_len4], True)
    heavy_row = dbl_game[game]
    T = (len(game) + 63) // 64
    pad = T * 64 - len(game)
    hr = np.concatenate([heavy_row, np.zeros(pad, bool)]).reshape(T, 64)
    heavy_tile = hr.mean(1) > 0.5
    cost = 1.0 + heavy_tile
    per_block = np.bincount(np.arange(T) % G, weights=cost, minlength=G)
    runs = np.flatnonzero(np.diff(np.concatenate([[0], heavy_tile.astype(int), [0]])))
    run_len = (runs[1::2] - runs[0::2])
    print("rows %d tiles %d heavy tiles %d (%.1f %%) in %d runs (median length %d); cost per workgroup mean %.1f max %.1f (+%.1f %%) min %.1f; "
          "ideal split by class: max %.1f" % (len(game), T, heavy_tile.sum(), 100 * heavy_tile.mean(), len(run_len), np.median(run_len) if len(run_len) else 0,
             per_block.mean(), per_block.max(), 100 * (per_block.max() / per_block.mean() - 1), per_block.min(),
             np.ceil((~heavy_tile).sum() / G) + 2 * np.ceil(heavy_tile.sum() / G)), flush=True)
