#!/usr/bin/env python3
"""End-to-end rehearsal of configs 4/5 on ONE GPU: batched self-play with the turn log, lock-step TD(λ)
replay (HIP learner kernels, bgamd_td_*; --host-learner = the PyTorch closed form), weights back into the env;
win rate against a uniformly random mover before and after, and against the reference's 100k-episode checkpoint.
Not the reference's training CLI (out of scope) -- a 60-line demonstration that the pieces compose.

    python examples/selfplay_train.py --games 512 --rounds 120
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/selfplay_train.py --games 4096
        (config 4/5 shape: every rank plays its own shard of games, ONE all-reduce of the 25 601-float update per
         training step keeps the replicas' weights identical; --dist-backend gloo lets ranks share a GPU for a rehearsal)
"""
import argparse
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before the HIP runtime starts: backgammon_env/__init__.py says why
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "backgammon-engine_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import backgammon_env as bg  # noqa: E402
from backgammon_env.arena import head_to_head  # noqa: E402
from backgammon_env.learner import DeviceTDLambdaLearner, TDLambdaLearner, play_round  # noqa: E402


def xavier_init(seed=0):
    """model.py:55-61: xavier_uniform(gain=0.1) weights, zero biases."""
    g = torch.Generator().manual_seed(seed)
    def xu(fan_out, fan_in):
        a = 0.1 * (6.0 / (fan_in + fan_out)) ** 0.5
        return (torch.rand((fan_out, fan_in), generator=g) * 2 - 1) * a
    return torch.cat([xu(128, 198).flatten(), torch.zeros(128), xu(1, 128).flatten(), torch.zeros(1)]).numpy()


def run_continuous(a, env, arena, L, group, world, prec, say, n_classic=0, t0=None, turns=0):
    """--continuous [--pipeline-rounds]: windows of self-play on every lane, the games that ended in a window replayed streamed through
    --slots; pipelined, window w is played while window w - 1 is replayed (learner on its own stream and host thread)."""
    import threading
    from backgammon_env.learner import ContinuousSelfPlay
    assert a.slots > 0 and not a.host_learner, "--continuous replays streamed through --slots on the device learner"
    # pipelined windows do not all-reduce the finished-game count (the learner's thread owns the process group): with more than one rank
    # the hold decision could then differ between ranks and leave one of them out of a replay's collectives -- refused
    assert not (world > 1 and a.pipeline_rounds and a.min_window_games > 0), "--min-window-games with --pipeline-rounds needs a single rank"
    dist = torch.distributed if world > 1 else None
    sp = ContinuousSelfPlay(env, ring_steps=a.ring_steps, episode=n_classic)      # (the classic rounds played episodes 0 .. n_classic - 1)
    L.set_delay(a.update_delay)
    side = torch.cuda.Stream()
    dev = torch.cuda.current_device()
    t0 = time.time() if t0 is None else t0
    games_done, dropped = n_classic * a.games * world, 0
    total_games = a.rounds * a.games * world          # ~ one game per lane and window once the lanes are out of step
    res = {}

    def replay(table, scale):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(side):
            res["out"] = L.replay_games(sp.rows, *table, slots=a.slots, group=group, batch_scale=scale)

    for b in range(a.burn_in_windows):                 # lanes out of step before the first window that counts (weights: the initial ones)
        if b == 0:
            env.load_weights(L.theta.cpu().numpy())
        sp.play(a.window_steps, epsilon=a.eps if a.eps is not None else a.eps_start, precision=prec)
        sp.finished()
    held, held_all = None, 0                           # finished games held back until --min-window-games (per rank, summed over the ranks) have accumulated
    pending, th = None, None
    n_win = a.rounds - n_classic
    for r in range(n_win + (1 if a.pipeline_rounds else 0)):
        L.update_learning_params(games_done // max(1, a.schedule_div))
        if a.lam is not None:
            L.lambda_decay = a.lam
        eps = a.eps if a.eps is not None else a.eps_start + (a.eps_end - a.eps_start) * min(1.0, games_done / total_games)
        sg = a.scale_games if a.scale_warmup <= 0 else min(a.scale_games, a.scale_warmup * 2.0 ** (r + n_classic))
        scale = min(1.0, sg / (a.slots * world))
        env.load_weights(L.theta.cpu().numpy())        # the weights the last finished replay left
        if a.pipeline_rounds and pending is not None:
            th = threading.Thread(target=replay, args=(pending, scale))
            th.start()
        table = None
        if r < n_win:
            sp.play(a.window_steps, epsilon=eps, precision=prec)
            table = sp.finished(keep_margin=a.window_steps if a.pipeline_rounds else 0)
            dropped += sp.dropped
            n_fin = torch.tensor([int(table[0].numel())], dtype=torch.int64, device="cuda")
            if dist is not None and not a.pipeline_rounds:
                dist.all_reduce(n_fin, group=group)
            else:                                      # (pipelined: the learner's thread owns the process group while it replays -- two threads
                n_fin *= world                         #  must not interleave collectives; every rank finishes ~ the same number of games)
            games_done += int(n_fin.item())
            if held is not None:
                table = tuple(torch.cat([h, t]) for h, t in zip(held, table))
                held = None
            # hold or replay is ONE decision for all ranks (a rank that holds while another replays would miss the replay's collectives):
            # taken from the all-reduced count -- the games of this window plus what every rank already holds
            held_all += int(n_fin.item())
            if held_all < a.min_window_games * world and r + 1 < n_win:
                held, table = table, None              # too few games for a replay of their own: they go with the next window's
            else:
                held_all = 0
        if a.pipeline_rounds:
            if th is not None:
                th.join()
                turns += res["out"][1]
                th = None
            pending = table
        elif table is not None:
            replay(table, scale)
            side.synchronize()
            turns += res["out"][1]
        if a.verbose or r % 4 == 3 or r >= n_win - 1:
            say(f"window {r + 1}: {games_done} games finished ({int(table[0].numel()) if table is not None else 0} in this window, mean len "
                f"{float(table[2].float().mean().item()) if table is not None and table[0].numel() else 0.0:.1f}), eps {eps:.3f}, " +
                (f"td loss {res['out'][0] / max(1, res['out'][1]):.5f} over {res['out'][1]} turns, {world * turns / (time.time() - t0):.0f} turns/s" if res else "no replay yet"),
                flush=True)
    torch.cuda.synchronize()
    say(f"{games_done} games, {turns} turns replayed per rank in {time.time() - t0:.2f} s; {dropped} games dropped (longer than the ring allows)", flush=True)
    w_after = L.theta.cpu().numpy()
    if world > 1:
        chk = torch.tensor([float(np.abs(w_after).sum()), -float(np.abs(w_after).sum())], dtype=torch.float64, device="cuda")
        dist.all_reduce(chk, op=dist.ReduceOp.MAX, group=group)
        assert chk[0].item() == -chk[1].item(), "replicas diverged"
        say("replicas identical across", world, "ranks")
    say("after: vs random", head_to_head(arena, w_after, None), flush=True)
    ref = os.path.join(ROOT, "tests", "golden", "tdgammonNEW100k.f32")
    if os.path.exists(ref):
        say("after: vs tdgammonNEW100k", head_to_head(arena, w_after, np.fromfile(ref, dtype=np.float32)), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=512, help="games per round (one per lane)")
    ap.add_argument("--rounds", type=int, default=120)
    ap.add_argument("--eps-start", type=float, default=0.1, help="exploration rate of the first round (train.py:446)")
    ap.add_argument("--eps-end", type=float, default=0.0, help="... decaying linearly to this over the run (train.py:531)")
    ap.add_argument("--eps", type=float, default=None, help="fixed exploration rate (overrides the linear decay)")
    ap.add_argument("--sub-round", type=int, default=0, help="replay the round in sub-rounds of this many games, weights "
                    "refreshed between them (0 = the whole round lock-step)")
    ap.add_argument("--slots", type=int, default=0, help="streamed replay: this many slots replay the round's games one after another "
                    "(every training step sums the updates of that many games at different plies; overrides --sub-round)")
    ap.add_argument("--scale-games", type=float, default=24.0, help="every game's update is scaled by min(1, this / games per "
                    "(sub-)round): 24 = the summed update of a reference-sized round (round_size = workers <= 24, train.py:307-312)")
    ap.add_argument("--scale-warmup", type=float, default=0.0, help="first round's --scale-games (doubling every round up to --scale-games): "
                    "the summed update of a large step overshoots while the net is random and every game pushes the same way (0 = no warm-up)")
    ap.add_argument("--arena", type=int, default=1024, help="lanes of the evaluation arena (2 games per lane, sides alternated)")
    ap.add_argument("--max-plies", type=int, default=400, help="turn log depth of a classic round: a game that is not over by then is not replayed.  400 is what the "
                    "quality gate runs with (tools/quality_r04.sh); round 5 found that with 600 the long games of the FIRST rounds from a random net -- hundreds of "
                    "turns of noise each, summed into one update -- push every loop, the classic one included, into the 'short games' attractor "
                    "(1.2 %% against tdgammonNEW100k instead of 53 %%: profiles/r05_training_quality.txt)")
    ap.add_argument("--lam", type=float, default=None, help="fixed lambda (default: the reference schedule, model.py:69-73)")
    ap.add_argument("--schedule-div", type=int, default=1, help="the reference's alpha / lambda schedule is a function of the episode count "
                    "(model.py:69-73, written for runs of ~1e5 episodes); it is evaluated at games_done // this")
    ap.add_argument("--continuous", action="store_true", help="continuous self-play (learner.ContinuousSelfPlay): lanes restart their next game the step "
                    "after the last one ended, turns go to a ring log by env step; a 'round' is --window-steps env steps of all lanes and the learner "
                    "replays the games that ended in it (needs --slots).  Games in flight go on under the refreshed weights: a documented deviation")
    ap.add_argument("--window-steps", type=int, default=84, help="env steps per window of --continuous (84 steps of n lanes ~ n finished games)")
    ap.add_argument("--ring-steps", type=int, default=1024, help="depth of the ring log of --continuous (games longer than ring - 2 windows are dropped)")
    ap.add_argument("--verbose", action="store_true", help="a line per round / window")
    ap.add_argument("--update-delay", type=int, default=0, help="1: the device learner applies every update one step late and a training step is ONE launch "
                    "(bgamd_td_set_delay: a documented deviation; streamed replays through 512 ... 4 096 slots)")
    ap.add_argument("--burn-in-windows", type=int, default=0, help="--continuous: windows played (and not replayed) before the first one that counts: all lanes "
                    "start at ply 0 together, so the games that END in the first windows are the short ones only; after ~3 windows the lanes are out of "
                    "step and a window's finished games are an unbiased sample")
    ap.add_argument("--min-window-games", type=int, default=0, help="--continuous: a window's finished games are held back and replayed together with the next "
                    "window's until at least this many have accumulated")
    ap.add_argument("--classic-rounds", type=int, default=0, help="--continuous: this many of --rounds are played first as classic rounds (one game per "
                    "lane to the end, then the replay: the reference's loop, train.py:527-547), the rest as windows: the first rounds of a run from random "
                    "weights are the fragile ones (DESIGN §7)")
    ap.add_argument("--seed", type=int, default=1, help="seed of the self-play env's dice")
    ap.add_argument("--pipeline-rounds", action="store_true", help="with --continuous: the learner replays window w-1 on its own stream (and host "
                    "thread) WHILE the env plays window w -- the self-play policy is one window staler than train.py:519-547's snapshot")
    ap.add_argument("--in-library-collective", action="store_true", help="multi-rank: the per-step all-reduce is issued by the library on the "
                    "learner's stream (bgamd_td_replay_allreduce, an RCCL communicator of the learner's own) instead of torch.distributed per step")
    ap.add_argument("--host-learner", action="store_true", help="PyTorch closed-form replay instead of the HIP kernels")
    ap.add_argument("--dist-backend", default="nccl")
    ap.add_argument("--precision", choices=("auto", "f32", "f16x2", "bf16"), default="auto",
                    help="value net of the self-play steps: f32 = incremental fp32; f16x2 = W1 as f16 hi+lo on the MFMA pipe (same "
                         "1e-5 parity class, 3e-7 measured) -- the faster one for small rounds (512 lanes: 0.039 vs 0.058 ms per "
                         "step); bf16 = config 5's speed mode (single bf16 weights on the MFMA pipe, outside the 1e-5 bound; the learner keeps "
                         "fp32 traces and weights); auto = f16x2 below 16 384 games per rank")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    group = None
    if world > 1:
        import torch.distributed as dist
        from backgammon_env.shard import shard_for_rank
        local = int(os.environ.get("LOCAL_RANK", 0))
        torch.cuda.set_device(local if a.dist_backend == "nccl" else local % torch.cuda.device_count())
        dist.init_process_group(a.dist_backend)
        group = dist.group.WORLD
        off, stride = shard_for_rank(rank, world, a.games)
        env = bg.VecGame(a.games, seed=a.seed, lane_offset=off, lane_stride=stride)
    else:
        env = bg.VecGame(a.games, seed=a.seed)
    arena = bg.VecGame(a.arena, seed=2)
    say = print if rank == 0 else (lambda *x, **k: None)
    prec = (bg.BF16 if a.precision == "bf16" else
            bg.F16X2 if a.precision == "f16x2" or (a.precision == "auto" and a.games < 16384) else bg.F32)
    if a.host_learner:
        L = TDLambdaLearner(xavier_init(), device="cuda", alpha=0.1, lam=0.7)
    else:
        L = DeviceTDLambdaLearner(xavier_init(), max_games=a.games, alpha=0.1, lam=0.7)
    if a.in_library_collective and not a.host_learner:
        L.init_collective(group)
    say("before: vs random", head_to_head(arena, L.theta.cpu().numpy(), None)["win_rate"], flush=True)
    t0, turns = time.time(), 0
    total_games = a.rounds * a.games * world
    first_dice = None
    n_classic = min(a.classic_rounds, a.rounds) if a.continuous else a.rounds
    for r in range(n_classic):
        if not a.host_learner:                               # the first --classic-rounds rounds are always replayed exactly (the fragile phase of a run)
            L.set_delay(a.update_delay if r >= a.classic_rounds else 0)
        games_done = r * a.games * world
        L.update_learning_params(games_done // max(1, a.schedule_div))
        if a.lam is not None:
            L.lambda_decay = a.lam
        # linear decay of the exploration rate across the run (train.py:531)
        eps = a.eps if a.eps is not None else a.eps_start + (a.eps_end - a.eps_start) * (games_done / total_games)
        env.load_weights(L.theta.cpu().numpy())
        rows, lengths, p1_won = play_round(env, max_plies=a.max_plies, epsilon=eps, precision=prec)   # round r = episode r: fresh dice
        if r < 2:                                            # every round must play NEW games
            d = env.dice().clone()
            assert first_dice is None or not torch.equal(d, first_dice), "two rounds rolled the same dice"
            first_dice = d
        sub = a.slots if a.slots > 0 else a.sub_round if a.sub_round > 0 else a.games
        sg = a.scale_games if a.scale_warmup <= 0 else min(a.scale_games, a.scale_warmup * 2.0 ** r)
        scale = min(1.0, sg / (min(sub, a.games) * world))
        if a.host_learner:
            sq, cnt = L.replay(env.encode_rows(rows), lengths, p1_won, group=group, batch_scale=scale)
        else:
            sq, cnt = L.replay_rows(rows, lengths, p1_won, group=group, batch_scale=scale, sub_round=a.sub_round, slots=a.slots)
        turns += cnt
        if a.verbose or r % 20 == 19 or r == a.rounds - 1:
            say(f"round {r + 1}: {(r + 1) * a.games * world} games, mean len {cnt / a.games:.1f}, td loss {sq / cnt:.5f}, "
                f"{world * turns / (time.time() - t0):.0f} turns/s", flush=True)
    if a.continuous:
        return run_continuous(a, env, arena, L, group, world, prec, say, n_classic, t0, turns)
    w_after = L.theta.cpu().numpy()
    if world > 1:                                             # the replicas must still hold the same weights
        chk = torch.tensor([float(np.abs(w_after).sum()), -float(np.abs(w_after).sum())], dtype=torch.float64, device="cuda")
        dist.all_reduce(chk, op=dist.ReduceOp.MAX, group=group)
        assert chk[0].item() == -chk[1].item(), "replicas diverged"
        say("replicas identical across", world, "ranks")
    say("after: vs random", head_to_head(arena, w_after, None), flush=True)
    ref = os.path.join(ROOT, "tests", "golden", "tdgammonNEW100k.f32")       # the reference's own 100k-episode checkpoint
    if os.path.exists(ref):
        say("after: vs tdgammonNEW100k", head_to_head(arena, w_after, np.fromfile(ref, dtype=np.float32)), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
