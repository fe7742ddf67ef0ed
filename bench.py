#!/usr/bin/env python3
"""bench.py -- self-play env steps/sec @65 536 concurrent games per MI355X (BASELINE.json metric).

A "step" is one batched env step: every live lane rolls, enumerates its legal turn sequences,
scores every afterstate with the 198->128->1 value net, applies the arg-max/arg-min and
auto-resets finished games (config 3 of BASELINE.json, weights tdgammonNEW100k -- SURVEY.md
explains why not bestModel.pth).  value = env steps (live-lane turns) of ALL ranks / wall time
of the slowest rank, states resident in HBM.  The K timed steps are ONE bgamd_env_run_greedy(K)
call (the same games as K step_greedy calls; consecutive steps share a launch); the dominant
kernel is bracketed with HIP events on every 8th of those steps (every 4th in runs shorter than 80 steps; an event pair costs ~4 us).

    python bench.py [--gpus N --steps K --warmup W]            # N > 1: this process starts the N ranks itself (launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      # or a launcher does

After the timed region (outside the contract's bracket) the line also carries `training_round`: configs 4/5's per-GPU share -- self-play
with the turn log + the TD(lambda) replay, with ONE all-reduce of the 25 601-float update per training step when there is more than one rank.
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "backgammon-engine_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# Streams of one process map onto GPU_MAX_HW_QUEUES (default 4) hardware queues; an RCCL communicator brings streams of its own.
# The step itself runs on ONE stream since round 4; the training loop's learner has a second one (replay beside the next window).
# Read when the HIP runtime starts, so it is set before anything touches the GPU.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

np = torch = dist = None                    # numpy / torch / torch.distributed: imported by _heavy_imports() in a RANK, never in the launching parent


def _heavy_imports():
    global np, torch, dist
    import numpy, torch as _torch, torch.distributed as _dist       # noqa: E401
    np, torch, dist = numpy, _torch, _dist


GAMES_PER_GPU = 65536
SEED = 20240603
FLOP_PER_ROW = 2 * 198 * 128 + 2 * 128          # 50 944, SURVEY.md §8d
PEAK = {"f32": 157.3, "bf16": 2500.0, "f16x2": 2500.0, "hbm": 8000.0}   # TFLOP/s dense MFMA, GB/s HBM3E (MI355X_MICROARCH.md)
PEAK_VALU_F32 = 157.3        # TFLOP/s fp32 vector (256 CUs x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz), same guide
PEAK_LDS_TBPS = 157.3        # ds_read_b128: 256 B/clk/CU x 256 CUs x 2.4 GHz (the guide measures ~150 with every CU streaming)
# executed fp32 operations of the incremental value net per row: 128 hidden units x (exp2, +1, rcp, fma = 5 flop) + the
# output unit; per (row, changed feature): one 128-float W1 column = 128 FMAs
FLOP_PER_ROW_EPILOGUE = 128 * 5 + 6
FLOP_PER_COLUMN = 256

def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--games", type=int, default=GAMES_PER_GPU, help="concurrent games per GPU")
    ap.add_argument("--burnin", type=int, default=160, help="untimed steps that de-phase the games (input preparation)")
    ap.add_argument("--precision", choices=("f32", "f32_dense", "f16x2", "bf16"), default="f32",
                    help="value-net arithmetic: f32 = fp32 FMAs, incremental hidden layer (headline); f32_dense = the dense fp32 "
                         "MFMA chain over every afterstate; f16x2 = f16 hi+lo weight split, fp32 accumulate (also inside the "
                         "1e-5 parity bound); bf16 = speed mode outside it")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (RCCL, one GPU per rank) | gloo (rehearsal: ranks may share a GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--training-round", dest="training_round", action="store_true", default=True,
                    help="(default) extra object in the line, measured AFTER the timed region: one training round of configs 4/5's per-GPU "
                         "share (self-play with the turn log + TD(lambda) replay; one all-reduce per training step when world > 1)")
    ap.add_argument("--no-training-round", dest="training_round", action="store_false",
                    help="skip it: its self-play launches the step's kernels on shrinking batches, and a rocprofv3 --stats summary "
                         "must average the timed workload only (tools/profile_round.sh and the A/B harness pass this)")
    ap.add_argument("--quick", action="store_true", help="A/B runs: skip the other value-net modes, the training round and the CPU baseline, sample U on fewer lanes")
    ap.add_argument("--training-round-timeout", type=float, default=300.0, help="seconds before a training round that does not finish is "
                    "abandoned (the line is printed without it)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="--gpus N > 1 started plainly: seconds before the parent's watchdog "
                    "kills the ranks it started and reports")
    return ap.parse_args(argv)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n_ranks, argv, child_cmd=None, timeout_s=1500.0, out=None, err=None):
    """`python bench.py --gpus N` started PLAINLY (no WORLD_SIZE in the environment): this process -- which has imported neither torch nor
    the HIP library and never touches the GPU -- starts the N ranks as CHILD processes (the reference starts its workers the same way: a
    spawn pool, train.py:324-325,495-496), one process group each, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 /
    MASTER_PORT set, relays rank 0's stdout (the ONE JSON line) to its own, every rank's stderr (and the other ranks' stdout) to its
    stderr, and returns the ranks' exit code: 0 only when every rank exited 0.  The first rank that fails takes the others down (SIGTERM to
    exactly the process groups started here, SIGKILL 10 s later); so does the watchdog after timeout_s (-> 124).  Nothing is exec'ed or
    retried.  child_cmd: the command of one rank (default: this interpreter on this file with the same arguments)."""
    out = out or sys.stdout
    err = err or sys.stderr
    cmd = list(child_cmd) if child_cmd else [sys.executable, os.path.abspath(__file__), *argv]
    port = _free_port()
    procs, pumps = [], []

    def pump(src, dst, tag):
        for line in iter(src.readline, b""):
            dst.write((tag + line.decode(errors="replace")) if tag else line.decode(errors="replace"))
            dst.flush()
        src.close()

    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_LAUNCHED_BY=str(os.getpid()))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True)
        procs.append(p)
        for src, dst, tag in ((p.stdout, out if r == 0 else err, "" if r == 0 else f"[rank {r}] "), (p.stderr, err, f"[rank {r}] ")):
            t = threading.Thread(target=pump, args=(src, dst, tag), daemon=True)
            t.start()
            pumps.append(t)

    def stop_all(sig):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)              # start_new_session: pgid == pid of the rank this process started
                except (ProcessLookupError, PermissionError):
                    pass

    t0, rc, why = time.monotonic(), 0, None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc, why = (bad[0][1] if bad[0][1] > 0 else 128 - bad[0][1]), f"rank {bad[0][0]} exited with {bad[0][1]}"
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() - t0 > timeout_s:
            rc, why = 124, f"watchdog: ranks still running after {timeout_s:.0f} s"
            break
        time.sleep(0.05)
    if why:
        stop_all(signal.SIGTERM)
        t1 = time.monotonic()
        while any(p.poll() is None for p in procs) and time.monotonic() - t1 < 10.0:
            time.sleep(0.05)
        stop_all(signal.SIGKILL)
        for p in procs:
            p.wait()
    for t in pumps:
        t.join(timeout=5.0)
    if why:
        err.write(f"bench.py launcher: {why}; exit codes by rank {[p.returncode for p in procs]}\n")
        err.flush()
    return rc


def host_description():
    """CPU model and core counts of the host the baseline ran on (SURVEY §8d ii)."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    quota = None                                   # a container's CPU share (cgroup v2 cpu.max / v1 cfs quota) caps what threads can use
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        usable = max(1, min(usable, int(round(quota))))
    return {"cpu_model": model, "nproc": os.cpu_count() or 1, "usable_cores": usable,
            "cgroup_cpu_quota": round(quota, 2) if quota else None}


def cpu_baseline(weights, budget_s=10.0):
    """The same workload on the host's cores, bounded samples (SURVEY §8d ii):
      * the oracle port (greedy, fp32, same Philox streams and weights) on ONE thread and on ALL usable cores -- one
        game shard (lane) per core, the way the reference parallelises self-play (one game per pool worker,
        train.py:324-325); bgo_lane_run is a foreign call, ctypes drops the GIL, so plain threads run in parallel;
      * when the prebuilt reference engine travelled with the snapshot, the unmodified reference's
        evaluateTurnSequences + tryMove loop (its move-gen, 84 % of its turn; its Python policy cannot travel)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    torch.set_num_threads(1)
    host = host_description()

    def run_lane(lane_id, budget):
        lane, steps, t0 = None, 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            _, _, _, lane = O.lane_run(SEED, lane_id, GAMES_PER_GPU, 200, 1, weights=weights, lane=lane, want_snap=False)
            steps += 200
        return steps, time.perf_counter() - t0

    steps, dt = run_lane(0, budget_s * 0.6)
    out = {"value": round(steps / dt, 1), "unit": "env steps/s", "cores": 1, "kind": "port",
           "sample": f"{steps} greedy fp32 env steps of lane 0 (auto-reset, same Philox streams, same weights), "
                     f"{dt:.1f} s, oracle/bg_oracle.c single thread", "host": host}
    nthr = max(1, host["usable_cores"])
    t0 = time.perf_counter()
    with ThreadPoolExecutor(nthr) as ex:
        res = list(ex.map(lambda k: run_lane(k, budget_s * 0.6), range(nthr)))
    wall = time.perf_counter() - t0
    tot = sum(r[0] for r in res)
    out["all_cores"] = {"value": round(tot / wall, 1), "unit": "env steps/s", "cores": nthr, "kind": "port",
                        "sample": f"{tot} greedy fp32 env steps, lanes 0..{nthr - 1} one per thread (one game shard per core), "
                                  f"{wall:.1f} s wall", "speedup_vs_1_thread": round(tot / wall / max(steps / dt, 1e-9), 2)}
    try:
        import importlib.util
        so = [f for f in os.listdir(os.path.join(ROOT, "oracle", "_ref"))
              if f.startswith("backgammon_env") and ("cpython-%d%d" % sys.version_info[:2]) in f]
        spec = importlib.util.spec_from_file_location("backgammon_env", os.path.join(ROOT, "oracle", "_ref", so[0]))
        rb = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(rb)
        p1, p2 = rb.Player("a", rb.PlayerType.PLAYER1), rb.Player("b", rb.PlayerType.PLAYER2)
        n, t0, gid = 0, time.perf_counter(), 0
        while time.perf_counter() - t0 < budget_s * 0.5:
            g = rb.Game(0); g.setPlayers(p1, p2)
            g.setTurn(O.lib().bgo_opening_turn(SEED, gid))
            ply = 0
            while time.perf_counter() - t0 < budget_s * 0.5:
                d1, d2, cu, _ = O.turn_randoms(SEED, gid, ply)
                t = g.getTurn()
                seqs, _ = g.evaluateTurnSequences(t, d1, d2)
                if seqs:
                    pl = g.getPlayers(t)
                    for o, d in seqs[(cu * len(seqs)) >> 32]:
                        g.tryMove(pl, abs(o - d), o, d)
                n += 1
                if g.is_game_over()[0]:
                    break
                g.setTurn(1 - t); ply += 1
            gid += 1
        dt = time.perf_counter() - t0
        out["reference_engine"] = {"value": round(n / dt, 1), "unit": "env steps/s", "cores": 1, "kind": "reference",
                                   "sample": f"{n} random-policy steps through the unmodified reference's "
                                             f"evaluateTurnSequences+tryMove (oracle/_ref, g++ -O2), {dt:.1f} s"}
    except Exception as e:  # the prebuilt reference module is optional on the GPU box
        out["reference_engine"] = {"value": None, "note": f"oracle/_ref not usable here: {type(e).__name__}"}
    return out


def training_round(bg, games, w, rank=0, world=1, backend="nccl", rehearsal=False):
    """Extra information (outside the timed region of the contract): one training round of configs 4/5's per-GPU share -- self-play
    of `games` games per rank with the turn log from a frozen snapshot (train.py:527-547), then the TD(lambda) replay of the round on the
    HIP learner kernels (bgamd_td_*): lock-step over the whole round, and streamed through 2 048 slots (the configuration the
    quality study recommends: DESIGN.md §6).  Second run of each (the first pays allocations).

    world > 1 (configs 4/5 as BASELINE lays them out; replaces the reference's spawn pool, train.py:324-325,495-496): every rank plays and
    replays ITS shard (global game ids: shard_for_rank), every training step's 25 601-float update is all-reduced ONCE -- issued by the
    library on the learner's own RCCL communicator (DeviceTDLambdaLearner.init_collective -> bgamd_td_replay_allreduce) with backend nccl,
    through torch.distributed with the gloo rehearsal backend (ranks sharing a GPU cannot form an RCCL communicator) -- and the replicas'
    weights are compared bit for bit at the end of every phase (MIN == MAX of a 64-bit checksum over the ranks).  Times are the MAX over
    the ranks between barriers, turns and updates the SUM: round_turns_per_s is the whole job's.
    rehearsal (ranks SHARING a GPU over gloo): the same route once through -- one pass per replay, two windows per loop -- because processes that
    share a GPU take turns on it: an all-reduce of a device tensor then costs a scheduling quantum (1 ms with two processes on the card, ~35 ms with
    three), and the numbers measure that, not the learner."""
    from backgammon_env.learner import ContinuousSelfPlay, DeviceTDLambdaLearner, play_round
    from backgammon_env.shard import shard_for_rank
    # BENCH_FORCE_DIST=1 on one rank rehearses the multi-rank route end to end: process group, the learner's own RCCL communicator beside torch's, the
    # in-library all-reduce of every training step (a world of one has nothing to move: what runs is every call the real thing makes)
    forced = world == 1 and os.environ.get("BENCH_FORCE_DIST") == "1" and dist.is_initialized()
    if forced:
        os.environ["BGAMD_FORCE_COLLECTIVE"] = "1"
    multi = world > 1 or forced
    group = dist.group.WORLD if multi else None
    cdev = torch.device("cuda", torch.cuda.current_device()) if (not multi or backend == "nccl") else torch.device("cpu")
    off, stride = shard_for_rank(rank, world, games)
    env = bg.VecGame(games, seed=5, lane_offset=off, lane_stride=stride)
    env.load_weights(w)

    def reduce(x, op):
        if not multi:
            return x
        t = torch.tensor([x], dtype=torch.float64 if isinstance(x, float) else torch.int64, device=cdev)
        dist.all_reduce(t, op=op, group=group)
        return t.item()

    coll = {"in_library": multi and backend == "nccl", "note": None}

    def learner(max_games):
        L = DeviceTDLambdaLearner(w, max_games=max_games, alpha=0.1, lam=0.7)
        if coll["in_library"]:
            ok = 1
            try:
                L.init_collective(group)                   # an RCCL communicator of the learner's own beside torch's
            except Exception as e:
                ok, coll["note"] = 0, repr(e)[:200]
            if reduce(ok, dist.ReduceOp.MIN) == 0:         # one decision for all ranks: the per-step all-reduce goes through torch.distributed instead
                L._comm = None
                coll["in_library"] = False
                coll["note"] = coll["note"] or "another rank could not create the learner's communicator"
        return L

    def timed(f):
        if multi:
            dist.barrier(group)
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
        return r, reduce(time.perf_counter() - t0, dist.ReduceOp.MAX if multi else None)

    def replicas_identical(L):
        """the ranks' weights, bit for bit: a 64-bit sum of the fp32 bit patterns, MIN == MAX over the ranks"""
        cs = int(L.theta.view(torch.int32).to(torch.int64).sum().item())
        return reduce(cs, dist.ReduceOp.MIN if multi else None) == reduce(cs, dist.ReduceOp.MAX if multi else None), cs

    L = learner(games)
    passes, n_win, first_win = (1, 2, 1) if rehearsal else (2, 8, 4)
    for _ in range(passes):
        (rows, lengths, won), dt_play = timed(lambda: play_round(env, max_plies=600, epsilon=0.05))
    turns = reduce(int(lengths.sum().item()), dist.ReduceOp.SUM if multi else None)
    out = {"games": games * world, "games_per_rank": games, "ranks": world, "turns": turns, "selfplay_with_turn_log_ms": round(1e3 * dt_play, 2),
           "collective": None}
    ident = True
    for name, kw in (("lockstep_whole_round", {}), ("streamed_2048_slots", {"slots": 2048})):
        for _ in range(passes):
            L.set_weights(w)
            (sq, cnt), dt = timed(lambda: L.replay_rows(rows, lengths, won, group=group,
                                                        batch_scale=24.0 / (world * (kw.get("slots") or games)), **kw))
        cnt = reduce(cnt, dist.ReduceOp.SUM if multi else None)
        same, cs = replicas_identical(L)
        ident = ident and same
        out[name] = {"replay_ms": round(1e3 * dt, 2), "td_updates_per_s": round(cnt / dt, 1),
                     "round_turns_per_s": round(turns / (dt + dt_play), 1), "weights_checksum": "%016x" % (cs & (2 ** 64 - 1))}
    del L
    # round 4: continuous self-play (every lane restarts the step after its game ended; ring log by env step) in windows of 84 steps of all
    # lanes -- about as many turns as a round -- with the replay of the games that ended in the window after it, and BESIDE the next window
    # (learner on its own stream and host thread; tools/train_pipeline.py is the stand-alone A/B, DESIGN.md §6 the numbers).  With more than one
    # rank every rank joins every window's collectives (replay_games handles a window without finished games).
    L = learner(2048)
    side = torch.cuda.Stream()
    dev_i = torch.cuda.current_device()
    for pipe in (False, True):
        sp = ContinuousSelfPlay(env, ring_steps=1024)
        res, times, t_play = {}, [], []

        def replay(tab):
            torch.cuda.set_device(dev_i)
            with torch.cuda.stream(side):
                res["r"] = L.replay_games(sp.rows, *tab, slots=2048, group=group, batch_scale=24.0 / (world * 2048))
            side.synchronize()
        pending = None
        for r in range(n_win):
            if multi:
                dist.barrier(group)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            th = None
            if pipe and pending is not None:
                th = threading.Thread(target=replay, args=(pending,)); th.start()
            sp.play(84, epsilon=0.05)
            tab = sp.finished(keep_margin=84 if pipe else 0)
            t_play.append(time.perf_counter() - t0)
            if pipe:
                if th is not None:
                    th.join()
                pending = tab
            else:
                replay(tab)
            torch.cuda.synchronize()
            if r >= first_win:
                times.append(time.perf_counter() - t0)
        ms = reduce(1e3 * float(np.median(times)), dist.ReduceOp.MAX if multi else None)
        trn = reduce(int(res["r"][1]), dist.ReduceOp.SUM if multi else None)
        same, cs = replicas_identical(L)
        ident = ident and same
        out["continuous_window_84_steps" + ("_replay_beside_the_next_window" if pipe else "")] = {
            "window_ms": round(ms, 2), "selfplay_ms": round(1e3 * float(np.median(t_play[first_win:])), 2) if not pipe else None,
            "turns_replayed": trn, "round_turns_per_s": round(trn / ms * 1e3, 1), "weights_checksum": "%016x" % (cs & (2 ** 64 - 1))}
        sp.close()
    out["replicas_identical"] = bool(ident)
    out["collective"] = ("none (one rank)" if not multi else "in-library ncclAllReduce (bgamd_td_replay_allreduce)" if coll["in_library"]
                         else "torch.distributed all_reduce over %s%s" % (backend, " (rehearsal)" if backend != "nccl" else "")) + \
                        ("" if not multi else ", one per training step")
    out["collective_note"] = coll["note"]
    if rehearsal:
        out["rehearsal"] = "ranks share a GPU: one pass per replay, two windows per loop; times are scheduling quanta, not the learner's"
    del L, env
    return out


def distinct_ratio(env, prec, sample_lanes=2048, samples=4):
    """Distinct afterstates U and raw reference-order candidates C per env step, measured after the timed region
    on lane samples of `samples` consecutive steps through the ordered enumeration (the throughput path never
    materialises the raw list)."""
    u = c = lanes = 0
    for k in range(samples):
        env.roll()
        offs, cnts, st, _, _ = env.enumerate()
        offs, cnts, st = offs.cpu().numpy(), cnts.cpu().numpy(), st.cpu().numpy()
        for lane in range(k, env.n, max(1, env.n // sample_lanes)):
            n = int(cnts[lane])
            lanes += 1
            if n:
                u += len(np.unique(st[offs[lane]:offs[lane] + n], axis=0))
                c += n
        env.step_greedy(precision=prec)
    return u / lanes, c / lanes              # distinct afterstates and raw candidates per env step


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse_args(argv)
    if a.quick:
        a.training_round = False

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # started plainly: this process becomes the launcher of the N ranks and nothing else (no torch, no HIP in it)
        sys.exit(launch_ranks(a.gpus, argv, timeout_s=a.launch_timeout))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: start it plainly (it launches its own ranks) or with --nproc-per-node {a.gpus}")
    _heavy_imports()
    dev_index = local_rank if a.dist_backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"      # the env switch rehearses the RCCL path on one rank
    if use_dist:
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.dist_backend)

    # always through build(): content-based (the library carries the digest of its sources), a no-op when fresh
    if rank == 0:
        import __graft_entry__
        __graft_entry__.build()
    if use_dist:
        dist.barrier()
    import backgammon_env as bg
    from backgammon_env.shard import aggregate, census, shard_for_rank

    w = np.fromfile(os.path.join(ROOT, "tests", "golden", "tdgammonNEW100k.f32"), dtype=np.float32)
    off, stride = shard_for_rank(rank, world, a.games)
    env = bg.VecGame(a.games, device=dev_index, seed=SEED, lane_offset=off, lane_stride=stride,
                     arena_rows=a.games * 512)
    env.load_weights(w)
    prec = {"f32": bg.F32, "f32_dense": bg.F32_DENSE, "bf16": bg.BF16, "f16x2": bg.F16X2}[a.precision]
    env.run_greedy(a.burnin + a.warmup, precision=prec)
    torch.cuda.synchronize()
    env.stats()                                   # raises on arena overflow
    env.reset_stats()
    if not a.no_kernel_timing:
        # the dominant kernel, bracketed live in the timed region on every 8th step (every 4th in a short run; an event
        # pair costs ~4 us of stream time: bracketing every launch slowed the timed region by 4 %, every 4th by 1.5 %)
        env.time_kernels(True, groups=("eval",), stride=8 if a.steps >= 80 else 4)
        env.kernel_times()

    def timed_region():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.run_greedy(a.steps, precision=prec)   # EXACTLY a.steps env steps (one call: consecutive steps share launches)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0             # this rank's steps are done; the MAX over ranks is taken below (aggregate)
        if use_dist:
            dist.barrier()                        # the closing barrier of the bracket: its own latency (an RCCL all-reduce + a host
        return dt                                 # round trip, ~0.4 ms of a 3 ms region) is not part of anybody's K steps

    # A region shorter than 10 ms (the driver's --steps 20 is 3 ms) is at the mercy of one scheduling hiccup: it is
    # then repeated -- every repetition is again EXACTLY a.steps steps between barrier + synchronize -- and the MEDIAN
    # region is the one reported (all ranks take the same decision from rank 0's first region).
    regions = [timed_region()]
    n_rep = torch.tensor([14 if regions[0] < 0.010 else 0], dtype=torch.int64, device=dev if a.dist_backend == "nccl" else "cpu")
    if use_dist:
        dist.broadcast(n_rep, src=0)
    for _ in range(int(n_rep.item())):
        regions.append(timed_region())
    elapsed = sorted(regions)[len(regions) // 2]

    kt = env.kernel_times() if not a.no_kernel_timing else None
    env.time_kernels(False)
    choice = env.kernel_choice()                 # what the timed region's steps launched (asked of the library, not read from the environment)
    st = env.stats()
    # the counters cover every region; all regions do the same amount of work (auto-reset keeps every lane live), so
    # the median region's share is 1 / len(regions) of each
    st = {k: (v / len(regions) if k != "error_flags" else v) for k, v in st.items()}
    if kt is not None:
        # the other kernel groups: a short extra pass on the same env (bracketing every group costs ~20 us per step,
        # which would distort the timed region; the dominant kernel's figure above comes from the timed region itself)
        env.time_kernels(True)
        env.kernel_times()
        env.run_greedy(50, precision=prec)
        kt2 = env.kernel_times()
        env.time_kernels(False)
        for k in kt2:
            if k != "eval":
                kt[k] = kt2[k]

    cdev = dev if a.dist_backend == "nccl" else None
    tot, t_max = aggregate({k: int(round(st[k])) for k in ("steps", "games_finished", "candidates_raw", "rows_evaluated", "ksteps_executed")},
                           elapsed, device=cdev)
    ranks_seen, per_rank_s = census(elapsed, device=cdev)      # how many ranks this process group really has, and each one's median region
    def build_line():
        """rank 0: everything of the line but the training round and the CPU baseline"""
        u_step, c_step = distinct_ratio(env, prec, samples=1 if a.quick else 4)
        # the other value-net modes on the same env (untimed region of the contract: extra information only)
        alt = {}
        if world == 1 and not a.quick:
            for name, pm in (("f32", bg.F32), ("f32_dense", bg.F32_DENSE), ("f16x2", bg.F16X2), ("bf16", bg.BF16)):
                if name == a.precision:
                    continue
                env.run_greedy(10, precision=pm)
                torch.cuda.synchronize()
                s0, t1 = env.stats()["steps"], time.perf_counter()
                env.run_greedy(100, precision=pm)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                alt[name] = {"env_steps_per_s": round((env.stats()["steps"] - s0) / dt, 1), "ms_per_step": round(10 * dt, 4)}
            alt["note"] = ("f32_dense = dense v_mfma_f32_32x32x2_f32 chain over every afterstate (the r01 v6 headline path); "
                           "f16x2 = W1 as f16 hi+lo (22 mantissa bits), exact products, fp32 accumulate: max |value - reference| 3e-7, "
                           "inside the 1e-5 parity bound like f32; bf16 = speed mode outside it (1.2e-3)")
        out = {
            "metric": "self-play env steps/sec @65k concurrent games", "value": round(tot["steps"] / t_max, 1),
            "unit": "env steps/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * t_max / a.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if a.precision == "f32_dense" else a.precision, "data": "synthetic",
            "config": {"workload": "config3: 65 536 concurrent games per MI355X, greedy 198->128->1 value net "
                                   "(tdgammonNEW100k weights), auto-reset, Philox dice",
                       "games_per_gpu": a.games, "burnin_steps": a.burnin, "parallelism": f"shard{world}",
                       "raw_candidates_per_step": round(c_step, 2), "distinct_afterstates_per_step": round(u_step, 2),
                       "staged_leaves_per_step": round(tot["candidates_raw"] / max(tot["steps"], 1), 2),
                       "rows_evaluated_per_step": round(tot["rows_evaluated"] / max(tot["steps"], 1), 2),
                       "games_finished": tot["games_finished"]},
            # the timed region, as measured: every region is EXACTLY `steps` steps; more than one when the first was < 10 ms
            "timed_regions": len(regions), "region_ms": {"min": round(1e3 * min(regions), 4), "median": round(1e3 * elapsed, 4),
                                                         "max": round(1e3 * max(regions), 4)},
            "source_hash": bg._capi.source_hash(), "library_build": bg._capi.load().bgamd_build_flags().decode(),
        }
        if kt:
            nl = a.steps                                  # launches of the value-net kernel in the timed region (one per step)
            # per-step GPU time of each kernel group: the value net as the mean of its bracketed launches in the timed
            # region; the others from the short extra pass (a group may be bracketed more than once per step)
            per = {k: (v["ms"] / (v["launches"] if k == "eval" else max(kt2["eval"]["launches"], 1)) if v["launches"] else 0.0)
                   for k, v in kt.items()}
            rows_l, raw_l, steps_l = st["rows_evaluated"] / nl, st["candidates_raw"] / nl, st["steps"] / nl
            fn_l, dn_l = st["leaf_parent_nodes"] / nl, st["doubles_inner_nodes"] / nl
            # distinct afterstates per launch: U per step sampled on 4 x 2 048 positions after the run, never more than the
            # rows the launch actually evaluated (every distinct afterstate is one of them)
            u_l = min(steps_l * u_step, rows_l)
            eval_tf = u_l * FLOP_PER_ROW / (per["eval"] * 1e-3) / 1e12 if per["eval"] else 0.0
            # algorithmic bytes: leaves = per leaf-parent 8 B node + 44 B state gather, per distinct
            # afterstate 40 B out; expand = per game 44 B in + per node 8 B out/in; apply = 52 B in + 60 B out per game
            merged = choice["expand"] == "expand_all_kernel"      # one launch for the doubles plies AND the leaf stage: both byte counts, one time
            dbl_bytes = (fn_l + 2 * dn_l) * 8 + dn_l * 44
            leaves_gbs = (fn_l * 52 + u_l * 40 + (dbl_bytes if merged else 0)) / (per["leaves"] * 1e-3) / 1e9 if per["leaves"] else 0.0
            expand_gbs = (steps_l * 52 + dbl_bytes) / (per["expand"] * 1e-3) / 1e9 if per["expand"] and not merged else 0.0
            ks_l = st.get("ksteps_executed", 0) / nl
            peak = PEAK["f32" if a.precision == "f32_dense" else a.precision]
            ev = {"bound": "mfma", "achieved": round(eval_tf, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(eval_tf / peak, 4),
                  "traffic": None, "avg_ms": round(per["eval"], 4), "rows_per_launch": int(rows_l), "distinct_per_launch": int(u_l)}
            if a.precision == "f32":
                # The incremental evaluator executes no MFMA: its work is fp32 VALU arithmetic on W1 columns gathered from
                # LDS, so THAT is the roof it is measured against: executed flop / kernel time over the fp32 vector peak.
                # (The dense-equivalent figure -- SURVEY 8d's 50 944 flop per distinct afterstate -- stays as
                # dense_equiv_tflops: it says how much faster than a perfect dense fp32 MFMA evaluation the stage is, and
                # is not a fraction of anything.)  The per-game root pass is its own kernel (timed in slot "root").
                exec_tf = (ks_l * FLOP_PER_COLUMN + rows_l * FLOP_PER_ROW_EPILOGUE) / (per["eval"] * 1e-3) / 1e12 if per["eval"] else 0.0
                root_inside = choice["root"] == "inside boundary_kernel<true>"     # no launch of its own: the boundary launch's time holds it
                root_tf = steps_l * FLOP_PER_ROW / (per["root"] * 1e-3) / 1e12 if per.get("root") and not root_inside else 0.0
                stage_ms = per["eval"] + (0.0 if root_inside else per.get("root", 0.0))
                lds_tbps = ks_l * 512 / (per["eval"] * 1e-3) / 1e12 if per["eval"] else 0.0
                occ = {}
                for occ_file in ("r05_valu_occupancy.json", "r04_valu_occupancy.json", "r03_valu_occupancy.json", "r02_valu_occupancy.json"):
                    try:                                      # VALU issue occupancy from the COMMITTED SQ counters (tools/valu_occupancy.py): a number
                        occ = json.load(open(os.path.join(ROOT, "profiles", occ_file)))     # of another run of this kernel, not of this one
                        occ["source"] = "profiles/%s: %s" % (occ_file, occ.get("source", ""))
                        break
                    except Exception:
                        continue
                mdelta = choice["eval"] == "eval_rows_mdelta_kernel"      # round 3's kernel for the same stage (experimental build + BGAMD_MFMA_DELTA=1)
                ev = {"bound": "valu", "achieved": round(exec_tf, 3), "peak": PEAK_VALU_F32, "unit": "TFLOP/s",
                      "frac": round(exec_tf / PEAK_VALU_F32, 4), "traffic": None, "avg_ms": round(per["eval"], 4),
                      "rows_per_launch": int(rows_l), "distinct_per_launch": int(u_l),
                      "kernel": choice["eval"], "w1_columns_per_row": round(ks_l / max(rows_l, 1), 3),
                      "flop_per_launch": int(ks_l * FLOP_PER_COLUMN + rows_l * FLOP_PER_ROW_EPILOGUE),
                      "dense_equiv_tflops": round(eval_tf, 2), "dense_equiv_vs_f32_mfma_peak": round(eval_tf / peak, 3),
                      # second resource: one 512-byte W1 column per (row, changed feature) out of LDS, ds_read_b128
                      "lds_gather_GB_per_launch": round(ks_l * 512 / 1e9, 3), "lds_gather_TBps": round(lds_tbps, 2),
                      "lds_peak_TBps": PEAK_LDS_TBPS, "lds_frac": round(lds_tbps / PEAK_LDS_TBPS, 4),
                      "valu_issue_occupancy": None if mdelta else occ.get("eval_rows_delta_kernel", {}).get("valu_issue_occupancy"),
                      "valu_issue_occupancy_source": None if mdelta else occ.get("source"),
                      "root_pass_kernel": choice["root"], "root_pass_on_second_stream": choice["root_on_second_stream"],
                      "root_pass_avg_ms": None if root_inside else round(per.get("root", 0.0), 4),
                      "root_pass_tflops": None if root_inside else round(root_tf, 2),
                      "root_pass_frac_of_f32_mfma_peak": None if root_inside else round(root_tf / peak, 4),
                      "value_net_stage_ms": round(stage_ms, 4),
                      "value_net_stage_dense_equiv_tflops": round(u_l * FLOP_PER_ROW / (stage_ms * 1e-3) / 1e12, 2) if stage_ms else None,
                      "note": "achieved = fp32 operations the kernel executes (128 FMAs per row and changed feature + 646 per row of "
                              "sigmoids and output unit) / kernel time, peak = fp32 vector peak: the kernel runs on the VALUs (no MFMA, "
                              "HBM at ~1.2 TB/s).  Its issue slots also carry what is not a flop -- LDS address arithmetic, list decoding, "
                              "half-rate transcendentals (8-cycle issue) -- which is why the VALU issue occupancy from the SQ counters "
                              "(valu_issue_occupancy, tools/valu_occupancy.py over profiles/) is far above frac."}
            else:
                exec_tf = ks_l * 4 * 4096 / (per["eval"] * 1e-3) / 1e12 if per["eval"] and a.precision == "f32_dense" else None
                ev.update({"kernel": choice["eval"],
                           "executed_mfma_tflops": round(exec_tf, 2) if exec_tf else None,
                           "executed_frac_of_peak": round(exec_tf / peak, 4) if exec_tf else None,
                           "live_ksteps_frac": round(ks_l / (max(rows_l, 1) / 32 * 99), 4) if exec_tf else None,
                           "note": "achieved = 50 944 flop x distinct afterstates / kernel time (SURVEY 8d); frac can exceed 1 because the "
                                   "kernel skips k-steps whose features are zero in every row of a tile -- executed_* is the MFMA work issued"})
            roofs = {
                "eval": ev,
                "leaves": {"kernel": "expand_all_kernel (doubles plies 2+3 and every leaf stage)" if merged else "expand_kernel<LEAF>", "bound": "hbm", "achieved": round(leaves_gbs, 2), "peak": PEAK["hbm"],
                           "unit": "GB/s", "frac": round(leaves_gbs / PEAK["hbm"], 5), "traffic": None, "avg_ms": round(per["leaves"], 4)},
                "expand": {"kernel": "(inside expand_all_kernel)" if merged else "doubles_kernel (plies 2+3 of the doubles turns)", "bound": "hbm", "achieved": round(expand_gbs, 2),
                           "peak": PEAK["hbm"], "unit": "GB/s", "frac": round(expand_gbs / PEAK["hbm"], 5), "traffic": None,
                           "avg_ms": round(per["expand"], 4)},
            }
            # HBM traffic per launch from the committed PMC passes (separate rocprofv3 --pmc runs; bench.py cannot
            # collect counters itself) -- profiles/r01_pmc_traffic.json, corrected as MI355X_MICROARCH.md prescribes
            for pmc_file in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))["kernels"]
                    for name, key in (("eval", roofs["eval"]["kernel"]), ("leaves", "expand_all_kernel" if merged else "expand_kernel<3>")):
                        if key in pmc:
                            roofs[name]["traffic"] = round(pmc[key]["traffic_MB"] * 1e6)
                            roofs[name]["traffic_source"] = f"profiles/{pmc_file} (bytes per launch; a committed PMC pass of this kernel, not measured in this run)"
                    break
                except Exception:
                    continue
            dom = max(roofs, key=lambda k: roofs[k]["avg_ms"])
            out["roofline"] = roofs[dom]
            out["kernels"] = dict(roofs, apply_avg_ms=round(per["apply"], 4),
                                  gpu_ms_per_step=round(per["eval"] + per.get("root", 0.0) + per["leaves"] + per["expand"] + per["apply"], 4))
            if a.precision == "f32" and occ:
                # the STEP's VALU issue occupancy: the three launches of a fused step, each one's occupancy weighted by its own duration in
                # the committed SQ pass it comes from (a number of that pass, like valu_issue_occupancy above; not measured in this run)
                parts = [occ.get(k) for k in ("expand_all_kernel", "eval_rows_delta_kernel", "boundary_kernel<true>")]
                if all(parts):
                    tt = sum(x["duration_us"] for x in parts)
                    out["kernels"]["valu_issue_occupancy_step"] = round(sum(x["duration_us"] * x["valu_issue_occupancy"] for x in parts) / tt, 3)
                    out["kernels"]["valu_issue_occupancy_step_source"] = occ.get("source")
        if alt:
            out["alt_modes"] = alt
        if world > 1:
            ms = [1e3 * t / a.steps for t in per_rank_s]
            out["per_rank_ms_per_step"] = {"min": round(min(ms), 4), "max": round(max(ms), 4), "by_rank": [round(x, 4) for x in ms]}
            out["dist_backend"] = a.dist_backend
            out["launched_by"] = "bench.py launch_ranks" if os.environ.get("BENCH_LAUNCHED_BY") else "external launcher"
        return out

    out = build_line() if rank == 0 else None
    # configs 4/5's per-GPU share, on EVERY rank (its replay's collectives are joined by all of them); outside the contract's bracket, and
    # never at the cost of the line: an exception becomes {"error": ...}, and a watchdog thread ends a round that does not finish (a rank
    # waiting in a collective another rank never joined) -- rank 0 prints the line it already has, every rank leaves with os._exit
    tr = None
    if a.training_round and a.games >= 4096:
        finished = threading.Event()
        line_lock = threading.Lock()                            # the watchdog and the main thread never both decide to print

        def give_up():
            with line_lock:
                if finished.is_set():
                    return
                msg = f"training round did not finish within {a.training_round_timeout:.0f} s on rank {rank}: abandoned (the timed region above is complete)"
                sys.stderr.write("bench.py: " + msg + "\n")
                sys.stderr.flush()
                if rank == 0:
                    out["training_round"] = {"error": msg}
                    print(json.dumps(out), flush=True)
                os._exit(0)
        dog = threading.Timer(a.training_round_timeout, give_up)
        dog.daemon = True
        dog.start()
        # (ranks that share a GPU -- the gloo rehearsal -- take turns on it and pay a host round trip per all-reduce: the rehearsal exercises the
        #  route on a smaller round; a measurement it is not)
        rehearsal = world > 1 and a.dist_backend != "nccl"
        tr_games = min(a.games, 4096) if rehearsal else a.games
        try:
            tr = training_round(bg, tr_games, w, rank, world, a.dist_backend, rehearsal=rehearsal)
        except Exception as e:
            import traceback
            traceback.print_exc()
            tr = {"error": repr(e)[:300]}
        if use_dist and "error" not in tr:
            dist.barrier()                                      # under the watchdog too: every rank is out of the round
        with line_lock:
            finished.set()
        dog.cancel()
    tr_failed = tr is not None and "error" in tr                # (the process group may be gone with the rank that failed: no more collectives)

    def leave():
        if use_dist and not tr_failed:
            dist.barrier()
            dist.destroy_process_group()
    if rank != 0:
        leave()
        return

    if tr is not None:
        out["training_round"] = tr
    if world == 1 and not a.no_cpu_baseline and not a.quick:
        out["cpu_baseline"] = cpu_baseline(w)
    print(json.dumps(out), flush=True)
    leave()


if __name__ == "__main__":
    main()
