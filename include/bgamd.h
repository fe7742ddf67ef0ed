/* bgamd.h -- C ABI of the MI355X-native batched backgammon env step (libbgamd.so).
 *
 * This is the drop-in boundary for ONE hot path of romanoshiliarhopoulos/Backgammon-Engine:
 * the self-play env step.  Each entry point names the reference interface it replaces
 * (paths relative to the reference root).  The reference crosses Python<->C++ through the
 * pybind11 module `backgammon_env` (cppsrc/backgammon_bindings.cpp:41-94); a maintainer binds
 * these symbols instead (ctypes stub in INTEGRATION.md, shipped as
 * backgammon-engine_amd/backgammon_env/).
 *
 * Conventions
 *   - plain C, no torch types.  Every `d_*` pointer is a DEVICE pointer owned by the caller
 *     (e.g. torch.Tensor.data_ptr()); `h_*` pointers are host pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All work is
 *     stream-ordered; nothing synchronises unless stated.
 *   - return value: 0 = ok, negative = BGAMD_E_* (bgamd_error_string()).
 *   - a state is int32[28] = [board24 (+P1/-P2), bar1, bar2, off1, off2]   (game.hpp:16-28);
 *     |count| <= 15.  turn: 0 = PLAYER1, 1 = PLAYER2 (player.hpp:14-18).
 *   - one host thread drives one env; one env lives on one device.
 */
#ifndef BGAMD_H
#define BGAMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bgamd_env bgamd_env;

enum {
    BGAMD_OK = 0,
    BGAMD_E_INVALID = -1,     /* bad argument                                         */
    BGAMD_E_HIP = -2,         /* a HIP runtime call failed (see bgamd_last_hip_error) */
    BGAMD_E_NODEVICE = -3,    /* no usable gfx950 device                              */
    BGAMD_E_ARENA = -4,       /* candidate arena overflow (raise arena_rows)          */
    BGAMD_E_STATE = -5,       /* a state had |count| > 15                             */
    BGAMD_E_NOWEIGHTS = -6,   /* greedy step / evaluate before bgamd_env_load_weights */
    BGAMD_E_DELTA = -7,       /* an afterstate differed from its root position in more features than a legal turn
                                 can change (incremental value net): the rows handed to it are not afterstates  */
    BGAMD_E_WEIGHTS = -8      /* a weight is not finite, or an fc1 weight does not fit the f16 hi + lo planes of the value
                                 net's root pass (|w| >= 65 504): nothing was loaded                            */
};

/* step flags */
enum {
    BGAMD_ROLL = 1,           /* draw this turn's dice from the Philox stream (Game::rollDice,
                                 game.cpp:665-670); without it the dice set by
                                 bgamd_env_set_dice are used (Game::setDice, game.cpp:9-12)     */
    BGAMD_AUTO_RESET = 2,     /* a finished game restarts (new episode, opening roll);
                                 without it a finished game stays finished and is skipped      */
    BGAMD_NO_FLIP = 4,        /* do not flip the turn / advance ply (make_move semantics,
                                 model.py:180-222, for the scalar Game surface)                */
    BGAMD_WANT_INDEX = 8,     /* greedy step: also report the chosen move's index into the
                                 reference-order list and the list length (walks the whole list
                                 per lane: parity/debug use, not the throughput path)           */
    BGAMD_ONLY_P1 = 16,       /* only lanes with PLAYER1 / PLAYER2 to move take part in this call  */
    BGAMD_ONLY_P2 = 32,       /*   (head-to-head play of two policies, train.py:262-277)           */
    BGAMD_WEIGHTS_SLOT1 = 64  /* greedy step evaluates with weight slot 1 instead of slot 0        */
};

/* value-net arithmetic.  F32: fp32-grade values (inside the 1e-5 parity bound, measured 1.8e-7).  In a greedy step the
 * hidden layer is evaluated INCREMENTALLY: one dense pass per game for the root position (W1 split exactly into three
 * bf16 terms on v_mfma_f32_32x32x16_bf16, fp32 accumulation; BGAMD_ROOT_F32=1 at env creation: the f32 MFMA chain
 * v_mfma_f32_32x32x2_f32 instead), then  a_row = a_root + Σ Δx_f · W1[:, f]  in fp32 FMAs over the few features an
 * afterstate changes (same real sum as the dense chain, different association, values agree to ~1e-7).
 * BGAMD_MFMA_DELTA=1 at env creation: that sum on v_mfma_f32_32x32x16_f16 over a K-compacted fixed-point W table
 * instead (csrc/bg_eval_mfma.h: exact sums, values within 1.4e-6; measured slower, opt-in).  F32_DENSE: the dense MFMA chain over every afterstate (what bgamd_evaluate always uses).  F16X2: W1 split into f16 hi + lo
 * (22 mantissa bits), exact products, fp32 accumulation on v_mfma_f32_32x32x16_f16 -- fp32-grade values (inside
 * the 1e-5 parity bound) on the fast matrix pipe.  BF16: single bf16 weights, speed mode outside the bound. */
enum { BGAMD_F32 = 0, BGAMD_BF16 = 1, BGAMD_F16X2 = 2, BGAMD_F32_DENSE = 3 };

int bgamd_version(void);
/* hex digest of the sources (csrc + this header) the library was compiled from: the Python binding compares it with
 * the sources on disk and refuses a stale build; bench.py prints it */
const char *bgamd_source_hash(void);
const char *bgamd_error_string(int code);
const char *bgamd_last_hip_error(void);
int bgamd_device_count(void);

/* ---- env lifetime ---------------------------------------------------------------------
 * Replaces: Game::Game(int) game.cpp:44-56 for n_games boards at once.
 * Lane g plays game_id = lane_offset + g + episode * lane_stride; dice/choice words are
 * Philox4x32-10(key = seed, counter = (game_id, ply, stream)) so a shard's games do not
 * depend on how many shards there are.  arena_rows = capacity of the candidate arena
 * (0 = default 256 rows per game, min 65 536).  The greedy step with the incremental value net (BGAMD_F32) uses it as four
 * arenas of arena_rows / 4 (non-doubles / doubles turn x no blot hit / hit): BGAMD_E_ARENA is raised when ONE of them runs
 * over in a step -- the distinct afterstates of a step average 17 per game over all four, 64 per game and arena are the default. */
int bgamd_env_create(bgamd_env **out, int64_t n_games, int device, uint64_t seed,
                     uint64_t lane_offset, uint64_t lane_stride, int64_t arena_rows);
int bgamd_env_destroy(bgamd_env *env);
int64_t bgamd_env_num_games(const bgamd_env *env);

/* All lanes: episode 0, ply 0, start position (Game::populateBoard game.cpp:240-252), turn from
 * the opening roll protocol of play_game (train.py:89-97) on the OPENING stream. */
int bgamd_env_reset(bgamd_env *env, void *stream);
/* The same for episode `episode` of every lane (game_id = lane_offset + lane + episode * lane_stride): a training
 * loop that plays one game per lane and round gives every round fresh dice this way -- play_game draws new dice for
 * every game (train.py:64-121, 527-547).  Counters are kept. */
int bgamd_env_reset_episode(bgamd_env *env, uint32_t episode, void *stream);
/* Gives an existing env the dice streams a freshly created one would have (seed, lane_offset, lane_stride as in
 * bgamd_env_create; lane_stride 0 = n_games) and resets it to episode 0.  Pooled one-lane envs of the scalar Game surface
 * (the reference constructs a Game per candidate, game.cpp:68-77) are re-seeded with it, so set_seed() decides the dice of
 * every Game created afterwards whether its env is new or comes from the pool. */
int bgamd_env_reseed(bgamd_env *env, uint64_t seed, uint64_t lane_offset, uint64_t lane_stride, void *stream);
/* Only the lanes with d_mask[g] != 0 restart, as the next episode of that lane (start position, opening roll of
 * the next global game id) -- what BGAMD_AUTO_RESET does for a finished game, on demand. */
int bgamd_env_reset_lanes(bgamd_env *env, const int32_t *d_mask /*[n]*/, void *stream);

/* ---- state access (getGameBoard/getJailedCount/getBornOffCount/getTurn/setGameBoard/
 *      setBorneOffPieces/setTurn, bindings.cpp:64-85) ------------------------------------- */
int bgamd_env_set_states(bgamd_env *env, const int32_t *d_states28, const int32_t *d_turn, void *stream);
int bgamd_env_get_states(bgamd_env *env, int32_t *d_states28, int32_t *d_turn, void *stream);
/* is_game_over (bindings.cpp:11-16) on the current boards: bit0 = over, bit1 = winner (0/1),
 * bit2 = lane frozen (finished without AUTO_RESET); bits4-5 = over/winner flags of the last step
 * (set even when the lane was auto-reset). d_states28 may be NULL in set_states (turn only). */
int bgamd_env_get_flags(bgamd_env *env, int32_t *d_flags, void *stream);
/* everything the scalar getters of bindings.cpp:64-93 read, in ONE launch: d_out[n][32] = state28 | turn | die1 | die2 |
 * flags (as bgamd_env_get_flags) */
int bgamd_env_snapshot(bgamd_env *env, int32_t *d_out, void *stream);
int bgamd_env_set_dice(bgamd_env *env, const int32_t *d_dice /*[n,2]*/, void *stream);   /* setDice      */
int bgamd_env_get_dice(bgamd_env *env, int32_t *d_dice /*[n,2]*/, void *stream);         /* get_last_dice */
/* roll_dice: dice of (game_id, ply) on the TURN stream; advance_ply != 0 post-increments ply so that
 * repeated calls draw fresh dice (the scalar Game surface), 0 leaves ply to the step functions. */
int bgamd_env_roll(bgamd_env *env, int advance_ply, void *stream);

/* ---- the scalar Game surface with HOST arguments, for ONE-LANE envs (n_games == 1): what a compiled binding of the
 * reference's module calls per method of bindings.cpp:62-93 (backgammon-engine_amd/pybind/backgammon_env_pybind.cpp is that
 * binding; the Python package's Game uses the same entry points).  Plain ints and host pointers in and out; each call
 * stages its arguments on the device, runs the same kernels as the vectorised entry points and synchronises.
 *   snapshot   : h_out[32] = state28 | turn | die1 | die2 | flags (as bgamd_env_get_flags)  -- getGameBoard, getTurn,
 *                getJailedCount, getBornOffCount, get_last_dice, is_game_over
 *   set_state  : h_state28 (or NULL = keep the board) and turn (-1 = keep)                    -- setGameBoard, setBorneOffPieces, setTurn
 *   set_dice / roll (roll draws the lane's next Philox dice and returns them)                -- setDice, roll_dice
 *   legal_moves: returns the number of (origin, dest) pairs written to h_pairs[26][2]         -- legalMoves (game.cpp:80-105)
 *   try_move   : returns 0 = moved, 1..7 = the reference's message in source order            -- tryMove (game.cpp:573-663)
 *   enumerate  : returns the number C of sequences of legalTurnSequences / evaluateTurnSequences (game.cpp:134-222) in
 *                reference order and fills the first min(C, cap) rows of h_states28[.][28], h_seq[.][4][2], h_len[.]
 *                (any may be NULL); call with cap = 0 for the count alone */
int bgamd_game_snapshot(bgamd_env *env, int32_t h_out[32]);
int bgamd_game_set_state(bgamd_env *env, const int32_t *h_state28, int turn);
int bgamd_game_set_dice(bgamd_env *env, int d1, int d2);
int bgamd_game_roll(bgamd_env *env, int32_t h_dice[2]);
int bgamd_game_legal_moves(bgamd_env *env, int player, int die, int8_t *h_pairs);
int bgamd_game_try_move(bgamd_env *env, int player, int dice, int origin, int dest);
int64_t bgamd_game_enumerate(bgamd_env *env, int player, int d1, int d2, int32_t *h_states28, int8_t *h_seq, int32_t *h_len,
                             int64_t cap);

/* ---- enumeration: Game::evaluateTurnSequences for every lane (game.cpp:193-222) -----------
 * Fills the env's candidate arena in REFERENCE ORDER (duplicates kept) for the lanes' current
 * turn and dice, or for the explicit (player, d1, d2) arguments of the reference call.  Afterwards:
 *   bgamd_env_candidates_info : d_offsets[n] (first row of lane g), d_counts[n]; returns total rows
 *                               (synchronises the stream)
 *   bgamd_env_candidates_read : rows [first, first+n_rows) unpacked to int32 states [n_rows,28] and
 *                               int8 sequences [n_rows,4,2] ((origin,dest), -1 padded), lengths. */
int bgamd_env_enumerate(bgamd_env *env, const int32_t *d_player /*[n] or NULL*/, const int32_t *d_dice /*[n,2] or NULL*/,
                        void *stream);
int64_t bgamd_env_candidates_info(bgamd_env *env, int64_t *d_offsets, int32_t *d_counts, void *stream);
int bgamd_env_candidates_read(bgamd_env *env, int64_t first, int64_t n_rows, int32_t *d_states28,
                              int8_t *d_seq, int32_t *d_seq_len, void *stream);

/* ---- the env step ---------------------------------------------------------------------------
 * One turn of every live lane: (roll) -> enumerate -> choose -> apply -> terminal check ->
 * (auto-reset | flip turn), i.e. the body of play_game's loop (train.py:103-121).
 *   random: uniform over the reference-order list, k = (u32 * C) >> 32 with u32 from d_choice_u32
 *           (per lane) or, when NULL, from the Philox TURN stream         (benchmark.py:54-61)
 *   greedy: TDLGammonModel.make_move (model.py:180-222): value net on every afterstate with the
 *           MOVER's turn bit, argmax (P1) / argmin (P2), first index wins ties; epsilon > 0 draws
 *           the exploration test and index from the TURN stream.  precision BGAMD_F32 | BGAMD_BF16.
 * Per-lane results of the last step: chosen reference-order index (-1 = no move), sequence, candidate
 * count, value of the chosen afterstate.  The greedy step reports index and count exactly only with
 * BGAMD_WANT_INDEX; without it index is 0 and count 1 when a move was made (-1 / 0 when none was). */
int bgamd_env_step_random(bgamd_env *env, int flags, const uint32_t *d_choice_u32, void *stream);
/* same result through the one-lane-per-game walk of the whole sequence tree (slow tail; cross-check) */
int bgamd_env_step_random_walk(bgamd_env *env, int flags, const uint32_t *d_choice_u32, void *stream);
int bgamd_env_load_weights(bgamd_env *env, const float *h_weights /* 25601: W1[128][198] b1 W2 b2 */);   /* slot 0 */
int bgamd_env_load_weights_slot(bgamd_env *env, int slot /* 0 | 1 */, const float *h_weights);
/* host only (no device needed): would bgamd_env_load_weights accept these 25 601 floats?  BGAMD_OK | BGAMD_E_WEIGHTS.  The reference's
 * model takes any fp32 state_dict (model.py:36-37); the value net's root pass here multiplies fc1.weight as f16 hi + f16 lo planes, exact
 * for 22 mantissa bits up to |w| < 65 504 -- a table beyond that (a diverged run) is refused loudly instead of evaluated as NaN. */
int bgamd_weights_check(const float *h_weights);
int bgamd_env_step_greedy(bgamd_env *env, int flags, float epsilon, int precision, void *stream);
/* n_steps greedy steps back to back (the loop body of play_game, train.py:103-121, n_steps times for every lane),
 * identical in effect to n_steps calls of bgamd_env_step_greedy; between two steps of a run the apply of one and the
 * roots of the next share a launch. */
int bgamd_env_run_greedy(bgamd_env *env, int flags, float epsilon, int precision, int64_t n_steps, void *stream);
int bgamd_env_last_choice(bgamd_env *env, int32_t *d_chosen, int32_t *d_count, int8_t *d_seq /*[n,4,2]*/,
                          int32_t *d_seq_len, float *d_value, void *stream);

/* (slot 8 below counts the 32-row x 2-feature MFMA steps of the dense f32 net, or the W1 columns added by the
 * incremental one) */
/* counters since create/reset_stats (synchronises): [steps, games_finished, p1_wins, candidates_raw,
 * rows_evaluated, error_flags, leaf_parent_nodes, doubles_inner_nodes, ksteps_executed (fp32 value net:
 * 32-row x 2-feature MFMA steps actually issued, of 99 per tile), reserved] */
int bgamd_env_stats(bgamd_env *env, uint64_t h_out[10]);
int bgamd_env_reset_stats(bgamd_env *env, void *stream);

/* single-checker surface for lane-wise moves: Game::tryMove (game.cpp:573-663) and
 * Game::legalMoves (game.cpp:80-105).  d_err: 0 ok, 1..7 = the reference's messages in source order. */
int bgamd_env_try_move(bgamd_env *env, const int32_t *d_player, const int32_t *d_dice, const int32_t *d_origin,
                       const int32_t *d_dest, int32_t *d_err, void *stream);
int bgamd_env_legal_moves(bgamd_env *env, const int32_t *d_player, const int32_t *d_die,
                          int32_t *d_n, int8_t *d_pairs /*[n,26,2]*/, void *stream);

/* diagnostics: (game, key | turn<<31) of every row the value net evaluated in the last greedy step (after the
 * pruning of commuting move orders: a few per cent of the rows are still copies); returns the row count (synchronises). */
int64_t bgamd_env_unique_rows_info(bgamd_env *env, void *d_info /* uint32[cap][2] */, int64_t cap, void *stream);
/* ... and the rows themselves with the value the net gave each: rows [first, first + n_rows) of that list as int32
 * states [n_rows][28] and float values [n_rows] (either may be NULL).  With BGAMD_F32 these are the outputs of the
 * incremental kernel, row by row -- what the parity tests compare with the reference model.  (Round 4: the step keeps its rows in four
 * arenas by kind of turn and by whether the moves hit a blot; both calls list them one arena after the other, i.e. `first` counts rows
 * of that list.  The order of the rows inside an arena is the order in which the leaf stage's workgroups allocated them: no meaning.) */
int bgamd_env_unique_rows_read(bgamd_env *env, int64_t first, int64_t n_rows, int32_t *d_states28, float *d_values,
                               void *stream);

/* ---- trajectory log for the learner (the list of encodings play_game returns, train.py:105-106,
 * kept as 32-byte rows: 8 bit planes, turn of the side to move in plane 0 bit 31).  When set, every
 * greedy step stores the PRE-move row of each live lane at d_rows[(ply*n + lane)*32 B]; plies >=
 * max_plies are dropped.  NULL disables.  bgamd_env_get_progress copies ply / episode per lane
 * (a finished, frozen lane has ply = T-1).  bgamd_encode_rows: rows -> float[n][198]. */
int bgamd_env_set_trajectory(bgamd_env *env, void *d_rows, int64_t max_plies);
int bgamd_env_get_progress(bgamd_env *env, int32_t *d_ply, int32_t *d_episode, void *stream);

/* ---- ring log of CONTINUOUS self-play ---------------------------------------------------------------------------------
 * play_game (train.py:64-121) plays one game to the end; a round of them (train.py:527-547) on n lanes ends with a long tail of
 * nearly empty steps (mean game 83 turns, longest ~450: a 65 536-game round is ~460 steps for 5.5 M turns).  With BGAMD_AUTO_RESET a
 * lane starts its next game (episode + 1 of the lane: fresh dice, fresh opening roll) the step after the last one ended, every lane
 * is busy at every step, and the log is indexed by the ENV STEP instead of the lane's ply:
 *   d_rows [ring_steps][n] x 32 B : the pre-move row of lane g at env step k (k counted from this call) sits in slot k % ring_steps
 *   d_end  [ring_steps][n] uint16 : 0, or -- when the turn logged in that slot was the LAST of its game -- the game's number of logged
 *                                   turns (<= 32 767) | winner << 15 (0 = PLAYER1).  The game's turns are the `turns` slots ending there.
 * Only greedy steps (bgamd_env_step_greedy / run_greedy) log into the ring; every lane must take part in every step (no
 * BGAMD_ONLY_P1/P2).  Every game is one episode of one lane: the same game bgamd_env_reset_episode + a run to the end plays with the
 * same weights.  A caller that refreshes the weights between runs lets the games in flight go on under the new ones (TD-Gammon's own
 * self-play changes the weights after every move).  NULL disables.  bgamd_env_trajectory_step: env steps logged since the call.
 * Enforced (round 5): with a ring set, a greedy step without BGAMD_AUTO_RESET or with BGAMD_ONLY_P1/P2 returns BGAMD_E_INVALID.
 * bgamd_env_trajectory_step is a HOST counter bumped when a run is ENQUEUED: read the end records on the stream the run was issued on (or
 * after synchronising it) -- ContinuousSelfPlay.finished() does the former. */
int bgamd_env_set_trajectory_ring(bgamd_env *env, void *d_rows, int64_t ring_steps, uint16_t *d_end);
int64_t bgamd_env_trajectory_step(const bgamd_env *env);
int bgamd_encode_rows(const void *d_rows, int64_t n, float *d_out198, void *stream);

/* ---- stateless operators ----------------------------------------------------------------------
 * _encode_states_np (model.py:111-144) and forward (model.py:63-67) on caller-provided states. */
int bgamd_encode(const int32_t *d_states28, const int32_t *d_turn, int64_t n, float *d_out198, void *stream);
int bgamd_evaluate(bgamd_env *env, const int32_t *d_states28, const int32_t *d_turn, int64_t n,
                   int precision, float *d_values, void *stream);                       /* weight slot 0 */
int bgamd_evaluate_slot(bgamd_env *env, int slot, const int32_t *d_states28, const int32_t *d_turn, int64_t n,
                        int precision, float *d_values, void *stream);
/* The same values through the INCREMENTAL fp32 path of the greedy step (make_move's forward over the afterstates of
 * a turn, model.py:209-211): afterstate i is evaluated as its root position's hidden layer (one dense pass per root,
 * root d_root_index[i] of the n_roots given with the MOVER's turn, model.py:209) plus the W1 columns of the features
 * in which it differs from that root.  A row that differs from its root in more features than a legal turn changes
 * raises BGAMD_E_DELTA at the next bgamd_env_stats.  n_roots <= the env's lanes, n <= its arena rows; the env's
 * arenas are used as scratch (do not interleave with a step in flight on another stream). */
int bgamd_evaluate_incremental(bgamd_env *env, int slot, const int32_t *d_root_states28, const int32_t *d_root_turn,
                               int64_t n_roots, const int32_t *d_states28, const int32_t *d_root_index, int64_t n,
                               float *d_values, void *stream);

/* kernel timing hook for bench.py: brackets kernel groups with HIP events on the launch stream;
 * bgamd_env_kernel_times returns accumulated milliseconds and launch counts since the last call
 * (synchronises).  slots: 0 ordered enumerate, 1 value net, 2 apply, 3 random step,
 * 4 roots+expand (plies 1-3), 5 leaf stage, 6 root term of the incremental value net when it is a launch of its own (the first step of a run; every step of an env
 * below 24 576 lanes (experimental build: or with BGAMD_ROOT_IN_BOUNDARY=0) -- otherwise it runs inside the boundary launch, slot 2; the experimental build's BGAMD_OVERLAP=1 forks the launch onto the env's
 * second stream beside the doubles plies, rounds 1-3's default).  enable: 0 = off, 1 = every group, (mask << 8) | (stride << 20) = only the groups
 * whose bit is set in mask, on every stride-th launch of a group (0 = every launch; an event pair costs ~4 us of
 * stream time, so a timed run samples). */
/* "default" or "experimental" (-DBGAMD_EXPERIMENTAL: the kernels that lost their A/B are compiled in and their switches honoured --
 * BGAMD_MFMA_DELTA, BGAMD_F16X2_RESIDENT, BGAMD_ROOT_RESIDENT=0, BGAMD_ROOT_F32, BGAMD_TD_FUSED=0; the default build ignores them).
 * bgamd_env_kernel_choice: what the last greedy step actually launched -- h_out[0] value net: 0 eval_rows_delta_kernel, 1 eval_rows_mdelta_kernel,
 * 2 eval_rows_f32_kernel, 3 eval_rows_f16x2_kernel, 4 eval_rows_d16_kernel, 5 eval_rows_bf16_kernel; h_out[1] root pass: 0 none, 1
 * root_hidden_resident_kernel, 2 root_hidden_bf16x3_kernel, 3 eval_rows_f32_kernel<root>, 4 no launch of its own: it ran inside the
 * boundary launch of the step before (boundary_kernel<true>, every step of a run but the first); h_out[2]: bit 0 = root pass on the env's
 * second stream, bit 1 = the expansion below the roots (doubles plies 2-3 + leaf stage) ran as ONE launch (expand_all_kernel; the default,
 * the experimental build's BGAMD_EXPAND_MERGED=0 brings doubles_kernel + expand_kernel<LEAF> back); h_out[3]: 1 = experimental build.  bench.py labels its kernels
 * from this, not from the environment. */
const char *bgamd_build_flags(void);
int bgamd_env_kernel_choice(bgamd_env *env, int32_t h_out[4]);
int bgamd_env_time_kernels(bgamd_env *env, int enable);
int bgamd_env_kernel_times(bgamd_env *env, double h_ms[8], uint64_t h_launches[8]);

/* states (28 ints, reference getGameBoard layout) + turn -> 32-byte rows, the trajectory log's format */
int bgamd_pack_rows(const int32_t *d_states28, const int32_t *d_turn, int64_t n, void *d_rows, void *stream);

/* ---- TD(lambda) learner -----------------------------------------------------------------------------
 * Replaces apply_td_updates (pysrc/TD(λ) model/train.py:124-172) with the eligibility traces of
 * model.py:48-53 (reset per game, train.py:539-540), as a lock-step replay: step t updates EVERY game that has a
 * turn t from the same weights, and the per-game updates  fp32(α δ_g) · e_g  are summed (one game = the
 * reference's own update, step for step; many games = mini-batch TD(λ), the documented deviation).
 * Weights: flat float[25601] = fc1.weight[128][198] | fc1.bias[128] | fc2.weight[128] | fc2.bias[1].
 *
 * begin : d_rows = the env's trajectory log [T][n_lanes] x 32 B; d_order = the n_games lanes to replay, ordered by
 *         DECREASING d_length[lane] (so the games still running at step t are a prefix of the order);
 *         d_length[lane] in 1..T = number of logged turns, d_p1_won[lane] = 1 if PLAYER1 won (the terminal target z,
 *         train.py:165).  The buffers must stay valid until the last step.  Traces start at zero.
 * step  : step t over the first n_active games of the order: forward of s_t and s_{t+1}, δ = V(s_{t+1}) - V(s_t)
 *         (z - V(s_t) on a game's last turn), e <- λ e + ∇V(s_t), update = Σ_g fp32(alpha · δ_g) e_g with alpha·δ
 *         formed in float64 (train.py:147).  d_update == NULL: the update is applied to the weights.  Otherwise it is
 *         written to d_update[25601] and NOT applied: the caller all-reduces it over the ranks (the one collective
 *         of a training step) and calls bgamd_td_apply.
 * replay: steps 0..n_steps-1 with h_n_active[t] games each, applied locally.
 * stats : Σ δ² and the number of (game, step) updates since begin (synchronises). */
typedef struct bgamd_td bgamd_td;
int bgamd_td_create(bgamd_td **out, int64_t max_games, int device);
int bgamd_td_destroy(bgamd_td *td);
int bgamd_td_set_weights(bgamd_td *td, const float *d_theta, void *stream);
int bgamd_td_get_weights(bgamd_td *td, float *d_theta, void *stream);
int bgamd_td_begin(bgamd_td *td, const void *d_rows, int64_t T, int64_t n_lanes, const int32_t *d_order,
                   int64_t n_games, const int32_t *d_length, const uint8_t *d_p1_won, void *stream);
/* Streamed replay of a round (replaces the loop over a round's games of train.py:536-547 for rounds far larger than the
 * reference's): n_slots slots replay the round's games one after another -- slot i plays the games (lanes of the log)
 * d_queue[d_queue_offsets[i] .. d_queue_offsets[i + 1]), a game's step 0 following the terminal step of the game before it in
 * the next training step -- so every training step sums the TD(lambda) updates of n_slots games at DIFFERENT plies, and the
 * round takes max_i (sum of slot i's game lengths) steps instead of (sub-rounds) x (longest game).  A slot's trace restarts
 * with each game (train.py:133 reset_eligibility_traces).  Then bgamd_td_step(t, n_slots, ...) for t = 0, 1, ... (steps past a
 * slot's last game add nothing for it); t is not bounded by T here.  One game per slot == bgamd_td_begin with that order. */
/* Host-only helper (no device needed): the schedule of a streamed replay.  h_length[n_lanes] = logged turns per lane (<= 0: not
 * replayed).  Games are dealt longest first, each to the slot with the fewest turns so far, and every slot plays its share in a fixed
 * pseudo-random order.  -> h_queue[games] lanes in slot-major play order, h_queue_offsets[n_slots + 1], the number of games and the
 * number of training steps (= the largest slot total).  bgamd_td_begin_stream takes these two arrays (copied to the device). */
int bgamd_td_stream_schedule(const int32_t *h_length, int64_t n_lanes, int64_t n_slots, int32_t *h_queue, int32_t *h_queue_offsets,
                             int64_t *h_n_games, int64_t *h_n_steps);
int bgamd_td_begin_stream(bgamd_td *td, const void *d_rows, int64_t T, int64_t n_lanes, const int32_t *d_queue,
                          const int32_t *d_queue_offsets, int64_t n_slots, const int32_t *d_length, const uint8_t *d_p1_won,
                          void *stream);
/* The same over a GAME TABLE and a ring log (continuous self-play, bgamd_env_set_trajectory_ring): queue entries are game ids
 * 0 .. n_games-1; game i sits in column d_game_lane[i] of the log, its first turn in ring slot d_game_start[i], its turn k in slot
 * (d_game_start[i] + k) % ring_steps; d_length / d_p1_won are indexed by game id.  bgamd_td_stream_schedule takes d_length as it is. */
int bgamd_td_begin_stream_games(bgamd_td *td, const void *d_rows, int64_t ring_steps, int64_t n_lanes, const int32_t *d_queue,
                                const int32_t *d_queue_offsets, int64_t n_slots, int64_t n_games, const int32_t *d_game_lane,
                                const int32_t *d_game_start, const int32_t *d_length, const uint8_t *d_p1_won, void *stream);
int bgamd_td_step(bgamd_td *td, int64_t t, int64_t n_active, double alpha, float lambda, float *d_update, void *stream);
int bgamd_td_apply(bgamd_td *td, const float *d_update, void *stream);
int bgamd_td_replay(bgamd_td *td, int64_t n_steps, const int64_t *h_n_active, double alpha, float lambda, void *stream);
/* Delayed update, opt-in (delay = 1; 0 = exact, the default): bgamd_td_replay applies the update of step t one step LATE -- step t + 1 runs on
 * the weights of step t plus the update of step t - 1 -- whenever the replay is a streamed one through a constant number of slots that takes the
 * fused training step with >= 201 workgroups (512 ... 4 096 slots on 256 CUs; anything else replays exactly).  The step's reduction then has a
 * whole step to happen in and a training step is ONE launch instead of two (csrc/bg_learner.h: the reduction of step t - 1 rides on the waves
 * that idle during step t's forward pass).  A documented deviation from train.py:136-170, where every update is applied before the next state
 * is evaluated: with thousands of games summed per step a one-step delay is a perturbation of the step ORDER, not of what is learned (quality:
 * profiles/r04_training_quality.txt) -- but ONE game no longer reproduces the reference's update step for step, so fixtures G6 / G9 hold for
 * delay = 0 only.  bgamd_td_step / bgamd_td_step_allreduce are always exact. */
int bgamd_td_set_delay(bgamd_td *td, int delay);
/* ---- the one collective of a multi-rank training step, issued by the library ("one all-reduce of 25 601 fp32 per
 * training step, RCCL over xGMI, in place, on the compute stream"; train.py:536-547 is the loop it scales).  The learner owns an RCCL
 * communicator: rank 0 calls bgamd_td_comm_unique_id, the 128 bytes reach the other ranks by whatever the launcher has
 * (torch.distributed broadcast, a file), every rank calls bgamd_td_comm_init (collective: returns when all ranks have).  RCCL is
 * resolved at run time (the librccl.so.1 already in the process, else the ROCm install's; BGAMD_RCCL_LIB overrides).
 *   step_allreduce  : bgamd_td_step with the update handed out -> ncclAllReduce(sum, in place) -> bgamd_td_apply, three enqueues on
 *                     `stream`, no host synchronisation; n_active = 0: the rank adds nothing at step t but joins the collective
 *   replay_allreduce: steps 0 .. n_steps-1 (the MAX over the ranks); this rank's own log covers the first n_own_steps of them */
int bgamd_td_comm_unique_id(uint8_t h_id[128]);
int bgamd_td_comm_init(bgamd_td *td, const uint8_t h_id[128], int rank, int world);
int bgamd_td_comm_destroy(bgamd_td *td);
int bgamd_td_step_allreduce(bgamd_td *td, int64_t t, int64_t n_active, double alpha, float lambda, void *stream);
int bgamd_td_replay_allreduce(bgamd_td *td, int64_t n_steps, const int64_t *h_n_active, int64_t n_own_steps, double alpha, float lambda,
                              void *stream);
/* (the readers below wait for the stream the replay was issued on, not for the device: a learner replaying on its own stream beside
 * an env at play does not wait for the env) */
int bgamd_td_stats(bgamd_td *td, double *h_sq_sum, int64_t *h_updates);
/* Traffic report of the column-sparse traces: Σ over the (game, step) updates since begin of the W1 trace columns that
 * were touched (of 198; a column = 128 floats, read + written).  BGAMD_TD_DENSE=1 in the environment at bgamd_td_create
 * keeps every column active (the dense pass: same results bit for bit, 198 columns per update).  Synchronises. */
int bgamd_td_active_columns(bgamd_td *td, uint64_t *h_columns);
/* ... and of the columns that were WRITTEN.  The stored trace is e / c with one scale c = Π λ for all games of the replay
 * (train.py:150-158's e <- λ e + ∇ becomes ê <- ê + ∇ / c), so a column whose feature is zero in s_t is read but not written;
 * c is folded back in by an ordinary pass when it leaves [2^-40, 2^40].  BGAMD_TD_LAZY=0 at bgamd_td_create: every step is an
 * ordinary pass (written = active).  Synchronises. */
int bgamd_td_written_columns(bgamd_td *td, uint64_t *h_columns);
/* Diagnostics: the replay's slots (lock-step: one per game) -> int32 h_out[n][6] = (lane, length, p1_won, start step) of the game
 * a slot holds (length 0: none), the slot's queue cursor (-1 in a lock-step replay), its (game, step) updates so far.  Synchronises. */
int bgamd_td_slots(bgamd_td *td, int32_t *h_out);
/* HIP-event time of the trace kernel since the last call: enable with bgamd_td_time(td, 1).  A step of 512 .. 4 096 slots runs its forward pass
 * INSIDE the trace launch (td_step_fused_kernel): the bracket then holds both, and bytes / time derived from it understate the trace pass
 * (BGAMD_TD_FUSE_STEP=0 at bgamd_td_create separates them again: forward kernel, trace kernel, reduce kernel). */
int bgamd_td_time(bgamd_td *td, int enable);
int bgamd_td_times(bgamd_td *td, double *h_trace_ms, uint64_t *h_launches, uint64_t *h_game_steps);

#ifdef __cplusplus
}
#endif
#endif
