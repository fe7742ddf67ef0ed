/* bg_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference self-play env step
 * (romanoshiliarhopoulos/Backgammon-Engine).  It exists to CHECK the HIP path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it.  Nothing under backgammon-engine_amd/ links, imports or calls it.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle_*.py)
 * against the reference's own known answers (cppsrc/tests.cpp) and against golden
 * vectors produced by the unmodified reference compiled into oracle/_ref/ (see
 * oracle/Makefile, tests/golden/make_golden.py).
 *
 * All file:line citations are relative to the reference repository root.
 */
#ifndef BG_ORACLE_H
#define BG_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Game state, cppsrc/game.hpp:37-44 + cppsrc/Pieces.hpp:8-12.
 * board[i] is point i+1: >0 PLAYER1 checkers, <0 PLAYER2 checkers. */
typedef struct bgo_state {
    int32_t board[24];
    int32_t bar[2];   /* "jailed"  Pieces::numPieces_p1/p2   */
    int32_t off[2];   /* "freed"   Pieces::freedPieces_p1/p2 */
    int32_t turn;     /* 0 = PLAYER1, 1 = PLAYER2            */
} bgo_state;

/* tryMove error codes, in the order the reference tests them (game.cpp:583-642). */
enum {
    BGO_OK = 0,
    BGO_ERR_INVALID_ORIGIN = 1,      /* "Invalid origin"                 */
    BGO_ERR_ORIGIN_RANGE = 2,        /* "Origin out of range"            */
    BGO_ERR_DEST_RANGE = 3,          /* "Destination out of range"       */
    BGO_ERR_DIRECTION = 4,           /* "Cannot move in that direction." */
    BGO_ERR_DICE_MISMATCH = 5,       /* "Move does not match dice."      */
    BGO_ERR_INVALID_DEST = 6,        /* "Invalid destination."           */
    BGO_ERR_BEAROFF_FROM_JAIL = 7    /* "Cannot bear off from jail"      */
};

#define BGO_MAX_SEQ_LEN 4

void bgo_init(bgo_state *s, int first_player);                 /* game.cpp:44-56,240-252 */
int  bgo_is_valid_origin(const bgo_state *s, int multi, int idx);              /* game.cpp:416-457 */
int  bgo_can_free_piece(const bgo_state *s, int multi, int dice, int origin);  /* game.cpp:488-557 */
int  bgo_is_valid_destination(const bgo_state *s, int multi, int idx, int dice, int origin); /* :459-485 */
int  bgo_legal_moves(const bgo_state *s, int player, int die, int32_t *out_pairs /* [26][2] */); /* :80-105 */
int  bgo_try_move(bgo_state *s, int player, int dice, int origin, int dest);   /* game.cpp:573-663 */
int  bgo_over(const bgo_state *s, int *winner);                                /* game.cpp:388-407 */

/* legalTurnSequences (game.cpp:134-191) + evaluateTurnSequences (game.cpp:193-222).
 * Writes up to `cap` entries in reference order; returns the full count C (may exceed cap).
 *   seq     [cap][4][2] int8  (origin,dest) pairs, unused slots = -1   (may be NULL)
 *   seq_len [cap]       int32                                           (may be NULL)
 *   states  [cap][28]   int32 afterstates [board24, bar1, bar2, off1, off2] (may be NULL) */
int64_t bgo_evaluate_turn_sequences(const bgo_state *s, int player, int d1, int d2,
                                    int64_t cap, int8_t *seq, int32_t *seq_len, int32_t *states);

/* 198-feature encoder, pysrc/TD(λ) model/model.py:111-144.  states [n][28] int32. */
void bgo_encode(const int32_t *states, int64_t n, int turn, float *out /* [n][198] */);

/* Value net forward, model.py:63-67.  Weights as flat fp32:
 * W1[128][198] | b1[128] | W2[128] | b2[1]  (the state_dict order, 25 601 floats). */
#define BGO_N_IN 198
#define BGO_N_HID 128
#define BGO_N_PARAMS (BGO_N_HID * BGO_N_IN + BGO_N_HID + BGO_N_HID + 1)
void bgo_forward_f32(const float *w, const float *x, int64_t n, float *out);
void bgo_forward_f64(const float *w, const float *x, int64_t n, double *out);

/* Counter RNG shared by the oracle and the HIP env: Philox4x32-10,
 * key = seed, counter = (game_id_lo, game_id_hi, ply, stream). */
void bgo_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                       uint32_t k0, uint32_t k1, uint32_t out[4]);
enum { BGO_STREAM_TURN = 0, BGO_STREAM_OPENING = 1 };
int  bgo_die_from_u32(uint32_t u);                       /* 1 + ((u*6)>>32)            */
int  bgo_opening_turn(uint64_t seed, uint64_t game_id);  /* train.py:89-97 on the stream */

/* One env step of one game (make_move model.py:180-222 + loop body train.py:109-120).
 * policy: 0 = uniform random over the reference-order list, 1 = greedy value net.
 * dice/rand words are explicit so a harness can inject them.
 * Returns C (candidate count); *chosen = index applied or -1; *over / *winner as is_game_over.
 * The turn is flipped when the game is not over (train.py:119-120). */
typedef struct bgo_step_out {
    int64_t n_candidates;
    int64_t chosen;
    int32_t over;
    int32_t winner;
    float   value;      /* value of the chosen afterstate (greedy), else 0 */
} bgo_step_out;
void bgo_step(bgo_state *s, int d1, int d2, int policy, uint32_t choice_u32,
              uint32_t eps_u32, float epsilon, const float *weights, bgo_step_out *out);

/* Full self-play of `n_steps` env steps of ONE lane with auto-reset, using the Philox
 * streams exactly as the HIP env does.  Records the state AFTER each step
 * (post-reset when the game ended) into snap[n_steps][30] =
 * [board24, bar1, bar2, off1, off2, turn, flags(bit0=over_this_step, bit1=winner)] (may be NULL).
 * Returns the number of finished games. */
typedef struct bgo_lane {
    bgo_state s;
    uint64_t  lane_id;     /* global lane index                                  */
    uint64_t  stride;      /* total lanes: game_id = lane_id + episode * stride  */
    uint64_t  episode;
    uint32_t  ply;
} bgo_lane;
void    bgo_lane_reset(bgo_lane *l, uint64_t seed, uint64_t lane_id, uint64_t stride);
int64_t bgo_lane_run(bgo_lane *l, uint64_t seed, int64_t n_steps, int policy, float epsilon,
                     const float *weights, int32_t *snap, int64_t *n_candidates_total);

#ifdef __cplusplus
}
#endif
#endif
