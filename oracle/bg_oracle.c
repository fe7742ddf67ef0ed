/* bg_oracle.c -- TEST INFRASTRUCTURE ONLY (see bg_oracle.h).
 *
 * Scalar C restatement of the reference env step.  Written fresh from the reference's
 * observable rules; each function names the reference lines it follows.  It is never on
 * the product path: the HIP kernels in backgammon-engine_amd/csrc are checked against it.
 */
#include "bg_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const int32_t k_start_board[24] = {
    /* cppsrc/game.cpp:251 */
    2, 0, 0, 0, 0, -5, 0, -3, 0, 0, 0, 5, -5, 0, 0, 0, 3, 0, 5, 0, 0, 0, 0, -2};

void bgo_init(bgo_state *s, int first_player)
{
    /* Game::Game(int) game.cpp:44-56: turn = parity, Pieces zeroed, populateBoard(). */
    memcpy(s->board, k_start_board, sizeof k_start_board);
    s->bar[0] = s->bar[1] = 0;
    s->off[0] = s->off[1] = 0;
    s->turn = (first_player % 2 == 0) ? 0 : 1;
}

/* game.cpp:416-457.  multi = +1 for PLAYER1, -1 for PLAYER2.
 * A mover with checkers on the bar may only move from its bar slot (0 for P1, 25 for P2). */
int bgo_is_valid_origin(const bgo_state *s, int multi, int idx)
{
    if (multi == -1) {
        if (s->bar[1] > 0)
            return idx == 25;
    } else if (multi == 1) {
        if (s->bar[0] > 0)
            return idx == 0;
    }
    if (idx < 1 || idx > 24)
        return 0;
    return s->board[idx - 1] * multi > 0;
}

/* game.cpp:488-557 including the asymmetric overrun rule. */
int bgo_can_free_piece(const bgo_state *s, int multi, int dice, int origin)
{
    int player = (multi == 1) ? 0 : 1;
    if (s->bar[player] != 0)
        return 0;
    for (int pt = 1; pt <= 24; pt++) {
        if (player == 0) {
            if (pt < 19 && s->board[pt - 1] > 0)
                return 0;                       /* P1 checker outside 19..24 */
        } else {
            if (pt > 6 && s->board[pt - 1] < 0)
                return 0;                       /* P2 checker outside 1..6   */
        }
    }
    if (player == 0) {
        if (dice > 25 - origin) {
            /* game.cpp:526-537: blocked by P1 checkers on points origin+1..24 */
            for (int i = origin; i <= 23; i++)
                if (s->board[i] > 0)
                    return 0;
        }
    } else {
        if (dice > origin) {
            /* game.cpp:542-553: blocked by ANY checker on points origin+1..7 */
            for (int i = origin; i <= 6; i++)
                if (s->board[i] != 0)
                    return 0;
        }
    }
    return 1;
}

/* game.cpp:459-485 */
int bgo_is_valid_destination(const bgo_state *s, int multi, int idx, int dice, int origin)
{
    if (idx == 0 || idx >= 25)
        return bgo_can_free_piece(s, multi, dice, origin);
    if (idx < 1 || idx > 24)
        return 0;
    /* own/empty point, or a single opposing checker (hit) */
    return s->board[idx - 1] * multi >= -1;
}

/* game.cpp:80-105: origins 0..25 ascending, destination clamped to [0,25]. */
int bgo_legal_moves(const bgo_state *s, int player, int die, int32_t *out_pairs)
{
    int n = 0;
    int multi = (player == 0) ? 1 : -1;
    for (int o = 0; o <= 25; o++) {
        if (!bgo_is_valid_origin(s, multi, o))
            continue;
        int d = o + multi * die;
        if (d > 25) d = 25;
        if (d < 0) d = 0;
        if (bgo_is_valid_destination(s, multi, d, die, o)) {
            if (out_pairs) {
                out_pairs[2 * n] = o;
                out_pairs[2 * n + 1] = d;
            }
            n++;
        }
    }
    return n;
}

/* Pieces::removeJailedPiece, Pieces.cpp:45-55 (decrements P2 when P1 has none). */
static void remove_jailed(bgo_state *s, int player)
{
    if (player == 0 && s->bar[0] > 0)
        s->bar[0] -= 1;
    else
        s->bar[1] -= 1;
}

/* game.cpp:573-663; check order and messages preserved as error codes. */
int bgo_try_move(bgo_state *s, int player, int dice, int origin, int dest)
{
    int multi = (player == 1) ? -1 : 1;
    if (!bgo_is_valid_origin(s, multi, origin))
        return BGO_ERR_INVALID_ORIGIN;
    if (origin < 0 || origin > 25)
        return BGO_ERR_ORIGIN_RANGE;
    if (dest < 0 || dest > 25)
        return BGO_ERR_DEST_RANGE;

    int diff = origin - dest;
    if (dest != 0 && dest != 25) {
        if (diff * (-multi) < 0)
            return BGO_ERR_DIRECTION;
        if (dice != abs(diff))
            return BGO_ERR_DICE_MISMATCH;
        if (!bgo_is_valid_destination(s, multi, dest, dice, origin))
            return BGO_ERR_INVALID_DEST;
        if (origin == 0 || origin == 25)
            remove_jailed(s, multi > 0 ? 0 : 1);
        else
            s->board[origin - 1] -= multi;
    }

    if (dest == 0 || dest == 25) {
        /* bear-off branch re-checks only that the origin is on the board */
        if (origin == 0 || origin == 25)
            return BGO_ERR_BEAROFF_FROM_JAIL;
        s->off[multi > 0 ? 0 : 1] += 1;
        s->board[origin - 1] -= multi;
        return BGO_OK;
    }
    if (s->board[dest - 1] * multi == -1) {
        s->board[dest - 1] = 0;                 /* hit the blot */
        s->bar[multi > 0 ? 1 : 0] += 1;
    }
    s->board[dest - 1] += multi;
    return BGO_OK;
}

/* game.cpp:388-407 */
int bgo_over(const bgo_state *s, int *winner)
{
    if (s->off[0] == 15) { if (winner) *winner = 0; return 1; }
    if (s->off[1] == 15) { if (winner) *winner = 1; return 1; }
    return 0;
}

/* ---- turn-sequence enumeration ------------------------------------------------ */

typedef struct emit_ctx {
    const bgo_state *root;
    int player;
    int64_t cap, count;
    int8_t *seq;
    int32_t *seq_len;
    int32_t *states;
} emit_ctx;

/* evaluateTurnSequences body, game.cpp:201-220: replay the sequence on a fresh copy with
 * die = |origin - dest| and pack [board24, bar1, bar2, off1, off2]. */
static void emit_sequence(emit_ctx *c, const int32_t (*mv)[2], int len)
{
    int64_t k = c->count++;
    if (k >= c->cap)
        return;
    if (c->seq) {
        int8_t *q = c->seq + k * 8;
        for (int i = 0; i < 4; i++) {
            q[2 * i] = (i < len) ? (int8_t)mv[i][0] : -1;
            q[2 * i + 1] = (i < len) ? (int8_t)mv[i][1] : -1;
        }
    }
    if (c->seq_len)
        c->seq_len[k] = len;
    if (c->states) {
        bgo_state sim = *c->root;
        for (int i = 0; i < len; i++)
            (void)bgo_try_move(&sim, c->player, abs(mv[i][0] - mv[i][1]), mv[i][0], mv[i][1]);
        int32_t *o = c->states + k * 28;
        memcpy(o, sim.board, 24 * sizeof(int32_t));
        o[24] = sim.bar[0]; o[25] = sim.bar[1];
        o[26] = sim.off[0]; o[27] = sim.off[1];
    }
}

/* collectDoubles, game.cpp:109-131: pre-order DFS, leaf at depth 4 or when stuck
 * (a stuck root yields ONE empty sequence -- SURVEY.md Q4). */
static void doubles_dfs(emit_ctx *c, const bgo_state *st, int die, int depth, int32_t (*cur)[2])
{
    int32_t mv[26][2];
    int n = bgo_legal_moves(st, c->player, die, &mv[0][0]);
    if (depth == 4 || n == 0) {
        emit_sequence(c, (const int32_t (*)[2])cur, depth);
        return;
    }
    for (int i = 0; i < n; i++) {
        bgo_state nx = *st;
        (void)bgo_try_move(&nx, c->player, die, mv[i][0], mv[i][1]);
        cur[depth][0] = mv[i][0];
        cur[depth][1] = mv[i][1];
        doubles_dfs(c, &nx, die, depth + 1, cur);
    }
}

/* One die-order block of legalTurnSequences, game.cpp:142-163 / :165-182. */
static void ordered_block(emit_ctx *c, int first_die, int second_die)
{
    int32_t m1[26][2], m2[26][2], cur[2][2];
    int n1 = bgo_legal_moves(c->root, c->player, first_die, &m1[0][0]);
    for (int i = 0; i < n1; i++) {
        bgo_state g1 = *c->root;
        (void)bgo_try_move(&g1, c->player, first_die, m1[i][0], m1[i][1]);
        cur[0][0] = m1[i][0]; cur[0][1] = m1[i][1];
        int n2 = bgo_legal_moves(&g1, c->player, second_die, &m2[0][0]);
        if (n2 == 0) {
            emit_sequence(c, (const int32_t (*)[2])cur, 1);   /* no max-dice rule: Q2 */
        } else {
            for (int j = 0; j < n2; j++) {
                cur[1][0] = m2[j][0]; cur[1][1] = m2[j][1];
                emit_sequence(c, (const int32_t (*)[2])cur, 2);
            }
        }
    }
}

int64_t bgo_evaluate_turn_sequences(const bgo_state *s, int player, int d1, int d2,
                                    int64_t cap, int8_t *seq, int32_t *seq_len, int32_t *states)
{
    emit_ctx c = {s, player, cap, 0, seq, seq_len, states};
    if (d1 != d2) {
        ordered_block(&c, d1, d2);      /* d1 first ... */
        ordered_block(&c, d2, d1);      /* ... then d2 first, no dedup (Q3) */
    } else {
        int32_t cur[4][2];
        doubles_dfs(&c, s, d1, 0, cur);
    }
    return c.count;
}

/* ---- encoder + value net ------------------------------------------------------ */

/* model.py:111-144.  Values are formed in double and stored as float, as numpy does. */
void bgo_encode(const int32_t *states, int64_t n, int turn, float *out)
{
    for (int64_t r = 0; r < n; r++) {
        const int32_t *st = states + r * 28;
        float *x = out + r * BGO_N_IN;
        memset(x, 0, BGO_N_IN * sizeof(float));
        for (int i = 0; i < 24; i++) {
            int v = st[i];
            int cnt = v < 0 ? -v : v;
            int base = 8 * i + (v > 0 ? 0 : 4);
            if (cnt >= 1) x[base + 0] = 1.0f;
            if (cnt >= 2) x[base + 1] = 1.0f;
            if (cnt >= 3) x[base + 2] = 1.0f;
            if (cnt >= 4) x[base + 3] = (float)((double)(cnt - 3) / 2.0);
        }
        x[192] = (turn == 0) ? 1.0f : 0.0f;
        x[193] = (turn == 0) ? 0.0f : 1.0f;
        x[194] = (float)((double)st[24] / 2.0);
        x[195] = (float)((double)st[25] / 2.0);
        x[196] = (float)((double)st[26] / 15.0);
        x[197] = (float)((double)st[27] / 15.0);
    }
}

/* model.py:63-67: sigmoid(fc2(sigmoid(fc1(x)))) in fp32, k-ascending accumulation. */
void bgo_forward_f32(const float *w, const float *x, int64_t n, float *out)
{
    const float *W1 = w, *b1 = w + BGO_N_HID * BGO_N_IN;
    const float *W2 = b1 + BGO_N_HID, *b2 = W2 + BGO_N_HID;
    for (int64_t r = 0; r < n; r++) {
        const float *xr = x + r * BGO_N_IN;
        float z = b2[0];
        for (int j = 0; j < BGO_N_HID; j++) {
            float a = b1[j];
            const float *wj = W1 + j * BGO_N_IN;
            for (int k = 0; k < BGO_N_IN; k++)
                a += wj[k] * xr[k];
            float h = 1.0f / (1.0f + expf(-a));
            z += W2[j] * h;
        }
        out[r] = 1.0f / (1.0f + expf(-z));
    }
}

void bgo_forward_f64(const float *w, const float *x, int64_t n, double *out)
{
    const float *W1 = w, *b1 = w + BGO_N_HID * BGO_N_IN;
    const float *W2 = b1 + BGO_N_HID, *b2 = W2 + BGO_N_HID;
    for (int64_t r = 0; r < n; r++) {
        const float *xr = x + r * BGO_N_IN;
        double z = b2[0];
        for (int j = 0; j < BGO_N_HID; j++) {
            double a = b1[j];
            const float *wj = W1 + j * BGO_N_IN;
            for (int k = 0; k < BGO_N_IN; k++)
                a += (double)wj[k] * (double)xr[k];
            z += (double)W2[j] / (1.0 + exp(-a));
        }
        out[r] = 1.0 / (1.0 + exp(-z));
    }
}

/* ---- counter RNG ---------------------------------------------------------------- */

void bgo_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                       uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

int bgo_die_from_u32(uint32_t u) { return 1 + (int)(((uint64_t)u * 6u) >> 32); }

/* Opening protocol of play_game, train.py:89-97: each side rolls a PAIR; repeat while the
 * sums tie; the larger sum moves first.  Attempt a draws Philox(counter=(gid, a, OPENING)). */
int bgo_opening_turn(uint64_t seed, uint64_t game_id)
{
    for (uint32_t attempt = 0;; attempt++) {
        uint32_t x[4];
        bgo_philox4x32_10((uint32_t)game_id, (uint32_t)(game_id >> 32), attempt,
                          BGO_STREAM_OPENING, (uint32_t)seed, (uint32_t)(seed >> 32), x);
        int p1 = bgo_die_from_u32(x[0]) + bgo_die_from_u32(x[1]);
        int p2 = bgo_die_from_u32(x[2]) + bgo_die_from_u32(x[3]);
        if (p1 != p2)
            return p1 > p2 ? 0 : 1;
    }
}

/* ---- env step ------------------------------------------------------------------- */

static __thread int32_t *g_states_buf;    /* per-thread scratch, grown on demand */
static __thread float *g_x_buf, *g_v_buf;
static __thread int64_t g_cap;

static void ensure_cap(int64_t c)
{
    if (c <= g_cap) return;
    int64_t nc = g_cap ? g_cap : 1024;
    while (nc < c) nc *= 2;
    g_states_buf = (int32_t *)realloc(g_states_buf, (size_t)nc * 28 * sizeof(int32_t));
    g_x_buf = (float *)realloc(g_x_buf, (size_t)nc * BGO_N_IN * sizeof(float));
    g_v_buf = (float *)realloc(g_v_buf, (size_t)nc * sizeof(float));
    g_cap = nc;
}

void bgo_step(bgo_state *s, int d1, int d2, int policy, uint32_t choice_u32,
              uint32_t eps_u32, float epsilon, const float *weights, bgo_step_out *out)
{
    int player = s->turn;
    ensure_cap(1024);
    int64_t C = bgo_evaluate_turn_sequences(s, player, d1, d2, g_cap, NULL, NULL, g_states_buf);
    if (C > g_cap) {
        ensure_cap(C);
        C = bgo_evaluate_turn_sequences(s, player, d1, d2, g_cap, NULL, NULL, g_states_buf);
    }
    out->n_candidates = C;
    out->chosen = -1;
    out->value = 0.0f;
    if (C > 0) {                                   /* model.py:202-203: empty list => no-op */
        int64_t idx;
        int explore = (policy == 0) ||
                      (epsilon > 0.0f && (float)(eps_u32 >> 8) * (1.0f / 16777216.0f) < epsilon);
        if (explore) {
            idx = (int64_t)(((uint64_t)choice_u32 * (uint64_t)C) >> 32);
        } else {
            /* model.py:209-213: mover's turn bit; argmax for P1, argmin for P2, first index */
            bgo_encode(g_states_buf, C, player, g_x_buf);
            bgo_forward_f32(weights, g_x_buf, C, g_v_buf);
            idx = 0;
            for (int64_t i = 1; i < C; i++) {
                if (player == 0 ? (g_v_buf[i] > g_v_buf[idx]) : (g_v_buf[i] < g_v_buf[idx]))
                    idx = i;
            }
            out->value = g_v_buf[idx];
        }
        const int32_t *a = g_states_buf + idx * 28;  /* afterstate == replaying the moves */
        memcpy(s->board, a, 24 * sizeof(int32_t));
        s->bar[0] = a[24]; s->bar[1] = a[25];
        s->off[0] = a[26]; s->off[1] = a[27];
        out->chosen = idx;
    }
    int w = -1;
    out->over = bgo_over(s, &w);                   /* train.py:113 */
    out->winner = w;
    if (!out->over)
        s->turn = 1 - s->turn;                     /* train.py:119-120 */
}

void bgo_lane_reset(bgo_lane *l, uint64_t seed, uint64_t lane_id, uint64_t stride)
{
    l->lane_id = lane_id;
    l->stride = stride;
    l->episode = 0;
    l->ply = 0;
    bgo_init(&l->s, 0);
    l->s.turn = bgo_opening_turn(seed, lane_id);
}

int64_t bgo_lane_run(bgo_lane *l, uint64_t seed, int64_t n_steps, int policy, float epsilon,
                     const float *weights, int32_t *snap, int64_t *n_candidates_total)
{
    int64_t finished = 0, ctot = 0;
    for (int64_t t = 0; t < n_steps; t++) {
        uint64_t gid = l->lane_id + l->episode * l->stride;
        uint32_t x[4];
        bgo_philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), l->ply, BGO_STREAM_TURN,
                          (uint32_t)seed, (uint32_t)(seed >> 32), x);
        bgo_step_out o;
        bgo_step(&l->s, bgo_die_from_u32(x[0]), bgo_die_from_u32(x[1]), policy, x[2], x[3],
                 epsilon, weights, &o);
        ctot += o.n_candidates;
        int flags = 0;
        if (o.over) {
            flags = 1 | (o.winner << 1);
            finished++;
            l->episode++;
            l->ply = 0;
            gid = l->lane_id + l->episode * l->stride;
            bgo_init(&l->s, 0);
            l->s.turn = bgo_opening_turn(seed, gid);
        } else {
            l->ply++;
        }
        if (snap) {
            int32_t *q = snap + t * 30;
            memcpy(q, l->s.board, 24 * sizeof(int32_t));
            q[24] = l->s.bar[0]; q[25] = l->s.bar[1];
            q[26] = l->s.off[0]; q[27] = l->s.off[1];
            q[28] = l->s.turn;
            q[29] = flags;
        }
    }
    if (n_candidates_total) *n_candidates_total = ctot;
    return finished;
}
