// bg_learner.h -- TD(lambda) learner kernels: the reference's apply_td_updates
// (pysrc/TD(λ) model/train.py:124-172) with the eligibility traces of model.py:48-53, as a replay in training steps over
// SLOTS: a slot holds one game at a time (gmeta: lane, length, winner, the step the game started at).  Lock-step replay
// (bgamd_td_begin): one game per slot, all starting at step 0, ordered by decreasing length, so the running ones are a
// prefix.  Streamed replay (bgamd_td_begin_stream): a slot takes the next game of its queue the step after its last one
// ended (td_advance_slot).  Per step t, for every slot with a running game:
//
//   forward pass       x_t and x_{t+1} from the 32-byte trajectory rows, h = σ(W1 x + b1), v = σ(W2 h + b2) for both, δ
//                      (train.py:136-141, terminal step train.py:165-166), g = v(1-v), the factors of the closed-form gradient
//                        ∇W1 = db1 ⊗ x, ∇b1 = db1 = g W2 ⊙ h ⊙ (1-h), ∇W2 = g h, ∇b2 = g
//                      written as one 272-float "factor row" per slot (the row of s_t | db1 | g·h | g), coef = fp32(α δ) with
//                      α δ formed in float64 first (train.py:147,160,169), the slot's ever-active feature mask.  By size of the step:
//                        td_forward_kernel<GB>     < 512 slots: GB slots per workgroup, thread per hidden unit, all on the VALUs
//                        td_forward_mfma_kernel    512 .. 24 575: W1 x on the matrix pipe (bf16 x 3, weights from the L2), epilogue fused
//                        traj_hidden_bf16x3_kernel + td_epilogue_wave_kernel   larger: the env's LDS-staged root pass, a wave per slot
//   trace pass         the HBM-bound one: e ← λ e + ∇ (model.py:52-53 semantics, train.py:150-158) and, in the same pass, the
//                      workgroup's share of Σ_g coef_g · e_g (train.py:159-161) -> partial sums; column-sparse and lazily
//                      scaled (below).  td_trace_kernel (slices x groups of games) for small steps, td_trace_wide_kernel
//                      (a workgroup per whole trace row, chunks of TD_CHUNK games) from 8 192 slots and where the chunks divide
//                      evenly, td_trace_pipe_kernel (the same, software-pipelined over the chunk's games) for mid-sized steps
//   td_reduce_kernel   Σ over the partial sums -> the 25 601-float update; θ += update (or hand it to the caller for
//                      the one all-reduce of the step), W1 re-transposed and re-split for the next forward
//   td_step_fused_kernel<FIRST, G>   (round 3) forward pass AND trace pass of a mid-sized step (512 .. 4 096 slots: what a
//                      streamed replay at the batch sizes the quality study allows runs) in ONE launch: a workgroup owns G = 1 .. 16
//                      slots, computes their forward pass on four of its waves and runs their pipelined whole-row trace pass behind
//                      one block barrier -- a training step is this launch + td_reduce_kernel
//
// Algorithmic bytes per (game, step): 2 · 25 601 · 4 = 204 808 B of trace traffic (SURVEY §8d "learner") for DENSE traces.
//
// Column-sparse traces (what runs): ∇W1 = db1 ⊗ x is zero in every column whose feature x_j is zero, and the encoder is
// sparse (<= 35 of 198 features are non-zero), so the W1 part of a game's trace is EXACTLY zero in the columns of the
// features that have never been non-zero in that game so far -- 63 % of them, averaged over the turns of a greedy game
// (89 of 198 features have been active by the end of one).  The trace is stored feature-major (column j = 128 contiguous
// floats), every game carries the bit mask of its ever-active features, and the pass touches only those columns: λ·0 + 0 is
// never read, never computed, never written.  Same arithmetic on the same numbers in the same order: results are
// bit-identical to the dense pass (BGAMD_TD_DENSE=1 keeps it for the test that says so), at ~0.37 of its HBM traffic.
//
// Lazily scaled traces (round 2, what runs unless BGAMD_TD_LAZY=0): every game of a replay starts at t = 0 and decays by the
// same λ per step, so the stored trace is ê = e / c with ONE host-side scale c = Π λ for all games.  e ← λ e + ∇ becomes
// ê ← ê + ∇ / c: a column whose feature is zero in s_t has ∇ = 0 and is READ (Σ_g coef_g c ê_g needs it) but NOT WRITTEN --
// ~30 of the ~72 active columns of a game are written per step.  When c falls below 2^-40 (every 78 steps at λ = 0.7) one
// ordinary pass folds it back in (ê ← c ê + ∇, c = 1); with λ so small that every step would, this IS the ordinary pass.
// Same real numbers, different rounding (fp32 relative 1e-7 per term): the parity tests' bounds hold unchanged.
#pragma once
#include "bg_board.h"
#include "bg_eval.h"

namespace bg {

constexpr int TD_P = 25601;              // W1[128][198] | b1[128] | W2[128] | b2
constexpr int TD_LD = 25664;             // trace row stride in floats (multiple of 64)
constexpr int TD_OFF_B1 = 25344, TD_OFF_W2 = 25472, TD_OFF_B2 = 25600;
// factor row of one game: the 32-byte row of s_t (8 words: the trace pass decodes x_j from it) | db1[128] | g·h[128] | g | pad
constexpr int TD_F_ROW = 0, TD_F_DB1 = 8, TD_F_GH = 136, TD_F_G = 264, TD_FLD = 272;
constexpr int TD_TRACE_THREADS = 256;
constexpr int TD_SLICES = (TD_LD / 4 + TD_TRACE_THREADS - 1) / TD_TRACE_THREADS;   // 26
#ifndef BG_TD_CHUNK
#define BG_TD_CHUNK 8
#endif
constexpr int TD_CHUNK = BG_TD_CHUNK;    // games staged in LDS at a time by the trace kernel
#ifndef BG_TD_MAX_GROUPS
#define BG_TD_MAX_GROUPS 256
#endif
constexpr int TD_MAX_GROUPS = BG_TD_MAX_GROUPS;
constexpr int TD_MASK_WORDS = 8;         // 198 feature bits in 7 words, padded to 32 B per game
// INTERNAL order of a trace row / partial-sum row (the parameter order of theta is fc1.weight[n][j] | b1 | W2 | b2):
//   [j * 128 + n] = W1[n][j] for j < 198 (feature-major: a column of W1 is 512 contiguous bytes), then b1, W2, b2 at the
//   same offsets as in parameter order
__host__ __device__ __forceinline__ int td_param_of_internal(int p)
{
    return p < TD_OFF_B1 ? (p & (N_HID - 1)) * N_IN + (p >> 7) : p;
}

struct TdView {
    float *theta;                        // [TD_LD] flat parameters
    float *w1t;                          // [198][128] transposed fc1.weight for the forward kernel
    float *e;                            // [max_games][TD_LD] traces, indexed by order position
    float *fac;                          // [max_games][TD_FLD]
    float *coef;                         // [max_games]
    double *sq;                          // [max_games] Σ δ² per game
    float *partial;                      // [TD_MAX_GROUPS][TD_LD]
    const uint4 *rows;                   // [T][n_lanes] x 2 uint4
    const int32_t *order;                // [n_games] lane index, by decreasing length
    const int32_t *length;               // [n_lanes]
    const uint8_t *p1_won;               // [n_lanes]
    int4 *gmeta;                         // [max_games] (lane, length, p1_won | log row of the first turn << 1, start step) of the game in slot i: one load, no chain
    // streamed replay (bgamd_td_begin_stream): slot i plays the games queue[qoff[i] .. qoff[i + 1]) one after another; qcur[i] = the
    // next one.  Lock-step replay: one game per slot, all starting at step 0 (queue == nullptr).
    const int32_t *queue, *qoff;
    int32_t *qcur;
    // game table of a streamed replay over a RING log (bgamd_td_begin_stream_games): queue entries are game ids; game id -> the lane whose
    // column of the log holds it and the ring slot of its first turn.  nullptr: game id = lane, first turn in row 0 (one game per lane).
    const int32_t *game_lane, *game_start;
    long long n_table;                   // entries of length / p1_won (and of the game table)
    unsigned int *nupd;                  // [max_games] (game, step) updates of the slot since begin
    uint16_t *wl3;                       // fc1.weight as three bf16 planes in the MFMA layout of bg_eval.h (refreshed with every update)
    uint2 *lut;                          // count -> 4 bf16 features
    float *hid;                          // [2 * max_games][128] W1 x + b1 of (s_t, s_{t+1}) per running game, from the MFMA pass
    uint32_t *amask;                     // [max_games][TD_MASK_WORDS] features that have been non-zero in the game so far
    uint32_t *anew;                      // [max_games][TD_MASK_WORDS] ... for the first time at the current step (trace column = 0: not read)
    unsigned int *act_cols;              // [max_games] Σ over the game's steps of its active W1 trace columns since begin (traffic report;
                                         //   per game: a single counter would be one same-address atomic per block and step)
    unsigned int *wr_cols;               // [max_games] ... of the W1 trace columns WRITTEN (lazily scaled traces: x_j != 0 or newly active)
    int full_step;                       // the step at hand is an ordinary pass: every active column is written
    int dense;                           // BGAMD_TD_DENSE=1: every column active from the first step (the dense pass)
    long long T, n_lanes, n_games;
};

__device__ __forceinline__ float td_sigmoid(float a) { return 1.0f / (1.0f + expf(-a)); }

// encoder feature j (model.py:111-144) of a 32-byte row held as 8 words (LDS or registers)
__device__ __forceinline__ float td_feature_value(const uint32_t *__restrict__ row, int j)
{
    if (j < 192) {
        const int pos = (j >> 3) + 1, base = 4 * ((j >> 2) & 1), level = j & 3;
        const int cnt = (int)(((row[base] >> pos) & 1u) | (((row[base + 1] >> pos) & 1u) << 1) | (((row[base + 2] >> pos) & 1u) << 2) |
                              (((row[base + 3] >> pos) & 1u) << 3));
        return level < 3 ? (cnt > level ? 1.0f : 0.0f) : (cnt > 3 ? 0.5f * (float)(cnt - 3) : 0.0f);
    }
    const int side = j & 1;                                    // 192/193 turn, 194/195 bar, 196/197 off
    const int turn = (row[0] & TURN_BIT) ? 1 : 0;
    if (j < 194) return turn == side ? 1.0f : 0.0f;
    const int pos = (j < 196) == (side == 0) ? 0 : 25;         // P1 bar = 0, P2 bar = 25; P1 off = 25, P2 off = 0
    const int base = 4 * side;
    const int cnt = (int)(((row[base] >> pos) & 1u) | (((row[base + 1] >> pos) & 1u) << 1) | (((row[base + 2] >> pos) & 1u) << 2) |
                          (((row[base + 3] >> pos) & 1u) << 3));
    return j < 196 ? 0.5f * (float)cnt : (float)cnt / 15.0f;
}

// the 8 feature bits "x_j != 0" of board point pt (both sides x four levels), or the 6 bits of the tail (pt == 24)
__device__ __forceinline__ uint32_t td_nonzero_bits(const uint32_t *__restrict__ row, int pt)
{
    uint32_t out = 0;
    if (pt < 24) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int pos = pt + 1, b = 4 * q;
            const uint32_t b0 = (row[b] >> pos) & 1u, b1 = (row[b + 1] >> pos) & 1u, b2 = (row[b + 2] >> pos) & 1u, b3 = (row[b + 3] >> pos) & 1u;
            const uint32_t ge1 = b0 | b1 | b2 | b3, ge2 = b1 | b2 | b3, ge3 = (b0 & b1) | b2 | b3, ge4 = b2 | b3;
            out |= (ge1 | (ge2 << 1) | (ge3 << 2) | (ge4 << 3)) << (4 * q);
        }
        return out;
    }
    const uint32_t turn = (row[0] & TURN_BIT) ? 1u : 0u;
    auto any = [&](int side, int pos) { const int b = 4 * side; return (((row[b] | row[b + 1] | row[b + 2] | row[b + 3]) >> pos) & 1u); };
    return (turn ^ 1u) | (turn << 1) | (any(0, 0) << 2) | (any(1, 25) << 3) | (any(0, 25) << 4) | (any(1, 0) << 5);
}
typedef float td_f32x4 __attribute__((ext_vector_type(4)));

__global__ void td_gather_kernel(TdView v)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.n_games) return;
    int lane = v.order[i];                               // caller data: kept inside the log whatever it says
    lane = lane < 0 ? 0 : (lane >= v.n_lanes ? (int)v.n_lanes - 1 : lane);
    int len = v.length[lane];
    len = len < 0 ? 0 : (len > v.T ? (int)v.T : len);
    v.gmeta[i] = make_int4(lane, len, v.p1_won[lane] ? 1 : 0, 0);
    v.nupd[i] = 0u;
    v.sq[i] = 0.0;
    v.act_cols[i] = 0u;
    v.wr_cols[i] = 0u;
}

// Slot i takes the next game of its queue that has any turns (streamed replay), starting at step `start`, or falls empty
// (length 0: a lock-step replay, or the queue is used up).
__device__ __forceinline__ void td_advance_slot(const TdView &v, long long i, int start)
{
    int lane = 0, len = 0, won = 0, first_row = 0;
    if (v.queue) {
        int c = v.qcur[i];
        const int end = v.qoff[i + 1];
        while (c < end && len == 0) {
            int id = v.queue[c++];                           // caller data: kept inside the tables whatever it says
            id = id < 0 ? 0 : (id >= v.n_table ? (int)v.n_table - 1 : id);
            lane = v.game_lane ? v.game_lane[id] : id;
            lane = lane < 0 ? 0 : (lane >= v.n_lanes ? (int)v.n_lanes - 1 : lane);
            len = v.length[id];
            len = len < 0 ? 0 : (len > v.T ? (int)v.T : len);
            won = v.p1_won[id] ? 1 : 0;
            first_row = v.game_start ? v.game_start[id] : 0;
            first_row = first_row < 0 ? 0 : (first_row >= v.T ? (int)v.T - 1 : first_row);
        }
        v.qcur[i] = c;
    }
    int4 gm;
    gm.x = len ? lane : 0; gm.y = len; gm.z = len ? (won | (first_row << 1)) : 0; gm.w = start;
    v.gmeta[i] = gm;
}

// log row of a game's step tl (the log is a ring of T rows when the game table names a first row; tl < length <= T)
__device__ __forceinline__ long long td_log_row(const int4 &gm, long long tl, long long T)
{
    const long long r = tl + (gm.z >> 1);
    return r >= T ? r - T : r;
}

__global__ void td_gather_stream_kernel(TdView v)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.n_games) return;                          // n_games = slots here
    v.qcur[i] = v.qoff[i];
    td_advance_slot(v, i, 0);
    v.nupd[i] = 0u;
    v.sq[i] = 0.0;
    v.act_cols[i] = 0u;
    v.wr_cols[i] = 0u;
}

// mask word 7 of a slot carries two flags for the trace pass: amask: the slot holds a running game; anew: it is the game's first step
// (nothing of the slot's trace is read: the previous game's is dead)
constexpr int TD_FLAG_WORD = 7;

// TD_GB games x {s_t, s_{t+1}} per workgroup of 128 threads (2 for small rounds: a shorter FMA chain per thread and
// more workgroups; 4 for large ones: half the W1 traffic)
// PRE = false: the whole forward pass on the VALUs (small rounds: one launch, latency is what counts).
// PRE = true : the hidden pre-activations W1 x + b1 come from traj_hidden_bf16x3_kernel (bg_eval.h: exact bf16 x 3 split of
//              W1 on the matrix pipe, the [2 G x 198] · [198 x 128] product of the step); this kernel is its epilogue.
template <int TD_GB, bool PRE>
__global__ __launch_bounds__(128) void td_forward_kernel(TdView v, long long t, long long n_active, double alpha)
{
    constexpr int NR = 2 * TD_GB;                         // rows: [s][game]
    __shared__ unsigned int s_colsg[TD_GB], s_wrg[TD_GB];
    __shared__ int adv[TD_GB];
    __shared__ __attribute__((aligned(16))) float xs[PRE ? 1 : N_IN][NR];
    __shared__ uint32_t srow[TD_GB][8];                   // the 32-byte row of s_t of every game of the block
    __shared__ uint32_t smask[TD_GB][TD_MASK_WORDS];
    __shared__ float hs[NR][N_HID + 1];
    __shared__ float outs[NR];
    __shared__ float gs[TD_GB];
    const int tid = threadIdx.x;
    const long long i0 = (long long)blockIdx.x * TD_GB;
    // this thread's fc1.weight column set (198 values) is fetched in ONE batch of independent loads -- a single L2
    // round trip instead of a chain of them -- and the FMAs then run out of registers
    float w[PRE ? 1 : N_IN];
    if (!PRE) {
#pragma unroll
        for (int j = 0; j < N_IN; ++j) w[j] = v.w1t[j * N_HID + tid];
    }

    // ---- the rows: thread = (row r, chunk c) ----
    constexpr int CH = 128 / NR;                          // chunks (threads per row)
    if (tid < TD_GB * TD_MASK_WORDS) (&smask[0][0])[tid] = 0;
    {
        const int r = tid & (NR - 1), c = tid / NR;
        const int s = r / TD_GB, g = r % TD_GB;
        const long long i = i0 + g;
        bool live = i < n_active;
        int lane = 0;
        long long tl = 0, lrow = 0;                       // the game's own step, its row of the log
        if (live && (!PRE || (s == 0 && c == 0))) {
            const int4 gm = v.gmeta[i];
            lane = gm.x;
            tl = t - gm.w;
            live = (tl + s) < gm.y && (tl + s) < v.T;     // s_{t+1} does not exist on the terminal step
            lrow = td_log_row(gm, tl + s, v.T);
        } else if (PRE) live = false;                     // the matrix-pipe pass has decoded the rows: only s_t is needed here, once
        uint32_t p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (live) {
            const uint4 *src = v.rows + (lrow * v.n_lanes + lane) * 2;
            const uint4 u0 = src[0], u1 = src[1];
            p[0] = u0.x; p[1] = u0.y; p[2] = u0.z; p[3] = u0.w; p[4] = u1.x; p[5] = u1.y; p[6] = u1.z; p[7] = u1.w;
        }
        if (s == 0 && c == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) srow[g][k] = p[k];
        }
        if (!PRE) {                                       // the VALU path needs the 198 floats of both states
            const Side sd[2] = {{{p[0], p[1], p[2], p[3]}}, {{p[4], p[5], p[6], p[7]}}};
#pragma unroll
            for (int k = 0; k < (24 + CH - 1) / CH; ++k) {
                const int pt = c + CH * k;
                if (pt >= 24) break;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int cn = count_at(sd[q], pt + 1);
                    xs[8 * pt + 4 * q + 0][r] = cn >= 1 ? 1.0f : 0.0f;
                    xs[8 * pt + 4 * q + 1][r] = cn >= 2 ? 1.0f : 0.0f;
                    xs[8 * pt + 4 * q + 2][r] = cn >= 3 ? 1.0f : 0.0f;
                    xs[8 * pt + 4 * q + 3][r] = cn >= 4 ? 0.5f * (float)(cn - 3) : 0.0f;
                }
            }
            if (c == 0) {
                const int turn = (p[0] & TURN_BIT) ? 1 : 0;
                xs[192][r] = live ? (turn == 0 ? 1.0f : 0.0f) : 0.0f;
                xs[193][r] = live ? (turn == 0 ? 0.0f : 1.0f) : 0.0f;
                xs[194][r] = 0.5f * (float)count_at(sd[0], 0);
                xs[195][r] = 0.5f * (float)count_at(sd[1], 25);
                xs[196][r] = (float)count_at(sd[0], 25) / 15.0f;
                xs[197][r] = (float)count_at(sd[1], 0) / 15.0f;
            }
        }
    }
    // the hidden pre-activations of the matrix-pipe pass and this thread's weights: requested before the first barrier
    const int n = tid;
    float acc[NR];
    if (PRE) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {                         // row r = [s][game] here, row 2 i + s in the MFMA pass's output
            const long long i = i0 + (r % TD_GB);
            acc[r] = i < n_active ? v.hid[(2 * i + r / TD_GB) * N_HID + n] : 0.0f;
        }
    }
    const float w2 = v.theta[TD_OFF_W2 + n];
    if (tid < TD_GB) { s_colsg[tid] = 0; s_wrg[tid] = 0; adv[tid] = 0; }
    __syncthreads();
    // ---- ever-active feature masks of the games: thread = (game, board point | tail) ORs its 8 (6) "x_j != 0" bits into the
    //      game's mask words; the trace pass touches only these columns ----
    if (tid < TD_GB * 25) {
        const int g = tid / 25, pt = tid % 25;
        if (i0 + g < n_active) {
            const uint32_t bits = td_nonzero_bits(srow[g], pt);
            if (bits) atomicOr(&smask[g][pt >> 2], bits << (8 * (pt & 3)));      // word = 4 points (8 features each); tail = word 6
        }
    }
    __syncthreads();
    if (tid < TD_GB * TD_MASK_WORDS) {
        const int g = tid / TD_MASK_WORDS, wd = tid % TD_MASK_WORDS;
        const long long i = i0 + g;
        if (i < n_active) {
            const int4 gm = v.gmeta[i];
            const bool run = (t - gm.w) < gm.y, first = t == gm.w;
            uint32_t now = smask[g][wd];
            const uint32_t valid = wd < 6 ? 0xFFFFFFFFu : (wd == 6 ? 0x3Fu : 0u);     // 198 = 6 * 32 + 6
            const uint32_t nz = now & valid;
            if (v.dense) now = valid;
            if (!run) now = 0u;
            const uint32_t old = (first || !run || wd == TD_FLAG_WORD) ? 0u : v.amask[i * TD_MASK_WORDS + wd];
            v.amask[i * TD_MASK_WORDS + wd] = wd == TD_FLAG_WORD ? (run ? 1u : 0u) : (old | now);
            v.anew[i * TD_MASK_WORDS + wd] = wd == TD_FLAG_WORD ? (first ? 1u : 0u) : (now & ~old);
            atomicAdd(&s_colsg[g], (unsigned int)__popc(old | now));
            atomicAdd(&s_wrg[g], (unsigned int)__popc(v.full_step ? (old | now) : (nz | (now & ~old))));
        }
    }

    // ---- hidden layer: thread n owns unit n for all 16 rows ----
    if (!PRE) {
        const float b = v.theta[TD_OFF_B1 + n];
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[r] = b;
#pragma unroll
        for (int j = 0; j < N_IN; ++j) {
            const float4 *xr = reinterpret_cast<const float4 *>(&xs[j][0]);
#pragma unroll
            for (int q = 0; q < NR / 4; ++q) {
                const float4 x4 = xr[q];
                acc[4 * q + 0] = fmaf(w[PRE ? 0 : j], x4.x, acc[4 * q + 0]);
                acc[4 * q + 1] = fmaf(w[PRE ? 0 : j], x4.y, acc[4 * q + 1]);
                acc[4 * q + 2] = fmaf(w[PRE ? 0 : j], x4.z, acc[4 * q + 2]);
                acc[4 * q + 3] = fmaf(w[PRE ? 0 : j], x4.w, acc[4 * q + 3]);
            }
        }
    }
    float h[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        h[r] = td_sigmoid(acc[r]);
        hs[r][n] = w2 * h[r];
    }
    __syncthreads();
    // ---- output unit: thread = (row r, part e) sums N_HID / CH products, then a CH-lane butterfly ----
    {
        const int r = tid / CH, e8 = tid % CH;
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < N_HID / CH; ++k) s += hs[r][e8 * (N_HID / CH) + k];
#pragma unroll
        for (int m = 1; m < CH; m <<= 1) s += __shfl_xor(s, m, 64);
        if (e8 == 0) outs[r] = td_sigmoid(s + v.theta[TD_OFF_B2]);
    }
    __syncthreads();
    // ---- δ, g, coef per game ----
    if (tid < TD_GB) {
        const long long i = i0 + tid;
        float g = 0.0f;
        if (i < n_active) {
            const int4 gm = v.gmeta[i];
            const long long tl = t - gm.w;
            if (tl < gm.y) {
                const float val = outs[tid], vnext = outs[TD_GB + tid];
                const float z = (gm.z & 1) ? 1.0f : 0.0f;
                const float delta = (tl + 1 >= gm.y) ? z - val : vnext - val;   // lengths never exceed the log (host check)
                g = val * (1.0f - val);
                v.coef[i] = (float)(alpha * (double)delta);
                v.sq[i] += (double)delta * (double)delta;
                v.nupd[i] += 1u;
                if (tl + 1 >= gm.y) adv[tid] = 1;                          // the slot takes its next game (below, after the last read of gmeta)
            } else v.coef[i] = 0.0f;
        }
        gs[tid] = g;
    }
    __syncthreads();
    // ---- factor rows ----
#pragma unroll
    for (int gq = 0; gq < TD_GB; ++gq) {
        const long long i = i0 + gq;
        if (i >= n_active) break;
        float *f = v.fac + i * TD_FLD;
        const float g = gs[gq], hh = h[gq];
        f[TD_F_DB1 + n] = (g * w2) * (1.0f - hh) * hh;
        f[TD_F_GH + n] = g * hh;
        if (n < 8) f[TD_F_ROW + n] = __uint_as_float(srow[gq][n]);
        if (n == 8) f[TD_F_G] = g;
    }
    if (tid < TD_GB && i0 + tid < n_active) {                                      // (the barriers above ordered the LDS adds)
        v.act_cols[i0 + tid] += s_colsg[tid];
        v.wr_cols[i0 + tid] += s_wrg[tid];
        if (adv[tid]) td_advance_slot(v, i0 + tid, (int)(t + 1));
    }
}

// Epilogue of the matrix-pipe forward pass, ONE WAVE PER GAME (4 games per 256-thread block, no LDS, no block barrier): lane l
// owns hidden units l and l + 64 of s_t and s_{t+1}; h = σ(·), the dot with W2 by a wave reduction, v, δ, g, coef, the
// factor row (32-byte row of s_t | db1 | g·h | g) and the game's ever-active feature mask.  The chain of dependent memory
// accesses is gmeta -> row; everything else is requested up front.  (The thread-per-unit kernel above, with its four block
// barriers and six dependent round trips per block, took 126 us per step at 65 536 games for 140 MB of traffic.)
__global__ __launch_bounds__(256) void td_epilogue_wave_kernel(TdView v, long long t, long long n_active, double alpha)
{
    const int lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_active) return;                                 // whole waves leave: no barrier below
    const int4 gm = v.gmeta[i];
    const long long tl = t - gm.w;                             // the game's own step
    if (tl >= gm.y) {                                          // a slot whose queue is used up (streamed replay): nothing to add
        if (lane < TD_MASK_WORDS) { v.amask[i * TD_MASK_WORDS + lane] = 0u; v.anew[i * TD_MASK_WORDS + lane] = 0u; }
        if (lane == 9) v.coef[i] = 0.0f;
        return;
    }
    const float *hp = v.hid + 2 * i * N_HID;
    const float a0 = hp[lane], a1 = hp[lane + 64], c0 = hp[N_HID + lane], c1 = hp[N_HID + lane + 64];     // s_t | s_{t+1}
    const float w20 = v.theta[TD_OFF_W2 + lane], w21 = v.theta[TD_OFF_W2 + lane + 64], b2 = v.theta[TD_OFF_B2];
    uint32_t old = 0;
    if (tl != 0 && lane < TD_FLAG_WORD) old = v.amask[i * TD_MASK_WORDS + lane];
    const uint4 *src = v.rows + (td_log_row(gm, tl, v.T) * v.n_lanes + gm.x) * 2;   // s_t exists for every running game
    const uint4 u0 = src[0], u1 = src[1];
    const uint32_t row[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
    const float h0 = td_sigmoid(a0), h1 = td_sigmoid(a1), k0 = td_sigmoid(c0), k1 = td_sigmoid(c1);
    float sv = w20 * h0 + w21 * h1, sn = w20 * k0 + w21 * k1;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { sv += __shfl_xor(sv, m, 64); sn += __shfl_xor(sn, m, 64); }
    const float val = td_sigmoid(sv + b2), vnext = td_sigmoid(sn + b2);
    const float z = (gm.z & 1) ? 1.0f : 0.0f;
    const float delta = (tl + 1 >= gm.y) ? z - val : vnext - val;
    const float g = val * (1.0f - val);
    float *f = v.fac + i * TD_FLD;
    f[TD_F_DB1 + lane] = (g * w20) * (1.0f - h0) * h0;
    f[TD_F_DB1 + lane + 64] = (g * w21) * (1.0f - h1) * h1;
    f[TD_F_GH + lane] = g * h0;
    f[TD_F_GH + lane + 64] = g * h1;
    if (lane < 8) f[TD_F_ROW + lane] = __uint_as_float(row[lane & 7]);
    if (lane == 8) f[TD_F_G] = g;
    if (lane == 9) { v.coef[i] = (float)(alpha * (double)delta); v.sq[i] += (double)delta * (double)delta; v.nupd[i] += 1u; }
    // ever-active feature mask: lane pt < 25 forms the 8 (6) "x_j != 0" bits of board point pt (the tail), a quad of lanes
    // is one 32-bit word
    uint32_t bits = lane < 25 ? td_nonzero_bits(row, lane) << (8 * (lane & 3)) : 0u;
    bits |= __shfl_xor(bits, 1, 64);
    bits |= __shfl_xor(bits, 2, 64);
    uint32_t now = __shfl(bits, 4 * (lane & 7), 64);           // lane w < 7 takes word w from lane 4 w
    unsigned int wr = 0;
    if (lane < TD_MASK_WORDS) {
        const uint32_t valid = lane < 6 ? 0xFFFFFFFFu : (lane == 6 ? 0x3Fu : 0u);
        const uint32_t nz = now & valid;
        now = v.dense ? valid : nz;
        v.amask[i * TD_MASK_WORDS + lane] = lane == TD_FLAG_WORD ? 1u : (old | now);
        v.anew[i * TD_MASK_WORDS + lane] = lane == TD_FLAG_WORD ? (tl == 0 ? 1u : 0u) : (now & ~old);
        wr = (unsigned int)__popc(v.full_step ? (old | now) : (nz | (now & ~old)));
    }
    unsigned int cols = lane < TD_MASK_WORDS ? (unsigned int)__popc(old | now) : 0u;
#pragma unroll
    for (int m = 4; m >= 1; m >>= 1) { cols += __shfl_xor(cols, m, 64); wr += __shfl_xor(wr, m, 64); }
    if (lane == 0) {
        v.act_cols[i] += cols; v.wr_cols[i] += wr;
        if (tl + 1 >= gm.y) td_advance_slot(v, i, (int)(t + 1));    // the slot takes its next game (streamed replay) or falls empty
    }
}

// The forward pass of a mid-sized step in ONE launch: traj_hidden_direct_kernel's product (a workgroup of four waves per 32-row tile =
// 16 slots x {s_t, s_{t+1}}, wave c = hidden units 32 c .. 32 c + 31, weight planes straight from the L2) with the epilogue on the
// accumulators -- σ, the W2 dot through LDS (32 rows x 128 products, 8 threads per row), δ, g, coef, the factor rows, the masks, the
// slot's next game: no `hid` round trip, no second launch (5.9 us + a kernel boundary of the ~49 us step at 2 048 slots).
constexpr int TD_FUSED_GAMES = 16;
__global__ __launch_bounds__(ROOT3D_THREADS) void td_forward_mfma_kernel(TdView v, long long t, long long n_active, double alpha)
{
    constexpr int G = TD_FUSED_GAMES;
    __shared__ uint2 sLut[16];
    __shared__ float sd[2 * G][N_HID + 1];
    __shared__ uint32_t srow[G][8], smask[G][TD_MASK_WORDS];
    __shared__ float outs[2 * G], gs[G];
    __shared__ unsigned int s_colsg[G], s_wrg[G];
    __shared__ int adv[G], lives[G];
    const int tid = threadIdx.x, lane = tid & 63, c = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long long i0 = (long long)blockIdx.x * G;
    const long long n_rows = 2 * n_active;
    if (tid < 16) sLut[tid] = v.lut[tid];
    if (tid < G) { s_colsg[tid] = 0; s_wrg[tid] = 0; adv[tid] = 0; lives[tid] = 0; }
    if (tid < G * TD_MASK_WORDS) (&smask[0][0])[tid] = 0;
    const TrajRowsFetch fetch{v.rows, v.gmeta, t, v.n_lanes, v.T};
    uint32_t p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool row_ok = false;
    if (i0 * 2 + r < n_rows) {
        uint4 u0, u1;
        if (fetch.get(i0 * 2 + r, u0, u1)) {
            row_ok = true;
            p[0] = u0.x; p[1] = u0.y; p[2] = u0.z; p[3] = u0.w; p[4] = u1.x; p[5] = u1.y; p[6] = u1.z; p[7] = u1.w;
        }
    }
    if (c == 0 && h == 0 && (r & 1) == 0) {                       // the 32-byte row of s_t of game r / 2 (zero: no running game)
#pragma unroll
        for (int k = 0; k < 8; ++k) srow[r >> 1][k] = p[k];
    }
    const int n = 32 * c + r;
    const float bb = v.theta[TD_OFF_B1 + n], w2n = v.theta[TD_OFF_W2 + n], b2 = v.theta[TD_OFF_B2];
    const uint4 *wp = reinterpret_cast<const uint4 *>(v.wl3) + (size_t)c * 64 + lane;
    uint4 w[K16_STEPS][3];
#pragma unroll
    for (int s = 0; s < K16_STEPS; ++s)
#pragma unroll
        for (int part = 0; part < 3; ++part) w[s][part] = wp[(size_t)part * ROOT3_PART_U4 + (size_t)s * 4 * 64];
    __syncthreads();
    const Side sa{{p[0], p[1], p[2], p[3]}}, sb{{p[4], p[5], p[6], p[7]}};
    floatx16 acc = {0};
#pragma unroll
    for (int s = 0; s < K16_STEPS; ++s) {
        union { uint4 u; bf16x8 v; } a;
        if (s < 12) {
            const int pos = 2 * s + h + 1;
            const uint2 l0 = sLut[count_at(sa, pos)], l1 = sLut[count_at(sb, pos)];
            a.u = make_uint4(l0.x, l0.y, l1.x, l1.y);
        } else {
            const int turn = (p[0] & TURN_BIT) ? 1 : 0;
            const uint32_t t0 = (turn == 0 && row_ok) ? 0x3F80u : 0u, t1 = (turn == 0 || !row_ok) ? 0u : 0x3F80u;
            const uint32_t bar1 = f32_to_bf16_rne(0.5f * (float)count_at(sa, 0)), bar2 = f32_to_bf16_rne(0.5f * (float)count_at(sb, 25));
            const uint32_t off1 = f32_to_bf16_rne((float)count_at(sa, 25)), off2 = f32_to_bf16_rne((float)count_at(sb, 0));
            a.u = h ? make_uint4(0, 0, 0, 0) : make_uint4(t0 | (t1 << 16), bar1 | (bar2 << 16), off1 | (off2 << 16), 0u);
        }
#pragma unroll
        for (int part = 0; part < 3; ++part) {
            union { uint4 u; bf16x8 v; } wv;
            wv.u = w[s][part];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, wv.v, acc, 0, 0, 0);
        }
    }
    // accumulator j of this lane = tile row (j & 3) + 8 (j >> 2) + 4 h (row 2 g + s: state s of game g), hidden unit n
    float hv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        hv[j] = td_sigmoid(acc[j] + bb);
        sd[(j & 3) + 8 * (j >> 2) + 4 * h][n] = w2n * hv[j];
    }
    // ever-active feature masks: thread = (game, two board points | tail)
    {
        const int g = tid >> 4, q = tid & 15;
        if (i0 + g < n_active) {
#pragma unroll
            for (int pt = q; pt < 25; pt += 16) {
                const uint32_t bits = td_nonzero_bits(srow[g], pt);
                if (bits) atomicOr(&smask[g][pt >> 2], bits << (8 * (pt & 3)));
            }
        }
    }
    __syncthreads();
    {   // output unit: thread = (row, eighth of the hidden layer)
        const int row = tid >> 3, e8 = tid & 7;
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < N_HID / 8; ++k) sum += sd[row][e8 * (N_HID / 8) + k];
        sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 4, 64);
        if (e8 == 0) outs[row] = td_sigmoid(sum + b2);
    }
    __syncthreads();
    if (tid < G) {                                                // δ, g, coef per game
        const long long i = i0 + tid;
        float g = 0.0f;
        if (i < n_active) {
            const int4 gm = v.gmeta[i];
            const long long tl = t - gm.w;
            if (tl < gm.y) {
                const float val = outs[2 * tid], vnext = outs[2 * tid + 1];
                const float z = (gm.z & 1) ? 1.0f : 0.0f;
                const float delta = (tl + 1 >= gm.y) ? z - val : vnext - val;
                g = val * (1.0f - val);
                v.coef[i] = (float)(alpha * (double)delta);
                v.sq[i] += (double)delta * (double)delta;
                v.nupd[i] += 1u;
                lives[tid] = 1 + (tl == 0 ? 1 : 0);
                if (tl + 1 >= gm.y) adv[tid] = 1;
            } else v.coef[i] = 0.0f;
        }
        gs[tid] = g;
    }
    __syncthreads();
    // factor rows of the running games: db1 | g·h from the accumulators of s_t (even tile rows = even j)
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        const int g = ((j & 3) + 8 * (j >> 2) + 4 * h) >> 1;
        if (lives[g]) {
            float *f = v.fac + (i0 + g) * TD_FLD;
            const float gg = gs[g], hh = hv[j];
            f[TD_F_DB1 + n] = (gg * w2n) * (1.0f - hh) * hh;
            f[TD_F_GH + n] = gg * hh;
        }
    }
    if (tid < G * 9) {
        const int g = tid / 9, k = tid % 9;
        if (lives[g]) {
            float *f = v.fac + (i0 + g) * TD_FLD;
            if (k < 8) f[TD_F_ROW + k] = __uint_as_float(srow[g][k]);
            else f[TD_F_G] = gs[g];
        }
    }
    if (tid < G * TD_MASK_WORDS) {
        const int g = tid / TD_MASK_WORDS, wd = tid % TD_MASK_WORDS;
        const long long i = i0 + g;
        if (i < n_active) {
            const bool run = lives[g] != 0, first = lives[g] == 2;
            uint32_t now = smask[g][wd];
            const uint32_t valid = wd < 6 ? 0xFFFFFFFFu : (wd == 6 ? 0x3Fu : 0u);
            const uint32_t nz = now & valid;
            now = v.dense ? valid : nz;
            if (!run) now = 0u;
            const uint32_t old = (first || !run || wd == TD_FLAG_WORD) ? 0u : v.amask[i * TD_MASK_WORDS + wd];
            v.amask[i * TD_MASK_WORDS + wd] = wd == TD_FLAG_WORD ? (run ? 1u : 0u) : (old | now);
            v.anew[i * TD_MASK_WORDS + wd] = wd == TD_FLAG_WORD ? (first ? 1u : 0u) : (now & ~old);
            atomicAdd(&s_colsg[g], (unsigned int)__popc(old | now));
            atomicAdd(&s_wrg[g], (unsigned int)__popc(v.full_step ? (old | now) : (run ? (nz | (now & ~old)) : 0u)));
        }
    }
    __syncthreads();
    if (tid < G && i0 + tid < n_active) {
        v.act_cols[i0 + tid] += s_colsg[tid];
        v.wr_cols[i0 + tid] += s_wrg[tid];
        if (adv[tid]) td_advance_slot(v, i0 + tid, (int)(t + 1));
    }
}

// grid (TD_SLICES, n_groups); block 256 threads x float4 of the trace (internal order); `ng` games per group.
// A thread owns four consecutive internal positions = ONE feature column j and four hidden units of it (or four of the
// dense tail b1 | W2 | b2), for every game of its group: a W1 column that is not active in a game is skipped for that game
// (no load, no store, nothing to add), one that became active at this step is written without being read.
template <bool FIRST>
__global__ __launch_bounds__(TD_TRACE_THREADS) void td_trace_kernel(TdView v, long long n_active, int ng, float emul, float ginv,
                                                                    float cmul, int full)
{
    // stored trace ê, scale c (host): ê_new = emul · ê_old + ginv · ∇, update += coef · cmul · ê_new.
    //   ordinary pass (full = 1): emul = λ c_old, ginv = 1, cmul = 1 (c_old = 1 with BGAMD_TD_LAZY=0: e ← λ e + ∇ to the bit)
    //   lazy pass     (full = 0): emul = 1, ginv = 1 / c, cmul = c; a column with x_j = 0 keeps its bits and is not stored
    // a game's factor row: its 32-byte row (x_j is decoded from it) | db1 | g·h | g
    constexpr int L_DB1 = TD_F_DB1, L_GH = TD_F_GH, L_G = TD_F_G, L_LD = TD_FLD;
    __shared__ __attribute__((aligned(16))) float fs[TD_CHUNK][L_LD];
    __shared__ float cs[TD_CHUNK];
    __shared__ uint32_t ms[TD_CHUNK][TD_MASK_WORDS], ns[TD_CHUNK][TD_MASK_WORDS];
    const int tid = threadIdx.x;
    const int p0 = (blockIdx.x * TD_TRACE_THREADS + tid) * 4;
    const bool in_row = p0 < TD_LD;
    const bool is_w1 = p0 < TD_OFF_B1;
    const int jw = is_w1 ? (p0 >> 7) : 0;                     // feature column of a W1 thread
    const int mword = jw >> 5;
    const uint32_t mbit = 1u << (jw & 31);
    const long long g0 = (long long)blockIdx.y * ng;
    // gradient of position p = fs[ia] * xj, xj = the thread's feature x_j (W1 block), 1 (b1 | W2 | b2) or 0 (padding)
    int ia[4];
    float xfix[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = p0 + k;
        xfix[k] = 1.0f;
        if (p < TD_OFF_B1) ia[k] = L_DB1 + (p & (N_HID - 1));
        else if (p < TD_OFF_W2) ia[k] = L_DB1 + (p - TD_OFF_B1);
        else if (p < TD_OFF_B2) ia[k] = L_GH + (p - TD_OFF_W2);
        else if (p == TD_OFF_B2) ia[k] = L_G;
        else { ia[k] = L_G; xfix[k] = 0.0f; }
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < ng; c += TD_CHUNK) {
        const long long gb = g0 + c;
        if (gb >= n_active) break;
        long long left = n_active - gb;
        if (left > ng - c) left = ng - c;
        const int m = left < TD_CHUNK ? (int)left : TD_CHUNK;
        __syncthreads();
        {
            const float4 *src = reinterpret_cast<const float4 *>(v.fac + gb * TD_FLD);
            float4 *dst = reinterpret_cast<float4 *>(&fs[0][0]);
            for (int q = tid; q < m * (TD_FLD / 4); q += TD_TRACE_THREADS) dst[q] = src[q];
            if (tid < m) cs[tid] = v.coef[gb + tid] * cmul;
            if (tid < m * TD_MASK_WORDS) {
                (&ms[0][0])[tid] = v.amask[gb * TD_MASK_WORDS + tid];
                (&ns[0][0])[tid] = v.anew[gb * TD_MASK_WORDS + tid];
            }
        }
        __syncthreads();
        // which of the chunk's games this thread's column exists in, and where it has to be read: the loads of the whole
        // chunk go out before the first use
        uint32_t act = 0, rd = 0;
        if (in_row) {
#pragma unroll
            for (int q = 0; q < TD_CHUNK; ++q)
                if (q < m) {
                    const bool a = is_w1 ? (ms[q][mword] & mbit) != 0u : (ms[q][TD_FLAG_WORD] & 1u) != 0u;     // b1 | W2 | b2: a running game
                    const bool fresh = FIRST || (is_w1 ? (ns[q][mword] & mbit) != 0u : (ns[q][TD_FLAG_WORD] & 1u) != 0u);
                    act |= (a ? 1u : 0u) << q;
                    rd |= ((a && !fresh) ? 1u : 0u) << q;
                }
        }
        td_f32x4 ev[TD_CHUNK];
#pragma unroll
        for (int q = 0; q < TD_CHUNK; ++q) {
            ev[q] = (td_f32x4){0.f, 0.f, 0.f, 0.f};
            if ((rd >> q) & 1u) {
                const td_f32x4 *ep = reinterpret_cast<const td_f32x4 *>(v.e + (gb + q) * TD_LD + p0);
                                ev[q] = *ep;                                  // default cache policy: this kernel serves the rounds whose active trace
                                                              // columns fit the Infinity Cache (< 8 192 running games; +3-4 % over nt)
            }
        }

#pragma unroll
        for (int q = 0; q < TD_CHUNK; ++q) {
            if ((act >> q) & 1u) {
                td_f32x4 x = ev[q];
                // (decoded per thread: one decode per (slice, game) shared through LDS costs a third barrier per chunk and
                //  was slower, 98.1 vs 96.2 ms per 65 536-game replay)
                const float xj = is_w1 ? td_feature_value(reinterpret_cast<const uint32_t *>(&fs[q][TD_F_ROW]), jw) : 1.0f;
                x.x = fmaf(emul, x.x, (fs[q][ia[0]] * (is_w1 ? xj : xfix[0])) * ginv);
                x.y = fmaf(emul, x.y, (fs[q][ia[1]] * (is_w1 ? xj : xfix[1])) * ginv);
                x.z = fmaf(emul, x.z, (fs[q][ia[2]] * (is_w1 ? xj : xfix[2])) * ginv);
                x.w = fmaf(emul, x.w, (fs[q][ia[3]] * (is_w1 ? xj : xfix[3])) * ginv);
                if (full || !is_w1 || xj != 0.0f || !((rd >> q) & 1u)) {      // (not read = activated at this step: written whatever x_j is)
                    td_f32x4 *ep = reinterpret_cast<td_f32x4 *>(v.e + (gb + q) * TD_LD + p0);
                                        *ep = x;
                }
                const float cf = cs[q];
                acc.x = fmaf(cf, x.x, acc.x);
                acc.y = fmaf(cf, x.y, acc.y);
                acc.z = fmaf(cf, x.z, acc.z);
                acc.w = fmaf(cf, x.w, acc.w);
            }
        }
    }
    if (in_row) *reinterpret_cast<float4 *>(v.partial + (long long)blockIdx.y * TD_LD + p0) = acc;
}

// The same pass for LARGE rounds: one workgroup of 512 threads owns the WHOLE trace row (25 664 floats = 13 float4 per thread,
// its partial sums in 52 registers) for a strided share of the running games, so a game's factor row is staged once per
// step instead of once per 1 024-float slice (26 x: at 65 536 games 1.85 GB per step through the L2s next to the 3.3 GB of
// trace traffic), and a thread's loads of a game (its active columns among 13) go out together, for two games at a time.
// Thread tid: hidden units 4 (tid & 31) .. + 3 of the columns j = 16 k + (tid >> 5), k = 0..12 -- side and thermometer level of
// its features are the same for every k, only the board point moves (2 k + (tid >> 8) + 1), so x_j is one shift of a word formed
// once per game.  k = 12: columns 192..197 (tid < 192), then b1 | W2 | b2 | padding (tid 192..271).
// Same arithmetic per element as td_trace_kernel; the partial sums group the games differently (sums differ by rounding).
#ifndef BG_TD_WIDE_W
#define BG_TD_WIDE_W 512
#endif
constexpr int TD_WIDE_THREADS = BG_TD_WIDE_W;                 // 512 or 1024
constexpr int TD_WIDE_KL = 6144 / TD_WIDE_THREADS;            // the k of the tail group (columns 192..197 | b1 | W2 | b2)
constexpr int TD_WIDE_K = TD_WIDE_KL + 1;
constexpr int TD_WIDE_CPG = TD_WIDE_THREADS / 32;             // columns per k
#ifndef BG_TD_WIDE_GPI
#define BG_TD_WIDE_GPI 2
#endif
// NT: nontemporal loads / stores (rounds whose traces stream from HBM); off where the active columns fit the Infinity Cache
template <bool FIRST, bool NT>
__global__ __launch_bounds__(TD_WIDE_THREADS) void td_trace_wide_kernel(TdView v, long long n_active, float emul, float ginv, float cmul, int full)
{
    constexpr int GPI = BG_TD_WIDE_GPI;
    __shared__ __attribute__((aligned(16))) float fs[TD_CHUNK][TD_FLD];
    __shared__ float cs[TD_CHUNK];
    __shared__ uint32_t ms[TD_CHUNK][TD_MASK_WORDS], ns[TD_CHUNK][TD_MASK_WORDS];
    const int tid = threadIdx.x;
    const int c = tid >> 5;                                   // column inside a group of TD_WIDE_CPG
    const int n0 = (tid & 31) * 4;                            // hidden units n0 .. n0 + 3
    const int base = 4 * ((c >> 2) & 1), level = c & 3;       // of the board features j = CPG k + c, k < KL
    const int pos0 = (c >> 3) + 1;                            // board position of column k: pos0 + CPG k / 8
    // k = 12: tail columns / dense tail
    const bool t_w1 = tid < 192, t_in = tid < 272;
    int ia[4];
    float xfix[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int p = (TD_WIDE_KL * TD_WIDE_THREADS + tid) * 4 + u;
        xfix[u] = 1.0f;
        if (p < TD_OFF_B1) ia[u] = TD_F_DB1 + (p & (N_HID - 1));
        else if (p < TD_OFF_W2) ia[u] = TD_F_DB1 + (p - TD_OFF_B1);
        else if (p < TD_OFF_B2) ia[u] = TD_F_GH + (p - TD_OFF_W2);
        else if (p == TD_OFF_B2) ia[u] = TD_F_G;
        else { ia[u] = TD_F_G; xfix[u] = 0.0f; }
    }
    td_f32x4 acc[TD_WIDE_K];
#pragma unroll
    for (int k = 0; k < TD_WIDE_K; ++k) acc[k] = (td_f32x4){0.f, 0.f, 0.f, 0.f};

    for (long long chunk = blockIdx.x; chunk * TD_CHUNK < n_active; chunk += gridDim.x) {
        const long long gb = chunk * TD_CHUNK;
        const long long left = n_active - gb;
        const int m = left < TD_CHUNK ? (int)left : TD_CHUNK;
        __syncthreads();
        {
            const float4 *src = reinterpret_cast<const float4 *>(v.fac + gb * TD_FLD);
            float4 *dst = reinterpret_cast<float4 *>(&fs[0][0]);
            for (int q = tid; q < m * (TD_FLD / 4); q += TD_WIDE_THREADS) dst[q] = src[q];
            if (tid < m) cs[tid] = v.coef[gb + tid] * cmul;
            if (tid < m * TD_MASK_WORDS) {
                (&ms[0][0])[tid] = v.amask[gb * TD_MASK_WORDS + tid];
                (&ns[0][0])[tid] = v.anew[gb * TD_MASK_WORDS + tid];
            }
        }
        __syncthreads();
        for (int q0 = 0; q0 < m; q0 += GPI) {
            uint32_t act[GPI], rd[GPI];
            td_f32x4 ev[GPI][TD_WIDE_K];
#pragma unroll
            for (int u = 0; u < GPI; ++u) {
                const int q = q0 + u;
                act[u] = 0; rd[u] = 0;
                if (q < m) {
#pragma unroll
                    for (int k = 0; k < TD_WIDE_K; ++k) {
                        const int sh = ((TD_WIDE_CPG * k) & 31) + c;           // column CPG k + c: its mask word and bit
                        bool a, fresh;
                        if (k < TD_WIDE_KL || t_w1) {
                            a = (ms[q][(TD_WIDE_CPG * k) >> 5] >> sh) & 1u;
                            fresh = FIRST || ((ns[q][(TD_WIDE_CPG * k) >> 5] >> sh) & 1u);
                        } else { a = t_in && (ms[q][TD_FLAG_WORD] & 1u); fresh = FIRST || (ns[q][TD_FLAG_WORD] & 1u); }
                        act[u] |= (a ? 1u : 0u) << k;
                        rd[u] |= ((a && !fresh) ? 1u : 0u) << k;
                    }
                }
                const float *eg = v.e + (gb + q) * TD_LD + tid * 4;
#pragma unroll
                for (int k = 0; k < TD_WIDE_K; ++k) {
                    ev[u][k] = (td_f32x4){0.f, 0.f, 0.f, 0.f};
                    if ((rd[u] >> k) & 1u) {
                        const td_f32x4 *ep = reinterpret_cast<const td_f32x4 *>(eg + k * (TD_WIDE_THREADS * 4));
                        ev[u][k] = NT ? __builtin_nontemporal_load(ep) : *ep;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < GPI; ++u) {
                const int q = q0 + u;
                if (q >= m) break;
                const uint32_t *rw = reinterpret_cast<const uint32_t *>(&fs[q][TD_F_ROW]);
                const uint32_t r0 = rw[base], r1 = rw[base + 1], r2 = rw[base + 2], r3 = rw[base + 3];
                // the positions where this thread's thermometer level is on (level 3: count >= 4, its value comes from the count)
                const uint32_t tw = level == 0 ? (r0 | r1 | r2 | r3) : level == 1 ? (r1 | r2 | r3) : level == 2 ? ((r0 & r1) | r2 | r3) : (r2 | r3);
                const float4 db = *reinterpret_cast<const float4 *>(&fs[q][TD_F_DB1 + n0]);
                const float cf = cs[q];
                float *eg = v.e + (gb + q) * TD_LD + tid * 4;
#pragma unroll
                for (int k = 0; k < TD_WIDE_K; ++k) {
                    if (!((act[u] >> k) & 1u)) continue;
                    td_f32x4 x = ev[u][k];
                    bool wr;
                    if (k < TD_WIDE_KL) {
                        const int pos = pos0 + (TD_WIDE_CPG / 8) * k;
                        float xj = (float)((tw >> pos) & 1u);
                        if (level == 3) {
                            const int cnt = (int)(((r0 >> pos) & 1u) | (((r1 >> pos) & 1u) << 1) | (((r2 >> pos) & 1u) << 2) | (((r3 >> pos) & 1u) << 3));
                            xj = cnt > 3 ? 0.5f * (float)(cnt - 3) : 0.0f;
                        }
                        x.x = fmaf(emul, x.x, (db.x * xj) * ginv);
                        x.y = fmaf(emul, x.y, (db.y * xj) * ginv);
                        x.z = fmaf(emul, x.z, (db.z * xj) * ginv);
                        x.w = fmaf(emul, x.w, (db.w * xj) * ginv);
                        wr = xj != 0.0f;
                    } else {
                        const float xj = t_w1 ? td_feature_value(rw, 192 + c) : 1.0f;
                        x.x = fmaf(emul, x.x, (fs[q][ia[0]] * (t_w1 ? xj : xfix[0])) * ginv);
                        x.y = fmaf(emul, x.y, (fs[q][ia[1]] * (t_w1 ? xj : xfix[1])) * ginv);
                        x.z = fmaf(emul, x.z, (fs[q][ia[2]] * (t_w1 ? xj : xfix[2])) * ginv);
                        x.w = fmaf(emul, x.w, (fs[q][ia[3]] * (t_w1 ? xj : xfix[3])) * ginv);
                        wr = !t_w1 || xj != 0.0f;
                    }
                    if (full || wr || !((rd[u] >> k) & 1u)) {
                        td_f32x4 *ep = reinterpret_cast<td_f32x4 *>(eg + k * (TD_WIDE_THREADS * 4));
                        if (NT) __builtin_nontemporal_store(x, ep); else *ep = x;
                    }
                    acc[k].x = fmaf(cf, x.x, acc[k].x);
                    acc[k].y = fmaf(cf, x.y, acc[k].y);
                    acc[k].z = fmaf(cf, x.z, acc[k].z);
                    acc[k].w = fmaf(cf, x.w, acc[k].w);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < TD_WIDE_K; ++k)
        if (k < TD_WIDE_KL || t_in)
            *reinterpret_cast<td_f32x4 *>(v.partial + (long long)blockIdx.x * TD_LD + (k * TD_WIDE_THREADS + tid) * 4) = acc[k];
}

// A 16-byte store that is written through to the memory side (sc1) instead of staying dirty in the XCD's L2 until the kernel ends:
// a step's trace pass leaves ~54 MB dirty (26 MB of partial sums + the written trace columns), and the write-back of dirty lines at
// the kernel boundary is serial time (MI355X_MICROARCH.md, "boundary": + B / 6 TB/s).  BG_TD_WT=0: plain stores.
#ifndef BG_TD_WT
#define BG_TD_WT 1
#endif
// INVARIANT the asm relies on: the compiler does not see this store (no alias analysis, no waitcnt bookkeeping for it), so NOTHING in the
// launch may read back an address stored through it -- issue() / process() of the trace passes read a trace row strictly before they
// store it and never again, the partial sums are written once at the end -- and vmcnt retires in order on gfx9, so the counted waits
// around it stay correct.  The sc1 modifier is gfx94x / gfx950 ISA: any other target takes the plain store.
__device__ __forceinline__ void td_store_wt(td_f32x4 *p, td_f32x4 x)
{
#if BG_TD_WT && (defined(__gfx942__) || defined(__gfx950__))
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(x) : "memory");
#else
    *p = x;
#endif
}

// The whole-row pass for MID-SIZED steps (a streamed replay through 1 024 .. 8 191 slots: what configs 4 / 5 run at the batch
// sizes the quality study allows), software-pipelined.  Such a step has one chunk of TD_CHUNK games per CU; td_trace_wide_kernel
// walks it in rounds of two games -- masks, loads, WAIT, arithmetic, stores -- and the rounds' latencies add up (26.7 us per step
// at 2 048 slots for 104 MB that sit in the Infinity Cache).  Here a workgroup has the CU to itself (8 waves, up to 256 VGPRs):
// the loads of game q + 1 (13 float4 per thread, its active columns among them) go out BEFORE game q is computed and stored, into
// a second register set.  Thread mapping, arithmetic and the order of the partial sums are td_trace_wide_kernel's: same bits.
template <bool FIRST>
__global__ __launch_bounds__(TD_WIDE_THREADS) void td_trace_pipe_kernel(TdView v, long long n_active, float emul, float ginv, float cmul, int full)
{
    static_assert(TD_WIDE_THREADS == 512, "the pipelined pass is written for 512-thread workgroups");
    __shared__ __attribute__((aligned(16))) float fs[TD_CHUNK][TD_FLD];
    __shared__ float cs[TD_CHUNK];
    __shared__ uint32_t ms[TD_CHUNK][TD_MASK_WORDS], ns[TD_CHUNK][TD_MASK_WORDS];
    const int tid = threadIdx.x;
    const int c = tid >> 5;
    const int n0 = (tid & 31) * 4;
    const int base = 4 * ((c >> 2) & 1), level = c & 3;
    const int pos0 = (c >> 3) + 1;
    const bool t_w1 = tid < 192, t_in = tid < 272;
    int ia[4];
    float xfix[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int p = (TD_WIDE_KL * TD_WIDE_THREADS + tid) * 4 + u;
        xfix[u] = 1.0f;
        if (p < TD_OFF_B1) ia[u] = TD_F_DB1 + (p & (N_HID - 1));
        else if (p < TD_OFF_W2) ia[u] = TD_F_DB1 + (p - TD_OFF_B1);
        else if (p < TD_OFF_B2) ia[u] = TD_F_GH + (p - TD_OFF_W2);
        else if (p == TD_OFF_B2) ia[u] = TD_F_G;
        else { ia[u] = TD_F_G; xfix[u] = 0.0f; }
    }
    td_f32x4 acc[TD_WIDE_K];
#pragma unroll
    for (int k = 0; k < TD_WIDE_K; ++k) acc[k] = (td_f32x4){0.f, 0.f, 0.f, 0.f};

    for (long long chunk = blockIdx.x; chunk * TD_CHUNK < n_active; chunk += gridDim.x) {
        const long long gb = chunk * TD_CHUNK;
        const long long left = n_active - gb;
        const int m = left < TD_CHUNK ? (int)left : TD_CHUNK;
        __syncthreads();
        {
            const float4 *src = reinterpret_cast<const float4 *>(v.fac + gb * TD_FLD);
            float4 *dst = reinterpret_cast<float4 *>(&fs[0][0]);
            for (int q = tid; q < m * (TD_FLD / 4); q += TD_WIDE_THREADS) dst[q] = src[q];
            if (tid < m) cs[tid] = v.coef[gb + tid] * cmul;
            if (tid < m * TD_MASK_WORDS) {
                (&ms[0][0])[tid] = v.amask[gb * TD_MASK_WORDS + tid];
                (&ns[0][0])[tid] = v.anew[gb * TD_MASK_WORDS + tid];
            }
        }
        __syncthreads();
        // which of the thread's 13 positions exist in game q (act) and have to be read (rd), and the reads themselves
        auto issue = [&](int q, uint32_t &act, uint32_t &rd, td_f32x4 (&ev)[TD_WIDE_K]) {
            act = 0; rd = 0;
            if (q < m) {
#pragma unroll
                for (int k = 0; k < TD_WIDE_K; ++k) {
                    const int sh = ((TD_WIDE_CPG * k) & 31) + c;
                    bool a, fresh;
                    if (k < TD_WIDE_KL || t_w1) {
                        a = (ms[q][(TD_WIDE_CPG * k) >> 5] >> sh) & 1u;
                        fresh = FIRST || ((ns[q][(TD_WIDE_CPG * k) >> 5] >> sh) & 1u);
                    } else { a = t_in && (ms[q][TD_FLAG_WORD] & 1u); fresh = FIRST || (ns[q][TD_FLAG_WORD] & 1u); }
                    act |= (a ? 1u : 0u) << k;
                    rd |= ((a && !fresh) ? 1u : 0u) << k;
                }
            }
            // EVERY one of the 13 loads is issued -- a position that is not read fetches one shared dummy line instead -- so that the
            // number of loads in flight is static and the wait before game q's arithmetic is a counted vmcnt that leaves game q + 1's
            // loads in flight (conditional loads made the compiler wait for vmcnt(0): no pipelining at all)
            const float *eg = v.e + (gb + (q < m ? q : 0)) * TD_LD + tid * 4;
            const float *dummy = v.partial + (size_t)TD_MAX_GROUPS * TD_LD;      // 16 zeroed bytes behind the partial sums: a position
                                                                                //   that is not read needs no select, it reads 0
#pragma unroll
            for (int k = 0; k < TD_WIDE_K; ++k) {
                const float *src = ((rd >> k) & 1u) ? eg + k * (TD_WIDE_THREADS * 4) : dummy;
                ev[k] = *reinterpret_cast<const td_f32x4 *>(src);
            }
        };
        auto process = [&](int q, uint32_t act, uint32_t rd, td_f32x4 (&ev)[TD_WIDE_K]) {
            const uint32_t *rw = reinterpret_cast<const uint32_t *>(&fs[q][TD_F_ROW]);
            const uint32_t r0 = rw[base], r1 = rw[base + 1], r2 = rw[base + 2], r3 = rw[base + 3];
            const uint32_t tw = level == 0 ? (r0 | r1 | r2 | r3) : level == 1 ? (r1 | r2 | r3) : level == 2 ? ((r0 & r1) | r2 | r3) : (r2 | r3);
            const float4 db = *reinterpret_cast<const float4 *>(&fs[q][TD_F_DB1 + n0]);
            const float4 dg = make_float4(db.x * ginv, db.y * ginv, db.z * ginv, db.w * ginv);   // thermometer levels: x_j is 0 or 1, and
            const float cf = cs[q];                                                             //   (db x_j) ginv = x_j ? db ginv : 0 to the bit
            float *eg = v.e + (gb + q) * TD_LD + tid * 4;
#pragma unroll
            for (int k = 0; k < TD_WIDE_K; ++k) {
                if (!((act >> k) & 1u)) continue;
                td_f32x4 x = ev[k];                                     // (zero where the position is not read: the dummy line)
                bool wr;
                if (k < TD_WIDE_KL) {
                    const int pos = pos0 + (TD_WIDE_CPG / 8) * k;
                    if (level == 3) {
                        const int cnt = (int)(((r0 >> pos) & 1u) | (((r1 >> pos) & 1u) << 1) | (((r2 >> pos) & 1u) << 2) | (((r3 >> pos) & 1u) << 3));
                        const float xj = cnt > 3 ? 0.5f * (float)(cnt - 3) : 0.0f;
                        x.x = fmaf(emul, x.x, (db.x * xj) * ginv);
                        x.y = fmaf(emul, x.y, (db.y * xj) * ginv);
                        x.z = fmaf(emul, x.z, (db.z * xj) * ginv);
                        x.w = fmaf(emul, x.w, (db.w * xj) * ginv);
                        wr = xj != 0.0f;
                    } else {
                        wr = ((tw >> pos) & 1u) != 0u;
                        x.x = fmaf(emul, x.x, wr ? dg.x : 0.0f);
                        x.y = fmaf(emul, x.y, wr ? dg.y : 0.0f);
                        x.z = fmaf(emul, x.z, wr ? dg.z : 0.0f);
                        x.w = fmaf(emul, x.w, wr ? dg.w : 0.0f);
                    }
                } else {
                    const float xj = t_w1 ? td_feature_value(rw, 192 + c) : 1.0f;
                    x.x = fmaf(emul, x.x, (fs[q][ia[0]] * (t_w1 ? xj : xfix[0])) * ginv);
                    x.y = fmaf(emul, x.y, (fs[q][ia[1]] * (t_w1 ? xj : xfix[1])) * ginv);
                    x.z = fmaf(emul, x.z, (fs[q][ia[2]] * (t_w1 ? xj : xfix[2])) * ginv);
                    x.w = fmaf(emul, x.w, (fs[q][ia[3]] * (t_w1 ? xj : xfix[3])) * ginv);
                    wr = !t_w1 || xj != 0.0f;
                }
                if (full || wr || !((rd >> k) & 1u)) td_store_wt(reinterpret_cast<td_f32x4 *>(eg + k * (TD_WIDE_THREADS * 4)), x);
                acc[k].x = fmaf(cf, x.x, acc[k].x);
                acc[k].y = fmaf(cf, x.y, acc[k].y);
                acc[k].z = fmaf(cf, x.z, acc[k].z);
                acc[k].w = fmaf(cf, x.w, acc[k].w);
            }
        };
        uint32_t actA, rdA, actB, rdB;
        td_f32x4 evA[TD_WIDE_K], evB[TD_WIDE_K];
        issue(0, actA, rdA, evA);
        for (int q0 = 0; q0 < m; q0 += 2) {
            issue(q0 + 1, actB, rdB, evB);                    // game q0 + 1 is on its way while game q0 is computed and stored
            process(q0, actA, rdA, evA);
            if (q0 + 1 >= m) break;
            issue(q0 + 2, actA, rdA, evA);
            process(q0 + 1, actB, rdB, evB);
        }
    }
#pragma unroll
    for (int k = 0; k < TD_WIDE_K; ++k)
        if (k < TD_WIDE_KL || t_in)
            td_store_wt(reinterpret_cast<td_f32x4 *>(v.partial + (long long)blockIdx.x * TD_LD + (k * TD_WIDE_THREADS + tid) * 4), acc[k]);
}

// A mid-sized training step in TWO launches instead of three: the forward pass of a chunk's TD_CHUNK slots (td_forward_mfma_kernel's
// product and epilogue, on a half-filled 32-row tile) runs in the trace workgroup that owns those slots, leaves the factor rows,
// coefficients and masks in LDS -- no `fac` / `coef` round trip through memory, no staging -- and the software-pipelined whole-row
// trace pass (td_trace_pipe_kernel) follows behind one block barrier.  The forward kernel of such a step is a latency chain of 10.8 us
// on 128 workgroups plus a kernel boundary; here every CU computes the 16 rows it needs itself while nothing else could run.
// Same arithmetic in the same order as the two kernels it replaces: same bits.
// G = slots per workgroup: the host picks the smallest of 1, 2, 4, 8, 16 that needs no more workgroups than there are CUs -- a step of
// 1 024 slots runs on all 256 CUs with four games each instead of on 128 with eight (its trace pass halves), a step of 4 096 slots keeps
// the two-launch form with a FULL 32-row tile in the forward pass instead of falling back to three launches.
// DELAY (round 4, bgamd_td_set_delay): the update of step t is applied one step LATE -- step t + 1 still runs on the weights step t ran on
// plus the update of step t - 1 -- and then a training step is ONE launch: the reduction of step t - 1's partial sums (td_reduce_kernel's work:
// 26 MB over 256 partial rows, 6.2 us and two kernel boundaries of the 39 us step) is done by the four waves of every workgroup that idle while
// waves 0-3 run the forward pass, 128 parameters per workgroup, into the OTHER weight buffer: this launch reads theta / wl3 / the partial sums
// of step t - 1 and writes th_next / wl3_next / its own partial sums, nobody reads what anybody writes.  A documented deviation (one game no
// longer reproduces the reference's update step for step): opt-in.
constexpr int TD_DELAY_SLICE = 128;                                              // parameters (internal order) per workgroup and slice: 4 cache lines
constexpr int TD_DELAY_SLICES = (TD_LD + TD_DELAY_SLICE - 1) / TD_DELAY_SLICE;   // 201 (the last one reads 64 floats past a row: the next row, or the zeroed tail)

// threads j = 0 .. 255 of a workgroup: float4 f = j & 31 of slice `slice`, partial rows r, r + 8, ... -> red[r][f]; call td_delay_finish after a barrier
__device__ __forceinline__ void td_delay_gather(const float *__restrict__ part_prev, int n_prev, int slice, int j, td_f32x4 (*red)[32])
{
    const int f = j & 31, r = j >> 5;
    td_f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const float *src = part_prev + (long long)slice * TD_DELAY_SLICE + f * 4;
    // TD_MAX_GROUPS / 8 = 32 rows per thread at most: all of a batch's loads go out before the first add (one round trip per 16 rows, not per row);
    // the order of the adds is fixed (ascending row): bit-reproducible
    static_assert(TD_MAX_GROUPS <= 256, "the delayed update's gather is written for <= 256 partial rows");
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        td_f32x4 x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int g = r + 8 * (16 * b + i);
            x[i] = (td_f32x4){0.f, 0.f, 0.f, 0.f};
            if (g < n_prev) x[i] = *reinterpret_cast<const td_f32x4 *>(src + (long long)g * TD_LD);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += x[i];
    }
    red[r][f] = s;
}
// threads j = 0 .. 127: parameter j of the slice: th_next = th_cur + the sum over the 8 row groups (fixed order: bit-reproducible), planes refreshed
__device__ __forceinline__ void td_delay_finish(const float *__restrict__ th_cur, float *__restrict__ th_next, uint16_t *__restrict__ wl3_next,
                                                float *__restrict__ w1t_next, int slice, int j, td_f32x4 (*red)[32])
{
    const int q = slice * TD_DELAY_SLICE + j;                                    // INTERNAL position
    float u = 0.0f;
#pragma unroll
    for (int g = 0; g < 8; ++g) u += red[g][j >> 2][j & 3];
    if (q < TD_P) {
        const int p = td_param_of_internal(q);
        const float th = th_cur[p] + u;
        th_next[p] = th;
        if (q < TD_OFF_B1) {
            if (w1t_next) w1t_next[q] = th;
            root3_store_weight(wl3_next, q & (N_HID - 1), q >> 7, th);
        }
    }
}

// after the last step of a delayed replay (and before its first: n_prev = 0 copies): theta_out = th_cur + the update still outstanding
__global__ __launch_bounds__(256) void td_delay_flush_kernel(const float *__restrict__ part_prev, int n_prev, const float *__restrict__ th_cur,
                                                             float *__restrict__ th_next, uint16_t *__restrict__ wl3_next, float *__restrict__ w1t_next)
{
    __shared__ td_f32x4 red[8][32];
    td_delay_gather(part_prev, n_prev, blockIdx.x, threadIdx.x, red);
    __syncthreads();
    if (threadIdx.x < TD_DELAY_SLICE) td_delay_finish(th_cur, th_next, wl3_next, w1t_next, blockIdx.x, threadIdx.x, red);
}

template <bool FIRST, int G, bool DELAY = false>
__global__ __launch_bounds__(TD_WIDE_THREADS) void td_step_fused_kernel(TdView v, long long t, long long n_active, double alpha, float emul, float ginv,
                                                                        float cmul, int full, const float *__restrict__ part_prev = nullptr,
                                                                        int n_prev = 0, float *__restrict__ th_next = nullptr,
                                                                        uint16_t *__restrict__ wl3_next = nullptr)
{
    __shared__ td_f32x4 dred[DELAY ? 8 : 1][32];
    static_assert(TD_WIDE_THREADS == 512 && G >= 1 && G <= 16 && (G & (G - 1)) == 0, "512-thread workgroups, 2 G rows in one 32-row tile");
    __shared__ __attribute__((aligned(16))) float fs[G][TD_FLD];
    __shared__ float cs[G];
    __shared__ uint32_t ms[G][TD_MASK_WORDS], ns[G][TD_MASK_WORDS];
    __shared__ uint2 sLut[16];
    __shared__ float sd[2 * G][N_HID + 1];
    __shared__ uint32_t srow[G][8], smask[G][TD_MASK_WORDS];
    __shared__ float outs[2 * G], gs[G];
    __shared__ unsigned int s_colsg[G], s_wrg[G];
    __shared__ int adv[G], lives[G];
    const int tid = threadIdx.x;
    // ---- trace-pass thread constants (td_trace_wide_kernel's mapping)
    const int c = tid >> 5;
    const int n0 = (tid & 31) * 4;
    const int base = 4 * ((c >> 2) & 1), level = c & 3;
    const int pos0 = (c >> 3) + 1;
    const bool t_w1 = tid < 192, t_in = tid < 272;
    int ia[4];
    float xfix[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int p = (TD_WIDE_KL * TD_WIDE_THREADS + tid) * 4 + u;
        xfix[u] = 1.0f;
        if (p < TD_OFF_B1) ia[u] = TD_F_DB1 + (p & (N_HID - 1));
        else if (p < TD_OFF_W2) ia[u] = TD_F_DB1 + (p - TD_OFF_B1);
        else if (p < TD_OFF_B2) ia[u] = TD_F_GH + (p - TD_OFF_W2);
        else if (p == TD_OFF_B2) ia[u] = TD_F_G;
        else { ia[u] = TD_F_G; xfix[u] = 0.0f; }
    }
    // ---- forward-pass thread constants (td_forward_mfma_kernel's mapping, waves 0-3 only)
    const bool fw = tid < 256;
    const int lane = tid & 63, fc = (tid >> 6) & 3;
    const int r = lane & 31, h = lane >> 5;
    const int n = 32 * fc + r;
    if (tid < 16) sLut[tid] = v.lut[tid];

    {   // ONE chunk per workgroup (the launch has a workgroup per chunk: n_active <= CUs x TD_CHUNK; larger steps take two launches) --
        // the forward pass's 156 weight registers and the trace pass's 156 of loads in flight + 52 of partial sums then never coexist
        const long long i0 = (long long)blockIdx.x * G;
        const long long left = n_active - i0;
        const int m = left < G ? (int)left : G;
        // what the epilogue below will need from memory and whose address is known now is requested now: the old column masks, the slots'
        // game records and running sums -- each of them was a load a block barrier waited for (forward pass 6.1 -> 5.x us per step)
        uint32_t pre_mask = 0u;
        int4 pre_gm = make_int4(0, 0, 0, 0);
        double pre_sq = 0.0;
        unsigned int pre_nupd = 0u;
        if (tid < G * TD_MASK_WORDS && i0 + tid / TD_MASK_WORDS < n_active) pre_mask = v.amask[(i0 + tid / TD_MASK_WORDS) * TD_MASK_WORDS + tid % TD_MASK_WORDS];
        if (tid < G && i0 + tid < n_active) { pre_gm = v.gmeta[i0 + tid]; pre_sq = v.sq[i0 + tid]; pre_nupd = v.nupd[i0 + tid]; }
        __syncthreads();                                             // sLut
        // =============================== forward pass of slots i0 .. i0 + 7 ===============================
        if (tid < G) { s_colsg[tid] = 0; s_wrg[tid] = 0; adv[tid] = 0; lives[tid] = 0; }
        if (tid < G * TD_MASK_WORDS) (&smask[0][0])[tid] = 0;
        float hv[16];
        float w2n = 0.0f;
        if (fw) {
            // the first K-steps' weight planes are requested BEFORE the slot's rows (game record -> row: two dependent loads): they do not
            // depend on them, and behind them they would wait a third round trip
            constexpr int W_AHEAD = 3;
            const uint4 *wp = reinterpret_cast<const uint4 *>(v.wl3) + (size_t)fc * 64 + lane;
            uint4 wahead[W_AHEAD][3];
#pragma unroll
            for (int s = 0; s < W_AHEAD; ++s)
#pragma unroll
                for (int part = 0; part < 3; ++part) wahead[s][part] = wp[(size_t)part * ROOT3_PART_U4 + (size_t)s * 4 * 64];
            const TrajRowsFetch fetch{v.rows, v.gmeta, t, v.n_lanes, v.T};
            uint32_t p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            bool row_ok = false;
            if (r < 2 * G && (i0 * 2 + r) < 2 * n_active) {
                uint4 u0, u1;
                if (fetch.get(i0 * 2 + r, u0, u1)) {
                    row_ok = true;
                    p[0] = u0.x; p[1] = u0.y; p[2] = u0.z; p[3] = u0.w; p[4] = u1.x; p[5] = u1.y; p[6] = u1.z; p[7] = u1.w;
                }
            }
            if (fc == 0 && h == 0 && (r & 1) == 0 && r < 2 * G) {
#pragma unroll
                for (int k = 0; k < 8; ++k) srow[r >> 1][k] = p[k];
            }
            const float bb = v.theta[TD_OFF_B1 + n];
            w2n = v.theta[TD_OFF_W2 + n];
            const Side sa{{p[0], p[1], p[2], p[3]}}, sb{{p[4], p[5], p[6], p[7]}};
            floatx16 a16 = {0};
#pragma unroll
            for (int s = 0; s < K16_STEPS; ++s) {
                union { uint4 u; bf16x8 v; } a;
                if (s < 12) {
                    const int pos = 2 * s + h + 1;
                    const uint2 l0 = sLut[count_at(sa, pos)], l1 = sLut[count_at(sb, pos)];
                    a.u = make_uint4(l0.x, l0.y, l1.x, l1.y);
                } else {
                    const int turn = (p[0] & TURN_BIT) ? 1 : 0;
                    const uint32_t t0 = (turn == 0 && row_ok) ? 0x3F80u : 0u, t1 = (turn == 0 || !row_ok) ? 0u : 0x3F80u;
                    const uint32_t bar1 = f32_to_bf16_rne(0.5f * (float)count_at(sa, 0)), bar2 = f32_to_bf16_rne(0.5f * (float)count_at(sb, 25));
                    const uint32_t off1 = f32_to_bf16_rne((float)count_at(sa, 25)), off2 = f32_to_bf16_rne((float)count_at(sb, 0));
                    a.u = h ? make_uint4(0, 0, 0, 0) : make_uint4(t0 | (t1 << 16), bar1 | (bar2 << 16), off1 | (off2 << 16), 0u);
                }
#pragma unroll
                for (int part = 0; part < 3; ++part) {
                    union { uint4 u; bf16x8 v; } wv;
                    wv.u = s < W_AHEAD ? wahead[s][part] : wp[(size_t)part * ROOT3_PART_U4 + (size_t)s * 4 * 64];
                    a16 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, wv.v, a16, 0, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                hv[j] = td_sigmoid(a16[j] + bb);
                const int row = (j & 3) + 8 * (j >> 2) + 4 * h;
                if (row < 2 * G) sd[row][n] = w2n * hv[j];
            }
        } else if (DELAY) {
            // waves 4-7, idle during the forward pass: step t - 1's partial sums -> the weights of step t + 1, one 128-parameter slice
            // per workgroup (the host takes this route only for launches of >= TD_DELAY_SLICES workgroups)
            if ((int)blockIdx.x < TD_DELAY_SLICES) td_delay_gather(part_prev, n_prev, blockIdx.x, tid - 256, dred);
        }
        __syncthreads();                                             // srow, sd (and the delayed update's row-group sums)
        if (DELAY && tid >= 256 && tid < 256 + TD_DELAY_SLICE && (int)blockIdx.x < TD_DELAY_SLICES)
            td_delay_finish(v.theta, th_next, wl3_next, nullptr, blockIdx.x, tid - 256, dred);
        if (tid < G * 16) {                                          // ever-active feature masks: thread = (game, two board points | tail)
            const int g = tid >> 4, q = tid & 15;
            if (i0 + g < n_active) {
#pragma unroll
                for (int pt = q; pt < 25; pt += 16) {
                    const uint32_t bits = td_nonzero_bits(srow[g], pt);
                    if (bits) atomicOr(&smask[g][pt >> 2], bits << (8 * (pt & 3)));
                }
            }
        }
        if (tid < 2 * G * 8) {                                       // output unit: thread = (row, eighth of the hidden layer)
            const int row = tid >> 3, e8 = tid & 7;
            float sum = 0.0f;
#pragma unroll
            for (int k = 0; k < N_HID / 8; ++k) sum += sd[row][e8 * (N_HID / 8) + k];
            sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 4, 64);
            if (e8 == 0) outs[row] = td_sigmoid(sum + v.theta[TD_OFF_B2]);
        }
        __syncthreads();
        if (tid < G) {                                               // delta, g, coef per game
            const long long i = i0 + tid;
            float g = 0.0f, cf = 0.0f;
            if (i < n_active) {
                const int4 gm = pre_gm;
                const long long tl = t - gm.w;
                if (tl < gm.y) {
                    const float val = outs[2 * tid], vnext = outs[2 * tid + 1];
                    const float z = (gm.z & 1) ? 1.0f : 0.0f;
                    const float delta = (tl + 1 >= gm.y) ? z - val : vnext - val;
                    g = val * (1.0f - val);
                    cf = (float)(alpha * (double)delta);
                    v.sq[i] = pre_sq + (double)delta * (double)delta;
                    v.nupd[i] = pre_nupd + 1u;
                    lives[tid] = 1 + (tl == 0 ? 1 : 0);
                    if (tl + 1 >= gm.y) adv[tid] = 1;
                }
                v.coef[i] = cf;
            }
            gs[tid] = g;
            cs[tid] = cf * cmul;
        }
        __syncthreads();
        // factor rows of the chunk's games, straight into the trace pass's LDS image (a slot without a running game: zeros)
        if (fw) {
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                const int row = (j & 3) + 8 * (j >> 2) + 4 * h;
                if (row < 2 * G) {
                    const int g = row >> 1;
                    const float gg = lives[g] ? gs[g] : 0.0f, hh = hv[j];
                    fs[g][TD_F_DB1 + n] = (gg * w2n) * (1.0f - hh) * hh;
                    fs[g][TD_F_GH + n] = gg * hh;
                }
            }
        }
        if (tid < G * 9) {
            const int g = tid / 9, k = tid % 9;
            if (k < 8) fs[g][TD_F_ROW + k] = __uint_as_float(srow[g][k]);
            else fs[g][TD_F_G] = lives[g] ? gs[g] : 0.0f;
        }
        if (tid < G * TD_MASK_WORDS) {
            const int g = tid / TD_MASK_WORDS, wd = tid % TD_MASK_WORDS;
            const long long i = i0 + g;
            uint32_t am = 0u, an = 0u;
            if (i < n_active) {
                const bool run = lives[g] != 0, first = lives[g] == 2;
                uint32_t now = smask[g][wd];
                const uint32_t valid = wd < 6 ? 0xFFFFFFFFu : (wd == 6 ? 0x3Fu : 0u);
                const uint32_t nz = now & valid;
                now = v.dense ? valid : nz;
                if (!run) now = 0u;
                const uint32_t old = (first || !run || wd == TD_FLAG_WORD) ? 0u : pre_mask;
                am = wd == TD_FLAG_WORD ? (run ? 1u : 0u) : (old | now);
                an = wd == TD_FLAG_WORD ? (first ? 1u : 0u) : (now & ~old);
                v.amask[i * TD_MASK_WORDS + wd] = am;
                v.anew[i * TD_MASK_WORDS + wd] = an;
                atomicAdd(&s_colsg[g], (unsigned int)__popc(old | now));
                atomicAdd(&s_wrg[g], (unsigned int)__popc(v.full_step ? (old | now) : (run ? (nz | (now & ~old)) : 0u)));
            }
            ms[g][wd] = am;
            ns[g][wd] = an;
        }
        __syncthreads();
        if (tid < G && i0 + tid < n_active) {
            v.act_cols[i0 + tid] += s_colsg[tid];
            v.wr_cols[i0 + tid] += s_wrg[tid];
            if (adv[tid]) td_advance_slot(v, i0 + tid, (int)(t + 1));
        }
        // =============================== trace pass of the same slots (td_trace_pipe_kernel) ===============================
        const long long gb = i0;
        td_f32x4 acc[TD_WIDE_K];
#pragma unroll
        for (int k = 0; k < TD_WIDE_K; ++k) acc[k] = (td_f32x4){0.f, 0.f, 0.f, 0.f};
        auto issue = [&](int q, uint32_t &act, uint32_t &rd, td_f32x4 (&ev)[TD_WIDE_K]) {
            act = 0; rd = 0;
            if (q < m) {
#pragma unroll
                for (int k = 0; k < TD_WIDE_K; ++k) {
                    const int sh = ((TD_WIDE_CPG * k) & 31) + c;
                    bool a, fresh;
                    if (k < TD_WIDE_KL || t_w1) {
                        a = (ms[q][(TD_WIDE_CPG * k) >> 5] >> sh) & 1u;
                        fresh = FIRST || ((ns[q][(TD_WIDE_CPG * k) >> 5] >> sh) & 1u);
                    } else { a = t_in && (ms[q][TD_FLAG_WORD] & 1u); fresh = FIRST || (ns[q][TD_FLAG_WORD] & 1u); }
                    act |= (a ? 1u : 0u) << k;
                    rd |= ((a && !fresh) ? 1u : 0u) << k;
                }
            }
            const float *eg = v.e + (gb + (q < m ? q : 0)) * TD_LD + tid * 4;
            const float *dummy = v.partial + (size_t)TD_MAX_GROUPS * TD_LD;      // 16 zeroed bytes behind the partial sums: a position
                                                                                //   that is not read needs no select, it reads 0
#pragma unroll
            for (int k = 0; k < TD_WIDE_K; ++k) {
                const float *src = ((rd >> k) & 1u) ? eg + k * (TD_WIDE_THREADS * 4) : dummy;
                ev[k] = *reinterpret_cast<const td_f32x4 *>(src);
            }
        };
        auto process = [&](int q, uint32_t act, uint32_t rd, td_f32x4 (&ev)[TD_WIDE_K]) {
            const uint32_t *rw = reinterpret_cast<const uint32_t *>(&fs[q][TD_F_ROW]);
            const uint32_t r0 = rw[base], r1 = rw[base + 1], r2 = rw[base + 2], r3 = rw[base + 3];
            const uint32_t tw = level == 0 ? (r0 | r1 | r2 | r3) : level == 1 ? (r1 | r2 | r3) : level == 2 ? ((r0 & r1) | r2 | r3) : (r2 | r3);
            const float4 db = *reinterpret_cast<const float4 *>(&fs[q][TD_F_DB1 + n0]);
            const float4 dg = make_float4(db.x * ginv, db.y * ginv, db.z * ginv, db.w * ginv);   // thermometer levels: x_j is 0 or 1, and
            const float cf = cs[q];                                                             //   (db x_j) ginv = x_j ? db ginv : 0 to the bit
            float *eg = v.e + (gb + q) * TD_LD + tid * 4;
#pragma unroll
            for (int k = 0; k < TD_WIDE_K; ++k) {
                if (!((act >> k) & 1u)) continue;
                td_f32x4 x = ev[k];                                     // (zero where the position is not read: the dummy line)
                bool wr;
                if (k < TD_WIDE_KL) {
                    const int pos = pos0 + (TD_WIDE_CPG / 8) * k;
                    if (level == 3) {
                        const int cnt = (int)(((r0 >> pos) & 1u) | (((r1 >> pos) & 1u) << 1) | (((r2 >> pos) & 1u) << 2) | (((r3 >> pos) & 1u) << 3));
                        const float xj = cnt > 3 ? 0.5f * (float)(cnt - 3) : 0.0f;
                        x.x = fmaf(emul, x.x, (db.x * xj) * ginv);
                        x.y = fmaf(emul, x.y, (db.y * xj) * ginv);
                        x.z = fmaf(emul, x.z, (db.z * xj) * ginv);
                        x.w = fmaf(emul, x.w, (db.w * xj) * ginv);
                        wr = xj != 0.0f;
                    } else {
                        wr = ((tw >> pos) & 1u) != 0u;
                        x.x = fmaf(emul, x.x, wr ? dg.x : 0.0f);
                        x.y = fmaf(emul, x.y, wr ? dg.y : 0.0f);
                        x.z = fmaf(emul, x.z, wr ? dg.z : 0.0f);
                        x.w = fmaf(emul, x.w, wr ? dg.w : 0.0f);
                    }
                } else {
                    const float xj = t_w1 ? td_feature_value(rw, 192 + c) : 1.0f;
                    x.x = fmaf(emul, x.x, (fs[q][ia[0]] * (t_w1 ? xj : xfix[0])) * ginv);
                    x.y = fmaf(emul, x.y, (fs[q][ia[1]] * (t_w1 ? xj : xfix[1])) * ginv);
                    x.z = fmaf(emul, x.z, (fs[q][ia[2]] * (t_w1 ? xj : xfix[2])) * ginv);
                    x.w = fmaf(emul, x.w, (fs[q][ia[3]] * (t_w1 ? xj : xfix[3])) * ginv);
                    wr = !t_w1 || xj != 0.0f;
                }
                if (full || wr || !((rd >> k) & 1u)) td_store_wt(reinterpret_cast<td_f32x4 *>(eg + k * (TD_WIDE_THREADS * 4)), x);
                acc[k].x = fmaf(cf, x.x, acc[k].x);
                acc[k].y = fmaf(cf, x.y, acc[k].y);
                acc[k].z = fmaf(cf, x.z, acc[k].z);
                acc[k].w = fmaf(cf, x.w, acc[k].w);
            }
        };
        uint32_t actA, rdA, actB, rdB;
        td_f32x4 evA[TD_WIDE_K], evB[TD_WIDE_K];
        issue(0, actA, rdA, evA);
        for (int q0 = 0; q0 < m; q0 += 2) {
            issue(q0 + 1, actB, rdB, evB);
            process(q0, actA, rdA, evA);
            if (q0 + 1 >= m) break;
            issue(q0 + 2, actA, rdA, evA);
            process(q0 + 1, actB, rdB, evB);
        }
#pragma unroll
        for (int k = 0; k < TD_WIDE_K; ++k)
            if (k < TD_WIDE_KL || t_in)
                td_store_wt(reinterpret_cast<td_f32x4 *>(v.partial + (long long)blockIdx.x * TD_LD + (k * TD_WIDE_THREADS + tid) * 4), acc[k]);
    }
}

// block 256 = 16 float4 (64 consecutive INTERNAL positions) x 16 group lanes: a thread sums every 16th partial row of its float4
// (a 16-byte load each: a quarter of the load instructions of the one-float-per-thread form, the same bytes), the 16 lanes meet
// in LDS in a fixed order.  upd (optional) receives the summed update; apply adds it to theta.
__global__ __launch_bounds__(256) void td_reduce_kernel(TdView v, int n_groups, float *upd, int apply)
{
    __shared__ td_f32x4 red[16][16];
    const int p4 = threadIdx.x & 15, gl = threadIdx.x >> 4;
    const int q0 = blockIdx.x * 64 + p4 * 4;                  // INTERNAL position (the partial sums' order); TD_LD is a multiple of 64
    td_f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
    for (int g = gl; g < n_groups; g += 16) s += *reinterpret_cast<const td_f32x4 *>(v.partial + (long long)g * TD_LD + q0);
    red[gl][p4] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int q = blockIdx.x * 64 + threadIdx.x;
        const int f4 = threadIdx.x >> 2, e = threadIdx.x & 3;
        float u = 0.0f;
#pragma unroll
        for (int g = 0; g < 16; ++g) u += red[g][f4][e];
        if (q < TD_P) {
            const int p = td_param_of_internal(q);           // parameter order of theta / the update handed out
            if (upd) upd[p] = u;
            if (apply) {
                const float th = v.theta[p] + u;
                v.theta[p] = th;
                if (q < TD_OFF_B1) {
                    v.w1t[q] = th;                           // w1t is [j][n]: the internal order of the W1 block
                    root3_store_weight(v.wl3, q & (N_HID - 1), q >> 7, th);
                }
            }
        }
    }
}

// theta += upd (after the caller's all-reduce), or theta = upd (set); refreshes the transposed fc1.weight
__global__ void td_apply_kernel(TdView v, const float *upd, int set)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= TD_P) return;
    const float th = set ? upd[p] : v.theta[p] + upd[p];
    v.theta[p] = th;
    if (p < TD_OFF_B1) {
        const int n = p / N_IN, j = p - n * N_IN;
        v.w1t[j * N_HID + n] = th;
        root3_store_weight(v.wl3, n, j, th);
    }
}

}  // namespace bg
