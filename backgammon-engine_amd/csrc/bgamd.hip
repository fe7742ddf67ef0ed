// bgamd.hip -- kernels of the env step + the C ABI declared in include/bgamd.h (gfx950 only).
//
// Data layout in HBM (one env = n_games lanes on one device):
//   planes   uint32[8][n]   bit-plane boards, SoA: a 64-lane wave reads 256 contiguous bytes per plane
//   meta     uint32[n]      bit0 turn | bits4-6 die1 | bits8-10 die2 | bit12 finished
//   ply, episode uint32[n]  Philox counter words (game_id = lane_offset + lane + episode*lane_stride)
//   arena    uint4[2*cap]   candidate afterstates, 32 B rows, reference order inside a lane's segment
//   values   float[cap]     value-net output per row
// Greedy step: roots -> stage<PLY2> -> stage<PLY3> -> stage<LEAF> (bg_staged_kernels.h) -> eval (bg_eval.h) -> apply.
// Random step: rnd_tasks -> rnd_count -> rnd_select (bg_random_kernels.h).  emit_kernel serves the ordered enumerate API.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <queue>
#include <algorithm>
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and prototypes only: the library is resolved at run time (rccl_api below), libbgamd.so does not link it

#include "../../include/bgamd.h"
#include "bg_board.h"
#include "bg_eval.h"
#include "bg_root_resident.h"
// Kernels that lost their same-box A/B (DESIGN.md §4 / NOTEBOOK.md) compile only with -DBGAMD_EXPERIMENTAL (the library
// __graft_entry__.build_experimental() makes; the tests marked gpu_experimental run against it): the K-compacted MFMA delta kernel
// (BGAMD_MFMA_DELTA=1), the dense f16 x 2 kernel with register-resident weights (BGAMD_F16X2_RESIDENT=1), the LDS-staged / f32-MFMA root
// passes (BGAMD_ROOT_RESIDENT=0, BGAMD_ROOT_F32=1), the learner's unfused matrix-pipe forward (BGAMD_TD_FUSED=0).  In the default build
// those switches are ignored and bgamd_build_flags() says so.
#ifdef BGAMD_EXPERIMENTAL
#include "bg_eval_mfma.h"
#include "bg_eval_dense16.h"
#endif
#include "bg_learner.h"
#include "bg_schedule.h"
#include "bg_movegen.h"
#include "bg_staged.h"

using namespace bg;

namespace {

// Env counters, one 128-byte line each (BG_STAT_STRIDE=1: packed, as up to round 2).  Atomics queue up at the memory side per LINE: the
// boundary launch's three statistics from 256 workgroups were 768 atomics on one line -- 4 us of its 18 (same-box A/B, round 3:
// boundary 0.0182 -> 0.0139 ms).  Further copies of the statistics per group of workgroups (8, 16) measured nothing on top.
#ifndef BG_STAT_STRIDE
#define BG_STAT_STRIDE 16
#endif
enum { C_ARENA_TOP = 0, C_STEPS = BG_STAT_STRIDE, C_FINISHED = 2 * BG_STAT_STRIDE, C_P1WINS = 3 * BG_STAT_STRIDE, C_CAND_RAW = 4 * BG_STAT_STRIDE,
       C_ROWS_EVAL = 5 * BG_STAT_STRIDE, C_ERR = 6 * BG_STAT_STRIDE, C_FNODES = 7 * BG_STAT_STRIDE, C_DNODES = 8 * BG_STAT_STRIDE,
       C_KSTEPS = 9 * BG_STAT_STRIDE, C_COUNT = 10 * BG_STAT_STRIDE };
enum { ERRF_ARENA = 1, ERRF_STATE = 2, ERRF_DELTA = 4 };
constexpr uint32_t META_FINISHED = 1u << 12;

struct EnvView {
    long long n;
    unsigned long long seed, lane_offset, lane_stride;
    long long cap;
    uint32_t *planes, *meta, *ply, *episode, *flags;
    uint32_t *cand_off, *cand_cnt;
    int32_t *chosen;
    uint32_t *chosen_seq;
    float *chosen_val;
    uint4 *rows;
    uint32_t *seqs;
    float *values;
    unsigned long long *counters;
    uint4 *traj;               // optional [max_plies][n][2]: pre-move row of every turn (train.py:105-106)
    long long traj_plies;
    // ring log of continuous self-play (bgamd_env_set_trajectory_ring): traj is [traj_ring][n][2] indexed by the ENV STEP (mod traj_ring),
    // not by the lane's ply -- a lane's column holds game after game -- and endrec[slot][lane] says whether the game whose turn was
    // logged in that slot ended with it: 0, or logged turns of the game | winner << 15.  log_slot / end_slot: the slots the roots /
    // the apply of THIS launch write (set by the host per launch).
    unsigned short *endrec;
    long long traj_ring, log_slot, end_slot;
};

__device__ __forceinline__ uint32_t meta_pack(int turn, int d1, int d2, bool fin)
{
    return (uint32_t)turn | ((uint32_t)d1 << 4) | ((uint32_t)d2 << 8) | (fin ? META_FINISHED : 0u);
}

__device__ __forceinline__ void load_planes(const EnvView &e, long long g, uint32_t (&p)[8])
{
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = e.planes[(long long)k * e.n + g];
}
__device__ __forceinline__ void store_planes(const EnvView &e, long long g, const uint32_t (&p)[8])
{
#pragma unroll
    for (int k = 0; k < 8; ++k) e.planes[(long long)k * e.n + g] = p[k];
}

__device__ __forceinline__ unsigned long long wave_sum_u32(uint32_t v)
{
    unsigned long long s = v;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)s, m, 64), hi = __shfl_xor((uint32_t)(s >> 32), m, 64);
        s += ((unsigned long long)hi << 32) | lo;
    }
    return s;
}

// Terminal check (game.cpp:388-407), auto-reset / turn flip (train.py:113-120), write-back.
// fwd (optional, [3]): the lane's meta / ply / episode as they stand after this call (p is updated in place) -- for a
// caller that goes on with the lane's next turn in the same kernel instead of reading the state back
__device__ __forceinline__ void finish_turn(const EnvView &e, long long g, uint32_t (&p)[8], int turn, int d1, int d2,
                                            uint32_t ply, uint32_t epi, int flags, bool live, uint32_t *fwd = nullptr)
{
    uint32_t oflags = 0;
    bool fin = false;
    const int oc = live ? over_code(p) : 0;
    if (e.endrec && g < e.n)                               // ring log: did the game whose turn sits in this step's slot end with it?
        e.endrec[e.end_slot * e.n + g] = oc ? (unsigned short)((ply + 1u > 0x7FFFu ? 0x7FFFu : ply + 1u) | ((uint32_t)(oc - 1) << 15)) : (unsigned short)0;
    if (oc) {
        oflags = 1u | ((uint32_t)(oc - 1) << 1);
        if (flags & BGAMD_AUTO_RESET) {
            epi += 1; ply = 0;
            constexpr StartPlanes sp = start_planes();
#pragma unroll
            for (int k = 0; k < 8; ++k) p[k] = sp.p[k];
            const unsigned long long gid = e.lane_offset + (unsigned long long)g + (unsigned long long)epi * e.lane_stride;
            turn = opening_turn(e.seed, gid);
        } else if (!(flags & BGAMD_NO_FLIP)) {
            fin = true; oflags |= 4u;
        }
    } else if (live && !(flags & BGAMD_NO_FLIP)) {
        turn ^= 1; ply += 1;
    }
    if (live) {
        store_planes(e, g, p);
        e.meta[g] = meta_pack(turn, d1, d2, fin);
        e.ply[g] = ply; e.episode[g] = epi; e.flags[g] = oflags;
        if (fwd) { fwd[0] = meta_pack(turn, d1, d2, fin); fwd[1] = ply; fwd[2] = epi; }
    }
    // counters: wave sums -> one LDS word each -> ONE global atomic per block and counter (same-address global
    // atomics retire at ~10 ns each: a per-wave atomic from 1 024 waves costs more than the kernel itself)
    __shared__ unsigned int s_stat[3];
    if (threadIdx.x < 3) s_stat[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long nsteps = wave_sum_u32(live ? 1u : 0u);
    const unsigned long long nfin = wave_sum_u32(oc ? 1u : 0u);
    const unsigned long long nw1 = wave_sum_u32(oc == 1 ? 1u : 0u);
    if ((threadIdx.x & 63) == 0) {
        if (nsteps) atomicAdd(&s_stat[0], (unsigned int)nsteps);
        if (nfin) atomicAdd(&s_stat[1], (unsigned int)nfin);
        if (nw1) atomicAdd(&s_stat[2], (unsigned int)nw1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_stat[0]) atomicAdd(&e.counters[C_STEPS], (unsigned long long)s_stat[0]);
        if (s_stat[1]) atomicAdd(&e.counters[C_FINISHED], (unsigned long long)s_stat[1]);
        if (s_stat[2]) atomicAdd(&e.counters[C_P1WINS], (unsigned long long)s_stat[2]);
    }
}

struct LaneCtx {
    uint32_t p[8];
    uint32_t meta, ply, epi;
    int turn, d1, d2;
    U4 x;
    bool live;
};

// lane_derive: everything lane_begin does after its loads (c.p, c.meta, c.ply, c.epi are in place)
__device__ __forceinline__ void lane_derive(const EnvView &e, long long g, int flags, LaneCtx &c)
{
    c.live = g < e.n;
    const long long gg = c.live ? g : 0;
    if (c.meta & META_FINISHED) c.live = false;
    c.turn = c.meta & 1;
    // head-to-head play (train.py:262-277): only the lanes whose side is to move take part in this call
    if (((flags & BGAMD_ONLY_P1) && c.turn != 0) || ((flags & BGAMD_ONLY_P2) && c.turn != 1)) c.live = false;
    const unsigned long long gid = e.lane_offset + (unsigned long long)gg + (unsigned long long)c.epi * e.lane_stride;
    c.x = philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), c.ply, STREAM_TURN, (uint32_t)e.seed, (uint32_t)(e.seed >> 32));
    if (flags & BGAMD_ROLL) { c.d1 = die_from_u32(c.x.x); c.d2 = die_from_u32(c.x.y); }
    else { c.d1 = (c.meta >> 4) & 7; c.d2 = (c.meta >> 8) & 7; }
    if (c.d1 < 1 || c.d1 > 6 || c.d2 < 1 || c.d2 > 6) { c.d1 = 1; c.d2 = 1; }   // Game::last_dice default {1,1}
}

__device__ __forceinline__ void lane_begin(const EnvView &e, long long g, int flags, LaneCtx &c)
{
    const long long gg = g < e.n ? g : 0;
    load_planes(e, gg, c.p);
    c.meta = e.meta[gg]; c.ply = e.ply[gg]; c.epi = e.episode[gg];
    lane_derive(e, g, flags, c);
}

__device__ __forceinline__ void split_sides(const uint32_t (&p)[8], int turn, Side &own, Side &opp)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        own.b[k] = turn ? p[4 + k] : p[k];
        opp.b[k] = turn ? p[k] : p[4 + k];
    }
}
__device__ __forceinline__ void join_sides(const Side &own, const Side &opp, int turn, uint32_t (&p)[8])
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        p[k] = turn ? opp.b[k] : own.b[k];
        p[4 + k] = turn ? own.b[k] : opp.b[k];
    }
}

// ---- random-policy step: everything in one kernel ----------------------------------------------
__global__ __launch_bounds__(64) void step_random_kernel(EnvView e, int flags, const uint32_t *__restrict__ choice)
{
    const long long g = (long long)blockIdx.x * 64 + threadIdx.x;
    LaneCtx c;
    lane_begin(e, g, flags, c);
    Side own, opp;
    split_sides(c.p, c.turn, own, opp);
    uint32_t C = 0;
    if (c.live) {
        CountVisitor cv;
        walk_sequences(own, opp, c.turn, c.d1, c.d2, cv);
        C = cv.n;
    }
    int32_t chosen = -1;
    uint32_t cseq = 0;
    if (C > 0) {
        const uint32_t u = choice ? choice[g] : c.x.z;
        const uint32_t k = (uint32_t)(((unsigned long long)u * C) >> 32);
        SelectVisitor sv(k);
        walk_sequences(own, opp, c.turn, c.d1, c.d2, sv);
        join_sides(sv.own, sv.opp, c.turn, c.p);
        chosen = (int32_t)k; cseq = sv.seq | (c.turn ? (1u << 29) : 0u);   // bit 29: mover moves down (unpack_seq)
    }
    if (c.live) { e.chosen[g] = chosen; e.chosen_seq[g] = cseq; e.cand_cnt[g] = C; e.chosen_val[g] = 0.0f; }
    const unsigned long long tot = wave_sum_u32(C);
    if (threadIdx.x == 0 && tot) atomicAdd(&e.counters[C_CAND_RAW], tot);
    finish_turn(e, g, c.p, c.turn, c.d1, c.d2, c.ply, c.epi, flags, c.live);
}

// ---- emit: count pass, wave-level bump allocation, emit pass ---------------------------------
__global__ __launch_bounds__(64) void emit_kernel(EnvView e, int flags, int with_seq, const int32_t *__restrict__ ov_player,
                                                  const int32_t *__restrict__ ov_dice)
{
    const long long g = (long long)blockIdx.x * 64 + threadIdx.x;
    const int lane = threadIdx.x;
    LaneCtx c;
    lane_begin(e, g, flags, c);
    if (g < e.n) {       // explicit (player, d1, d2) of legalTurnSequences / evaluateTurnSequences
        if (ov_player) c.turn = ov_player[g] == 1 ? 1 : 0;
        if (ov_dice) {
            const int a = ov_dice[2 * g], b = ov_dice[2 * g + 1];
            if (a >= 1 && a <= 6 && b >= 1 && b <= 6) { c.d1 = a; c.d2 = b; } else c.live = false;
        }
    }
    Side own, opp;
    split_sides(c.p, c.turn, own, opp);
    uint32_t C = 0;
    if (c.live) {
        CountVisitor cv;
        walk_sequences(own, opp, c.turn, c.d1, c.d2, cv);
        C = cv.n;
    }
    // inclusive scan over the wave
    uint32_t incl = C;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const uint32_t total = __shfl(incl, 63, 64);
    unsigned long long base = 0;
    if (lane == 0 && total) {
        base = atomicAdd(&e.counters[C_ARENA_TOP], (unsigned long long)total);
        atomicAdd(&e.counters[C_CAND_RAW], (unsigned long long)total);
    }
    base = ((unsigned long long)__shfl((uint32_t)(base >> 32), 0, 64) << 32) | __shfl((uint32_t)base, 0, 64);
    const bool overflow = base + total > (unsigned long long)e.cap;
    if (overflow) {
        if (lane == 0) atomicOr(&e.counters[C_ERR], (unsigned long long)ERRF_ARENA);
        C = 0;
    }
    const unsigned long long my_off = overflow ? 0ull : base + incl - C;
    if (C > 0) {
        EmitVisitor ev(e.rows + 2 * my_off, with_seq ? e.seqs + my_off : nullptr, c.turn);
        walk_sequences(own, opp, c.turn, c.d1, c.d2, ev);
    }
    if (g < e.n) {
        e.cand_off[g] = (uint32_t)my_off;
        e.cand_cnt[g] = C;
        if ((flags & BGAMD_ROLL) && c.live) e.meta[g] = meta_pack(c.turn, c.d1, c.d2, false);
    }
}

// ---- small state kernels ---------------------------------------------------------------------------
// mask == nullptr: every lane restarts at episode 0.  Otherwise only the lanes with mask != 0 restart, as the NEXT
// episode of that lane (what the auto-reset of a finished game does)
__global__ void reset_kernel(EnvView e, const int32_t *__restrict__ mask, uint32_t episode)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= e.n) return;
    if (mask && mask[g] == 0) return;
    constexpr StartPlanes sp = start_planes();
    uint32_t p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) p[k] = sp.p[k];
    store_planes(e, g, p);
    const uint32_t epi = mask ? e.episode[g] + 1u : episode;
    const unsigned long long gid = e.lane_offset + (unsigned long long)g + (unsigned long long)epi * e.lane_stride;
    e.meta[g] = meta_pack(opening_turn(e.seed, gid), 1, 1, false);
    e.ply[g] = 0; e.episode[g] = epi; e.flags[g] = 0;
    e.cand_off[g] = 0; e.cand_cnt[g] = 0; e.chosen[g] = -1; e.chosen_seq[g] = 0; e.chosen_val[g] = 0.0f;
}

__global__ void set_states_kernel(EnvView e, const int32_t *__restrict__ st, const int32_t *__restrict__ turn)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= e.n) return;
    if (st) {
        uint32_t p[8];
        int bad = 0;
        int32_t s[28];
#pragma unroll
        for (int i = 0; i < 28; ++i) s[i] = st[g * 28 + i];
        planes_from_state28(s, p, &bad);
        if (bad) atomicOr(&e.counters[C_ERR], (unsigned long long)ERRF_STATE);
        store_planes(e, g, p);
    }
    const uint32_t m = e.meta[g];
    const int t = turn ? (turn[g] & 1) : (int)(m & 1);
    e.meta[g] = (m & ~(1u | META_FINISHED)) | (uint32_t)t;
}

__global__ void get_states_kernel(EnvView e, int32_t *__restrict__ st, int32_t *__restrict__ turn)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= e.n) return;
    uint32_t p[8];
    load_planes(e, g, p);
    if (st) {
        int32_t s[28];
        state28_from_planes(p, s);
#pragma unroll
        for (int i = 0; i < 28; ++i) st[g * 28 + i] = s[i];
    }
    if (turn) turn[g] = (int32_t)(e.meta[g] & 1);
}

__global__ void get_flags_kernel(EnvView e, int32_t *__restrict__ out)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= e.n) return;
    uint32_t p[8];
    load_planes(e, g, p);
    const int oc = over_code(p);          // is_game_over on the CURRENT board (game.cpp:388-407)
    out[g] = (oc ? (1 | ((oc - 1) << 1)) : 0) | ((e.meta[g] & META_FINISHED) ? 4 : 0) | (int32_t)((e.flags[g] & 3u) << 4);
}

__global__ void snapshot_kernel(EnvView e, int32_t *__restrict__ out)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= e.n) return;
    uint32_t p[8];
    load_planes(e, g, p);
    int32_t s[28];
    state28_from_planes(p, s);
    int32_t *o = out + g * 32;
#pragma unroll
    for (int i = 0; i < 28; ++i) o[i] = s[i];
    const uint32_t m = e.meta[g];
    const int oc = over_code(p);
    o[28] = (int32_t)(m & 1); o[29] = (int32_t)((m >> 4) & 7); o[30] = (int32_t)((m >> 8) & 7);
    o[31] = (oc ? (1 | ((oc - 1) << 1)) : 0) | ((m & META_FINISHED) ? 4 : 0) | (int32_t)((e.flags[g] & 3u) << 4);
}

__global__ void dice_kernel(EnvView e, const int32_t *__restrict__ in, int32_t *__restrict__ out, int roll)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= e.n) return;
    uint32_t m = e.meta[g];
    if (in || roll) {
        int d1, d2;
        if (roll) {
            const unsigned long long gid = e.lane_offset + (unsigned long long)g + (unsigned long long)e.episode[g] * e.lane_stride;
            const U4 x = philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), e.ply[g], STREAM_TURN, (uint32_t)e.seed, (uint32_t)(e.seed >> 32));
            d1 = die_from_u32(x.x); d2 = die_from_u32(x.y);
            if (roll == 2) e.ply[g] += 1;      // scalar Game::rollDice: every call draws fresh dice
        } else { d1 = in[2 * g] & 7; d2 = in[2 * g + 1] & 7; }
        m = (m & ~0x770u) | ((uint32_t)d1 << 4) | ((uint32_t)d2 << 8);
        e.meta[g] = m;
    }
    if (out) { out[2 * g] = (m >> 4) & 7; out[2 * g + 1] = (m >> 8) & 7; }
}

__global__ void cand_info_kernel(EnvView e, int64_t *__restrict__ offs, int32_t *__restrict__ cnts)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= e.n) return;
    if (offs) offs[g] = (int64_t)e.cand_off[g];
    if (cnts) cnts[g] = (int32_t)e.cand_cnt[g];
}

__device__ __forceinline__ void unpack_seq(uint32_t q, int8_t *out8, int32_t *len_out)
{
    const int len = (q >> 20) & 7, dA = (q >> 23) & 7, dB = (q >> 26) & 7;
    // direction is recovered from the move itself: the mover's sign is not stored, so the
    // caller passes it through bit 29 (1 = PLAYER2 moves down)
    const int down = (q >> 29) & 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int o = -1, d = -1;
        if (k < len) {
            o = (q >> (5 * k)) & 31;
            const int die = (k & 1) ? dB : dA;
            d = down ? o - die : o + die;
            d = d < 0 ? 0 : (d > 25 ? 25 : d);
        }
        out8[2 * k] = (int8_t)o; out8[2 * k + 1] = (int8_t)d;
    }
    if (len_out) *len_out = len;
}

__global__ void rows_read_kernel(const uint4 *__restrict__ rows, const uint32_t *__restrict__ seqs, long long first,
                                 long long n, int32_t *__restrict__ st, int8_t *__restrict__ seq, int32_t *__restrict__ len)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long r = first + i;
    const uint4 u0 = rows[2 * r], u1 = rows[2 * r + 1];
    const uint32_t p[8] = {u0.x & ~TURN_BIT, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
    if (st) {
        int32_t s[28];
        state28_from_planes(p, s);
#pragma unroll
        for (int k = 0; k < 28; ++k) st[i * 28 + k] = s[k];
    }
    if (seq) {
        const uint32_t q = seqs[r] | ((u0.x & TURN_BIT) ? (1u << 29) : 0u);
        unpack_seq(q, seq + i * 8, len ? len + i : nullptr);
    }
}

// (b_base > 0: four arenas -- the logical rows are the rows of arena 0, then those of arena 1, ...; tops = the step's counters)
__global__ void rows_values_kernel(const uint4 *__restrict__ rows, const float *__restrict__ values, long long first, long long n,
                                   int32_t *__restrict__ st, float *__restrict__ val, const unsigned long long *__restrict__ tops, long long b_base)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long long r = first + i;
#ifdef BG_EVAL_WGCLOCK                                         // (diagnostic build, tools/eval_wg_clock.py: physical rows)
    b_base = 0;
#endif
    if (b_base > 0) {
        long long k = 0, rest = r;
        for (; k < N_ARENAS; ++k) {
            long long c = (long long)tops[arena_counter((int)k)];
            if (c > b_base) c = b_base;
            if (rest < c) break;
            rest -= c;
        }
        if (k == N_ARENAS) {                                    // past the rows the step produced (every arena clamped, the last one too):
            if (st)                                             // zeros, never a read beyond the arenas
                for (int j = 0; j < 28; ++j) st[i * 28 + j] = 0;
            if (val) val[i] = 0.0f;
            return;
        }
        r = k * b_base + rest;
    }
    if (st) {
        const uint4 u0 = rows[2 * r], u1 = rows[2 * r + 1];
        const uint32_t p[8] = {u0.x & ~TURN_BIT, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
        int32_t s[28];
        state28_from_planes(p, s);
#pragma unroll
        for (int k = 0; k < 28; ++k) st[i * 28 + k] = s[k];
    }
    if (val) val[i] = values[r];
}

__global__ void last_choice_kernel(EnvView e, int32_t *chosen, int32_t *count, int8_t *seq, int32_t *len, float *val)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= e.n) return;
    if (chosen) chosen[g] = e.chosen[g];
    if (count) count[g] = (int32_t)e.cand_cnt[g];
    if (val) val[g] = e.chosen_val[g];
    if (seq) {
        // the mover of the last step: turn was flipped unless the game ended / NO_FLIP; the packed
        // sequence carries the direction in bit 29, set by the step kernels' callers below
        unpack_seq(e.chosen_seq[g], seq + g * 8, len ? len + g : nullptr);
    }
}

__global__ void pack_rows_kernel(const int32_t *__restrict__ st, const int32_t *__restrict__ turn, long long n,
                                 uint4 *__restrict__ rows, unsigned long long *err)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t s[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) s[k] = st[i * 28 + k];
    uint32_t p[8];
    int bad = 0;
    planes_from_state28(s, p, &bad);
    if (bad && err) atomicOr(err, (unsigned long long)ERRF_STATE);
    const int t = turn ? (turn[i] & 1) : 0;
    rows[2 * i] = make_uint4(p[0] | (t ? TURN_BIT : 0u), p[1], p[2], p[3]);
    rows[2 * i + 1] = make_uint4(p[4], p[5], p[6], p[7]);
}

// rows of the stateless incremental operator: afterstate i belongs to root root_index[i] and carries that root's turn bit
__global__ void pack_child_rows_kernel(const int32_t *__restrict__ st, const int32_t *__restrict__ root_index, long long n,
                                       long long n_roots, const uint4 *__restrict__ root_rows, uint4 *__restrict__ rows,
                                       uint2 *__restrict__ info, unsigned long long *err)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t s[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) s[k] = st[i * 28 + k];
    uint32_t p[8];
    int bad = 0;
    planes_from_state28(s, p, &bad);
    long long r = root_index[i];
    if (r < 0 || r >= n_roots) { bad = 1; r = 0; }
    if (bad && err) atomicOr(err, (unsigned long long)ERRF_STATE);
    const uint32_t tb = root_rows[2 * r].x & TURN_BIT;
    rows[2 * i] = make_uint4(p[0] | tb, p[1], p[2], p[3]);
    rows[2 * i + 1] = make_uint4(p[4], p[5], p[6], p[7]);
    info[i] = make_uint2((uint32_t)r, (uint32_t)(i & 0x00FFFFFF) | (tb ? 0x80000000u : 0u));
}

__global__ void encode_states_kernel(const int32_t *__restrict__ st, const int32_t *__restrict__ turn, long long n,
                                     float *__restrict__ out)
{
    const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    int32_t s[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) s[k] = st[row * 28 + k];
    uint32_t p[8];
    planes_from_state28(s, p, nullptr);
    const Side sd[2] = {{{p[0], p[1], p[2], p[3]}}, {{p[4], p[5], p[6], p[7]}}};
    float *x = out + row * N_IN;
    for (int i = 0; i < 24; ++i) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = count_at(sd[q], i + 1);
            x[8 * i + 4 * q + 0] = c >= 1 ? 1.0f : 0.0f;
            x[8 * i + 4 * q + 1] = c >= 2 ? 1.0f : 0.0f;
            x[8 * i + 4 * q + 2] = c >= 3 ? 1.0f : 0.0f;
            x[8 * i + 4 * q + 3] = c >= 4 ? 0.5f * (float)(c - 3) : 0.0f;
        }
    }
    const int t = turn ? (turn[row] & 1) : 0;
    x[192] = t == 0 ? 1.0f : 0.0f;
    x[193] = t == 0 ? 0.0f : 1.0f;
    x[194] = 0.5f * (float)count_at(sd[0], 0);
    x[195] = 0.5f * (float)count_at(sd[1], 25);
    x[196] = (float)count_at(sd[0], 25) / 15.0f;
    x[197] = (float)count_at(sd[1], 0) / 15.0f;
}

// Game::tryMove with the reference's check order (game.cpp:583-662); err codes 1..7
__global__ void try_move_kernel(EnvView e, const int32_t *__restrict__ player, const int32_t *__restrict__ dice,
                                const int32_t *__restrict__ origin, const int32_t *__restrict__ dest, int32_t *__restrict__ err)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= e.n) return;
    uint32_t p[8];
    load_planes(e, g, p);
    const int pl = player[g] == 1 ? 1 : 0, d = dice[g], o = origin[g], t = dest[g];
    Side own, opp;
    split_sides(p, pl, own, opp);
    const uint32_t barbit = pl ? (1u << 25) : 1u;
    const uint32_t own_any = any_of(own);
    int code = 0;
    // isValidOrigin (game.cpp:416-457)
    bool vo;
    if (own_any & barbit) vo = (o == (pl ? 25 : 0));
    else vo = (o >= 1 && o <= 24) && ((own_any >> o) & 1u);
    if (!vo) code = 1;
    else if (o < 0 || o > 25) code = 2;
    else if (t < 0 || t > 25) code = 3;
    else {
        const int diff = o - t;
        if (t != 0 && t != 25) {
            const int adiff = diff < 0 ? -diff : diff;
            if ((pl ? -diff : diff) > 0) code = 4;            // diff * (-multi) < 0
            else if (d != adiff) code = 5;
            else if ((ge2_of(opp) >> t) & 1u) code = 6;       // t in 1..24 here
            else {
                dec_at(own, 1u << o);                          // bar or point: same plane arithmetic
                const uint32_t md = 1u << t;
                const uint32_t hm = md & opp.b[0] & ~ge2_of(opp);
                opp.b[0] ^= hm;
                inc_at(opp, hm ? (pl ? 1u : (1u << 25)) : 0u);
                inc_at(own, md);
            }
        } else {
            // bear-off branch (game.cpp:636-648): only "origin on the board" is re-checked;
            // the checker leaves the origin and the MOVER's freed counter grows, whatever t is
            if (o == 0 || o == 25) code = 7;
            else { dec_at(own, 1u << o); inc_at(own, pl ? 1u : (1u << 25)); }
        }
    }
    if (code == 0) {
        join_sides(own, opp, pl, p);
        store_planes(e, g, p);
    }
    err[g] = code;
}

__global__ void legal_moves_kernel(EnvView e, const int32_t *__restrict__ player, const int32_t *__restrict__ die,
                                   int32_t *__restrict__ n_out, int8_t *__restrict__ pairs)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= e.n) return;
    uint32_t p[8];
    load_planes(e, g, p);
    const int pl = player[g] == 1 ? 1 : 0, d = die[g];
    Side own, opp;
    split_sides(p, pl, own, opp);
    uint32_t m = (d >= 1 && d <= 6) ? legal_origins(own, opp, pl, d) : 0u;
    int n = 0;
    while (m) {
        const int o = __ffs(m) - 1; m &= m - 1;
        int t = pl ? o - d : o + d;
        t = t < 0 ? 0 : (t > 25 ? 25 : t);
        pairs[(g * 26 + n) * 2] = (int8_t)o; pairs[(g * 26 + n) * 2 + 1] = (int8_t)t;
        ++n;
    }
    n_out[g] = n;
}

#include "bg_staged_kernels.h"
#include "bg_random_kernels.h"

}  // namespace

// ================================================================================================
//                                           C ABI
// ================================================================================================
static thread_local std::string g_hip_err;

#define HIPCHK(call)                                                                   \
    do {                                                                               \
        hipError_t _e = (call);                                                        \
        if (_e != hipSuccess) {                                                        \
            g_hip_err = std::string(#call) + ": " + hipGetErrorString(_e);             \
            return BGAMD_E_HIP;                                                        \
        }                                                                              \
    } while (0)

// every entry point that takes an env (or a learner) first makes its device current: one process may drive envs on
// several devices, and the caller's current device need not be the env's
#define ENV_GUARD(env)                                   \
    do {                                                 \
        if (!(env)) return BGAMD_E_INVALID;              \
        HIPCHK(hipSetDevice((env)->device));             \
    } while (0)

struct bgamd_env {
    int device = 0;
    EnvView v{};
    StagedView sv{};
    RandomView rv{};
    float *d_w[2] = {nullptr, nullptr};    // raw weights 25601, two slots (head-to-head: one per side)
    float4 *d_wl[2] = {nullptr, nullptr};  // fp32 MFMA layout [99][64]
    float4 *d_wt[2] = {nullptr, nullptr};  // W1^T [198][132] for the incremental evaluator
    uint4 *d_wm[2] = {nullptr, nullptr};   // W1^T as f16 hi | lo dwords [198][4][32] for the MFMA delta kernel (bg_eval_mfma.h)
    bool wm_ok[2] = {false, false};        // the slot's table fits f16 (else the VALU delta kernel evaluates that slot)
    bool mfma_delta = false;               // BGAMD_MFMA_DELTA=1: eval_rows_mdelta_kernel instead of eval_rows_delta_kernel
    uint4 *d_wl3[2] = {nullptr, nullptr};  // bf16 hi | mid | lo split, bf16 MFMA layout x 3 (root term, rounds 1-4; -DBG_ROOT_F16X2=0)
    uint4 *d_wr2[2] = {nullptr, nullptr};  // f16 hi | lo split in the same layout (the root pass, bg_root_resident.h)
    uint4 *d_wl16[2] = {nullptr, nullptr}; // bf16 MFMA layout [13][4][64] x 8 bf16
    uint4 *d_wlx2[2] = {nullptr, nullptr}; // f16 hi | lo split, same layout twice
    uint4 *d_wd16[2] = {nullptr, nullptr}; // the same split with -log2(e) and b1 folded in: the register-resident dense kernel (bg_eval_dense16.h)
    bool d16 = false;                      // BGAMD_F16X2_RESIDENT=1: eval_rows_d16_kernel (W1 resident in registers) for BGAMD_F16X2 -- measured 6 % slower than round 1's eval_rows_f16x2_kernel
    uint2 *d_lut = nullptr;                // count -> 4 bf16 features
    uint2 *d_lut16 = nullptr;              // count -> 4 f16 features
    bool has_weights[2] = {false, false};
    int n_cu = 256;
    hipStream_t side = nullptr;            // second stream: the root pass of the value net runs beside the doubles plies
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;   //   (fork after roots_kernel, join before the incremental kernel)

    unsigned long long *tops_base = nullptr;   // [2][T_COUNT]; sv.tops points at the set of the last step
    // ring log of continuous self-play (bgamd_env_set_trajectory_ring): greedy steps log by env step; traj_step counts the steps logged
    uint4 *ring_rows = nullptr;
    unsigned short *ring_end = nullptr;
    long long ring_steps = 0;
    long long traj_step = 0;
    int32_t *d_scalar = nullptr;           // device staging of the scalar (host-argument) surface: args at [0..63], results behind
    void *d_tmp = nullptr;                 // ... and of its enumerate call (states | seq | len), grown on demand
    size_t tmp_bytes = 0;
    int choice[4] = {0, 0, 0, 0};           // kernels of the last greedy step (bgamd_env_kernel_choice)
    bool list_shards = true;               // the roots' node lists in LIST_SHARDS parts (BGAMD_LIST_SHARDS=0: one list each)
    bool split_arena = true;               // the greedy step (incremental value net) keeps hit-free rows and the others in two arenas (BGAMD_SPLIT_ARENA=0: one)
    bool expand_merged = true;             // doubles plies + leaf stage in one launch (expand_all_kernel); BGAMD_EXPAND_MERGED=0: two launches
    int expand_parts = 3;                  // timing experiments (BGAMD_EXPAND_PARTS): 1 = only the doubles turns are expanded, 2 = only the others
    int expand_dbl_npb = 128;              // ply-1 nodes a doubles workgroup takes per iteration (BGAMD_EXPAND_DBL_NPB: 64 .. 512)
    int expand_dbl_pct = 0;                // share of that launch's workgroups that takes the doubles turns (BGAMD_EXPAND_DBL_PCT: 5 .. 95; 0 = by env
                                           // size: GreedyRun::init), the others take the non-doubles leaf stage
    bool root_in_boundary = true;          // inside a run the root pass of step t + 1 runs in the boundary launch of step t (bg_root_resident.h);
                                           //   BGAMD_ROOT_IN_BOUNDARY=0: a launch of its own every step, as up to round 3
    bool overlap = true;                   // BGAMD_NO_OVERLAP=1: everything on the caller's stream; BGAMD_OVERLAP=1: second stream for small envs too
    bool root_f32_mfma = false;            // root term by the f32 MFMA chain instead of the bf16 x 3 split (BGAMD_ROOT_F32=1)
    bool root_resident = true;             // the bf16 x 3 root pass with W1 resident in registers (same bits as the LDS-staged kernel, BGAMD_ROOT_RESIDENT=0).
                                           //   The leaf stage starts less obstructed beside it: with the counters packed (rounds 2-3a) the value net paid that
                                           //   back (step 0.1501 vs 0.1504 ms); with every counter on its own line it does not (leaf stage 0.0302 -> 0.0260,
                                           //   doubles plies 0.0180 -> 0.0195, step 0.1458 -> 0.1445 ms, same box, three interleaved runs)
    // kernel timing
    unsigned timing = 0;                   // bit k: bracket kernel group k with HIP events
    unsigned timing_stride = 1;            // ... on every timing_stride-th launch of the group (an event pair costs ~4 us)
    unsigned timing_seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<hipEvent_t> ev;            // pairs
    std::vector<int> ev_kind;
    size_t ev_used = 0;
    double t_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t t_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

namespace {
inline dim3 grid1(long long n, int bs) { return dim3((unsigned)((n + bs - 1) / bs)); }

int flush_events(bgamd_env *env)
{
    for (size_t i = 0; i < env->ev_used; ++i) {
        HIPCHK(hipEventSynchronize(env->ev[2 * i + 1]));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, env->ev[2 * i], env->ev[2 * i + 1]));
        env->t_ms[env->ev_kind[i]] += ms;
        env->t_n[env->ev_kind[i]] += 1;
    }
    env->ev_used = 0;
    return BGAMD_OK;
}

struct KTimer {
    bgamd_env *env; hipStream_t s; size_t slot; bool on;
    KTimer(bgamd_env *e, hipStream_t st, int kind) : env(e), s(st), slot(0), on((e->timing >> kind) & 1u)
    {
        if (on && e->timing_stride > 1) on = (e->timing_seen[kind]++ % e->timing_stride) == 0;
        if (!on) return;
        if (env->ev_used * 2 + 2 > env->ev.size()) {
            if (env->ev.size() >= 2 * 8192) { flush_events(env); }
            else {
                hipEvent_t x[2] = {nullptr, nullptr};
                if (hipEventCreate(&x[0]) != hipSuccess || hipEventCreate(&x[1]) != hipSuccess) {
                    if (x[0]) hipEventDestroy(x[0]);
                    on = false;                          // no event pair: this launch goes untimed
                    return;
                }
                env->ev.push_back(x[0]); env->ev.push_back(x[1]);
                env->ev_kind.push_back(0);
            }
        }
        if (env->ev_used * 2 + 2 > env->ev.size()) { on = false; return; }
        slot = env->ev_used;
        if (hipEventRecord(env->ev[2 * slot], s) != hipSuccess) { on = false; return; }
        env->ev_used++;
        env->ev_kind[slot] = kind;
    }
    // a pair whose second record failed would poison flush_events: record it on the null stream as a last resort
    ~KTimer() { if (on && hipEventRecord(env->ev[2 * slot + 1], s) != hipSuccess) hipEventRecord(env->ev[2 * slot + 1], nullptr); }
};
}  // namespace

extern "C" {

int bgamd_version(void) { return 200; }

#ifndef BGAMD_SRC_HASH
#define BGAMD_SRC_HASH "unknown"
#endif
// the marker lets the build script read the digest out of the file without loading it
const char *bgamd_source_hash(void) { static const char tag[] = "BGAMD_SRC_HASH=" BGAMD_SRC_HASH; return tag + 15; }

const char *bgamd_error_string(int code)
{
    switch (code) {
    case BGAMD_OK: return "ok";
    case BGAMD_E_INVALID: return "invalid argument";
    case BGAMD_E_HIP: return "HIP runtime error";
    case BGAMD_E_NODEVICE: return "no usable gfx950 device";
    case BGAMD_E_ARENA: return "candidate arena overflow";
    case BGAMD_E_STATE: return "state with |count| > 15";
    case BGAMD_E_NOWEIGHTS: return "weights not loaded";
    case BGAMD_E_DELTA: return "incremental value net: a row differs from its root in more features than a legal turn changes";
    case BGAMD_E_WEIGHTS: return "weights refused: a value is not finite or an fc1 weight is outside the f16 hi + lo range (|w| >= 65504)";
    default: return "unknown error";
    }
}
const char *bgamd_last_hip_error(void) { return g_hip_err.c_str(); }

int bgamd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int env_allocate(bgamd_env *env, int64_t n_games, uint64_t seed, uint64_t lane_offset, uint64_t lane_stride,
                        int64_t arena_rows);

int bgamd_env_create(bgamd_env **out, int64_t n_games, int device, uint64_t seed, uint64_t lane_offset,
                     uint64_t lane_stride, int64_t arena_rows)
{
    if (!out || n_games <= 0 || n_games > (1ll << 30)) return BGAMD_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return BGAMD_E_NODEVICE;
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return BGAMD_E_NODEVICE;
    bgamd_env *env = new bgamd_env();
    env->device = device;
    env->n_cu = prop.multiProcessorCount;
    int rc = env_allocate(env, n_games, seed, lane_offset, lane_stride, arena_rows);
    if (rc == BGAMD_OK) rc = bgamd_env_reset(env, nullptr);
    if (rc != BGAMD_OK) {                       // free whatever was allocated before the failure
        const std::string why = g_hip_err;
        bgamd_env_destroy(env);
        g_hip_err = why;
        return rc;
    }
#ifdef BGAMD_EXPERIMENTAL
    // every BGAMD_* switch reads the same way: set and atoi() != 0 (bench.py mirrors it through bgamd_env_kernel_choice, not the environment)
    auto on = [](const char *name) { const char *v = getenv(name); return v != nullptr && atoi(v) != 0; };
    auto off = [](const char *name) { const char *v = getenv(name); return v != nullptr && atoi(v) == 0; };
    env->root_f32_mfma = on("BGAMD_ROOT_F32");
    env->root_resident = !off("BGAMD_ROOT_RESIDENT");
    // round 3's K-compacted MFMA delta kernel (bg_eval_mfma.h) is correct and canonical but measured 10 % slower than the VALU
    // kernel on the same box (NOTEBOOK.md): opt-in, experimental build only
    env->mfma_delta = on("BGAMD_MFMA_DELTA");
    env->d16 = on("BGAMD_F16X2_RESIDENT");
#endif
    // Where the root pass runs (round 4, profiles/r04_lanes_32768_anatomy.txt, r04_ab_root_pass_in_boundary.txt): inside the boundary launch from 24 576 lanes
    // -- there a boundary workgroup owns 128 games (BROOT_GPW), so 32 768 lanes still put a workgroup on every CU: 94.5 vs 98.7 us per step with the pass as a
    // launch of its own (65 536: 143.2 vs 143.6; 16 384: 72.0 vs 71.7, a tie: smaller envs keep the launch) -- and always on the caller's stream.
    env->root_in_boundary = n_games >= 24576 && LANE_NT == BROOT_THREADS;
    env->overlap = false;
#ifdef BGAMD_EXPERIMENTAL
    // Round 5: the launch structures that lost their A/B (rounds 1-4) are honoured by the EXPERIMENTAL build only, as bit-identity references
    // (tests/test_gpu_experimental.py): the root pass forced in / out of the boundary launch (BGAMD_ROOT_IN_BOUNDARY) or forked onto the env's second
    // stream (BGAMD_OVERLAP), the expansion as two launches (BGAMD_EXPAND_MERGED=0), one row arena (BGAMD_SPLIT_ARENA=0), unsharded node lists
    // (BGAMD_LIST_SHARDS=0), and the expansion launch's timing knobs.  The default build reads none of them.
    {
        const char *rib = getenv("BGAMD_ROOT_IN_BOUNDARY");
        if (rib) env->root_in_boundary = atoi(rib) != 0 && LANE_NT == BROOT_THREADS;
        const char *xm = getenv("BGAMD_EXPAND_MERGED");
        env->expand_merged = xm ? atoi(xm) != 0 : true;
        const char *xp = getenv("BGAMD_EXPAND_DBL_PCT");
        if (xp && atoi(xp) >= 5 && atoi(xp) <= 95) env->expand_dbl_pct = atoi(xp);
        const char *ls = getenv("BGAMD_LIST_SHARDS");
        env->list_shards = ls ? atoi(ls) != 0 : true;
        const char *sa = getenv("BGAMD_SPLIT_ARENA");
        env->split_arena = sa ? atoi(sa) != 0 : true;
        const char *xq = getenv("BGAMD_EXPAND_PARTS");
        if (xq && atoi(xq) >= 1 && atoi(xq) <= 3) env->expand_parts = atoi(xq);
        const char *xn = getenv("BGAMD_EXPAND_DBL_NPB");
        if (xn && atoi(xn) >= 64 && atoi(xn) <= XALL_NT) env->expand_dbl_npb = atoi(xn) & ~63;
    }
    env->overlap = getenv("BGAMD_NO_OVERLAP") == nullptr && getenv("BGAMD_OVERLAP") != nullptr && atoi(getenv("BGAMD_OVERLAP")) != 0;
#endif
    if (hipStreamCreateWithFlags(&env->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&env->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&env->ev_join, hipEventDisableTiming) != hipSuccess) {
        g_hip_err = "side stream / events";
        bgamd_env_destroy(env);
        return BGAMD_E_HIP;
    }
    *out = env;
    return BGAMD_OK;
}

static int env_allocate(bgamd_env *env, int64_t n_games, uint64_t seed, uint64_t lane_offset, uint64_t lane_stride,
                        int64_t arena_rows)
{
    EnvView &v = env->v;
    v.n = n_games; v.seed = seed; v.lane_offset = lane_offset; v.lane_stride = lane_stride ? lane_stride : (uint64_t)n_games;
    long long cap = arena_rows > 0 ? arena_rows : n_games * 256;
    if (cap < 65536) cap = 65536;
    if (cap > (1ll << 31) - 64) cap = (1ll << 31) - 64;
    v.cap = cap;
    const size_t n = (size_t)n_games;
    HIPCHK(hipMalloc(&v.planes, n * 8 * 4));
    HIPCHK(hipMalloc(&v.meta, n * 4));
    HIPCHK(hipMalloc(&v.ply, n * 4));
    HIPCHK(hipMalloc(&v.episode, n * 4));
    HIPCHK(hipMalloc(&v.flags, n * 4));
    HIPCHK(hipMalloc(&v.cand_off, n * 4));
    HIPCHK(hipMalloc(&v.cand_cnt, n * 4));
    HIPCHK(hipMalloc(&v.chosen, n * 4));
    HIPCHK(hipMalloc(&v.chosen_seq, n * 4));
    HIPCHK(hipMalloc(&v.chosen_val, n * 4));
    HIPCHK(hipMalloc(&v.rows, (size_t)cap * 32));
    HIPCHK(hipMalloc(&v.seqs, (size_t)cap * 4));
    HIPCHK(hipMalloc(&v.values, (size_t)cap * 4));
    HIPCHK(hipMalloc(&v.counters, C_COUNT * 8));
    for (int k = 0; k < 2; ++k) {
        HIPCHK(hipMalloc(&env->d_w[k], N_PARAMS * 4));
        HIPCHK(hipMalloc(&env->d_wl[k], EVAL_LDS_BYTES));
        HIPCHK(hipMalloc(&env->d_wt[k], DELTA_W_FLOATS * 4));
#ifdef BGAMD_EXPERIMENTAL
        HIPCHK(hipMalloc(&env->d_wm[k], MD_W_BYTES));
#endif
        HIPCHK(hipMalloc(&env->d_wl3[k], 3 * EVAL16_W_BYTES));
        HIPCHK(hipMalloc(&env->d_wr2[k], 2 * EVAL16_W_BYTES));
        HIPCHK(hipMalloc(&env->d_wl16[k], EVAL16_W_BYTES));
        HIPCHK(hipMalloc(&env->d_wlx2[k], EVAL16X2_W_BYTES));
#ifdef BGAMD_EXPERIMENTAL
        HIPCHK(hipMalloc(&env->d_wd16[k], EVAL16X2_W_BYTES));
#endif
    }
    HIPCHK(hipMalloc(&env->d_lut16, EVAL16_LUT_BYTES));
    HIPCHK(hipFuncSetAttribute((const void *)eval_rows_f16x2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, EVAL16X2_LDS_TOTAL));
    HIPCHK(hipMalloc(&env->d_lut, EVAL16_LUT_BYTES));
    HIPCHK(hipFuncSetAttribute((const void *)eval_rows_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, EVAL16_LDS_TOTAL));
    {   // staged greedy step (bg_staged.h): node lists, per-workgroup staging, unique arena
        StagedView &sv = env->sv;
        const long long ng = n_games;
        sv.cap_d1 = ng * 15 < 4096 ? 4096 : ng * 15;
        sv.cap_d2 = ng * 64 < 4096 ? 4096 : ng * 64;
        sv.cap_f = ng * 256 < 16384 ? 16384 : ng * 256;
        sv.cap_rows = cap;
        sv.b_base = 0;
        sv.shards = 1;
        HIPCHK(hipMalloc(&sv.d1, (size_t)sv.cap_d1 * sizeof(Node)));
        HIPCHK(hipMalloc(&sv.d2, (size_t)sv.cap_d2 * sizeof(Node)));
        HIPCHK(hipMalloc(&sv.f, (size_t)sv.cap_f * sizeof(Node)));
        sv.cap_f2 = sv.cap_f;
        HIPCHK(hipMalloc(&sv.f2, (size_t)sv.cap_f2 * sizeof(Node)));
        HIPCHK(hipMalloc(&sv.u_rows, (size_t)cap * 32));
        HIPCHK(hipMalloc(&sv.u_info, (size_t)cap * sizeof(uint2)));
        // an overflowing step (flagged, raised by stats) leaves holes in the arena: they must name a valid game
        HIPCHK(hipMemset(sv.u_rows, 0, (size_t)cap * 32));
        HIPCHK(hipMemset(sv.u_info, 0, (size_t)cap * sizeof(uint2)));
        HIPCHK(hipMalloc(&sv.best, n * 8));
        HIPCHK(hipMalloc(&sv.root_rows, n * 32));
        HIPCHK(hipMalloc(&sv.root_hidden, n * N_HID * 4));
        HIPCHK(hipMalloc(&sv.tops, 2 * T_COUNT * 8));             // two sets: consecutive steps of a run alternate
        HIPCHK(hipMemset(sv.tops, 0, 2 * T_COUNT * 8));
        env->tops_base = sv.tops;
        RandomView &rv = env->rv;                    // bounded random-policy step (bg_random_kernels.h)
        rv.cap = sv.cap_f;                           // own task list and counter: a step boundary of a multi-step run
        HIPCHK(hipMalloc(&rv.tasks, (size_t)rv.cap * sizeof(Node)));   // reads the tasks while the next roots write sv.f
        HIPCHK(hipMalloc(&rv.top, 8));
        HIPCHK(hipMemset(rv.top, 0, 8));
        HIPCHK(hipMalloc(&rv.task_count, (size_t)rv.cap * 4));
        HIPCHK(hipMalloc(&rv.task_off, n * 4));
        HIPCHK(hipMalloc(&rv.task_n, n * 4));
    }
    HIPCHK(hipMemset(v.counters, 0, C_COUNT * 8));
    HIPCHK(hipFuncSetAttribute((const void *)eval_rows_f32_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, EVAL_LDS_TOTAL));
#ifdef BGAMD_EXPERIMENTAL
    HIPCHK(hipFuncSetAttribute((const void *)eval_rows_f32_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, EVAL_LDS_TOTAL));
#endif
    HIPCHK(hipFuncSetAttribute((const void *)eval_rows_delta_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DELTA_LDS_TOTAL));
    HIPCHK(hipFuncSetAttribute((const void *)boundary_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, BROOT_LDS_BYTES));
#ifdef BGAMD_EXPERIMENTAL
    HIPCHK(hipFuncSetAttribute((const void *)eval_rows_mdelta_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MD_LDS_TOTAL));
    HIPCHK(hipFuncSetAttribute((const void *)root_hidden_bf16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ROOT3_LDS_TOTAL));
#endif
    return BGAMD_OK;
}

int bgamd_env_destroy(bgamd_env *env)
{
    ENV_GUARD(env);
    hipDeviceSynchronize();
    EnvView &v = env->v;
    void *ptrs[] = {v.planes, v.meta, v.ply, v.episode, v.flags, v.cand_off, v.cand_cnt, v.chosen, v.chosen_seq,
                    v.chosen_val, v.rows, v.seqs, v.values, v.counters, env->d_w[0], env->d_wl[0], env->d_wl16[0], env->d_w[1], env->d_wl[1], env->d_wl16[1], env->d_lut, env->d_wlx2[0], env->d_wlx2[1], env->d_wd16[0], env->d_wd16[1], env->d_lut16, env->d_wt[0], env->d_wt[1], env->d_wm[0], env->d_wm[1], env->d_wl3[0], env->d_wl3[1], env->d_wr2[0], env->d_wr2[1], env->sv.root_rows, env->sv.root_hidden,
                    env->sv.d1, env->sv.d2, env->sv.f, env->sv.f2, env->sv.u_rows, env->sv.u_info, env->sv.best, env->tops_base, env->rv.tasks, env->rv.top, env->rv.task_count, env->rv.task_off, env->rv.task_n};
    for (void *p : ptrs) if (p) hipFree(p);
    if (env->d_scalar) hipFree(env->d_scalar);
    if (env->d_tmp) hipFree(env->d_tmp);
    for (hipEvent_t e : env->ev) hipEventDestroy(e);
    if (env->ev_fork) hipEventDestroy(env->ev_fork);
    if (env->ev_join) hipEventDestroy(env->ev_join);
    if (env->side) hipStreamDestroy(env->side);
    delete env;
    return BGAMD_OK;
}

int64_t bgamd_env_num_games(const bgamd_env *env) { return env ? env->v.n : 0; }

int bgamd_env_reset(bgamd_env *env, void *stream)
{
    ENV_GUARD(env);
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipSetDevice(env->device));
    hipLaunchKernelGGL(reset_kernel, grid1(env->v.n, 256), dim3(256), 0, s, env->v, (const int32_t *)nullptr, 0u);
    HIPCHK(hipMemsetAsync(env->v.counters, 0, C_COUNT * 8, s));
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

int bgamd_env_reset_episode(bgamd_env *env, uint32_t episode, void *stream)
{
    ENV_GUARD(env);
    hipLaunchKernelGGL(reset_kernel, grid1(env->v.n, 256), dim3(256), 0, (hipStream_t)stream, env->v, (const int32_t *)nullptr, episode);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

int bgamd_env_reseed(bgamd_env *env, uint64_t seed, uint64_t lane_offset, uint64_t lane_stride, void *stream)
{
    ENV_GUARD(env);
    env->v.seed = seed;
    env->v.lane_offset = lane_offset;
    env->v.lane_stride = lane_stride ? lane_stride : (uint64_t)env->v.n;
    return bgamd_env_reset_episode(env, 0u, stream);
}

int bgamd_env_reset_lanes(bgamd_env *env, const int32_t *d_mask, void *stream)
{
    if (!env || !d_mask) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(env->device));
    hipLaunchKernelGGL(reset_kernel, grid1(env->v.n, 256), dim3(256), 0, (hipStream_t)stream, env->v, d_mask, 0u);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

int bgamd_env_set_states(bgamd_env *env, const int32_t *d_states28, const int32_t *d_turn, void *stream)
{
    if (!env || (!d_states28 && !d_turn)) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(env->device));
    hipLaunchKernelGGL(set_states_kernel, grid1(env->v.n, 128), dim3(128), 0, (hipStream_t)stream, env->v, d_states28, d_turn);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

int bgamd_env_get_states(bgamd_env *env, int32_t *d_states28, int32_t *d_turn, void *stream)
{
    ENV_GUARD(env);
    hipLaunchKernelGGL(get_states_kernel, grid1(env->v.n, 128), dim3(128), 0, (hipStream_t)stream, env->v, d_states28, d_turn);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

int bgamd_env_get_flags(bgamd_env *env, int32_t *d_flags, void *stream)
{
    if (!env || !d_flags) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(env->device));
    hipLaunchKernelGGL(get_flags_kernel, grid1(env->v.n, 256), dim3(256), 0, (hipStream_t)stream, env->v, d_flags);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

int bgamd_env_snapshot(bgamd_env *env, int32_t *d_out, void *stream)
{
    if (!env || !d_out) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(env->device));
    hipLaunchKernelGGL(snapshot_kernel, grid1(env->v.n, 128), dim3(128), 0, (hipStream_t)stream, env->v, d_out);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

// ---- scalar surface with host arguments (one-lane envs) ----
static int launch_emit(bgamd_env *env, int flags, int with_seq, const int32_t *ov_player, const int32_t *ov_dice, hipStream_t s);
static int check_err_flags(bgamd_env *env, unsigned long long f);
static int scalar_guard(bgamd_env *env)
{
    if (!env || env->v.n != 1) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(env->device));
    if (!env->d_scalar) HIPCHK(hipMalloc(&env->d_scalar, 1024 * 4));
    return BGAMD_OK;
}

int bgamd_game_snapshot(bgamd_env *env, int32_t h_out[32])
{
    int rc = scalar_guard(env);
    if (rc || !h_out) return rc ? rc : BGAMD_E_INVALID;
    hipLaunchKernelGGL(snapshot_kernel, dim3(1), dim3(64), 0, nullptr, env->v, env->d_scalar + 64);
    HIPCHK(hipMemcpy(h_out, env->d_scalar + 64, 32 * 4, hipMemcpyDeviceToHost));
    return BGAMD_OK;
}

int bgamd_game_set_state(bgamd_env *env, const int32_t *h_state28, int turn)
{
    int rc = scalar_guard(env);
    if (rc || (!h_state28 && turn < 0)) return rc ? rc : BGAMD_E_INVALID;
    int32_t h[29];
    if (h_state28) memcpy(h, h_state28, 28 * 4);
    h[28] = turn;
    HIPCHK(hipMemcpy(env->d_scalar, h, 29 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(set_states_kernel, dim3(1), dim3(64), 0, nullptr, env->v, h_state28 ? (const int32_t *)env->d_scalar : nullptr,
                       turn >= 0 ? (const int32_t *)(env->d_scalar + 28) : nullptr);
    HIPCHK(hipDeviceSynchronize());
    return BGAMD_OK;
}

int bgamd_game_set_dice(bgamd_env *env, int d1, int d2)
{
    int rc = scalar_guard(env);
    if (rc) return rc;
    const int32_t h[2] = {d1, d2};
    HIPCHK(hipMemcpy(env->d_scalar, h, 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(dice_kernel, dim3(1), dim3(64), 0, nullptr, env->v, (const int32_t *)env->d_scalar, (int32_t *)nullptr, 0);
    HIPCHK(hipDeviceSynchronize());
    return BGAMD_OK;
}

int bgamd_game_roll(bgamd_env *env, int32_t h_dice[2])
{
    int rc = scalar_guard(env);
    if (rc || !h_dice) return rc ? rc : BGAMD_E_INVALID;
    hipLaunchKernelGGL(dice_kernel, dim3(1), dim3(64), 0, nullptr, env->v, (const int32_t *)nullptr, env->d_scalar + 64, 2);
    HIPCHK(hipMemcpy(h_dice, env->d_scalar + 64, 8, hipMemcpyDeviceToHost));
    return BGAMD_OK;
}

int bgamd_game_legal_moves(bgamd_env *env, int player, int die, int8_t *h_pairs)
{
    int rc = scalar_guard(env);
    if (rc || !h_pairs) return rc ? rc : BGAMD_E_INVALID;
    const int32_t h[2] = {player, die};
    HIPCHK(hipMemcpy(env->d_scalar, h, 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(legal_moves_kernel, dim3(1), dim3(64), 0, nullptr, env->v, (const int32_t *)env->d_scalar, (const int32_t *)(env->d_scalar + 1),
                       env->d_scalar + 64, (int8_t *)(env->d_scalar + 80));
    int32_t out[1 + 13];
    HIPCHK(hipMemcpy(out, env->d_scalar + 64, 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(h_pairs, env->d_scalar + 80, 52, hipMemcpyDeviceToHost));
    return out[0];
}

int bgamd_game_try_move(bgamd_env *env, int player, int dice, int origin, int dest)
{
    int rc = scalar_guard(env);
    if (rc) return rc;
    const int32_t h[4] = {player, dice, origin, dest};
    HIPCHK(hipMemcpy(env->d_scalar, h, 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(try_move_kernel, dim3(1), dim3(64), 0, nullptr, env->v, (const int32_t *)env->d_scalar, (const int32_t *)(env->d_scalar + 1),
                       (const int32_t *)(env->d_scalar + 2), (const int32_t *)(env->d_scalar + 3), env->d_scalar + 64);
    int32_t code = 0;
    HIPCHK(hipMemcpy(&code, env->d_scalar + 64, 4, hipMemcpyDeviceToHost));
    return code;
}

int64_t bgamd_game_enumerate(bgamd_env *env, int player, int d1, int d2, int32_t *h_states28, int8_t *h_seq, int32_t *h_len, int64_t cap)
{
    int rc = scalar_guard(env);
    if (rc || cap < 0) return rc ? rc : BGAMD_E_INVALID;
    const int32_t h[3] = {player, d1, d2};
    HIPCHK(hipMemcpy(env->d_scalar, h, 12, hipMemcpyHostToDevice));
    rc = launch_emit(env, 0, 1, (const int32_t *)env->d_scalar, (const int32_t *)(env->d_scalar + 1), nullptr);
    if (rc) return rc;
    unsigned long long hc[C_COUNT];
    HIPCHK(hipMemcpy(hc, env->v.counters, sizeof hc, hipMemcpyDeviceToHost));
    rc = check_err_flags(env, hc[C_ERR]);
    if (rc) return rc;
    const long long C = (long long)hc[C_ARENA_TOP];
    const long long m = C < cap ? C : cap;
    if (m > 0 && (h_states28 || h_seq || h_len)) {
        const size_t need = (size_t)m * (28 * 4 + 8 + 4);
        if (need > env->tmp_bytes) {
            if (env->d_tmp) hipFree(env->d_tmp);
            env->d_tmp = nullptr; env->tmp_bytes = 0;
            HIPCHK(hipMalloc(&env->d_tmp, need));
            env->tmp_bytes = need;
        }
        int32_t *d_st = (int32_t *)env->d_tmp;
        int32_t *d_len = d_st + (size_t)m * 28;
        int8_t *d_seq = (int8_t *)(d_len + m);
        hipLaunchKernelGGL(rows_read_kernel, grid1(m, 128), dim3(128), 0, nullptr, env->v.rows, env->v.seqs, 0ll, m, d_st, d_seq, d_len);
        if (h_states28) HIPCHK(hipMemcpy(h_states28, d_st, (size_t)m * 28 * 4, hipMemcpyDeviceToHost));
        if (h_len) HIPCHK(hipMemcpy(h_len, d_len, (size_t)m * 4, hipMemcpyDeviceToHost));
        if (h_seq) HIPCHK(hipMemcpy(h_seq, d_seq, (size_t)m * 8, hipMemcpyDeviceToHost));
    }
    HIPCHK(hipGetLastError());
    return (int64_t)C;
}

int bgamd_env_set_dice(bgamd_env *env, const int32_t *d_dice, void *stream)
{
    if (!env || !d_dice) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(env->device));
    hipLaunchKernelGGL(dice_kernel, grid1(env->v.n, 256), dim3(256), 0, (hipStream_t)stream, env->v, d_dice, (int32_t *)nullptr, 0);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}
int bgamd_env_get_dice(bgamd_env *env, int32_t *d_dice, void *stream)
{
    if (!env || !d_dice) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(env->device));
    hipLaunchKernelGGL(dice_kernel, grid1(env->v.n, 256), dim3(256), 0, (hipStream_t)stream, env->v, (const int32_t *)nullptr, d_dice, 0);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}
int bgamd_env_roll(bgamd_env *env, int advance_ply, void *stream)
{
    ENV_GUARD(env);
    hipLaunchKernelGGL(dice_kernel, grid1(env->v.n, 256), dim3(256), 0, (hipStream_t)stream, env->v, (const int32_t *)nullptr, (int32_t *)nullptr, advance_ply ? 2 : 1);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

static int launch_emit(bgamd_env *env, int flags, int with_seq, const int32_t *ov_player, const int32_t *ov_dice, hipStream_t s)
{
    HIPCHK(hipMemsetAsync(&env->v.counters[C_ARENA_TOP], 0, 8, s));
    {
        KTimer t(env, s, 0);
        hipLaunchKernelGGL(emit_kernel, grid1(env->v.n, 64), dim3(64), 0, s, env->v, flags, with_seq, ov_player, ov_dice);
    }
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

int bgamd_env_enumerate(bgamd_env *env, const int32_t *d_player, const int32_t *d_dice, void *stream)
{
    ENV_GUARD(env);
    return launch_emit(env, 0, 1, d_player, d_dice, (hipStream_t)stream);
}

static int check_err_flags(bgamd_env *env, unsigned long long f)
{
    (void)env;
    if (f & ERRF_ARENA) return BGAMD_E_ARENA;
    if (f & ERRF_STATE) return BGAMD_E_STATE;
    if (f & ERRF_DELTA) return BGAMD_E_DELTA;
    return BGAMD_OK;
}

int64_t bgamd_env_candidates_info(bgamd_env *env, int64_t *d_offsets, int32_t *d_counts, void *stream)
{
    ENV_GUARD(env);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cand_info_kernel, grid1(env->v.n, 256), dim3(256), 0, s, env->v, d_offsets, d_counts);
    unsigned long long h[C_COUNT];
    HIPCHK(hipMemcpyAsync(h, env->v.counters, sizeof h, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const int rc = check_err_flags(env, h[C_ERR]);
    if (rc) return rc;
    return (int64_t)h[C_ARENA_TOP];
}

int bgamd_env_candidates_read(bgamd_env *env, int64_t first, int64_t n_rows, int32_t *d_states28, int8_t *d_seq,
                              int32_t *d_seq_len, void *stream)
{
    if (!env || first < 0 || n_rows < 0 || first + n_rows > env->v.cap) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(env->device));
    if (n_rows == 0) return BGAMD_OK;
    hipLaunchKernelGGL(rows_read_kernel, grid1(n_rows, 128), dim3(128), 0, (hipStream_t)stream, env->v.rows, env->v.seqs,
                       (long long)first, (long long)n_rows, d_states28, d_seq, d_seq_len);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

int bgamd_env_step_random(bgamd_env *env, int flags, const uint32_t *d_choice_u32, void *stream)
{
    ENV_GUARD(env);
    hipStream_t s = (hipStream_t)stream;
    const long long n = env->v.n;
    HIPCHK(hipMemsetAsync(env->rv.top, 0, 8, s));
    {
        KTimer t(env, s, 3);
        hipLaunchKernelGGL(rnd_tasks_kernel, grid1(n, 256), dim3(256), 0, s, env->v, env->rv, flags, -1.0f);
        long long b = (n * 64 + 255) / 256;
        const long long lim = (long long)env->n_cu * 8;
        hipLaunchKernelGGL(rnd_count_kernel, dim3((unsigned)(b > lim ? lim : b)), dim3(256), 0, s, env->v, env->rv);
        hipLaunchKernelGGL(rnd_select_kernel, grid1(n, 256), dim3(256), 0, s, env->v, env->rv, flags, d_choice_u32);
    }
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

// the whole-tree walk (one lane per game, count pass + select pass); kept as the reference implementation of
// the bounded kernels above -- tests compare the two
int bgamd_env_step_random_walk(bgamd_env *env, int flags, const uint32_t *d_choice_u32, void *stream)
{
    ENV_GUARD(env);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(step_random_kernel, grid1(env->v.n, 64), dim3(64), 0, s, env->v, flags, d_choice_u32);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

int bgamd_env_load_weights(bgamd_env *env, const float *h_weights) { return bgamd_env_load_weights_slot(env, 0, h_weights); }

// host only: do the 25 601 floats survive the value net's weight layouts?  (b1, W2, b2 stay fp32: they must be finite; W1 must fit the
// f16 hi + lo planes of the root pass, bg_root_resident.h)
int bgamd_weights_check(const float *h_weights)
{
    if (!h_weights) return BGAMD_E_INVALID;
    for (int i = N_HID * N_IN; i < N_PARAMS; ++i)
        if (!(h_weights[i] - h_weights[i] == 0.0f)) return BGAMD_E_WEIGHTS;
    std::vector<uint16_t> planes((size_t)3 * K16_STEPS * 4 * 64 * 8);
    return relayout_w1_f16x2_root(h_weights, planes.data()) < 0 ? BGAMD_E_WEIGHTS : BGAMD_OK;
}

int bgamd_env_load_weights_slot(bgamd_env *env, int slot, const float *h_weights)
{
    if (!env || !h_weights || slot < 0 || slot > 1) return BGAMD_E_INVALID;
    if (int rc = bgamd_weights_check(h_weights)) return rc;      // before anything of the slot is overwritten
    HIPCHK(hipSetDevice(env->device));
    std::vector<float> wl((size_t)K_STEPS * 64 * 4);
    relayout_w1_f32(h_weights, wl.data());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(env->d_w[slot], h_weights, N_PARAMS * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(env->d_wl[slot], wl.data(), EVAL_LDS_BYTES, hipMemcpyHostToDevice));
    std::vector<float> wt((size_t)DELTA_W_FLOATS);
    relayout_w1_delta(h_weights, wt.data());
    HIPCHK(hipMemcpy(env->d_wt[slot], wt.data(), DELTA_W_FLOATS * 4, hipMemcpyHostToDevice));
#ifdef BGAMD_EXPERIMENTAL
    std::vector<uint32_t> wmd((size_t)MD_W_DWORDS);
    env->wm_ok[slot] = relayout_w1_mdelta(h_weights, wmd.data()) >= 0;
    HIPCHK(hipMemcpy(env->d_wm[slot], wmd.data(), MD_W_BYTES, hipMemcpyHostToDevice));
#endif
    std::vector<uint16_t> wl3((size_t)3 * K16_STEPS * 4 * 64 * 8);
    relayout_w1_bf16x3(h_weights, wl3.data());
    HIPCHK(hipMemcpy(env->d_wl3[slot], wl3.data(), 3 * EVAL16_W_BYTES, hipMemcpyHostToDevice));
    relayout_w1_f16x2_root(h_weights, wl3.data());
    HIPCHK(hipMemcpy(env->d_wr2[slot], wl3.data(), 2 * EVAL16_W_BYTES, hipMemcpyHostToDevice));
    std::vector<uint16_t> wl16((size_t)K16_STEPS * 4 * 64 * 8);
    relayout_w1_bf16(h_weights, wl16.data());
    uint32_t lut[32];
    make_count_lut(lut);
    HIPCHK(hipMemcpy(env->d_wl16[slot], wl16.data(), EVAL16_W_BYTES, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(env->d_lut, lut, EVAL16_LUT_BYTES, hipMemcpyHostToDevice));
    std::vector<uint16_t> wx2((size_t)2 * K16_STEPS * 4 * 64 * 8);
    relayout_w1_f16x2(h_weights, wx2.data());
    make_count_lut_f16(lut);
    HIPCHK(hipMemcpy(env->d_wlx2[slot], wx2.data(), EVAL16X2_W_BYTES, hipMemcpyHostToDevice));
#ifdef BGAMD_EXPERIMENTAL
    relayout_w1_d16(h_weights, wx2.data());
    HIPCHK(hipMemcpy(env->d_wd16[slot], wx2.data(), EVAL16X2_W_BYTES, hipMemcpyHostToDevice));
#endif
    HIPCHK(hipMemcpy(env->d_lut16, lut, EVAL16_LUT_BYTES, hipMemcpyHostToDevice));
    env->has_weights[slot] = true;
    return BGAMD_OK;
}

static int launch_eval(bgamd_env *env, int slot, int precision, const unsigned long long *n_rows_ptr, long long n_rows_imm,
                       const uint4 *rows, float *values, const uint2 *info, unsigned long long *best, hipStream_t s)
{
    if (!env->has_weights[slot]) return BGAMD_E_NOWEIGHTS;
    if (precision != BGAMD_F32 && precision != BGAMD_BF16 && precision != BGAMD_F16X2 && precision != BGAMD_F32_DENSE)
        return BGAMD_E_INVALID;
    const float *b1 = env->d_w[slot] + N_HID * N_IN, *w2 = b1 + N_HID, *b2 = w2 + N_HID;
    // every workgroup first stages W1 in LDS: no more workgroups than the rows can use (a greedy step averages ~18
    // rows per game; more than the estimate only means more tiles per wave)
    const long long est_rows = n_rows_ptr ? env->v.n * 24 : n_rows_imm;
    long long eb = (est_rows + (EVAL_THREADS / 64) * 32 - 1) / ((EVAL_THREADS / 64) * 32);
    eb = eb < 1 ? 1 : (eb > env->n_cu ? env->n_cu : eb);
    const dim3 egrid((unsigned)eb);
#ifdef BGAMD_EXPERIMENTAL
    if (precision == BGAMD_F16X2 && env->d16) {
        KTimer t(env, s, 1);
        long long tiles = (est_rows + 31) / 32;
        const long long per_cu = BG_D16_WAVES;                                                     // 4-wave workgroups per CU
        long long wg = tiles < 1 ? 1 : (tiles > per_cu * env->n_cu ? per_cu * env->n_cu : tiles);
        hipLaunchKernelGGL(eval_rows_d16_kernel<2>, dim3((unsigned)wg), dim3(D16_THREADS), D16_LDS_BYTES, s, rows, n_rows_ptr,
                           n_rows_imm, n_rows_ptr ? &env->v.counters[C_ROWS_EVAL] : (unsigned long long *)nullptr,
                           (const uint4 *)env->d_wd16[slot], (const uint2 *)env->d_lut16, w2, b2, values, info, best,
                           n_rows_ptr ? &env->v.counters[C_KSTEPS] : (unsigned long long *)nullptr, (unsigned long long *)nullptr, 0);
    } else
#endif
    if (precision == BGAMD_F16X2) {
        KTimer t(env, s, 1);
        hipLaunchKernelGGL(eval_rows_f16x2_kernel, egrid, dim3(EVAL_THREADS), EVAL16X2_LDS_TOTAL, s, rows, n_rows_ptr,
                           n_rows_imm, n_rows_ptr ? &env->v.counters[C_ROWS_EVAL] : (unsigned long long *)nullptr,
                           (const uint4 *)env->d_wlx2[slot], (const uint2 *)env->d_lut16, b1, w2, b2, values, info, best);
    } else if (precision == BGAMD_BF16) {
        KTimer t(env, s, 1);
        hipLaunchKernelGGL(eval_rows_bf16_kernel, egrid, dim3(EVAL_THREADS), EVAL16_LDS_TOTAL, s, rows, n_rows_ptr,
                           n_rows_imm, n_rows_ptr ? &env->v.counters[C_ROWS_EVAL] : (unsigned long long *)nullptr,
                           (const uint4 *)env->d_wl16[slot], (const uint2 *)env->d_lut, b1, w2, b2, values, info, best);
    } else {
        KTimer t(env, s, 1);
        hipLaunchKernelGGL(eval_rows_f32_kernel<false>, egrid, dim3(EVAL_THREADS), EVAL_LDS_TOTAL, s, rows, n_rows_ptr,
                           n_rows_imm, n_rows_ptr ? &env->v.counters[C_ROWS_EVAL] : (unsigned long long *)nullptr,
                           (const float4 *)env->d_wl[slot], b1, w2, b2, values, info, best,
                           n_rows_ptr ? &env->v.counters[C_KSTEPS] : (unsigned long long *)nullptr);
    }
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

}  // extern "C"

namespace {
// The streams one greedy step is issued on: everything on `gen`, except the root pass of the incremental value net, which
// goes to `root` (the env's side stream when it overlaps the doubles plies, else the same stream).
// (A CU-partitioned variant -- move generation of one sub-batch on 64-128 masked CUs beside the value net of another on the
// rest -- was built on this and measured 2.2-4x SLOWER: the "latency-bound" stage kernels need every CU's wave slots,
// their time scales with 1/CUs.  profiles/r02_cu_partition_group_run.txt)
struct StepStreams {
    hipStream_t gen, root;
    int n_cu;
};

// State of a run of greedy steps on one env (what bgamd_env_run_greedy kept in locals)
struct GreedyRun {
    bgamd_env *env;
    int flags, slot, precision, parity = 0;
    float epsilon;
    bool incremental;
    long long xall_nd = 1, xall_nl = 1;    // expand_all_kernel: workgroups for the doubles turns / for the non-doubles leaf stage
    StagedView sv;
    EnvView ev;                            // the env's view as this run's launches take it (ring log: the slots of the step at hand)
    long long cur_step = 0;                // ring log: the env step being played
    bool root_ready = false;               // the boundary launch of the step before has already run this step's root pass
    const float *b1, *w2, *b2;

    int init(bgamd_env *e, int fl, float eps, int prec)
    {
        env = e; flags = fl; epsilon = eps; precision = prec;
        slot = (fl & BGAMD_WEIGHTS_SLOT1) ? 1 : 0;
        if (!env->has_weights[slot]) return BGAMD_E_NOWEIGHTS;
        if (prec != BGAMD_F32 && prec != BGAMD_BF16 && prec != BGAMD_F16X2 && prec != BGAMD_F32_DENSE) return BGAMD_E_INVALID;
        incremental = prec == BGAMD_F32;
        b1 = env->d_w[slot] + N_HID * N_IN; w2 = b1 + N_HID; b2 = w2 + N_HID;
        sv = env->sv;
        sv.tops = env->tops_base;
        // two row arenas (by class) for the incremental value net; the other value-net kernels read one
        sv.b_base = (incremental && env->split_arena) ? (env->sv.cap_rows / N_ARENAS) & ~63ll : 0;
        {   // grid of expand_all_kernel: two 512-thread workgroups per CU are resident, the first xall_nd take the doubles turns (one of each kind
            // per CU at the default share); the roots' lists in LIST_SHARDS parts when both counts divide
#ifdef BG_XALL_WG_PER_CU                                                                     // (the round-5 co-residency experiment: fewer, so that one fits beside a value-net workgroup)
            const long long n = env->v.n, slots = (long long)BG_XALL_WG_PER_CU * (long long)env->n_cu;
#else
            const long long n = env->v.n, slots = (1024 / XALL_NT) * (long long)env->n_cu;       // 16 waves of expand_all_kernel per CU
#endif
            long long nd = (n * 4 + 63) / 64;                       // doubles: ~64 ply-1 nodes per workgroup in a small env (a sixth of the lanes x <= 15)
            // the share of workgroups for the doubles turns: BGAMD_EXPAND_DBL_PCT, else by env size -- the doubles chains are what the launch waits
            // for, and the fewer lanes, the more so (32 768 lanes: 70 % 22.9 us, 50 % 25.1; 65 536: 62 % 28.5, 50 % 29.2; 16 384: 50 % 20.7, 63 % 21.3)
            const int pct = env->expand_dbl_pct > 0 ? env->expand_dbl_pct : (n >= 49152 ? 62 : (n >= 24576 ? 70 : 50));
            long long nd_lim = slots * pct / 100;
            if (nd_lim >= LIST_SHARDS) nd_lim -= nd_lim % LIST_SHARDS;          // (both counts multiples of the list parts: a full launch keeps the lists in parts)
            nd = nd < 1 ? 1 : (nd > nd_lim ? nd_lim : nd);
            long long nl = (n * 16 + XALL_NT - 1) / XALL_NT;
            const long long nl_lim = slots - nd_lim < 1 ? 1 : slots - nd_lim;
            nl = nl < 1 ? 1 : (nl > nl_lim ? nl_lim : nl);
            xall_nd = nd; xall_nl = nl;
            sv.shards = (env->expand_merged && env->list_shards && nd % LIST_SHARDS == 0 && nl % LIST_SHARDS == 0) ? LIST_SHARDS : 1;
        }
#ifdef BGAMD_EXPERIMENTAL
        if (env->mfma_delta && env->wm_ok[slot]) sv.b_base = 0;
#endif
        parity = 0;
        root_ready = false;
        ev = env->v;
        if (env->ring_rows) {
            // the ring log's contract (include/bgamd.h): every lane takes part in every logged step and restarts itself -- a game is the
            // `length` contiguous slots that end at its end record.  Anything else would make the game table point at other games' rows.
            if (!(fl & BGAMD_AUTO_RESET) || (fl & (BGAMD_ONLY_P1 | BGAMD_ONLY_P2))) return BGAMD_E_INVALID;
            ev.traj = env->ring_rows; ev.traj_plies = env->ring_steps; ev.traj_ring = env->ring_steps; ev.endrec = env->ring_end;
        }
        return BGAMD_OK;
    }
    // ring log: the roots about to be launched log env step traj_step (the step they begin)
    void next_log_slot()
    {
        if (!ev.traj_ring) return;
        cur_step = env->traj_step++;
        ev.log_slot = cur_step % ev.traj_ring;
    }
    dim3 egrid(long long max_items, int mode, int n_cu) const    // expand_kernel: 48 B of LDS per thread, 2 048 threads per CU
    {
        const int nt = expand_threads(mode);
        long long b = (max_items + nt - 1) / nt;
        // one workgroup per CU for the 1 024-thread leaf stage (82 VGPRs: that is what fits; a second round of workgroups
        // would start its latency chain from scratch, a second iteration of the same workgroup has its nodes prefetched)
        const long long lim = (long long)n_cu * (nt >= 1024 ? 1 : 2048 / nt);
        return dim3((unsigned)(b < 1 ? 1 : (b > lim ? lim : b)));
    }
    // two sets of list counters: step t uses set t & 1.  Inside a run the apply of step t and the roots of step t+1
    // share one launch (boundary_kernel); the value-net kernel of step t clears the set those roots allocate from.
    int begin(hipStream_t s)
    {
        HIPCHK(hipMemsetAsync(env->tops_base, 0, 2 * T_COUNT * 8, s));
        KTimer t(env, s, 4);
        next_log_slot();
        hipLaunchKernelGGL(roots_kernel, grid1(env->v.n, LANE_NT), dim3(LANE_NT), 0, s, ev, sv, flags);
        return BGAMD_OK;
    }
    int step(const StepStreams &ss, bool more, bool first_of_run = true)
    {
        (void)first_of_run;
        const long long n = env->v.n;
        hipStream_t s = ss.gen;
        env->choice[0] = incremental ? ((env->mfma_delta && env->wm_ok[slot]) ? 1 : 0)
                                     : (precision == BGAMD_F32_DENSE ? 2 : precision == BGAMD_F16X2 ? (env->d16 ? 4 : 3) : 5);
        const bool own_root_launch = incremental && !root_ready;
        env->choice[2] = ((own_root_launch && ss.root != s) ? 1 : 0) | (env->expand_merged ? 2 : 0);
        env->choice[1] = !incremental ? 0 : (!own_root_launch ? 4 : (env->root_f32_mfma ? 3 : (env->root_resident ? 1 : 2)));      // 4: it ran inside the boundary launch before
        if (own_root_launch) {
            // The value net's root pass (one dense W1 x + b1 per GAME) needs only the root rows the roots just wrote.
            // It runs on a second stream beside the doubles plies -- small latency-bound launches that leave
            // most of the chip idle -- and is joined before the incremental kernel.
            hipStream_t s2 = ss.root;
            if (s2 != s) {
                HIPCHK(hipEventRecord(env->ev_fork, s));
                HIPCHK(hipStreamWaitEvent(s2, env->ev_fork, 0));
            }
#if defined(BG_ABL_STEP) && (BG_ABL_STEP & 1)            // ablation (timing only: the values go stale): no root pass after a run's first step
            if (first_of_run)
#endif
            {
                KTimer t(env, s2, 6);
#ifdef BGAMD_EXPERIMENTAL
                if (env->root_f32_mfma)
                    hipLaunchKernelGGL(eval_rows_f32_kernel<true>, dim3(ss.n_cu), dim3(EVAL_THREADS), EVAL_LDS_TOTAL, s2,
                                       (const uint4 *)sv.root_rows, (const unsigned long long *)nullptr, n, (unsigned long long *)nullptr,
                                       (const float4 *)env->d_wl[slot], b1, w2, b2, sv.root_hidden, (const uint2 *)nullptr,
                                       (unsigned long long *)nullptr, (unsigned long long *)nullptr);
                else if (!env->root_resident) {
                    long long blocks = ((n + 31) / 32 + ROOT3_THREADS / 64 - 1) / (ROOT3_THREADS / 64);
                    if (blocks > ss.n_cu) blocks = ss.n_cu;
                    hipLaunchKernelGGL(root_hidden_bf16x3_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(ROOT3_THREADS),
                                       ROOT3_LDS_TOTAL, s2, (const uint4 *)sv.root_rows, n, (const uint4 *)env->d_wl3[slot],
                                       (const uint2 *)env->d_lut, b1, sv.root_hidden);
                } else
#endif
                {
                    long long blocks = (n + 31) / 32;
                    if (blocks > 2ll * ss.n_cu) blocks = 2ll * ss.n_cu;                 // two 4-wave workgroups per CU (256 VGPRs each wave)
                    hipLaunchKernelGGL(root_hidden_resident_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(ROOTR_THREADS),
                                       ROOTR_LDS_BYTES, s2, (const uint4 *)sv.root_rows, n, (const uint4 *)(ROOT_F16X2 ? env->d_wr2[slot] : env->d_wl3[slot]),
                                       (const uint2 *)(ROOT_F16X2 ? env->d_lut16 : env->d_lut), b1, sv.root_hidden);
                }
            }
            if (s2 != s) HIPCHK(hipEventRecord(env->ev_join, s2));
        }
        StagedView sv_next = sv;
        sv_next.tops = env->tops_base + (parity ^ 1) * T_COUNT;
#ifdef BGAMD_EXPERIMENTAL
        if (!env->expand_merged) {                             // rounds 1-4's two launches (bit-identity reference)
            {
                KTimer t(env, s, 4);
                hipLaunchKernelGGL(doubles_kernel, egrid(n * 4, MODE_PLY2, ss.n_cu), dim3(expand_threads(MODE_PLY2)), 0, s, ev, sv);
            }
            {
                KTimer t(env, s, 5);
                hipLaunchKernelGGL(expand_kernel<MODE_LEAF>, egrid(n * 16, MODE_LEAF, ss.n_cu), dim3(expand_threads(MODE_LEAF)), 0, s, ev, sv);
            }
        } else
#endif
        {
            // one launch: the doubles turns' plies 2, 3 and leaf stage on the first workgroups, the non-doubles leaf stage on the others
            // (two 512-thread workgroups per CU are resident: one of each kind per CU at the default share)
            KTimer t(env, s, 5);
            const long long nd = xall_nd, grid = xall_nd + xall_nl;
            hipLaunchKernelGGL(expand_all_kernel, dim3((unsigned)grid), dim3(XALL_NT), 0, s, ev, sv, (unsigned)nd,
                               (unsigned)env->expand_dbl_npb, (unsigned)env->expand_parts, more ? sv_next.tops : (unsigned long long *)nullptr,
                               (int)T_COUNT);
        }
        // inside a run the apply of this step and the roots of the next share a launch; the counter set those roots allocate from is cleared
        // earlier in THIS step, while nothing uses it: by the expansion launch (every value-net mode), else by the incremental value net
        const bool fused = more && (incremental || env->expand_merged);
        hipStream_t se = s;
        if (incremental) {
            if (own_root_launch && ss.root != s) HIPCHK(hipStreamWaitEvent(s, env->ev_join, 0));
            KTimer t(env, se, 1);
            // every workgroup first copies W1^T (117 KB) into LDS: small envs get only as many as their rows can use
            long long dblocks = (n * 24 + DELTA_THREADS - 1) / DELTA_THREADS;
            dblocks = dblocks < 1 ? 1 : (dblocks > ss.n_cu ? ss.n_cu : dblocks);
#ifdef BGAMD_EXPERIMENTAL
            if (env->mfma_delta && env->wm_ok[slot])
                hipLaunchKernelGGL(eval_rows_mdelta_kernel, dim3((unsigned)dblocks), dim3(MD_THREADS), MD_LDS_TOTAL, se,
                                   (const uint4 *)sv.u_rows, (const unsigned long long *)&sv.tops[T_U], (long long)sv.cap_rows, &env->v.counters[C_ROWS_EVAL],
                                   (const uint4 *)env->d_wm[slot], w2, b2, (const uint4 *)sv.root_rows, (const float *)sv.root_hidden,
                                   env->v.values, (const uint2 *)sv.u_info, sv.best, &env->v.counters[C_KSTEPS],
                                   fused ? sv_next.tops : (unsigned long long *)nullptr, (int)T_COUNT, &env->v.counters[C_ERR],
                                   (unsigned long long)ERRF_DELTA);
            else
#endif
            hipLaunchKernelGGL(eval_rows_delta_kernel, dim3((unsigned)dblocks), dim3(DELTA_THREADS), DELTA_LDS_TOTAL, se,
                               (const uint4 *)sv.u_rows, (const unsigned long long *)&sv.tops[T_U], (long long)sv.cap_rows, &env->v.counters[C_ROWS_EVAL],
                               (const float4 *)env->d_wt[slot], w2, b2, (const uint4 *)sv.root_rows, (const float *)sv.root_hidden,
                               env->v.values, (const uint2 *)sv.u_info, sv.best, &env->v.counters[C_KSTEPS],
                               fused ? sv_next.tops : (unsigned long long *)nullptr, (int)T_COUNT, &env->v.counters[C_ERR],
                               (unsigned long long)ERRF_DELTA, sv.b_base > 0 ? (const unsigned long long *)&sv.tops[T_UB] : (const unsigned long long *)nullptr,
                               (long long)sv.b_base);                 // (the counters of arenas 1, 2, 3 follow one another from T_UB)
        } else {
            const int rc = launch_eval(env, slot, precision, &sv.tops[T_U], sv.cap_rows, sv.u_rows, env->v.values, sv.u_info, sv.best, se);
            if (rc) return rc;
        }
        ExploreView xv{env->rv.tasks, env->rv.task_count, env->rv.task_off, env->rv.task_n};
        if (epsilon > 0.0f) {                              // exploring lanes pick through their counted tasks (bounded work)
            KTimer t(env, s, 3);
            HIPCHK(hipMemsetAsync(env->rv.top, 0, 8, s));
            hipLaunchKernelGGL(rnd_tasks_kernel, grid1(n, 256), dim3(256), 0, s, ev, env->rv, flags & ~BGAMD_ROLL, epsilon);
            long long b = (n * 4 + 255) / 256;
            const long long lim = (long long)ss.n_cu * 8;
            hipLaunchKernelGGL(rnd_count_kernel, dim3((unsigned)(b > lim ? lim : b)), dim3(256), 0, s, ev, env->rv);
        }
        {
            KTimer t(env, s, 2);
            ev.end_slot = ev.traj_ring ? cur_step % ev.traj_ring : 0;      // the apply half closes the step the last roots began
            if (fused) {
                next_log_slot();                                           // ... and the roots half begins the next one
                root_ready = incremental && env->root_in_boundary;
#ifdef BGAMD_EXPERIMENTAL
                if (env->root_f32_mfma || !env->root_resident) root_ready = false;          // (the other root passes exist as launches only)
#endif
                if (root_ready)
                    hipLaunchKernelGGL(boundary_kernel<true>, grid1(n, BROOT_GPW), dim3(LANE_NT), BROOT_LDS_BYTES, s, ev, sv, sv_next, xv, flags, epsilon,
                                       (const uint4 *)(ROOT_F16X2 ? env->d_wr2[slot] : env->d_wl3[slot]), (const uint2 *)(ROOT_F16X2 ? env->d_lut16 : env->d_lut), b1);
                else
                    hipLaunchKernelGGL(boundary_kernel<false>, grid1(n, LANE_NT), dim3(LANE_NT), 0, s, ev, sv, sv_next, xv, flags, epsilon,
                                       (const uint4 *)nullptr, (const uint2 *)nullptr, (const float *)nullptr);
            } else {
                root_ready = false;
                hipLaunchKernelGGL(apply_kernel, grid1(n, LANE_NT), dim3(LANE_NT), 0, s, ev, sv, xv, flags, epsilon);
            }
        }
        env->sv.tops = sv.tops;                                // the set whose T_U describes the last evaluated rows
        env->sv.b_base = sv.b_base;
        if (fused) { parity ^= 1; sv = sv_next; }
        else if (more) {                                       // dense value-net modes: plain per-step sequence
            HIPCHK(hipMemsetAsync(sv.tops, 0, T_COUNT * 8, s));
            KTimer t(env, s, 4);
            next_log_slot();
            hipLaunchKernelGGL(roots_kernel, grid1(n, LANE_NT), dim3(LANE_NT), 0, s, ev, sv, flags);
        }
        return BGAMD_OK;
    }
};

}  // namespace

extern "C" {

int bgamd_env_run_greedy(bgamd_env *env, int flags, float epsilon, int precision, int64_t n_steps, void *stream)
{
    if (!env || n_steps < 0) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(env->device));
    GreedyRun run;
    int rc = run.init(env, flags, epsilon, precision);
    if (rc) return rc;
    if (n_steps == 0) return BGAMD_OK;
    hipStream_t s = (hipStream_t)stream;
    const StepStreams ss{s, (run.incremental && env->overlap) ? env->side : s, env->n_cu};
    rc = run.begin(s);
    if (rc) return rc;
    for (int64_t step = 0; step < n_steps; ++step) {
        rc = run.step(ss, step + 1 < n_steps, step == 0);
        if (rc) return rc;
    }
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

int bgamd_env_step_greedy(bgamd_env *env, int flags, float epsilon, int precision, void *stream)
{
    return bgamd_env_run_greedy(env, flags, epsilon, precision, 1, stream);
}

int bgamd_env_last_choice(bgamd_env *env, int32_t *d_chosen, int32_t *d_count, int8_t *d_seq, int32_t *d_seq_len,
                          float *d_value, void *stream)
{
    ENV_GUARD(env);
    hipLaunchKernelGGL(last_choice_kernel, grid1(env->v.n, 256), dim3(256), 0, (hipStream_t)stream, env->v, d_chosen, d_count,
                       d_seq, d_seq_len, d_value);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

int bgamd_env_stats(bgamd_env *env, uint64_t h_out[10])
{
    if (!env || !h_out) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(env->device));
    HIPCHK(hipDeviceSynchronize());
    unsigned long long h[C_COUNT];
    HIPCHK(hipMemcpy(h, env->v.counters, sizeof h, hipMemcpyDeviceToHost));
    h_out[0] = h[C_STEPS]; h_out[1] = h[C_FINISHED]; h_out[2] = h[C_P1WINS];
    h_out[3] = h[C_CAND_RAW]; h_out[4] = h[C_ROWS_EVAL]; h_out[5] = h[C_ERR];
    h_out[6] = h[C_FNODES]; h_out[7] = h[C_DNODES]; h_out[8] = h[C_KSTEPS]; h_out[9] = 0;
    return check_err_flags(env, h[C_ERR]);
}

int bgamd_env_reset_stats(bgamd_env *env, void *stream)
{
    ENV_GUARD(env);
    HIPCHK(hipMemsetAsync(env->v.counters, 0, C_COUNT * 8, (hipStream_t)stream));
    return BGAMD_OK;
}

int bgamd_env_try_move(bgamd_env *env, const int32_t *d_player, const int32_t *d_dice, const int32_t *d_origin,
                       const int32_t *d_dest, int32_t *d_err, void *stream)
{
    if (!env || !d_player || !d_dice || !d_origin || !d_dest || !d_err) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(env->device));
    hipLaunchKernelGGL(try_move_kernel, grid1(env->v.n, 128), dim3(128), 0, (hipStream_t)stream, env->v, d_player, d_dice,
                       d_origin, d_dest, d_err);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

int bgamd_env_legal_moves(bgamd_env *env, const int32_t *d_player, const int32_t *d_die, int32_t *d_n, int8_t *d_pairs,
                          void *stream)
{
    if (!env || !d_player || !d_die || !d_n || !d_pairs) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(env->device));
    hipLaunchKernelGGL(legal_moves_kernel, grid1(env->v.n, 128), dim3(128), 0, (hipStream_t)stream, env->v, d_player, d_die,
                       d_n, d_pairs);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

int64_t bgamd_env_unique_rows_info(bgamd_env *env, void *d_info, int64_t cap, void *stream)
{
    ENV_GUARD(env);
    hipStream_t s = (hipStream_t)stream;
    const long long bb = env->sv.b_base;                   // four arenas: the rows of the first, then those of the second, ...
    const int na = bb > 0 ? N_ARENAS : 1;
    unsigned long long cnt[N_ARENAS] = {0, 0, 0, 0};
    for (int k = 0; k < na; ++k) HIPCHK(hipMemcpyAsync(&cnt[k], &env->sv.tops[arena_counter(k)], 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const long long cap_a = bb > 0 ? bb : env->sv.cap_rows;
    long long total = 0, done = 0;
    for (int k = 0; k < na; ++k) {
        if ((long long)cnt[k] > cap_a) cnt[k] = (unsigned long long)cap_a;
        total += (long long)cnt[k];
        const long long m = (long long)cnt[k] < cap - done ? (long long)cnt[k] : cap - done;
        if (d_info && m > 0)
            HIPCHK(hipMemcpyAsync((uint2 *)d_info + done, env->sv.u_info + (long long)k * bb, (size_t)m * 8, hipMemcpyDeviceToDevice, s));
        done += m > 0 ? m : 0;
    }
    return (int64_t)total;
}

int bgamd_env_unique_rows_read(bgamd_env *env, int64_t first, int64_t n_rows, int32_t *d_states28, float *d_values, void *stream)
{
    if (!env || first < 0 || n_rows < 0 || first + n_rows > env->sv.cap_rows) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(env->device));
    if (n_rows == 0 || (!d_states28 && !d_values)) return BGAMD_OK;
    hipLaunchKernelGGL(rows_values_kernel, grid1(n_rows, 128), dim3(128), 0, (hipStream_t)stream, (const uint4 *)env->sv.u_rows,
                       (const float *)env->v.values, (long long)first, (long long)n_rows, d_states28, d_values,
                       (const unsigned long long *)env->sv.tops, (long long)env->sv.b_base);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

int bgamd_env_set_trajectory(bgamd_env *env, void *d_rows, int64_t max_plies)
{
    if (!env || (d_rows && max_plies <= 0)) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(env->device));
    env->v.traj = (uint4 *)d_rows;
    env->v.traj_plies = d_rows ? max_plies : 0;
    return BGAMD_OK;
}

int bgamd_env_set_trajectory_ring(bgamd_env *env, void *d_rows, int64_t ring_steps, uint16_t *d_end)
{
    if (!env || (d_rows && (ring_steps <= 0 || !d_end))) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(env->device));
    env->ring_rows = (uint4 *)d_rows;
    env->ring_end = d_rows ? (unsigned short *)d_end : nullptr;
    env->ring_steps = d_rows ? ring_steps : 0;
    env->traj_step = 0;
    return BGAMD_OK;
}

int64_t bgamd_env_trajectory_step(const bgamd_env *env) { return env ? env->traj_step : 0; }

int bgamd_env_get_progress(bgamd_env *env, int32_t *d_ply, int32_t *d_episode, void *stream)
{
    ENV_GUARD(env);
    hipStream_t s = (hipStream_t)stream;
    if (d_ply) HIPCHK(hipMemcpyAsync(d_ply, env->v.ply, (size_t)env->v.n * 4, hipMemcpyDeviceToDevice, s));
    if (d_episode) HIPCHK(hipMemcpyAsync(d_episode, env->v.episode, (size_t)env->v.n * 4, hipMemcpyDeviceToDevice, s));
    return BGAMD_OK;
}

int bgamd_encode_rows(const void *d_rows, int64_t n, float *d_out198, void *stream)
{
    if (!d_rows || !d_out198 || n < 0) return BGAMD_E_INVALID;
    if (n == 0) return BGAMD_OK;
    hipLaunchKernelGGL(encode_rows_kernel, grid1(n, 128), dim3(128), 0, (hipStream_t)stream, (const uint4 *)d_rows, (long long)n,
                       d_out198);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

int bgamd_encode(const int32_t *d_states28, const int32_t *d_turn, int64_t n, float *d_out198, void *stream)
{
    if (!d_states28 || !d_out198 || n < 0) return BGAMD_E_INVALID;
    if (n == 0) return BGAMD_OK;
    hipLaunchKernelGGL(encode_states_kernel, grid1(n, 128), dim3(128), 0, (hipStream_t)stream, d_states28, d_turn, (long long)n,
                       d_out198);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

int bgamd_evaluate(bgamd_env *env, const int32_t *d_states28, const int32_t *d_turn, int64_t n, int precision,
                   float *d_values, void *stream)
{
    return bgamd_evaluate_slot(env, 0, d_states28, d_turn, n, precision, d_values, stream);
}

int bgamd_evaluate_slot(bgamd_env *env, int slot, const int32_t *d_states28, const int32_t *d_turn, int64_t n, int precision,
                        float *d_values, void *stream)
{
    if (!env || slot < 0 || slot > 1 || !d_states28 || !d_values || n < 0 || n > env->v.cap) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(env->device));
    if (!env->has_weights[slot]) return BGAMD_E_NOWEIGHTS;
    if (n == 0) return BGAMD_OK;
    hipStream_t s = (hipStream_t)stream;
    // the candidate arena doubles as scratch for caller-provided states
    hipLaunchKernelGGL(pack_rows_kernel, grid1(n, 128), dim3(128), 0, s, d_states28, d_turn, (long long)n, env->v.rows,
                       &env->v.counters[C_ERR]);
    return launch_eval(env, slot, precision, nullptr, (long long)n, env->v.rows, d_values, nullptr, nullptr, s);
}

int bgamd_evaluate_incremental(bgamd_env *env, int slot, const int32_t *d_root_states28, const int32_t *d_root_turn,
                               int64_t n_roots, const int32_t *d_states28, const int32_t *d_root_index, int64_t n,
                               float *d_values, void *stream)
{
    if (!env || slot < 0 || slot > 1 || !d_root_states28 || !d_root_turn || !d_states28 || !d_root_index || !d_values ||
        n_roots <= 0 || n_roots > env->v.n || n < 0 || n > env->sv.cap_rows)
        return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(env->device));
    if (!env->has_weights[slot]) return BGAMD_E_NOWEIGHTS;
    if (n == 0) return BGAMD_OK;
    hipStream_t s = (hipStream_t)stream;
    StagedView &sv = env->sv;
    const float *b1 = env->d_w[slot] + N_HID * N_IN, *w2 = b1 + N_HID, *b2 = w2 + N_HID;
    // the env's root / afterstate arenas double as scratch, as the candidate arena does for bgamd_evaluate; the rows are written
    // linearly from 0: a later unique_rows_read must not remap them through the arena counters of an earlier greedy step
    sv.b_base = 0;
    hipLaunchKernelGGL(pack_rows_kernel, grid1(n_roots, 128), dim3(128), 0, s, d_root_states28, d_root_turn, (long long)n_roots,
                       sv.root_rows, &env->v.counters[C_ERR]);
    hipLaunchKernelGGL(pack_child_rows_kernel, grid1(n, 128), dim3(128), 0, s, d_states28, d_root_index, (long long)n,
                       (long long)n_roots, (const uint4 *)sv.root_rows, sv.u_rows, sv.u_info, &env->v.counters[C_ERR]);
    HIPCHK(hipMemsetAsync(sv.best, 0, (size_t)n_roots * 8, s));
#ifdef BGAMD_EXPERIMENTAL
    if (!env->root_resident) {
        long long blocks = ((n_roots + 31) / 32 + ROOT3_THREADS / 64 - 1) / (ROOT3_THREADS / 64);
        if (blocks > env->n_cu) blocks = env->n_cu;
        hipLaunchKernelGGL(root_hidden_bf16x3_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(ROOT3_THREADS), ROOT3_LDS_TOTAL, s,
                           (const uint4 *)sv.root_rows, (long long)n_roots, (const uint4 *)env->d_wl3[slot], (const uint2 *)env->d_lut, b1,
                           sv.root_hidden);
    } else
#endif
    {
        long long blocks = (n_roots + 31) / 32;
        if (blocks > 2ll * env->n_cu) blocks = 2ll * env->n_cu;
        hipLaunchKernelGGL(root_hidden_resident_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(ROOTR_THREADS), ROOTR_LDS_BYTES, s,
                           (const uint4 *)sv.root_rows, (long long)n_roots, (const uint4 *)(ROOT_F16X2 ? env->d_wr2[slot] : env->d_wl3[slot]), (const uint2 *)(ROOT_F16X2 ? env->d_lut16 : env->d_lut), b1,
                           sv.root_hidden);
    }
    long long dblocks = (n + DELTA_THREADS - 1) / DELTA_THREADS;
    dblocks = dblocks < 1 ? 1 : (dblocks > env->n_cu ? env->n_cu : dblocks);
    KTimer t(env, s, 1);
#ifdef BGAMD_EXPERIMENTAL
    if (env->mfma_delta && env->wm_ok[slot])
        hipLaunchKernelGGL(eval_rows_mdelta_kernel, dim3((unsigned)dblocks), dim3(MD_THREADS), MD_LDS_TOTAL, s, (const uint4 *)sv.u_rows,
                           (const unsigned long long *)nullptr, (long long)n, (unsigned long long *)nullptr, (const uint4 *)env->d_wm[slot],
                           w2, b2, (const uint4 *)sv.root_rows, (const float *)sv.root_hidden, d_values, (const uint2 *)sv.u_info, sv.best,
                           (unsigned long long *)nullptr, (unsigned long long *)nullptr, 0, &env->v.counters[C_ERR],
                           (unsigned long long)ERRF_DELTA);
    else
#endif
    hipLaunchKernelGGL(eval_rows_delta_kernel, dim3((unsigned)dblocks), dim3(DELTA_THREADS), DELTA_LDS_TOTAL, s, (const uint4 *)sv.u_rows,
                       (const unsigned long long *)nullptr, (long long)n, (unsigned long long *)nullptr, (const float4 *)env->d_wt[slot],
                       w2, b2, (const uint4 *)sv.root_rows, (const float *)sv.root_hidden, d_values, (const uint2 *)sv.u_info, sv.best,
                       (unsigned long long *)nullptr, (unsigned long long *)nullptr, 0, &env->v.counters[C_ERR],
                       (unsigned long long)ERRF_DELTA, (const unsigned long long *)nullptr, 0ll);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

int bgamd_env_time_kernels(bgamd_env *env, int enable)
{
    ENV_GUARD(env);
    if (!enable && env->timing) flush_events(env);
    env->timing = enable == 1 ? 0xFFu : ((unsigned)enable >> 8) & 0xFFu;   // 1 = every group, (mask << 8) = chosen groups
    env->timing_stride = ((unsigned)enable >> 20) & 0xFFu;                 // (stride << 20): every stride-th launch only
    if (env->timing_stride == 0) env->timing_stride = 1;
    for (unsigned &c : env->timing_seen) c = 0;
    if (env->timing)                                      // events are created here, not inside somebody's timed region
        while (env->ev.size() < 2 * 1024) {
            hipEvent_t x;
            HIPCHK(hipEventCreate(&x));
            env->ev.push_back(x);
            if (env->ev.size() % 2 == 0) env->ev_kind.push_back(0);
        }
    return BGAMD_OK;
}

const char *bgamd_build_flags(void)
{
#ifdef BGAMD_EXPERIMENTAL
    return "experimental";
#else
    return "default";
#endif
}

int bgamd_env_kernel_choice(bgamd_env *env, int32_t h_out[4])
{
    if (!env || !h_out) return BGAMD_E_INVALID;
    for (int i = 0; i < 3; ++i) h_out[i] = env->choice[i];
#ifdef BGAMD_EXPERIMENTAL
    h_out[3] = 1;
#else
    h_out[3] = 0;
#endif
    return BGAMD_OK;
}

int bgamd_env_kernel_times(bgamd_env *env, double h_ms[8], uint64_t h_launches[8])
{
    ENV_GUARD(env);
    const int rc = flush_events(env);
    if (rc) return rc;
    for (int i = 0; i < 8; ++i) {
        if (h_ms) h_ms[i] = env->t_ms[i];
        if (h_launches) h_launches[i] = env->t_n[i];
        env->t_ms[i] = 0; env->t_n[i] = 0;
    }
    return BGAMD_OK;
}

int bgamd_pack_rows(const int32_t *d_states28, const int32_t *d_turn, int64_t n, void *d_rows, void *stream)
{
    if (!d_states28 || !d_rows || n < 0) return BGAMD_E_INVALID;
    if (n == 0) return BGAMD_OK;
    hipLaunchKernelGGL(pack_rows_kernel, grid1(n, 128), dim3(128), 0, (hipStream_t)stream, d_states28, d_turn, (long long)n,
                       (uint4 *)d_rows, (unsigned long long *)nullptr);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

// ------------------------------------------- TD(lambda) learner -------------------------------------------
}  // extern "C"

struct bgamd_td {
    int device = 0;
    long long max_games = 0;
    TdView v{};
    bool has_weights = false, begun = false;
    int n_cu = 256;
    long long mfma_min = 24576;            // running games from which the forward pass uses the LDS-staged matrix-pipe kernel (BGAMD_TD_MFMA_MIN;
                                           //   measured equal to the direct one from there up, slower below: 153 vs 132 ms per round at 3 072 slots)
    bool fused = true;                     // ... with its epilogue in the same launch (BGAMD_TD_FUSED=0: two launches)
    long long direct_min = 512;            // ... from which it runs as one workgroup per 32-row tile, weights from the L2 (BGAMD_TD_DIRECT_MIN)
    long long nt_min = 8192;               // ... from which the whole-row trace pass uses nontemporal loads / stores (BGAMD_TD_NT_MIN)
    long long wide_min = 8192;             // running games from which the trace pass uses the whole-row workgroups (BGAMD_TD_WIDE_MIN)
    bool pipe = true;                      // mid-sized steps: the software-pipelined whole-row pass (BGAMD_TD_PIPE=0: td_trace_wide_kernel)
    bool fuse_step = true;                 // ... with the forward pass of the same slots in the same launch (BGAMD_TD_FUSE_STEP=0: two launches)
    long long fuse_min = 512;              // ... from this many running slots (measured: 512 slots 27 vs 31 us per step, 256 slots 28 vs 26) up to 16 per CU (BGAMD_TD_FUSE_MIN)
    int fuse_g = 0;                        // ... slots per workgroup of that launch: 0 = by step size (BGAMD_TD_FUSE_G = 1, 2, 4, 8, 16)
    long long slice_ng = 0;                // BGAMD_TD_NG: games per group of the slice kernel at mid-sized steps (0: as many groups as allowed)
    bool no_wide_even = false;             // BGAMD_TD_NO_WIDE_EVEN=1: mid-sized steps never take the whole-row kernels
    bool lazy = true;                      // lazily scaled traces (bg_learner.h); BGAMD_TD_LAZY=0: e <- λ e + ∇ every step
    double scale = 1.0;                    // c: stored trace = e / c, the same for every game of the replay
    bool stream_mode = false;              // bgamd_td_begin_stream: slots take game after game, steps are not bounded by the log length
    hipStream_t last_stream = nullptr;     // the stream the replay at hand is issued on: what the readers below wait for (a learner replaying on
                                           //   its own stream beside an env at play must not wait for the env: no device-wide synchronisation)
    // the ONE collective of a multi-rank training step issued by the library: an RCCL communicator of the learner's own (bgamd_td_comm_init)
    void *rccl = nullptr;                  // dlopen handle of librccl.so.1 (the one already in the process when there is one)
    void *comm = nullptr;                  // ncclComm_t
    int comm_rank = 0, comm_world = 0;
    float *d_upd = nullptr;                // [TD_P] the step's update, all-reduced in place
    // delayed update (bgamd_td_set_delay): the update of step t applied one step late, a training step = ONE launch (bg_learner.h)
    int delay = 0;
    float *theta2[2] = {nullptr, nullptr};      // [TD_LD] the weights a step reads / the next step's
    uint16_t *wl3_2[2] = {nullptr, nullptr};    // their bf16 x 3 planes
    float *partial2 = nullptr;                 // the second set of partial sums (+ zeroed tail): step t writes set t & 1 (set 0 = v.partial)
    bool timing = false;
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    double trace_ms = 0;
    uint64_t trace_launches = 0, trace_game_steps = 0;
};

namespace {
// device -> host on the replay's own stream (never the null stream: a learner beside an env at play waits for its own work only)
hipError_t td_read(bgamd_td *td, void *dst, const void *src, size_t bytes)
{
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, td->last_stream);
    return e != hipSuccess ? e : hipStreamSynchronize(td->last_stream);
}
int td_flush(bgamd_td *td)
{
    for (size_t i = 0; i < td->ev_used; ++i) {
        HIPCHK(hipEventSynchronize(td->ev[2 * i + 1]));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, td->ev[2 * i], td->ev[2 * i + 1]));
        td->trace_ms += ms;
    }
    td->ev_used = 0;
    return BGAMD_OK;
}
}  // namespace

extern "C" {

int bgamd_td_create(bgamd_td **out, int64_t max_games, int device)
{
    if (!out || max_games <= 0 || max_games > (1ll << 22)) return BGAMD_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return BGAMD_E_NODEVICE;
    HIPCHK(hipSetDevice(device));
    bgamd_td *td = new bgamd_td();
    td->device = device;
    td->max_games = max_games;
    TdView &v = td->v;
    auto fail = [&](int rc) { bgamd_td_destroy(td); return rc; };
#define TDALLOC(ptr, bytes)                                                         \
    do {                                                                            \
        hipError_t _e = hipMalloc((void **)&(ptr), (size_t)(bytes));                \
        if (_e != hipSuccess) {                                                     \
            g_hip_err = std::string("hipMalloc(" #ptr "): ") + hipGetErrorString(_e); \
            return fail(BGAMD_E_HIP);                                               \
        }                                                                           \
    } while (0)
    TDALLOC(v.theta, TD_LD * 4);
    TDALLOC(v.w1t, N_IN * N_HID * 4);
    TDALLOC(v.e, (size_t)max_games * TD_LD * 4);
    TDALLOC(v.fac, (size_t)max_games * TD_FLD * 4);
    TDALLOC(v.coef, (size_t)max_games * 4);
    TDALLOC(v.sq, (size_t)max_games * 8);
    TDALLOC(v.gmeta, (size_t)max_games * 16);
    TDALLOC(v.partial, ((size_t)TD_MAX_GROUPS * TD_LD + 64) * 4);          // + a zeroed line: the dummy source of the pipelined trace pass
    HIPCHK(hipMemset(v.partial + (size_t)TD_MAX_GROUPS * TD_LD, 0, 64 * 4));
    TDALLOC(v.amask, (size_t)max_games * TD_MASK_WORDS * 4);
    TDALLOC(v.anew, (size_t)max_games * TD_MASK_WORDS * 4);
    TDALLOC(v.act_cols, (size_t)max_games * 4);
    TDALLOC(v.wr_cols, (size_t)max_games * 4);
    TDALLOC(v.nupd, (size_t)max_games * 4);
    TDALLOC(v.qcur, (size_t)max_games * 4);
    TDALLOC(v.wl3, 3 * EVAL16_W_BYTES);
    TDALLOC(v.lut, EVAL16_LUT_BYTES);
    TDALLOC(v.hid, (size_t)max_games * 2 * N_HID * 4);
#undef TDALLOC
    {
        uint32_t lut[32];
        make_count_lut(lut);
        HIPCHK(hipMemcpy(v.lut, lut, EVAL16_LUT_BYTES, hipMemcpyHostToDevice));
        HIPCHK(hipMemset(v.wl3, 0, 3 * EVAL16_W_BYTES));          // the padding lanes of the tail K-step stay zero
        HIPCHK(hipFuncSetAttribute((const void *)traj_hidden_bf16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ROOT3_LDS_TOTAL));
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        td->n_cu = prop.multiProcessorCount;
        td->mfma_min = getenv("BGAMD_TD_MFMA_MIN") ? atoll(getenv("BGAMD_TD_MFMA_MIN")) : 24576;
    }
    v.dense = getenv("BGAMD_TD_DENSE") != nullptr ? 1 : 0;
    if (getenv("BGAMD_TD_WIDE_MIN")) td->wide_min = atoll(getenv("BGAMD_TD_WIDE_MIN"));
    if (getenv("BGAMD_TD_NT_MIN")) td->nt_min = atoll(getenv("BGAMD_TD_NT_MIN"));
    td->pipe = !(getenv("BGAMD_TD_PIPE") && atoi(getenv("BGAMD_TD_PIPE")) == 0);
    if (getenv("BGAMD_TD_NG")) td->slice_ng = atoll(getenv("BGAMD_TD_NG"));
    td->fuse_step = !(getenv("BGAMD_TD_FUSE_STEP") && atoi(getenv("BGAMD_TD_FUSE_STEP")) == 0);
    if (getenv("BGAMD_TD_FUSE_MIN")) td->fuse_min = atoll(getenv("BGAMD_TD_FUSE_MIN"));
    if (getenv("BGAMD_TD_FUSE_G")) {
        const int g = atoi(getenv("BGAMD_TD_FUSE_G"));
        td->fuse_g = (g == 1 || g == 2 || g == 4 || g == 8 || g == 16) ? g : 0;
    }
    td->no_wide_even = getenv("BGAMD_TD_NO_WIDE_EVEN") != nullptr && atoi(getenv("BGAMD_TD_NO_WIDE_EVEN")) != 0;
#ifdef BGAMD_EXPERIMENTAL
    td->fused = !(getenv("BGAMD_TD_FUSED") && atoi(getenv("BGAMD_TD_FUSED")) == 0);      // (the unfused matrix-pipe forward: experimental build only)
#endif
    if (getenv("BGAMD_TD_DIRECT_MIN")) td->direct_min = atoll(getenv("BGAMD_TD_DIRECT_MIN"));
    td->lazy = !(getenv("BGAMD_TD_LAZY") && atoi(getenv("BGAMD_TD_LAZY")) == 0);
    HIPCHK(hipMemset(v.act_cols, 0, (size_t)max_games * 4));
    HIPCHK(hipMemset(v.wr_cols, 0, (size_t)max_games * 4));
    HIPCHK(hipMemset(v.theta, 0, TD_LD * 4));
    HIPCHK(hipMemset(v.sq, 0, (size_t)max_games * 8));
    *out = td;
    return BGAMD_OK;
}

int bgamd_td_destroy(bgamd_td *td)
{
    if (!td) return BGAMD_OK;
    hipSetDevice(td->device);
    hipDeviceSynchronize();
    TdView &v = td->v;
    void *ptrs[] = {v.theta, v.w1t, v.e, v.fac, v.coef, v.sq, v.partial, v.gmeta, v.amask, v.anew, v.act_cols, v.wr_cols, v.nupd, v.qcur, v.wl3, v.lut, v.hid};
    for (void *p : ptrs) if (p) hipFree(p);
    for (hipEvent_t e : td->ev) hipEventDestroy(e);
    bgamd_td_comm_destroy(td);
    if (td->d_upd) hipFree(td->d_upd);
    for (int k = 0; k < 2; ++k) { if (td->theta2[k]) hipFree(td->theta2[k]); if (td->wl3_2[k]) hipFree(td->wl3_2[k]); }
    if (td->partial2) hipFree(td->partial2);
    delete td;
    return BGAMD_OK;
}

int bgamd_td_set_weights(bgamd_td *td, const float *d_theta, void *stream)
{
    if (!td || !d_theta) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(td->device));
    hipLaunchKernelGGL(td_apply_kernel, grid1(TD_P, 256), dim3(256), 0, (hipStream_t)stream, td->v, d_theta, 1);
    HIPCHK(hipGetLastError());
    td->has_weights = true;
    return BGAMD_OK;
}

int bgamd_td_get_weights(bgamd_td *td, float *d_theta, void *stream)
{
    if (!td || !d_theta) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(td->device));
    if (!td->has_weights) return BGAMD_E_NOWEIGHTS;
    HIPCHK(hipMemcpyAsync(d_theta, td->v.theta, (size_t)TD_P * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return BGAMD_OK;
}

int bgamd_td_begin(bgamd_td *td, const void *d_rows, int64_t T, int64_t n_lanes, const int32_t *d_order, int64_t n_games,
                   const int32_t *d_length, const uint8_t *d_p1_won, void *stream)
{
    if (!td || !d_rows || !d_order || !d_length || !d_p1_won || T <= 0 || n_lanes <= 0 || n_games < 0 ||
        n_games > td->max_games || n_games > n_lanes)
        return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(td->device));
    TdView &v = td->v;
    v.rows = (const uint4 *)d_rows;
    v.order = d_order;
    v.length = d_length;
    v.p1_won = d_p1_won;
    v.T = T; v.n_lanes = n_lanes; v.n_games = n_games;
    v.queue = nullptr; v.qoff = nullptr;
    v.game_lane = nullptr; v.game_start = nullptr; v.n_table = n_lanes;
    td->last_stream = (hipStream_t)stream;
    if (n_games > 0) {
        hipLaunchKernelGGL(td_gather_kernel, grid1(n_games, 256), dim3(256), 0, (hipStream_t)stream, v);
        HIPCHK(hipGetLastError());
    }
    td->begun = true;
    td->stream_mode = false;
    return BGAMD_OK;
}

// Host-only, plain C++ (csrc/bg_schedule.h: also compiled with -fsanitize=address,undefined by tests/test_sanitizers_cpu.py)
int bgamd_td_stream_schedule(const int32_t *h_length, int64_t n_lanes, int64_t n_slots, int32_t *h_queue, int32_t *h_queue_offsets,
                             int64_t *h_n_games, int64_t *h_n_steps)
{
    return bg::td_stream_schedule(h_length, n_lanes, n_slots, h_queue, h_queue_offsets, h_n_games, h_n_steps) ? BGAMD_E_INVALID : BGAMD_OK;
}

static int td_begin_stream_impl(bgamd_td *td, const void *d_rows, int64_t T, int64_t n_lanes, const int32_t *d_queue,
                                const int32_t *d_queue_offsets, int64_t n_slots, int64_t n_table, const int32_t *d_game_lane,
                                const int32_t *d_game_start, const int32_t *d_length, const uint8_t *d_p1_won, void *stream)
{
    if (!td || !d_rows || !d_queue || !d_queue_offsets || !d_length || !d_p1_won || T <= 0 || n_lanes <= 0 || n_table <= 0 || n_slots < 0 ||
        n_slots > td->max_games || T > 0x3FFFFFFFll)
        return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(td->device));
    TdView &v = td->v;
    v.rows = (const uint4 *)d_rows;
    v.order = nullptr;
    v.length = d_length;
    v.p1_won = d_p1_won;
    v.T = T; v.n_lanes = n_lanes; v.n_games = n_slots;
    v.queue = d_queue; v.qoff = d_queue_offsets;
    v.game_lane = d_game_lane; v.game_start = d_game_start; v.n_table = n_table;
    if (n_slots > 0) {
        hipLaunchKernelGGL(td_gather_stream_kernel, grid1(n_slots, 256), dim3(256), 0, (hipStream_t)stream, v);
        HIPCHK(hipGetLastError());
    }
    td->begun = true;
    td->stream_mode = true;
    td->last_stream = (hipStream_t)stream;
    return BGAMD_OK;
}

int bgamd_td_begin_stream(bgamd_td *td, const void *d_rows, int64_t T, int64_t n_lanes, const int32_t *d_queue,
                          const int32_t *d_queue_offsets, int64_t n_slots, const int32_t *d_length, const uint8_t *d_p1_won, void *stream)
{
    return td_begin_stream_impl(td, d_rows, T, n_lanes, d_queue, d_queue_offsets, n_slots, n_lanes, nullptr, nullptr, d_length, d_p1_won, stream);
}

int bgamd_td_begin_stream_games(bgamd_td *td, const void *d_rows, int64_t ring_steps, int64_t n_lanes, const int32_t *d_queue,
                                const int32_t *d_queue_offsets, int64_t n_slots, int64_t n_games, const int32_t *d_game_lane,
                                const int32_t *d_game_start, const int32_t *d_length, const uint8_t *d_p1_won, void *stream)
{
    if (!d_game_lane || !d_game_start) return BGAMD_E_INVALID;
    return td_begin_stream_impl(td, d_rows, ring_steps, n_lanes, d_queue, d_queue_offsets, n_slots, n_games, d_game_lane, d_game_start, d_length,
                                d_p1_won, stream);
}

int bgamd_td_step(bgamd_td *td, int64_t t, int64_t n_active, double alpha, float lambda, float *d_update, void *stream)
{
    if (!td || !td->begun || t < 0 || (t >= td->v.T && !td->stream_mode) || t > 0x7FFFFFF0ll || n_active < 0 || n_active > td->v.n_games)
        return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(td->device));
    if (!td->has_weights) return BGAMD_E_NOWEIGHTS;
    hipStream_t s = (hipStream_t)stream;
    td->last_stream = s;
    if (n_active == 0) {                 // nothing to add, but the caller's collective still needs a defined buffer
        if (d_update) HIPCHK(hipMemsetAsync(d_update, 0, (size_t)TD_P * 4, s));
        return BGAMD_OK;
    }
    // the scale of the stored traces (bg_learner.h): t = 0 writes ∇ at c = 1; afterwards c <- λ c, folded back in by an ordinary
    // pass when it leaves [2^-40, 2^40] (λ > 1 is the caller's business, but it must not overflow either)
    float emul = lambda, ginv = 1.0f, cmul = 1.0f;
    int full = 1;
    if (t == 0) td->scale = 1.0;
    else {
        const double c = (double)lambda * td->scale;
        if (td->lazy && c >= 0x1p-40 && c <= 0x1p40) { td->scale = c; emul = 1.0f; ginv = (float)(1.0 / c); cmul = (float)c; full = 0; }
        else { emul = (float)c; td->scale = 1.0; }
    }
    td->v.full_step = full;
    const TdView &v = td->v;
    // mid-sized steps whose trace pass takes the pipelined whole-row kernel: forward pass and trace pass in ONE launch (bg_learner.h)
    // slots per workgroup: the smallest of 1, 2, 4, 8, 16 that asks for no more workgroups than CUs (BGAMD_TD_FUSE_G pins it)
    const long long fuse_groups_max = td->n_cu < TD_MAX_GROUPS ? td->n_cu : TD_MAX_GROUPS;
    int fuse_g = td->fuse_g > 0 ? td->fuse_g : 1;
    if (td->fuse_g <= 0) while (fuse_g < 16 && (n_active + fuse_g - 1) / fuse_g > fuse_groups_max) fuse_g *= 2;
    // (BGAMD_TD_FUSED=0 / BGAMD_TD_DIRECT_MIN choose the forward kernel: a step that is to run the unfused or the VALU forward pass cannot take the
    //  launch that contains the fused one)
    const bool fused_step = td->fuse_step && td->pipe && td->fused && n_active >= td->direct_min && !td->no_wide_even && n_active >= td->fuse_min && n_active < td->mfma_min &&
                            n_active < td->nt_min && (n_active + fuse_g - 1) / fuse_g <= fuse_groups_max;
    if (fused_step) {
    } else if (n_active >= td->mfma_min) {
        // the [2 G x 198] · [198 x 128] product of the step on the matrix pipe (exact bf16 x 3 split of fc1.weight, fp32
        // accumulation: the env's root pass), then the epilogue per game
        const long long n_rows = 2 * n_active;
        long long blocks = ((n_rows + 31) / 32 + ROOT3_THREADS / 64 - 1) / (ROOT3_THREADS / 64);
        if (blocks > td->n_cu) blocks = td->n_cu;
        hipLaunchKernelGGL(traj_hidden_bf16x3_kernel, dim3((unsigned)blocks), dim3(ROOT3_THREADS), ROOT3_LDS_TOTAL, s, v.rows,
                           (const int4 *)v.gmeta, (long long)t, v.n_lanes, v.T, n_rows, (const uint4 *)v.wl3, (const uint2 *)v.lut,
                           (const float *)(v.theta + TD_OFF_B1), v.hid);
        hipLaunchKernelGGL(td_epilogue_wave_kernel, grid1(n_active, 4), dim3(256), 0, s, v, (long long)t, (long long)n_active, alpha);
    } else if (n_active >= td->direct_min && td->fused) {
        // mid-sized steps: the product and its epilogue in one launch (bg_learner.h)
        hipLaunchKernelGGL(td_forward_mfma_kernel, grid1(n_active, TD_FUSED_GAMES), dim3(ROOT3D_THREADS), 0, s, v, (long long)t, (long long)n_active, alpha);
#ifdef BGAMD_EXPERIMENTAL
    } else if (n_active >= td->direct_min) {
        // mid-sized steps: the same product, a workgroup per 32-row tile and the weight planes straight from the L2 (bg_eval.h)
        const long long n_rows = 2 * n_active;
        hipLaunchKernelGGL(traj_hidden_direct_kernel, dim3((unsigned)((n_rows + 31) / 32)), dim3(ROOT3D_THREADS), 0, s, v.rows,
                           (const int4 *)v.gmeta, (long long)t, v.n_lanes, v.T, n_rows, (const uint4 *)v.wl3, (const uint2 *)v.lut,
                           (const float *)(v.theta + TD_OFF_B1), v.hid);
        hipLaunchKernelGGL(td_epilogue_wave_kernel, grid1(n_active, 4), dim3(256), 0, s, v, (long long)t, (long long)n_active, alpha);
#endif
    } else if (n_active <= 8192)
        hipLaunchKernelGGL((td_forward_kernel<2, false>), grid1(n_active, 2), dim3(128), 0, s, v, (long long)t, (long long)n_active, alpha);
    else
        hipLaunchKernelGGL((td_forward_kernel<4, false>), grid1(n_active, 4), dim3(128), 0, s, v, (long long)t, (long long)n_active, alpha);
    // games per group: >= 4, and at most TD_MAX_GROUPS groups
    long long ng = (n_active + TD_MAX_GROUPS - 1) / TD_MAX_GROUPS;
#ifndef BG_TD_MIN_NG
#define BG_TD_MIN_NG 4
#endif
    if (ng < BG_TD_MIN_NG) ng = BG_TD_MIN_NG;
    if (td->slice_ng > 0 && n_active >= 512 && n_active < td->wide_min) {       // mid-sized steps on the slice kernel: fewer, larger groups
        ng = td->slice_ng;                                                       //   (fewer partial rows for the reduce kernel to read)
        if ((n_active + ng - 1) / ng > TD_MAX_GROUPS) ng = (n_active + TD_MAX_GROUPS - 1) / TD_MAX_GROUPS;
    }
    int n_groups = (int)((n_active + ng - 1) / ng);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (td->timing) {
        if (td->ev_used * 2 + 2 > td->ev.size()) {
            if (td->ev.size() >= 2 * 4096) { const int rc = td_flush(td); if (rc) return rc; }
            else for (int i = 0; i < 2; ++i) { hipEvent_t x; HIPCHK(hipEventCreate(&x)); td->ev.push_back(x); }
        }
        e0 = td->ev[2 * td->ev_used]; e1 = td->ev[2 * td->ev_used + 1];
        td->ev_used++;
        HIPCHK(hipEventRecord(e0, s));
    }
    // whole-row workgroups take chunks of TD_CHUNK games: below wide_min they pay only when the chunks divide evenly over the CUs
    // (a streamed replay through 2 048 or 4 096 slots: 143 vs 147 and 110 vs 118 ms per 65 536-game round)
    const long long per_wave_of_blocks = (long long)td->n_cu * TD_CHUNK;
    const bool wide_even = !td->no_wide_even && n_active >= per_wave_of_blocks && td->wide_min > per_wave_of_blocks &&
                           n_active * 20 >= ((n_active + per_wave_of_blocks - 1) / per_wave_of_blocks) * per_wave_of_blocks * 19;
    if (n_active >= td->wide_min || wide_even || fused_step) {
        // large rounds: a workgroup per whole trace row and strided chunks of games (bg_learner.h)
        n_groups = (int)((n_active + TD_CHUNK - 1) / TD_CHUNK);
        if (n_groups > td->n_cu) n_groups = td->n_cu;
        if (n_groups > TD_MAX_GROUPS) n_groups = TD_MAX_GROUPS;
        const bool nt = n_active >= td->nt_min;
        if (fused_step) {
            n_groups = (int)((n_active + fuse_g - 1) / fuse_g);
#define BG_FUSED_LAUNCH(G)                                                                                                                   \
    do {                                                                                                                                     \
        if (t == 0)                                                                                                                          \
            hipLaunchKernelGGL((td_step_fused_kernel<true, G>), dim3(n_groups), dim3(TD_WIDE_THREADS), 0, s, v, (long long)t,                  \
                               (long long)n_active, alpha, emul, ginv, cmul, 1);                                                             \
        else                                                                                                                                 \
            hipLaunchKernelGGL((td_step_fused_kernel<false, G>), dim3(n_groups), dim3(TD_WIDE_THREADS), 0, s, v, (long long)t,                 \
                               (long long)n_active, alpha, emul, ginv, cmul, full);                                                          \
    } while (0)
            switch (fuse_g) {
                case 1: BG_FUSED_LAUNCH(1); break;
                case 2: BG_FUSED_LAUNCH(2); break;
                case 4: BG_FUSED_LAUNCH(4); break;
                case 8: BG_FUSED_LAUNCH(8); break;
                default: BG_FUSED_LAUNCH(16); break;
            }
#undef BG_FUSED_LAUNCH
        } else if (td->pipe && !nt && n_active <= (long long)td->n_cu * TD_CHUNK * 4) {
            // mid-sized steps (at most a few chunks per CU): the software-pipelined whole-row pass (bg_learner.h)
            if (t == 0)
                hipLaunchKernelGGL((td_trace_pipe_kernel<true>), dim3(n_groups), dim3(TD_WIDE_THREADS), 0, s, v, (long long)n_active, emul, ginv, cmul, 1);
            else
                hipLaunchKernelGGL((td_trace_pipe_kernel<false>), dim3(n_groups), dim3(TD_WIDE_THREADS), 0, s, v, (long long)n_active, emul, ginv, cmul, full);
        } else if (t == 0)
            hipLaunchKernelGGL((td_trace_wide_kernel<true, true>), dim3(n_groups), dim3(TD_WIDE_THREADS), 0, s, v, (long long)n_active, emul, ginv, cmul, 1);
        else if (nt)
            hipLaunchKernelGGL((td_trace_wide_kernel<false, true>), dim3(n_groups), dim3(TD_WIDE_THREADS), 0, s, v, (long long)n_active, emul, ginv, cmul, full);
        else
            hipLaunchKernelGGL((td_trace_wide_kernel<false, false>), dim3(n_groups), dim3(TD_WIDE_THREADS), 0, s, v, (long long)n_active, emul, ginv, cmul, full);
    } else if (t == 0)
        hipLaunchKernelGGL(td_trace_kernel<true>, dim3(TD_SLICES, n_groups), dim3(TD_TRACE_THREADS), 0, s, v, (long long)n_active,
                           (int)ng, emul, ginv, cmul, 1);
    else
        hipLaunchKernelGGL(td_trace_kernel<false>, dim3(TD_SLICES, n_groups), dim3(TD_TRACE_THREADS), 0, s, v, (long long)n_active,
                           (int)ng, emul, ginv, cmul, full);
    if (td->timing) {
        HIPCHK(hipEventRecord(e1, s));
        td->trace_launches++;
        td->trace_game_steps += (uint64_t)n_active;
    }
    hipLaunchKernelGGL(td_reduce_kernel, grid1(TD_P, 64), dim3(256), 0, s, v, n_groups, d_update, d_update ? 0 : 1);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

int bgamd_td_apply(bgamd_td *td, const float *d_update, void *stream)
{
    if (!td || !d_update) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(td->device));
    if (!td->has_weights) return BGAMD_E_NOWEIGHTS;
    hipLaunchKernelGGL(td_apply_kernel, grid1(TD_P, 256), dim3(256), 0, (hipStream_t)stream, td->v, d_update, 0);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

}  // extern "C"

namespace {
// slots per workgroup of the fused launch for a step of n_active slots (0: the step does not take the fused launch)
int td_fuse_g_for(const bgamd_td *td, long long n_active)
{
    const long long fuse_groups_max = td->n_cu < TD_MAX_GROUPS ? td->n_cu : TD_MAX_GROUPS;
    int fuse_g = td->fuse_g > 0 ? td->fuse_g : 1;
    if (td->fuse_g <= 0) while (fuse_g < 16 && (n_active + fuse_g - 1) / fuse_g > fuse_groups_max) fuse_g *= 2;
    const bool ok = td->fuse_step && td->pipe && td->fused && n_active >= td->direct_min && !td->no_wide_even && n_active >= td->fuse_min &&
                    n_active < td->mfma_min && n_active < td->nt_min && (n_active + fuse_g - 1) / fuse_g <= fuse_groups_max;
    return ok ? fuse_g : 0;
}

// The delayed replay: every step ONE launch of td_step_fused_kernel<., ., DELAY = true> (bg_learner.h).  Step t reads weight buffer t & 1 and the
// partial sums of step t - 1, writes weight buffer (t + 1) & 1 and its own partial sums (set t & 1); one flush launch at the end applies the last
// update and leaves the result in the learner's canonical buffers.
int td_replay_delayed(bgamd_td *td, int64_t n_steps, long long k, int fuse_g, double alpha, float lambda, hipStream_t s)
{
    if (!td->theta2[0]) {
        for (int b = 0; b < 2; ++b) {
            HIPCHK(hipMalloc((void **)&td->theta2[b], (size_t)TD_LD * 4));
            HIPCHK(hipMemset(td->theta2[b], 0, (size_t)TD_LD * 4));
            HIPCHK(hipMalloc((void **)&td->wl3_2[b], 3 * EVAL16_W_BYTES));
            HIPCHK(hipMemset(td->wl3_2[b], 0, 3 * EVAL16_W_BYTES));
        }
        HIPCHK(hipMalloc((void **)&td->partial2, ((size_t)TD_MAX_GROUPS * TD_LD + 64) * 4));
        HIPCHK(hipMemset(td->partial2, 0, ((size_t)TD_MAX_GROUPS * TD_LD + 64) * 4));
    }
    td->last_stream = s;
    HIPCHK(hipMemcpyAsync(td->theta2[0], td->v.theta, (size_t)TD_LD * 4, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(td->wl3_2[0], td->v.wl3, 3 * EVAL16_W_BYTES, hipMemcpyDeviceToDevice, s));
    float *part[2] = {td->v.partial, td->partial2};
    const int n_groups = (int)((k + fuse_g - 1) / fuse_g);
    for (int64_t t = 0; t < n_steps; ++t) {
        float emul = lambda, ginv = 1.0f, cmul = 1.0f;                 // (the scale of the stored traces: as bgamd_td_step)
        int full = 1;
        if (t == 0) td->scale = 1.0;
        else {
            const double c = (double)lambda * td->scale;
            if (td->lazy && c >= 0x1p-40 && c <= 0x1p40) { td->scale = c; emul = 1.0f; ginv = (float)(1.0 / c); cmul = (float)c; full = 0; }
            else { emul = (float)c; td->scale = 1.0; }
        }
        TdView v = td->v;
        v.full_step = full;
        v.theta = td->theta2[t & 1];
        v.wl3 = td->wl3_2[t & 1];
        v.partial = part[t & 1];
        const float *pprev = part[(t + 1) & 1];
        const int n_prev = t == 0 ? 0 : n_groups;
        float *thn = td->theta2[(t + 1) & 1];
        uint16_t *wln = td->wl3_2[(t + 1) & 1];
#define BG_DELAY_LAUNCH(G)                                                                                                                   \
    do {                                                                                                                                     \
        if (t == 0)                                                                                                                          \
            hipLaunchKernelGGL((td_step_fused_kernel<true, G, true>), dim3(n_groups), dim3(TD_WIDE_THREADS), 0, s, v, (long long)t,           \
                               (long long)k, alpha, emul, ginv, cmul, 1, pprev, n_prev, thn, wln);                                           \
        else                                                                                                                                 \
            hipLaunchKernelGGL((td_step_fused_kernel<false, G, true>), dim3(n_groups), dim3(TD_WIDE_THREADS), 0, s, v, (long long)t,          \
                               (long long)k, alpha, emul, ginv, cmul, full, pprev, n_prev, thn, wln);                                        \
    } while (0)
        switch (fuse_g) {
            case 1: BG_DELAY_LAUNCH(1); break;
            case 2: BG_DELAY_LAUNCH(2); break;
            case 4: BG_DELAY_LAUNCH(4); break;
            case 8: BG_DELAY_LAUNCH(8); break;
            default: BG_DELAY_LAUNCH(16); break;
        }
#undef BG_DELAY_LAUNCH
    }
    // the update of the last step is still outstanding: theta = buffer n_steps & 1 + the sum of the last step's partial sums
    hipLaunchKernelGGL(td_delay_flush_kernel, dim3(TD_DELAY_SLICES), dim3(256), 0, s, (const float *)part[(n_steps + 1) & 1], n_steps ? n_groups : 0,
                       (const float *)td->theta2[n_steps & 1], td->v.theta, td->v.wl3, td->v.w1t);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}
}  // namespace

extern "C" {

int bgamd_td_set_delay(bgamd_td *td, int delay)
{
    if (!td || delay < 0 || delay > 1) return BGAMD_E_INVALID;
    td->delay = delay;
    return BGAMD_OK;
}

int bgamd_td_replay(bgamd_td *td, int64_t n_steps, const int64_t *h_n_active, double alpha, float lambda, void *stream)
{
    if (!td || !h_n_active || n_steps < 0 || (td->begun && !td->stream_mode && n_steps > td->v.T)) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(td->device));
    if (td->delay && td->begun && td->stream_mode && td->has_weights && n_steps > 0) {
        // the one-launch step exists for the steps that take the fused launch with at least TD_DELAY_SLICES workgroups: a streamed replay through a
        // constant number of slots in that range (512 ... 4 096 on 256 CUs).  Anything else replays exactly, update by update.
        bool same = true;
        for (int64_t t = 1; t < n_steps; ++t) same = same && h_n_active[t] == h_n_active[0];
        const long long k = h_n_active[0];
        const int g = same && k > 0 && k <= td->v.n_games ? td_fuse_g_for(td, k) : 0;
        if (g > 0 && (k + g - 1) / g >= TD_DELAY_SLICES) return td_replay_delayed(td, n_steps, k, g, alpha, lambda, (hipStream_t)stream);
    }
    for (int64_t t = 0; t < n_steps; ++t) {
        if (h_n_active[t] == 0) continue;
        const int rc = bgamd_td_step(td, t, h_n_active[t], alpha, lambda, nullptr, stream);
        if (rc != BGAMD_OK) return rc;
    }
    return BGAMD_OK;
}

int bgamd_td_stats(bgamd_td *td, double *h_sq_sum, int64_t *h_updates)
{
    ENV_GUARD(td);
    HIPCHK(hipSetDevice(td->device));
    HIPCHK(hipStreamSynchronize(td->last_stream));
    if (h_sq_sum) {
        std::vector<double> sq((size_t)td->v.n_games);
        if (!sq.empty()) HIPCHK(td_read(td, sq.data(), td->v.sq, sq.size() * 8));
        double acc = 0;
        for (double x : sq) acc += x;
        *h_sq_sum = acc;
    }
    if (h_updates) {
        std::vector<unsigned int> c((size_t)td->v.n_games);
        if (!c.empty()) HIPCHK(td_read(td, c.data(), td->v.nupd, c.size() * 4));
        int64_t tot = 0;
        for (unsigned int x : c) tot += x;
        *h_updates = tot;
    }
    return BGAMD_OK;
}

int bgamd_td_active_columns(bgamd_td *td, uint64_t *h_columns)
{
    if (!td || !h_columns) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(td->device));
    HIPCHK(hipStreamSynchronize(td->last_stream));
    std::vector<unsigned int> c((size_t)td->v.n_games);
    if (!c.empty()) HIPCHK(td_read(td, c.data(), td->v.act_cols, c.size() * 4));
    uint64_t tot = 0;
    for (unsigned int x : c) tot += x;
    *h_columns = tot;
    return BGAMD_OK;
}

int bgamd_td_slots(bgamd_td *td, int32_t *h_out)
{
    if (!td || !h_out) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(td->device));
    HIPCHK(hipStreamSynchronize(td->last_stream));
    const size_t n = (size_t)td->v.n_games;
    std::vector<int32_t> gm(4 * n), qc(n);
    std::vector<unsigned int> nu(n);
    if (n) {
        HIPCHK(td_read(td, gm.data(), td->v.gmeta, n * 16));
        HIPCHK(td_read(td, qc.data(), td->v.qcur, n * 4));
        HIPCHK(td_read(td, nu.data(), td->v.nupd, n * 4));
    }
    for (size_t i = 0; i < n; ++i) {
        for (int k = 0; k < 4; ++k) h_out[6 * i + k] = gm[4 * i + k];
        h_out[6 * i + 2] &= 1;                                 // (the word also carries the game's first log row)
        h_out[6 * i + 4] = td->stream_mode ? qc[i] : -1;
        h_out[6 * i + 5] = (int32_t)nu[i];
    }
    return BGAMD_OK;
}

int bgamd_td_written_columns(bgamd_td *td, uint64_t *h_columns)
{
    if (!td || !h_columns) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(td->device));
    HIPCHK(hipStreamSynchronize(td->last_stream));
    std::vector<unsigned int> c((size_t)td->v.n_games);
    if (!c.empty()) HIPCHK(td_read(td, c.data(), td->v.wr_cols, c.size() * 4));
    uint64_t tot = 0;
    for (unsigned int x : c) tot += x;
    *h_columns = tot;
    return BGAMD_OK;
}

int bgamd_td_time(bgamd_td *td, int enable)
{
    ENV_GUARD(td);
    if (!enable && td->timing) { const int rc = td_flush(td); if (rc) return rc; }
    td->timing = enable != 0;
    return BGAMD_OK;
}

int bgamd_td_times(bgamd_td *td, double *h_trace_ms, uint64_t *h_launches, uint64_t *h_game_steps)
{
    ENV_GUARD(td);
    const int rc = td_flush(td);
    if (rc) return rc;
    if (h_trace_ms) *h_trace_ms = td->trace_ms;
    if (h_launches) *h_launches = td->trace_launches;
    if (h_game_steps) *h_game_steps = td->trace_game_steps;
    td->trace_ms = 0; td->trace_launches = 0; td->trace_game_steps = 0;
    return BGAMD_OK;
}

// ---- the ONE collective of a multi-rank training step, issued by the library -----------------------------------------------
// one all-reduce (sum) of the 25 601-float update per training step, in place, on the compute stream.  Rounds 1-3 drove
// it from Python (bgamd_td_step -> torch.distributed.all_reduce -> bgamd_td_apply): 9.5 us of host time per step for the dispatch
// alone.  Here the learner owns an RCCL communicator and a step is three enqueues from C on ONE stream: the step's kernels with the
// update handed out, ncclAllReduce in place, the apply kernel.  RCCL is resolved at run time -- the librccl.so.1 that is already in the
// process (PyTorch brings its own) or the one of the ROCm install -- so libbgamd.so has no link-time dependency on it.
}  // extern "C"

namespace {
struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi *rccl_api()
{
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.h ? &api : nullptr;
    tried = true;
    const char *names[] = {getenv("BGAMD_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)                              // the copy that is already loaded wins: two RCCLs in one process is one too many
        if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char *n : names)
        if (n && !h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) { g_hip_err = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : ""); return nullptr; }
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))dlsym(h, "ncclAllReduce");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GetErrorString) {
        g_hip_err = "librccl: a symbol is missing";
        return nullptr;
    }
    api.h = h;
    return &api;
}

#define RCCLCHK(api, call)                                                                  \
    do {                                                                                    \
        ncclResult_t _r = (call);                                                           \
        if (_r != ncclSuccess) {                                                            \
            g_hip_err = std::string(#call) + ": " + (api)->GetErrorString(_r);              \
            return BGAMD_E_HIP;                                                             \
        }                                                                                   \
    } while (0)
}  // namespace

extern "C" {

int bgamd_td_comm_unique_id(uint8_t h_id[128])
{
    if (!h_id) return BGAMD_E_INVALID;
    RcclApi *api = rccl_api();
    if (!api) return BGAMD_E_HIP;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    RCCLCHK(api, api->GetUniqueId(&id));
    memcpy(h_id, &id, 128);
    return BGAMD_OK;
}

int bgamd_td_comm_init(bgamd_td *td, const uint8_t h_id[128], int rank, int world)
{
    if (!td || !h_id || world < 1 || rank < 0 || rank >= world) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(td->device));
    RcclApi *api = rccl_api();
    if (!api) return BGAMD_E_HIP;
    bgamd_td_comm_destroy(td);
    if (!td->d_upd) HIPCHK(hipMalloc((void **)&td->d_upd, (size_t)TD_P * 4));
    ncclUniqueId id;
    memcpy(&id, h_id, 128);
    ncclComm_t comm = nullptr;
    RCCLCHK(api, api->CommInitRank(&comm, world, id, rank));
    td->comm = comm; td->comm_rank = rank; td->comm_world = world;
    return BGAMD_OK;
}

int bgamd_td_comm_destroy(bgamd_td *td)
{
    if (!td) return BGAMD_E_INVALID;
    if (td->comm) {
        RcclApi *api = rccl_api();
        hipSetDevice(td->device);
        hipDeviceSynchronize();
        if (api) api->CommDestroy((ncclComm_t)td->comm);
        td->comm = nullptr; td->comm_world = 0;
    }
    return BGAMD_OK;
}

int bgamd_td_step_allreduce(bgamd_td *td, int64_t t, int64_t n_active, double alpha, float lambda, void *stream)
{
    if (!td || !td->comm || !td->d_upd) return BGAMD_E_INVALID;
    HIPCHK(hipSetDevice(td->device));
    hipStream_t s = (hipStream_t)stream;
    if (n_active > 0) {
        const int rc = bgamd_td_step(td, t, n_active, alpha, lambda, td->d_upd, stream);
        if (rc) return rc;
    } else {                                                 // this rank has nothing at step t: it still joins the collective
        if (!td->has_weights) return BGAMD_E_NOWEIGHTS;
        HIPCHK(hipMemsetAsync(td->d_upd, 0, (size_t)TD_P * 4, s));
        td->last_stream = s;
    }
    RcclApi *api = rccl_api();
    RCCLCHK(api, api->AllReduce(td->d_upd, td->d_upd, (size_t)TD_P, ncclFloat32, ncclSum, (ncclComm_t)td->comm, s));
    hipLaunchKernelGGL(td_apply_kernel, grid1(TD_P, 256), dim3(256), 0, s, td->v, (const float *)td->d_upd, 0);
    HIPCHK(hipGetLastError());
    return BGAMD_OK;
}

int bgamd_td_replay_allreduce(bgamd_td *td, int64_t n_steps, const int64_t *h_n_active, int64_t n_own_steps, double alpha, float lambda,
                              void *stream)
{
    if (!td || n_steps < 0 || n_own_steps < 0 || n_own_steps > n_steps || (n_own_steps > 0 && !h_n_active)) return BGAMD_E_INVALID;
    for (int64_t t = 0; t < n_steps; ++t) {
        const int rc = bgamd_td_step_allreduce(td, t, t < n_own_steps ? h_n_active[t] : 0, alpha, lambda, stream);
        if (rc != BGAMD_OK) return rc;
    }
    return BGAMD_OK;
}

}  // extern "C"
