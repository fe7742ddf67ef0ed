// bg_board.h -- bit-plane board and the single-checker rules, device side (gfx950).
//
// One game = 8 x uint32 "planes":  p[0..3] = PLAYER1 count bits 0..3, p[4..7] = PLAYER2 count
// bits 0..3.  Bit i (0..25) of plane k is bit k of the checker count at absolute position i:
//
//     PLAYER1:  0 = bar ("jail"),  1..24 = points,  25 = borne off ("freed")
//     PLAYER2: 25 = bar,           1..24 = points,   0 = borne off
//
// so that every move of the reference -- bar entry (origin 0 / 25), plain move, bear-off
// (destination clamped to 25 / 0, cppsrc/game.cpp:89-97) -- is "count[origin]--, count[dest]++"
// on the mover's planes, with destination = origin +/- die.  Counts are <= 15 (a nibble).
// Bits 26..31 of the planes are spare; candidate rows carry the mover's turn in p[0] bit 31.
//
// The rules below restate cppsrc/game.cpp (isValidOrigin :416-457, isValidDestination :459-485,
// canFreePiece :488-557, legalMoves :80-105, tryMove :573-663, over :388-407) as mask algebra;
// the per-origin ascending order of legalMoves is recovered by popping the lowest set bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bg {

constexpr uint32_t PTS = 0x01FFFFFEu;      // bits 1..24
constexpr uint32_t POS = 0x03FFFFFFu;      // bits 0..25
constexpr uint32_t TURN_BIT = 0x80000000u; // in p[0] of candidate rows

struct Side { uint32_t b[4]; };            // one player's four count planes

__host__ __device__ constexpr uint32_t plane_of(const int (&cnt)[26], int k)
{
    uint32_t m = 0;
    for (int i = 0; i < 26; ++i) m |= (uint32_t)((cnt[i] >> k) & 1) << i;
    return m;
}

// start position, cppsrc/game.cpp:251
//   P1: 2 on pt 1, 5 on 12, 3 on 17, 5 on 19;   P2: 5 on 6, 3 on 8, 5 on 13, 2 on 24
struct StartPlanes { uint32_t p[8]; };
__host__ __device__ constexpr StartPlanes start_planes()
{
    int a[26] = {0}, b[26] = {0};
    a[1] = 2; a[12] = 5; a[17] = 3; a[19] = 5;
    b[6] = 5; b[8] = 3; b[13] = 5; b[24] = 2;
    return StartPlanes{{plane_of(a, 0), plane_of(a, 1), plane_of(a, 2), plane_of(a, 3),
                        plane_of(b, 0), plane_of(b, 1), plane_of(b, 2), plane_of(b, 3)}};
}

__device__ __forceinline__ uint32_t any_of(const Side &s) { return s.b[0] | s.b[1] | s.b[2] | s.b[3]; }
__device__ __forceinline__ uint32_t ge2_of(const Side &s) { return s.b[1] | s.b[2] | s.b[3]; }

__device__ __forceinline__ int count_at(const Side &s, int pos)
{
    return (int)(((s.b[0] >> pos) & 1u) | (((s.b[1] >> pos) & 1u) << 1) |
                 (((s.b[2] >> pos) & 1u) << 2) | (((s.b[3] >> pos) & 1u) << 3));
}

// count[pos] += 1 for the single-bit (or empty) mask m: ripple carry over the planes
__device__ __forceinline__ void inc_at(Side &s, uint32_t m)
{
    uint32_t c = s.b[0] & m; s.b[0] ^= m;
    uint32_t c2 = s.b[1] & c; s.b[1] ^= c;
    uint32_t c3 = s.b[2] & c2; s.b[2] ^= c2;
    s.b[3] ^= c3;
}
// count[pos] -= 1 (count must be > 0): ripple borrow
__device__ __forceinline__ void dec_at(Side &s, uint32_t m)
{
    s.b[0] ^= m; uint32_t br = s.b[0] & m;
    s.b[1] ^= br; br &= s.b[1];
    s.b[2] ^= br; br &= s.b[2];
    s.b[3] ^= br;
}

// legalMoves(player, die) as a mask of origins (absolute positions 0..25).
//   own/opp are the mover's / opponent's planes, pl = 0 (PLAYER1, moves up) or 1 (PLAYER2, down).
__device__ __forceinline__ uint32_t legal_origins(const Side &own, const Side &opp, int pl, int d)
{
    const uint32_t own_any = any_of(own);
    const uint32_t blk = ge2_of(opp) & PTS;              // >= 2 opposing checkers: game.cpp:472-484
    const uint32_t barbit = pl ? (1u << 25) : 1u;
    uint32_t occ = own_any & (PTS | barbit);
    occ = (occ & barbit) ? barbit : occ;                 // on the bar: only the bar may move (:416-441)

    // plain moves and bar entry: destination stays on the board
    const uint32_t hit_blk = pl ? (blk << d) : (blk >> d);
    const uint32_t range = pl ? ((POS << (d + 1)) & POS) : ((1u << (25 - d)) - 1u);
    uint32_t mask = occ & ~hit_blk & range;

    // bear-off (destination clamps to 25 / 0): canFreePiece, game.cpp:488-557
    const uint32_t outside = pl ? 0x03FFFF80u : 0x0007FFFFu;   // P2: bits 7..25, P1: bits 0..18
    if ((occ & outside) == 0 && occ != 0) {
        const int exact_pos = pl ? d : 25 - d;
        uint32_t bear = occ & (1u << exact_pos);
        if (pl == 0) {
            // overrun: legal only from the HIGHEST occupied P1 point (:526-537)
            const int hi = 31 - __clz(occ);
            if (hi > exact_pos) bear |= 1u << hi;
        } else {
            // overrun: no checker of EITHER colour on points origin+1..7 (:542-553)
            const uint32_t any7 = (own_any | any_of(opp)) & 0xFEu;
            const int hi = 31 - __clz(any7);             // any7 != 0 because occ != 0 lies in 1..6
            if (hi < exact_pos) bear |= occ & (1u << hi);
        }
        mask |= bear;
    }
    return mask;
}

// tryMove for a move already known legal (game.cpp:623-660): leave origin, hit a blot, land / bear off.
__device__ __forceinline__ void apply_move(Side &own, Side &opp, int pl, int o, int d)
{
    int dest = pl ? (o - d) : (o + d);
    dest = dest < 0 ? 0 : (dest > 25 ? 25 : dest);
    const uint32_t md = 1u << dest;
    dec_at(own, 1u << o);
    inc_at(own, md);
    const uint32_t hm = md & PTS & opp.b[0] & ~ge2_of(opp);    // exactly one opposing checker
    opp.b[0] ^= hm;
    inc_at(opp, hm ? (pl ? 1u : (1u << 25)) : 0u);             // opponent's bar
}

// over(), game.cpp:388-407: P1 first.  returns 0 = not over, 1 = P1 won, 2 = P2 won
__device__ __forceinline__ int over_code(const uint32_t (&p)[8])
{
    const uint32_t f1 = p[0] & p[1] & p[2] & p[3] & (1u << 25);   // P1 off == 15
    const uint32_t f2 = p[4] & p[5] & p[6] & p[7] & 1u;           // P2 off == 15
    return f1 ? 1 : (f2 ? 2 : 0);
}

// ---- int32[28] <-> planes ([board24, bar1, bar2, off1, off2], game.hpp:19-23) ------------
__device__ __forceinline__ void planes_from_state28(const int32_t *s, uint32_t (&p)[8], int *bad)
{
    int a[26], b[26];
#pragma unroll
    for (int i = 0; i < 26; ++i) { a[i] = 0; b[i] = 0; }
    int err = 0;
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        const int v = s[i];
        if (v > 15 || v < -15) err = 1;
        if (v > 0) a[i + 1] = v & 15; else b[i + 1] = (-v) & 15;
    }
    if ((unsigned)s[24] > 15u || (unsigned)s[25] > 15u || (unsigned)s[26] > 15u || (unsigned)s[27] > 15u) err = 1;
    a[0] = s[24] & 15; b[25] = s[25] & 15; a[25] = s[26] & 15; b[0] = s[27] & 15;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t ma = 0, mb = 0;
#pragma unroll
        for (int i = 0; i < 26; ++i) {
            ma |= (uint32_t)((a[i] >> k) & 1) << i;
            mb |= (uint32_t)((b[i] >> k) & 1) << i;
        }
        p[k] = ma; p[4 + k] = mb;
    }
    if (bad) *bad = err;
}

__device__ __forceinline__ void state28_from_planes(const uint32_t (&p)[8], int32_t *s)
{
    Side a{{p[0], p[1], p[2], p[3]}}, b{{p[4], p[5], p[6], p[7]}};
#pragma unroll
    for (int i = 0; i < 24; ++i)
        s[i] = count_at(a, i + 1) - count_at(b, i + 1);
    s[24] = count_at(a, 0);
    s[25] = count_at(b, 25);
    s[26] = count_at(a, 25);
    s[27] = count_at(b, 0);
}

// ---- Philox4x32-10, key = seed, counter = (game_id_lo, game_id_hi, ply, stream) ----------
// (same stream definition as oracle/bg_oracle.c, which holds the known answers)
struct U4 { uint32_t x, y, z, w; };
__host__ __device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                     uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}
enum { STREAM_TURN = 0, STREAM_OPENING = 1 };
__host__ __device__ __forceinline__ int die_from_u32(uint32_t u) { return 1 + (int)(((uint64_t)u * 6u) >> 32); }

// opening protocol of play_game, pysrc/TD(λ) model/train.py:89-97
__host__ __device__ __forceinline__ int opening_turn(uint64_t seed, uint64_t gid)
{
    for (uint32_t attempt = 0;; ++attempt) {
        const U4 x = philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), attempt, STREAM_OPENING,
                                   (uint32_t)seed, (uint32_t)(seed >> 32));
        const int s1 = die_from_u32(x.x) + die_from_u32(x.y);
        const int s2 = die_from_u32(x.z) + die_from_u32(x.w);
        if (s1 != s2) return s1 > s2 ? 0 : 1;
    }
}

// A block barrier that orders LDS traffic only: __syncthreads() is a workgroup-scope fence over ALL address spaces, i.e. every wave
// first waits for its global stores to be acknowledged (s_waitcnt vmcnt(0)).  Where a barrier only orders LDS records -- the expansion phases between
// writing rows that nobody in the launch reads back, the learner's forward pass with its prefetches in flight -- this one lets global traffic run on.
__device__ __forceinline__ void barrier_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

}  // namespace bg
