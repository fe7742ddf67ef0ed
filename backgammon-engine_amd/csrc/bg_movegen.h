// bg_movegen.h -- turn-sequence enumeration (legalTurnSequences / evaluateTurnSequences,
// cppsrc/game.cpp:134-222, collectDoubles :109-131) as a per-lane depth-first walk over the
// bit-plane board, one game per lane, leaves visited in the reference's order:
//
//   non-doubles: all {m1}/{m1,m2} with d1 first, then all with d2 first; no dedup, no max-dice
//                rule; a stuck position yields NO sequence                     
//   doubles:     pre-order DFS to depth 4, leaf when depth == 4 or no move is left; a stuck
//                root yields ONE empty sequence
//
// A visitor sees every leaf (afterstate planes + packed origins) and may stop the walk.
#pragma once
#include "bg_board.h"

namespace bg {

// packed sequence: origins 5 bits each (bits 0..19), length (bits 20..22),
// first die (bits 23..25), second die (bits 26..28).  move k uses the first die when k is even.
__host__ __device__ __forceinline__ uint32_t seq_pack(uint32_t origins, int len, int dA, int dB)
{
    return origins | ((uint32_t)len << 20) | ((uint32_t)dA << 23) | ((uint32_t)dB << 26);
}

// Walk all leaves.  V::leaf(own, opp, origins, len, dA, dB) returns true to stop.
template <class V>
__device__ __forceinline__ void walk_sequences(const Side &own0, const Side &opp0, int pl, int d1, int d2, V &v)
{
    const bool dbl = (d1 == d2);
    const int npass = dbl ? 1 : 2;
    for (int pass = 0; pass < npass; ++pass) {
        const int dA = pass ? d2 : d1;     // die of moves 0 and 2
        const int dB = pass ? d1 : d2;     // die of moves 1 and 3 (== dA for doubles)
        uint32_t m0 = legal_origins(own0, opp0, pl, dA);
        if (m0 == 0) {
            if (dbl && v.leaf(own0, opp0, 0u, 0, dA, dB)) return;
            continue;
        }
        while (m0) {
            const int o0 = __ffs(m0) - 1; m0 &= m0 - 1;
            Side own1 = own0, opp1 = opp0;
            apply_move(own1, opp1, pl, o0, dA);
            const uint32_t q1 = (uint32_t)o0;
            uint32_t m1 = legal_origins(own1, opp1, pl, dB);
            if (m1 == 0) {
                if (v.leaf(own1, opp1, q1, 1, dA, dB)) return;
                continue;
            }
            while (m1) {
                const int o1 = __ffs(m1) - 1; m1 &= m1 - 1;
                Side own2 = own1, opp2 = opp1;
                apply_move(own2, opp2, pl, o1, dB);
                const uint32_t q2 = q1 | ((uint32_t)o1 << 5);
                uint32_t m2 = dbl ? legal_origins(own2, opp2, pl, dA) : 0u;
                if (m2 == 0) {
                    if (v.leaf(own2, opp2, q2, 2, dA, dB)) return;
                    continue;
                }
                while (m2) {
                    const int o2 = __ffs(m2) - 1; m2 &= m2 - 1;
                    Side own3 = own2, opp3 = opp2;
                    apply_move(own3, opp3, pl, o2, dA);
                    const uint32_t q3 = q2 | ((uint32_t)o2 << 10);
                    uint32_t m3 = legal_origins(own3, opp3, pl, dA);
                    if (m3 == 0) {
                        if (v.leaf(own3, opp3, q3, 3, dA, dB)) return;
                        continue;
                    }
                    while (m3) {
                        const int o3 = __ffs(m3) - 1; m3 &= m3 - 1;
                        Side own4 = own3, opp4 = opp3;
                        apply_move(own4, opp4, pl, o3, dA);
                        if (v.leaf(own4, opp4, q3 | ((uint32_t)o3 << 15), 4, dA, dB)) return;
                    }
                }
            }
        }
    }
}

struct CountVisitor {
    uint32_t n = 0;
    __device__ __forceinline__ bool leaf(const Side &, const Side &, uint32_t, int, int, int) { ++n; return false; }
};

// keeps the k-th leaf (reference order)
struct SelectVisitor {
    uint32_t n = 0, k;
    Side own, opp;
    uint32_t seq = 0;
    __device__ explicit SelectVisitor(uint32_t k_) : k(k_) {}
    __device__ __forceinline__ bool leaf(const Side &a, const Side &b, uint32_t origins, int len, int dA, int dB)
    {
        if (n++ == k) { own = a; opp = b; seq = seq_pack(origins, len, dA, dB); return true; }
        return false;
    }
};

// writes every leaf as a 32-byte candidate row (P1 planes, P2 planes; mover's turn in p[0] bit 31)
struct EmitVisitor {
    uint4 *rows;       // this lane's first row
    uint32_t *seqs;    // may be null
    uint32_t n = 0;
    int pl;
    __device__ EmitVisitor(uint4 *r, uint32_t *s, int pl_) : rows(r), seqs(s), pl(pl_) {}
    __device__ __forceinline__ bool leaf(const Side &a, const Side &b, uint32_t origins, int len, int dA, int dB)
    {
        const Side &s1 = pl ? b : a;       // PLAYER1 planes
        const Side &s2 = pl ? a : b;       // PLAYER2 planes
        rows[2 * n]     = make_uint4(s1.b[0] | (pl ? TURN_BIT : 0u), s1.b[1], s1.b[2], s1.b[3]);
        rows[2 * n + 1] = make_uint4(s2.b[0], s2.b[1], s2.b[2], s2.b[3]);
        if (seqs) seqs[n] = seq_pack(origins, len, dA, dB);
        ++n;
        return false;
    }
};

}  // namespace bg
