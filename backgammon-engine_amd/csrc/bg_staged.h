// bg_staged.h -- load-balanced greedy step: the turn-sequence tree of every game is expanded one
// ply per kernel ("node" = game + move prefix), so no lane ever walks more than one node:
//
//   roots_kernel         lane per game : roll, first-ply moves      -> D1 (doubles) / F (leaf parents)
//   expand_kernel<PLY2/3> lane per node, then lane per child : doubles ply 2 and 3          -> D2 / F
//   expand_kernel<LEAF>  lane per F node, then lane per afterstate (<= 15 each)             -> rows + (game, key)
//                        (of two orders of the same commuting moves only the smaller-key one is expanded)
//   eval kernel     (bg_eval.h)   : value per unique row, atomicMax of (value, ~key) per game
//   apply_kernel    lane per game : decode the winning key, replay its <= 4 moves, terminal/reset
//
// Reference order is carried by the KEY instead of by position: key = pass | o0 | o1 | o2 | o3 | len
// compares exactly like the index into legalTurnSequences' list (cppsrc/game.cpp:134-191: d1-first
// block then d2-first block, ascending origins, DFS pre-order for doubles), so "first index wins
// ties" (model.py:212-213) is "smallest key wins": duplicates can be dropped anywhere, and a duplicate that is kept
// (identical row, identical value) cannot change the result.
#pragma once
#include "bg_board.h"

namespace bg {

// key layout (bit 31 is used for the mover's turn inside row info only)
constexpr int KEY_PASS_SHIFT = 23;
__host__ __device__ __forceinline__ int key_len(uint32_t k) { return (int)(k & 7u); }
__host__ __device__ __forceinline__ int key_pass(uint32_t k) { return (int)((k >> KEY_PASS_SHIFT) & 1u); }
__host__ __device__ __forceinline__ int key_origin(uint32_t k, int i) { return (int)((k >> (18 - 5 * i)) & 31u); }
__host__ __device__ __forceinline__ uint32_t key_child(uint32_t k, int o)
{
    const int len = key_len(k);
    return (k & ~7u) | ((uint32_t)o << (18 - 5 * len)) | (uint32_t)(len + 1);
}

struct Node { uint32_t game, key; };

// A value every lane of the wave holds alike (a total or a list position read back from LDS): v_readfirstlane tells the compiler so, and the value
// lives in SGPRs from there on.  Round 5: the expansion kernels are built AT their register cap (128 VGPRs, 4 waves per SIMD) and were spilling
// block-uniform totals and bases that sat in vector registers for whole phases.
__device__ __forceinline__ uint32_t uniform_u32(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long x)
{
    return ((unsigned long long)uniform_u32((uint32_t)(x >> 32)) << 32) | (unsigned long long)uniform_u32((uint32_t)x);
}

template <bool LDS_ONLY>
__device__ __forceinline__ void block_barrier()
{
    if (LDS_ONLY) barrier_lds();
    else __syncthreads();
}

// ---- workgroup exclusive scan (NW waves); returns the prefix, *total = sum over the block ----
// LEAD_SYNC = false: the caller guarantees a barrier since the last read of s_wave / s_slot
// LDS_ONLY: the barriers order LDS only (above)
template <int NW = 4, bool LEAD_SYNC = true, bool LDS_ONLY = false>
__device__ __forceinline__ uint32_t block_scan_256(uint32_t v, uint32_t *total, uint32_t *s_wave /*[NW]*/)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (LEAD_SYNC) block_barrier<LDS_ONLY>();        // s_wave may still be read by the previous scan
    if (lane == 63) s_wave[wv] = incl;
    block_barrier<LDS_ONLY>();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t x = s_wave[w];
        if (w < wv) base += x;
        tot += x;
    }
    *total = uniform_u32(tot);
    return base + incl - v;
}

// bump allocation of `total` entries for the whole block (thread 0 does the atomic)
template <bool LEAD_SYNC = true, bool LDS_ONLY = false>
__device__ __forceinline__ unsigned long long block_alloc(unsigned long long *top, uint32_t total, unsigned long long *s_slot)
{
    if (LEAD_SYNC) block_barrier<LDS_ONLY>();
    if (threadIdx.x == 0) *s_slot = total ? atomicAdd(top, (unsigned long long)total) : 0ull;
    block_barrier<LDS_ONLY>();
    return uniform_u64(*s_slot);
}

// two lists at once, both atomics in flight together: the block waits for one round trip instead of two.  (Inline: the
// compiler's atomic optimiser reads the first result back through v_readfirstlane before it issues the second.)  Both
// are issued unconditionally -- for callers whose totals are almost never zero.
__device__ __forceinline__ void block_alloc2(unsigned long long *topA, uint32_t totA, unsigned long long *topB, uint32_t totB,
                                             unsigned long long (&s_slot)[2], unsigned long long &baseA, unsigned long long &baseB)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a, b;
        asm volatile("global_atomic_add_x2 %0, %2, %3, off sc0\n\t"
                     "global_atomic_add_x2 %1, %4, %5, off sc0\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b)
                     : "v"(topA), "v"((unsigned long long)totA), "v"(topB), "v"((unsigned long long)totB)
                     : "memory");
        s_slot[0] = a; s_slot[1] = b;
    }
    __syncthreads();
    baseA = uniform_u64(s_slot[0]); baseB = uniform_u64(s_slot[1]);
}

// A RETURNING 64-bit add whose result the COMPILER tracks: its s_waitcnt goes where the result is first used, not where the atomic is issued, so
// the ~1.1 us round trip of a block allocation can run under work that does not need the base yet (round 5: the first successor of every lane of
// an expansion phase; the staging of the root pass in the boundary launch).  The opaque zero offset makes the pointer divergent for the compiler:
// its atomic optimiser then leaves the instruction alone (it would read the result back through v_readfirstlane on the spot).
__device__ __forceinline__ unsigned long long atomic_add_deferred(unsigned long long *top, unsigned long long v)
{
    uint32_t zero = 0;
    asm volatile("" : "+v"(zero));
    return __hip_atomic_fetch_add(top + zero, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct StagedView {
    Node *d1, *d2, *f;                    // node lists
    long long cap_d1, cap_d2, cap_f;
    Node *f2;                             // leaf parents of the doubles turns when one launch expands everything (expand_all_kernel)
    long long cap_f2;
    int shards;                           // 1, or LIST_SHARDS: the roots' lists (D1, F) as that many lists of cap / LIST_SHARDS entries with a counter each
                                          // (workgroup b of the roots appends to lists b % shards; workgroup w of a kind of expand_all_kernel reads
                                          // lists w % shards): 512 workgroups reach their allocation at the same moment, and atomics on one address
                                          // are served one after the other
    uint4 *u_rows;                        // unique rows
    uint2 *u_info;
    long long cap_rows;
    long long b_base;                     // > 0: the arena is FOUR of b_base rows each -- by the turn's kind (non-doubles, doubles) and by whether the
                                          // row's moves hit a blot: arena 2 * doubles + hit, counters T_U, T_UB, T_UB + stride, ... -- and the incremental
                                          // value net gives every workgroup the same number of tiles of each kind
    uint4 *root_rows;                     // [n] the position every game is in at the start of the step (mover's turn bit)
    float *root_hidden;                   // [n][128] W1 x_root + b1 (incremental evaluator, bg_eval.h)
    unsigned long long *best;             // [n] (ordered value bits << 32) | ~key ; 0 = no candidate
    unsigned long long *tops;             // [T_COUNT]
};
// The list counters live on their own 128-byte lines: the roots' two allocations (T_F and T_D1, from every workgroup at the same moment)
// and the stages' queue up at the memory side per LINE, not per address (BG_CTR_STRIDE=1: the four counters in one 32-byte group, as
// up to round 2; same-box A/B: doubles plies 0.0200 -> 0.0177 ms)
#ifndef BG_CTR_STRIDE
#define BG_CTR_STRIDE 16
#endif
enum { T_D1 = 0, T_D2 = BG_CTR_STRIDE, T_F = 2 * BG_CTR_STRIDE, T_U = 3 * BG_CTR_STRIDE, T_F2 = 4 * BG_CTR_STRIDE, T_UB = 5 * BG_CTR_STRIDE /* arenas 1, 2, 3 */,
       T_FS = 8 * BG_CTR_STRIDE /* F lists 1, 2, 3 */, T_DS = 11 * BG_CTR_STRIDE /* D1 lists 1, 2, 3 */, T_COUNT = 14 * BG_CTR_STRIDE };
constexpr int LIST_SHARDS = 4;
__host__ __device__ __forceinline__ int f_counter(int k) { return k == 0 ? T_F : T_FS + (k - 1) * BG_CTR_STRIDE; }
__host__ __device__ __forceinline__ int d1_counter(int k) { return k == 0 ? T_D1 : T_DS + (k - 1) * BG_CTR_STRIDE; }
constexpr int N_ARENAS = 4;
__host__ __device__ __forceinline__ int arena_counter(int k) { return k == 0 ? T_U : T_UB + (k - 1) * BG_CTR_STRIDE; }

}  // namespace bg
