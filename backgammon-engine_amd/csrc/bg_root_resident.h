// bg_root_resident.h -- the value net's ROOT PASS (one dense W1 x + b1 per game: the root term of the incremental evaluator, bg_eval.h)
// with the weights resident in registers: the default since round 3 (model.py:63-67's first layer on the position every game is in at
// the start of its turn, model.py:209).  Split out of bg_eval_dense16.h in round 4, when the kernels that lost their A/B moved behind
// -DBGAMD_EXPERIMENTAL: this one won its A/B.
#pragma once
#include "bg_eval.h"

namespace bg {

constexpr int D16_XBUF_U4 = K16_STEPS * 64;             // uint4 per tile: [13 K-steps][64 lanes]

// The root pass's arithmetic (round 4): W1 as f16 hi + f16 lo (22 mantissa bits; relayout_w1_f16x2_root), the features exact in f16
// (thermometer bits, halves, checker counts with the 1/15 folded into the weights), products exact, fp32 accumulation on
// v_mfma_f32_32x32x16_f16: TWO MFMAs per (K-step, tile) where the bf16 hi + mid + lo split (rounds 1-4, -DBG_ROOT_F16X2=0) takes three --
// the pass is MFMA-bound inside the boundary launch (2 waves per SIMD x 312 MFMAs).  Row-level values against the reference model:
// tests/test_gpu_round2.py::test_delta_kernel_rows_vs_reference_values, test_gpu_round3.py::test_bar_positions_rows_vs_reference_values.
#ifndef BG_ROOT_F16X2
#define BG_ROOT_F16X2 1
#endif
#if BG_ROOT_F16X2
constexpr int ROOT_PLANES = 2;
typedef f16x8 root_vec8;
#define BG_ROOT_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
__device__ __forceinline__ uint32_t root_half_bits(float x) { return (uint32_t)f16_bits((_Float16)x); }
constexpr uint32_t ROOT_ONE = 0x3C00u;
#else
constexpr int ROOT_PLANES = 3;
typedef bf16x8 root_vec8;
#define BG_ROOT_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0)
__device__ __forceinline__ uint32_t root_half_bits(float x) { return (uint32_t)f32_to_bf16_rne(x); }
constexpr uint32_t ROOT_ONE = 0x3F80u;
#endif
// the plane set and the count LUT the root pass takes: (f16 hi | lo, f16 LUT) or (bf16 hi | mid | lo, bf16 LUT)
constexpr bool ROOT_F16X2 = BG_ROOT_F16X2 != 0;

// W1 for the f16 x 2 root pass: the bf16 x 3 layout's positions (relayout_w1_bf16x3), two planes.
// -> 0, or -1 when a weight does not survive the split: not finite, |w| >= 65 504 (f16 hi would be inf and lo NaN: every value of the
// slot NaN), or hi + lo further from w than 2^-21 |w| + 2^-24 (the lo plane's quantum).  bgamd_env_load_weights_slot refuses such a
// table (BGAMD_E_WEIGHTS) instead of playing garbage in the mode advertised as reference parity.
inline int relayout_w1_f16x2_root(const float *w1, uint16_t *wl)
{
    static const int tail_map[8] = {192, 193, 194, 195, 196, 197, -1, -1};
    int rc = 0;
    for (int s = 0; s < K16_STEPS; ++s)
        for (int c = 0; c < 4; ++c)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    int f = 16 * s + 8 * (l >> 5) + j;
                    if (s == 12) f = (l >> 5) == 0 ? tail_map[j] : -1;
                    float w = f >= 0 && f < N_IN ? w1[(32 * c + (l & 31)) * N_IN + f] : 0.0f;
                    if (f >= 196) w = w / 15.0f;
                    const float aw = w < 0.0f ? -w : w;
                    if (!(aw < 65504.0f)) rc = -1;                         // (NaN fails the comparison too)
                    const _Float16 hi = (_Float16)w;
                    const _Float16 lo = (_Float16)(w - (float)hi);
                    const float res = w - ((float)hi + (float)lo);
                    if (!((res < 0.0f ? -res : res) <= aw * 4.76837158e-7f + 5.96046448e-8f)) rc = -1;
                    const size_t o = (((size_t)s * 4 + c) * 64 + l) * 8 + j;
                    wl[o] = f16_bits(hi);
                    wl[(size_t)ROOT3_PART_U4 * 8 + o] = f16_bits(lo);
                }
    return rc;
}

// ================================ the root pass with the weights resident in registers =====================================
// root_hidden_bf16x3_kernel (bg_eval.h, rounds 1-2; experimental build only now) stages the three bf16 planes of W1 (160 KB) through LDS in two K phases per workgroup
// and needs 21 us for 4 us worth of MFMAs.  Same product, same MFMA sequence per accumulator (K-step ascending, planes hi, mid,
// lo inside a step: the same bits), organised like eval_rows_d16_kernel (bg_eval_dense16.h): a workgroup of four waves, wave c keeps the three
// planes of ITS 32 hidden units in 156 registers for the whole launch, the 13 A operands of a 32-game tile are decoded once
// into LDS (a quarter of the K-steps per wave), rows arrive by LDS-DMA two tiles ahead, one block barrier per tile.
constexpr int ROOTR_THREADS = 256;
constexpr int ROOTR_LDS_BYTES = 2 * D16_XBUF_U4 * 16 + EVAL16_LUT_BYTES + 3 * 1024;

__global__ __launch_bounds__(ROOTR_THREADS, 2) void root_hidden_resident_kernel(
    const uint4 *__restrict__ rows, long long n_rows, const uint4 *__restrict__ wl3, const uint2 *__restrict__ lut,
    const float *__restrict__ b1, float *__restrict__ hidden)
{
    extern __shared__ uint4 sRR[];
    uint4 *sX = sRR;                                                      // [2][13][64]
    uint2 *sLut = reinterpret_cast<uint2 *>(sX + 2 * D16_XBUF_U4);
    uint4 *sRows = reinterpret_cast<uint4 *>(sLut + 16);                  // [3][64]
    if (threadIdx.x < 16) sLut[threadIdx.x] = lut[threadIdx.x];
    const int lane = threadIdx.x & 63, c = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    const long long n_tiles = (n_rows + 31) >> 5;
    union WF { uint4 u; root_vec8 v; };
    WF wf[ROOT_PLANES][K16_STEPS];
#pragma unroll
    for (int p = 0; p < ROOT_PLANES; ++p)
#pragma unroll
        for (int s = 0; s < K16_STEPS; ++s) wf[p][s].u = wl3[(size_t)p * ROOT3_PART_U4 + ((size_t)s * 4 + c) * 64 + lane];
    constexpr float NL2E = -1.44269504088896340736f;
    const float bb = b1[32 * c + r];
    __syncthreads();

    auto fetch = [&](long long tile, int slot) {
        if (c == 0 && tile * 32 + (lane >> 1) < n_rows)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rows + tile * 64 + lane),
                                             (__attribute__((address_space(3))) void *)(sRows + slot * 64), 16, 0, 0);
    };
    // the A operands wave c decodes for a tile: K-steps c, c + 4, c + 8, and the tail step for wave 0 (exactly root3_body's operands)
    auto stage = [&](long long tile, int buf, int slot) {
        uint32_t p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const bool row_ok = tile < n_tiles && tile * 32 + r < n_rows;
        if (row_ok) {
            const uint4 u0 = sRows[slot * 64 + 2 * r], u1 = sRows[slot * 64 + 2 * r + 1];
            p[0] = u0.x; p[1] = u0.y; p[2] = u0.z; p[3] = u0.w; p[4] = u1.x; p[5] = u1.y; p[6] = u1.z; p[7] = u1.w;
        }
        const Side sa{{p[0], p[1], p[2], p[3]}}, sb{{p[4], p[5], p[6], p[7]}};
        uint4 *dst = sX + buf * D16_XBUF_U4 + lane;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int pos = 2 * (c + 4 * q) + h + 1;
            const uint2 l0 = sLut[count_at(sa, pos)], l1 = sLut[count_at(sb, pos)];
            dst[(c + 4 * q) * 64] = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
        if (c == 0) {
            const int turn = (p[0] & TURN_BIT) ? 1 : 0;
            const uint32_t t0 = (turn == 0 && row_ok) ? ROOT_ONE : 0u, t1 = (turn == 0 || !row_ok) ? 0u : ROOT_ONE;
            const uint32_t bar1 = root_half_bits(0.5f * (float)count_at(sa, 0)), bar2 = root_half_bits(0.5f * (float)count_at(sb, 25));
            const uint32_t off1 = root_half_bits((float)count_at(sa, 25)), off2 = root_half_bits((float)count_at(sb, 0));
            dst[12 * 64] = h ? make_uint4(0, 0, 0, 0) : make_uint4(t0 | (t1 << 16), bar1 | (bar2 << 16), off1 | (off2 << 16), 0u);
        }
    };

    long long tile = blockIdx.x;
    fetch(tile, 0);
    fetch(tile + gridDim.x, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stage(tile, 0, 0);
    __syncthreads();
    int it = 0, slot = 0;
    for (; tile < n_tiles; tile += gridDim.x, ++it) {
        const int buf = it & 1;
        const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
        fetch(tile + 2 * (long long)gridDim.x, slot2);
        stage(tile + gridDim.x, buf ^ 1, slot1);
        floatx16 acc = {0};
        const uint4 *xp = sX + buf * D16_XBUF_U4 + lane;
#pragma unroll
        for (int s = 0; s < K16_STEPS; ++s) {
            WF x;
            x.u = xp[s * 64];
#pragma unroll
            for (int p = 0; p < ROOT_PLANES; ++p) acc = BG_ROOT_MFMA(x.v, wf[p][s].v, acc);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const long long orow = tile * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
            if (orow < n_rows) hidden[orow * N_HID + 32 * c + r] = NL2E * (acc[j] + bb);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // wave 0: the rows it requested have landed
        __syncthreads();
        slot = slot1;
    }
}


// ---- the same pass INSIDE the step's boundary launch (round 4) ----------------------------------------------------------------------
// As a launch of its own the root pass costs the step ~25 us at 65 536 lanes (same-box ablation, profiles/r04_ab_root_pass_in_boundary.txt:
// 0.1441 -> 0.1186 ms per step without it): ~9 us of work, the rest a kernel boundary on the critical path (or, forked onto a second stream,
// two events and the CUs it takes from the doubles plies and the leaf stage), and the value net reading 33 MB that were written just before.
// The workgroup of the boundary launch that applies step t and builds the roots of step t + 1 for its 256 games has those 256 root rows in
// hand: it goes on with THEIR root pass -- 8 tiles of 32 games, wave c = hidden units 32 c .. 32 c + 31, the same operands, the same MFMA
// sequence per accumulator (K-step ascending; planes hi, mid, lo inside a step) and the same epilogue as root_hidden_resident_kernel:
// the same bits.  Rows come from LDS (the roots half of the launch puts them there), no launch, no event, no round trip.
constexpr int BROOT_THREADS = 256;                           // = LANE_NT of the boundary launch: four waves
// games per workgroup of the boundary launch when it carries the root pass: 128 -- half the threads idle through the lane-per-game halves, 4 tiles per
// workgroup, two workgroups per CU at 65 536 lanes and one on EVERY CU at 32 768 (-DBG_BROOT_GPW=256: a game per thread, 8 tiles; same-box A/B:
// 65 536 lanes 0.1432 vs 0.1429 ms per step, 32 768 lanes 0.0945 vs 0.0994)
#ifndef BG_BROOT_GPW
#define BG_BROOT_GPW 128
#endif
constexpr int BROOT_GPW = BG_BROOT_GPW;
constexpr int BROOT_TILES = BROOT_GPW / 32;
// all 8 tiles' A operands are staged at once (one block barrier, then 8 x 39 MFMAs per wave back to back): a first version that staged
// tile by tile behind a barrier each -- root_hidden_resident_kernel's loop -- cost the launch 16.7 us (a chain of 8 x (LDS round trip,
// 39 dependent MFMAs, stores, barrier) on one wave per SIMD) where the MFMAs are 4.8
constexpr int BROOT_LDS_BYTES = BROOT_TILES * D16_XBUF_U4 * 16 + EVAL16_LUT_BYTES + BROOT_GPW * 32;

struct BRootWeights { uint4 w[ROOT_PLANES][K16_STEPS]; };    // wave c's planes of hidden units 32 c .. 32 c + 31: 104 registers (f16 x 2; bf16 x 3: 156)

// requested BEFORE the roots half of the launch (they do not depend on it): in flight while the roots scan and allocate
__device__ __forceinline__ void broot_load_weights(const uint4 *__restrict__ wl3, BRootWeights &wf)
{
    const int lane = threadIdx.x & 63, c = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#pragma unroll
    for (int p = 0; p < ROOT_PLANES; ++p)
#pragma unroll
#if defined(BG_ABL_STEP) && (BG_ABL_STEP & 8)
        for (int s = 0; s < K16_STEPS; ++s) wf.w[p][s] = make_uint4(lane + s, p, c, 0u);
#else
        for (int s = 0; s < K16_STEPS; ++s) wf.w[p][s] = wl3[(size_t)p * ROOT3_PART_U4 + ((size_t)s * 4 + c) * 64 + lane];
#endif
}

// The pass in two halves (round 5: the boundary launch collects its roots' list allocations between them).
// boundary_root_stage: every thread of the workgroup calls it with ITS game's root row (32 bytes; zero for a lane past the env); g0 = the
// workgroup's first game.  The rows go to LDS, every wave decodes its share of the tiles' A operands.  -> the workgroup's number of tiles.
// Ends WITHOUT a barrier: the caller's next barrier (roots_collect has one) or boundary_root_compute's own orders the operands.
__device__ __forceinline__ int boundary_root_stage(uint4 row0, uint4 row1, long long g0, long long n_games, const uint2 *__restrict__ lut)
{
    extern __shared__ uint4 sBR[];
    uint4 *sX = sBR;                                                      // [8 tiles][13][64]
    uint2 *sLut = reinterpret_cast<uint2 *>(sX + BROOT_TILES * D16_XBUF_U4);
    uint4 *sRows = reinterpret_cast<uint4 *>(sLut + 16);                  // [games per workgroup][2]
    if (threadIdx.x < 16) sLut[threadIdx.x] = lut[threadIdx.x];
    if (threadIdx.x < BROOT_GPW) {
        sRows[2 * threadIdx.x] = row0;
        sRows[2 * threadIdx.x + 1] = row1;
    }
    const int lane = threadIdx.x & 63, c = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    long long left = n_games - g0;
    const int n_tiles = left <= 0 ? 0 : (int)((left < BROOT_GPW ? left : BROOT_GPW) + 31) >> 5;     // workgroup-uniform
    __syncthreads();
    // the A operands wave c decodes for a tile: K-steps c, c + 4, c + 8, and the tail step for wave 0 (exactly root_hidden_resident_kernel's)
    for (int tile = 0; tile < n_tiles; ++tile) {
        uint32_t p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const bool row_ok = g0 + tile * 32 + r < n_games;
        if (row_ok) {
            const uint4 u0 = sRows[2 * (tile * 32 + r)], u1 = sRows[2 * (tile * 32 + r) + 1];
            p[0] = u0.x; p[1] = u0.y; p[2] = u0.z; p[3] = u0.w; p[4] = u1.x; p[5] = u1.y; p[6] = u1.z; p[7] = u1.w;
        }
        const Side sa{{p[0], p[1], p[2], p[3]}}, sb{{p[4], p[5], p[6], p[7]}};
        uint4 *dst = sX + tile * D16_XBUF_U4 + lane;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int pos = 2 * (c + 4 * q) + h + 1;
            const uint2 l0 = sLut[count_at(sa, pos)], l1 = sLut[count_at(sb, pos)];
            dst[(c + 4 * q) * 64] = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
        if (c == 0) {
            const int turn = (p[0] & TURN_BIT) ? 1 : 0;
            const uint32_t t0 = (turn == 0 && row_ok) ? ROOT_ONE : 0u, t1 = (turn == 0 || !row_ok) ? 0u : ROOT_ONE;
            const uint32_t bar1 = root_half_bits(0.5f * (float)count_at(sa, 0)), bar2 = root_half_bits(0.5f * (float)count_at(sb, 25));
            const uint32_t off1 = root_half_bits((float)count_at(sa, 25)), off2 = root_half_bits((float)count_at(sb, 0));
            dst[12 * 64] = h ? make_uint4(0, 0, 0, 0) : make_uint4(t0 | (t1 << 16), bar1 | (bar2 << 16), off1 | (off2 << 16), 0u);
        }
    }
    return n_tiles;
}

__device__ __forceinline__ void boundary_root_compute(const BRootWeights &wf, int n_tiles, long long g0, long long n_games,
                                                      const float *__restrict__ b1, float *__restrict__ hidden)
{
    extern __shared__ uint4 sBR[];
    uint4 *sX = sBR;
    const int lane = threadIdx.x & 63, c = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int r = lane & 31, h = lane >> 5;
    constexpr float NL2E = -1.44269504088896340736f;
    const float bb = b1[32 * c + r];
    __syncthreads();
    union WF { uint4 u; root_vec8 v; };
    // two tiles per iteration: two independent accumulator chains and twice the LDS reads in flight -- with ONE wave per SIMD nothing else
    // covers an MFMA chain's and an LDS read's latency (a tile at a time: 2.1 us per tile and wave against 1.25 in the stand-alone kernel,
    // whose two workgroups per CU cover each other).  Per accumulator the sequence is unchanged: K-step ascending, planes hi, mid, lo.
#if defined(BG_ABL_STEP) && (BG_ABL_STEP & 2)
    if (n_tiles > 100)
#endif
    for (int tile = 0; tile < n_tiles; tile += 2) {
        floatx16 acc0 = {0}, acc1 = {0};
        const uint4 *xp0 = sX + tile * D16_XBUF_U4 + lane;
        const uint4 *xp1 = sX + (tile + 1 < n_tiles ? tile + 1 : tile) * D16_XBUF_U4 + lane;
#pragma unroll
        for (int s = 0; s < K16_STEPS; ++s) {
            WF x0, x1;
            x0.u = xp0[s * 64];
            x1.u = xp1[s * 64];
#pragma unroll
            for (int p = 0; p < ROOT_PLANES; ++p) {
                WF wv;
                wv.u = wf.w[p][s];
                acc0 = BG_ROOT_MFMA(x0.v, wv.v, acc0);
                acc1 = BG_ROOT_MFMA(x1.v, wv.v, acc1);
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const long long orow = g0 + tile * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
#if defined(BG_ABL_STEP) && (BG_ABL_STEP & 4)
            if (acc0[j] == 1.2345f || acc1[j] == 1.2345f) hidden[orow * N_HID + 32 * c + r] = 0.0f;
#else
            if (orow < n_games) hidden[orow * N_HID + 32 * c + r] = NL2E * (acc0[j] + bb);
            if (tile + 1 < n_tiles && orow + 32 < n_games) hidden[(orow + 32) * N_HID + 32 * c + r] = NL2E * (acc1[j] + bb);
#endif
        }
    }
}

}  // namespace bg
