// bg_eval_dense16.h -- the dense f16 hi + lo value net with the WEIGHTS RESIDENT IN REGISTERS (round 3).
//
// Same arithmetic as eval_rows_f16x2_kernel (bg_eval.h): W1 = f16 hi + f16 lo (22 significand bits), every encoder feature
// exact in f16 (off/15 fed as its own hi + lo pair), exact products, fp32 accumulation on v_mfma_f32_32x32x16_f16, then the
// hidden sigmoid, the W2 dot and the output sigmoid (model.py:63-67, encoder model.py:111-144) and the per-game arg-max
// (model.py:209-213).  What changes is where the operands live:
//
//   * a workgroup is FOUR waves; wave c owns the hidden units 32 c .. 32 c + 31 and keeps ITS slice of W1 -- 13 K-steps x
//     (hi, lo) x 4 VGPRs = 104 registers per lane -- for the whole launch.  The round-1 kernel re-read all of W1 from LDS for
//     every 32-row tile (104 KB per tile, 3.7 GB of LDS reads per launch) and ran at 2.3 x its MFMA time; here the weights
//     are read once per wave.
//   * orientation: the weights are the A operand (rows of the product = hidden units), the encoded afterstates the B
//     operand (columns = rows of the tile).  In the C layout the ROW OF THE TILE is on the lane and the 16 registers are
//     hidden units: the W2 dot is an in-lane FMA chain -- no cross-lane reduction but one add of the two lane halves -- and
//     the four waves' partial sums meet in LDS.
//   * the B operand (the 198 features of a row, never materialised anywhere else) is decoded ONCE per tile: each wave
//     decodes a quarter of the K-steps into LDS, all four read every K-step back (13 KB per tile, double-buffered, one
//     block barrier per tile).
//   * -log2(e) is folded into W1 and b1; b1 rides in a spare K slot of the tail step as the weight of a constant-1 feature:
//     the epilogue is exp2, + 1, rcp, fma per hidden unit.
//   * a K-step (two board points, both sides) on which NO row of the tile has a checker multiplies by zero: skipped (exact:
//     its terms are +0).  Executed steps stay in ascending order, so a row's value depends on that row alone.
#pragma once
#include "bg_eval.h"
#include "bg_root_resident.h"      // D16_XBUF_U4 and the root pass organised like this kernel (the default root pass: always built)

namespace bg {

constexpr int D16_THREADS = 256;                        // 4 waves: one per 32-unit column tile
#ifndef BG_D16_WAVES
#define BG_D16_WAVES 3                                   // waves per SIMD the register allocation aims at (= workgroups per CU)
#endif
constexpr int D16_LDS_BYTES = 2 * D16_XBUF_U4 * 16 + 2 * 4 * 32 * 4 + 2 * EVAL16_LUT_BYTES + N_HID * 4 + 3 * 1024;     // 26 624 + 1 024 + 256 + 512 + 3 072 (three tiles of rows)

// Wl[part][s][c][l][j] = f16 part of  -log2(e) · W1[32c + (l&31)][feature(s, l>>5, j)];  tail step (s = 12): the h = 0 lanes
// carry [turn0, turn1, bar1/2, bar2/2, off1.hi, off1.lo, off2.hi, off2.lo] (as relayout_w1_f16x2), the h = 1 lanes' first
// slot the constant-1 feature whose weight is -log2(e) · b1
inline void relayout_w1_d16(const float *w /*25601: W1 | b1 | W2 | b2*/, uint16_t *wl)
{
    static const int tail_map[8] = {192, 193, 194, 195, 196, 196, 197, 197};
    const float NL2E = -1.44269504088896340736f;
    const float *w1 = w, *b1 = w + N_HID * N_IN;
    for (int s = 0; s < K16_STEPS; ++s)
        for (int c = 0; c < 4; ++c)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int u = 32 * c + (l & 31), hh = l >> 5;
                    float x = 0.0f;
                    if (s < 12) x = w1[u * N_IN + 16 * s + 8 * hh + j];
                    else if (hh == 0) x = w1[u * N_IN + tail_map[j]];
                    else if (j == 0) x = b1[u];
                    x *= NL2E;
                    const _Float16 hi = (_Float16)x;
                    const _Float16 lo = (_Float16)(x - (float)hi);
                    const size_t o = (((size_t)s * 4 + c) * 64 + l) * 8 + j;
                    wl[o] = f16_bits(hi);
                    wl[(size_t)K16_STEPS * 4 * 64 * 8 + o] = f16_bits(lo);
                }
}

template <int N_PLANES>   // 2: f16 hi + lo (fp32-grade); 1: hi only (a speed mode, like bf16)
__global__ __launch_bounds__(D16_THREADS, BG_D16_WAVES) void eval_rows_d16_kernel(
    const uint4 *__restrict__ rows, const unsigned long long *__restrict__ n_rows_ptr, long long n_rows_imm,
    unsigned long long *__restrict__ rows_eval_counter, const uint4 *__restrict__ wl, const uint2 *__restrict__ lut,
    const float *__restrict__ w2, const float *__restrict__ b2p, float *__restrict__ values, const uint2 *__restrict__ info,
    unsigned long long *__restrict__ best, unsigned long long *__restrict__ ksteps_counter,
    unsigned long long *__restrict__ zero_words, int n_zero_words)
{
    if (zero_words && blockIdx.x == 0 && (int)threadIdx.x < n_zero_words) zero_words[threadIdx.x] = 0ull;
    extern __shared__ uint4 sD16[];
    uint4 *sX = sD16;                                                     // [2][13][64]
    float *sRed = reinterpret_cast<float *>(sX + 2 * D16_XBUF_U4);        // [2][4][32]
    uint2 *sLut = reinterpret_cast<uint2 *>(sRed + 2 * 4 * 32);
    uint2 *sLutT = sLut + 16;                                             // count k -> (f16 of k / 2, f16 hi | lo << 16 of k / 15)
    if (threadIdx.x < 16) {
        sLut[threadIdx.x] = lut[threadIdx.x];
        const float o = (float)threadIdx.x / 15.0f;
        const _Float16 oh = (_Float16)o;
        sLutT[threadIdx.x] = make_uint2((uint32_t)f16_bits((_Float16)(0.5f * (float)threadIdx.x)),
                                        (uint32_t)f16_bits(oh) | ((uint32_t)f16_bits((_Float16)(o - (float)oh)) << 16));
    }
    const int lane = threadIdx.x & 63, c = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // (uniform, and known to be)
    const int r = lane & 31, h = lane >> 5;

    long long n_rows = n_rows_imm;
    if (n_rows_ptr) { const long long cc = (long long)*n_rows_ptr; n_rows = cc < n_rows_imm ? cc : n_rows_imm; }
    if (rows_eval_counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(rows_eval_counter, (unsigned long long)n_rows);
    const long long n_tiles = (n_rows + 31) >> 5;

    // this wave's slice of W1, for the whole launch
    union WF { uint4 u; f16x8 v; };
    WF wf[N_PLANES][K16_STEPS];
#pragma unroll
    for (int p = 0; p < N_PLANES; ++p)
#pragma unroll
        for (int s = 0; s < K16_STEPS; ++s) wf[p][s].u = wl[((size_t)(p * K16_STEPS + s) * 4 + c) * 64 + lane];
    // W2 in the order the accumulator registers hold the hidden units (register j of lane half h = unit 32 c + (j & 3) +
    // 8 (j >> 2) + 4 h): sW2[c][h][j], read back as four uniform 16-byte reads per tile (16 registers would cost a wave per SIMD)
    float *sW2 = reinterpret_cast<float *>(sLutT + 16);
    uint4 *sRows = reinterpret_cast<uint4 *>(sW2 + N_HID);               // [3][64]: the 32 rows of a tile, two uint4 each
    if (threadIdx.x < N_HID) {
        const int cc = threadIdx.x >> 5, hh = (threadIdx.x >> 4) & 1, j = threadIdx.x & 15;
        sW2[threadIdx.x] = w2[32 * cc + (j & 3) + 8 * (j >> 2) + 4 * hh];
    }
    const f32x4_t *w2v = reinterpret_cast<const f32x4_t *>(sW2 + (c * 2 + h) * 16);
    const float b2 = *b2p;
    __syncthreads();

    uint32_t ksteps = 0;
    // B operand of K-step s < 12 for (row r of the tile, lane half h): the 8 features of board point 2 s + h
    auto decode = [&](const Side &sa, const Side &sb, int s) -> uint4 {
        const int pos = 2 * s + h + 1;
        const uint2 e0 = sLut[count_at(sa, pos)], e1 = sLut[count_at(sb, pos)];
        return make_uint4(e0.x, e0.y, e1.x, e1.y);
    };
    // the K-steps wave c decodes: c, c + 4, c + 8, and the tail step for wave 0
    // The rows of a tile come in by ONE LDS-DMA instruction (64 lanes x 16 bytes = its 32 rows), issued by wave 0 two tiles
    // ahead: nobody holds them in registers while they travel, and their latency is two iterations away from their use.
    // (Lanes whose row lies behind n_rows request nothing: stage() reads such rows as empty.)
    auto fetch = [&](long long tile, int slot) {
        if (c == 0 && tile * 32 + (lane >> 1) < n_rows)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rows + tile * 64 + lane),
                                             (__attribute__((address_space(3))) void *)(sRows + slot * 64), 16, 0, 0);
    };
    auto stage = [&](long long tile, int buf, int slot) {
        uint32_t p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (tile < n_tiles && tile * 32 + r < n_rows) {
            const uint4 u0 = sRows[slot * 64 + 2 * r], u1 = sRows[slot * 64 + 2 * r + 1];
            p[0] = u0.x; p[1] = u0.y; p[2] = u0.z; p[3] = u0.w; p[4] = u1.x; p[5] = u1.y; p[6] = u1.z; p[7] = u1.w;
        }
        const Side sa{{p[0], p[1], p[2], p[3]}}, sb{{p[4], p[5], p[6], p[7]}};
        uint4 *dst = sX + buf * D16_XBUF_U4 + lane;
#pragma unroll
        for (int q = 0; q < 3; ++q) dst[(c + 4 * q) * 64] = decode(sa, sb, c + 4 * q);
        if (c == 0) {
            // [turn0, turn1, bar1/2, bar2/2, off1.hi, off1.lo, off2.hi, off2.lo] on the h = 0 lanes (the counts through a table:
            // x = k / 2 and the f16 hi + lo pair of k / 15 as round 1's kernel computes them); the h = 1 lanes carry the constant 1
            const uint32_t one = 0x3C00u;
            const uint32_t t0 = (p[0] & TURN_BIT) ? 0u : one, t1 = (p[0] & TURN_BIT) ? one : 0u;
            const uint32_t bar1 = sLutT[count_at(sa, 0)].x, bar2 = sLutT[count_at(sb, 25)].x;
            const uint32_t off1 = sLutT[count_at(sa, 25)].y, off2 = sLutT[count_at(sb, 0)].y;
            dst[12 * 64] = h ? make_uint4(one, 0u, 0u, 0u) : make_uint4(t0 | (t1 << 16), bar1 | (bar2 << 16), off1, off2);
        }
        // which board points hold a checker of either side in ANY row of the tile (bit i = point i, i = 1 .. 24)
        uint32_t occ = (p[0] | p[1] | p[2] | p[3] | p[4] | p[5] | p[6] | p[7]) & PTS;
        occ |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)occ, 0x111, 0xF, 0xF, true);      // row_shr 1, 2, 4, 8: lane 15 of every
        occ |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)occ, 0x112, 0xF, 0xF, true);      // 16-lane row holds the row's OR
        occ |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)occ, 0x114, 0xF, 0xF, true);
        occ |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)occ, 0x118, 0xF, 0xF, true);
        return (uint32_t)(__builtin_amdgcn_readlane((int)occ, 15) | __builtin_amdgcn_readlane((int)occ, 31));   // (lanes 32-63: the same rows)
    };

    long long tile = blockIdx.x;
    fetch(tile, 0);
    fetch(tile + gridDim.x, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t live_next = stage(tile, 0, 0);
    __syncthreads();
    int it = 0, slot = 0;                                      // slot = it % 3: the row buffer of THIS tile
    for (; tile < n_tiles; tile += gridDim.x, ++it) {
        const int buf = it & 1;
        const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot1 == 2 ? 0 : slot1 + 1;
        const uint32_t live = live_next;
        fetch(tile + 2 * (long long)gridDim.x, slot2);         // rows of the tile after next (its buffer was last read an iteration ago)
        live_next = stage(tile + gridDim.x, buf ^ 1, slot1);   // the NEXT tile's operand, into the other buffer
        floatx16 acc = {0};
        const uint4 *xp = sX + buf * D16_XBUF_U4 + lane;
#pragma unroll
        for (int s = 0; s < K16_STEPS; ++s) {
            if (s == 12 || ((live >> (2 * s + 1)) & 3u)) {                   // wave-uniform
                WF x;
                x.u = xp[s * 64];
                if (N_PLANES == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[N_PLANES - 1][s].v, x.v, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][s].v, x.v, acc, 0, 0, 0);
                ksteps += 1;
            }
        }
        // hidden sigmoids and this wave's share of the W2 dot: the accumulators hold -log2(e) (W1 x + b1)
        float part = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4_t wq = w2v[q];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                part = __builtin_fmaf(wq[i], __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[4 * q + i])), part);
        }
        part += __shfl_xor(part, 32, 64);                      // the two lane halves hold the other 16 units of the row
        if (h == 0) sRed[(buf * 4 + c) * 32 + r] = part;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // wave 0: the rows it requested have landed
        __syncthreads();
        slot = slot1;
        if (c == (it & 3)) {                                   // one wave finishes the tile: 32 rows on its lanes 0 .. 31
            const float *rp = sRed + buf * 4 * 32 + r;
            const float sum = (rp[0] + rp[32]) + (rp[64] + rp[96]);
            const long long orow = tile * 32 + r;
            const bool vrow = h == 0 && orow < n_rows;
            const float v = fast_sigmoid(sum + b2);
            if (vrow) values[orow] = v;
            if (info) {
                const uint2 inf = vrow ? info[orow] : make_uint2(0u, 0u);
                best_atomic_max(best, inf.x, v, inf.y, vrow, 32);
            }
        }
    }
    __shared__ unsigned int s_ks;
    if (threadIdx.x == 0) s_ks = 0;
    __syncthreads();
    if (lane == 0 && ksteps) atomicAdd(&s_ks, ksteps);
    __syncthreads();
    if (ksteps_counter && threadIdx.x == 0 && s_ks) atomicAdd(ksteps_counter, (unsigned long long)s_ks);
}

}  // namespace bg
