// bg_eval_mfma.h -- the incremental value net with its delta on the matrix pipe (round 3).
//
// Same mathematics as eval_rows_delta_kernel (bg_eval.h): a_row = a_root + Σ_{changed features f} Δx_f · W1[:, f], then the
// hidden sigmoid, the W2 dot, the output sigmoid and the per-game arg-max (model.py:63-67, 209-213).  What changes is WHO adds
// the W1 columns.  The VALU kernel lets every row gather its ~5 columns privately from LDS (2.6 KB of LDS reads per row, 128
// accumulators per lane).  Here the rows of ONE GAME that sit next to each other in the arena (a "piece": a run of equal game
// ids inside a 64-row chunk) share one K-compacted product on v_mfma_f32_32x32x16_f16:
//
//     D[rows x Kc] · Wc[Kc x 128],   Kc = the UNION of the features any row of the piece changes (measured: mean 14.6 per
//                                     game, p99 34, over 3 868 turns of the reference checkpoint's greedy play: DESIGN.md §4)
//
//   * K-compaction: the eight difference masks of a row (three thermometer levels + the (n-3)/2 feature per side, bar and
//     borne-off bits riding in the fourth word) are OR-ed over the piece; a feature's slot is its rank in that union
//     (prefix popcounts), so equal rows of a piece get equal A rows and bit-identical values.
//   * W1 = f16 hi + f16 lo of a FIXED-POINT table (relayout_w1_mdelta: all sums exact, values independent of the arena
//     layout), interleaved per dword in LDS as [feature][column tile c][unit n] -- hi and lo are two ADJACENT k-slots of the
//     MFMA, so a dword read from LDS is a B-operand register as it stands and the A operand carries every multiplier twice.
//     One K-step = 8 features.  The multipliers are small integers (thermometer flips ±1; (n-3)/2, bar/2 and off/15 carry
//     their factor in the W row): exact in f16, every product exact in fp32, fp32 accumulation.
//   * C layout: the hidden unit lives on the lane, the row in the register index -- so a piece's root term is 4 floats per
//     lane (not 512 B per row), applied as  sigmoid = rcp(1 + 2^acc · 2^root)  with 2^root formed once per piece, and rows a
//     piece does not have cost no epilogue (piece row 2j + h sits in accumulator register j of lane half h).
//   * per row: the W2 dot is reduced over the 32 lanes of a half by DPP adds; values go back to row order through LDS and the
//     output sigmoid + per-game maximum run on all 64 rows of the chunk at once.
//
// Everything a wave shares between its lanes (A image, slot -> feature map, values) is wave-private LDS: no block barrier
// after the weights are staged.
#pragma once
#include "bg_eval.h"

namespace bg {

#ifndef BG_MD_THREADS
#define BG_MD_THREADS 1024
#endif
constexpr int MD_THREADS = BG_MD_THREADS;               // 16 waves per CU (4 per SIMD): <= 128 VGPRs
constexpr int MD_WAVES = MD_THREADS / 64;
constexpr int MD_SLOTS = 32;                            // compact feature slots of a piece (4 K-steps of 8 features)
constexpr int MD_W_DWORDS = N_IN * N_HID;               // [198][4][32] dwords: f16 hi | f16 lo << 16
constexpr int MD_W_BYTES = MD_W_DWORDS * 4;             // 101 376
constexpr int MD_IMG_STRIDE = MD_SLOTS + 4;             // 36: rows 4 apart would share all their LDS banks at a stride of 32
constexpr int MD_IMG_BYTES = 64 * MD_IMG_STRIDE;        // [64 rows][32 slots + pad]: HIGH byte of the f16 multiplier (the low byte is 0)
constexpr int MD_MAP_BYTES = 64 * 16;                   // [run start lane][16]: feature index of a slot; a piece owns 16 per row
constexpr int MD_VAL_BYTES = 64 * 4;
constexpr int MD_WAVE_BYTES = MD_IMG_BYTES + MD_MAP_BYTES + MD_VAL_BYTES;       // 3 584
constexpr int MD_LDS_TOTAL = MD_W_BYTES + N_HID * 4 + MD_WAVES * MD_WAVE_BYTES; // 144 896
constexpr int MD_MAX_ENTRIES = 16;                      // as DELTA_MAX: a legal turn changes at most 13 features

// W table of the kernel: entry (f, unit) = -log2(e) · W1[unit][f] · s_f with s = 1 for the thermometer features, 1/2 for
// (n-3)/2 and bar/2, 1/15 for off/15 (so every multiplier is an integer), as FIXED POINT with a quantum 2^-Q per hidden unit,
// split into two f16 halves hi + lo (11 + 11 bits).
//
// Why fixed point: the MFMA rounds its running sum, so with a floating split the last bits of a row's value depend on which
// slots its features landed in -- on the other rows of its piece, that is on where the leaf stage happened to put the game in the
// arena.  Values must not depend on that (duplicates of an afterstate, shards of an env, two runs of the same seed have to agree
// to the bit).  With every term a multiple of 2^-Q and every partial sum below 2^(24-Q) all sums are EXACT in fp32 in any order:
// the product is then the exact integer sum, whatever the matrix pipe's association.  Q follows from a bound on what a legal
// turn can add up for that unit: a turn is at most 16 unit changes of a count (4 checkers x origin, landing point, hit point,
// opponent's bar), a thermometer feature takes one of them, the (n-3)/2, bar and borne-off features up to four.
// The price is the table's precision: 2^-(Q+1) per entry, Q = 18..21 for the reference checkpoint (values within 1.4e-6 of the
// unquantised chain on 40 k afterstates; round 2's VALU kernel, BGAMD_VALU_DELTA=1, keeps fp32 weights).
// Returns the smallest Q used, or -1 when the table cannot be held (the caller then keeps the VALU kernel for that slot).
constexpr int MD_MIN_Q = 14;
inline int relayout_w1_mdelta(const float *w1 /*[128][198]*/, uint32_t *wm /*[198][4][32]*/)
{
    const double NL2E = -1.44269504088896340736;
    int min_q = 24;
    for (int u = 0; u < N_HID; ++u) {
        double t[N_IN], top[16];
        for (int k = 0; k < 16; ++k) top[k] = 0.0;
        double mx = 0.0;
        for (int f = 0; f < N_IN; ++f) {
            double sc = 1.0;
            int mult = 1;
            if (f < 192) { if ((f & 3) == 3) { sc = 0.5; mult = 4; } }
            else if (f == 194 || f == 195) { sc = 0.5; mult = 4; }
            else if (f >= 196) { sc = 1.0 / 15.0; mult = 4; }
            t[f] = NL2E * (double)w1[u * N_IN + f] * sc;
            const double a = t[f] < 0 ? -t[f] : t[f];
            if (!(a <= 1e30)) return -1;                                   // inf / NaN weights
            if (a > mx) mx = a;
            for (int r = 0; r < mult; ++r) {                                // keep the 16 largest items
                int k = 15;
                if (a <= top[k]) break;
                while (k > 0 && top[k - 1] < a) { top[k] = top[k - 1]; --k; }
                top[k] = a;
            }
        }
        double bound = 0.0;
        for (int k = 0; k < 16; ++k) bound += top[k];
        bound *= 1.0 + 1.0 / 256.0;                                         // |hi| + |lo| may exceed |hi + lo| by 2^-10
        int q = 24;
        while (q >= 0 && (bound * (double)(1u << q) > 16777216.0 || mx * (double)(1u << q) >= 4190000.0)) --q;
        if (q < MD_MIN_Q) return -1;
        min_q = q < min_q ? q : min_q;
        const double scale = (double)(1u << q), inv = 1.0 / scale;
        for (int f = 0; f < N_IN; ++f) {
            const long long ti = (long long)(t[f] * scale + (t[f] < 0 ? -0.5 : 0.5));      // |ti| < 2^22
            long long hi = (ti >= 0 ? (ti + 1024) / 2048 : -((-ti + 1024) / 2048)) * 2048;  // |hi / 2048| <= 2048: exact in f16
            const long long lo = ti - hi;                                                   // |lo| <= 1024
            const _Float16 h16 = (_Float16)(float)((double)hi * inv);
            const _Float16 l16 = (_Float16)(float)((double)lo * inv);
            wm[f * N_HID + (u >> 5) * 32 + (u & 31)] = (uint32_t)f16_bits(h16) | ((uint32_t)f16_bits(l16) << 16);
        }
    }
    return min_q;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float md_dpp(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xF, false));
}
// sum over the 32 lanes of each wave half; the total is valid on lanes 16-31 and 48-63
__device__ __forceinline__ float md_half_sum(float x)
{
    x += md_dpp<0xB1, 0xF>(x);      // quad_perm [1,0,3,2]
    x += md_dpp<0x4E, 0xF>(x);      // quad_perm [2,3,0,1]
    x += md_dpp<0x141, 0xF>(x);     // row_half_mirror
    x += md_dpp<0x140, 0xF>(x);     // row_mirror: every lane of a 16-lane row holds the row's sum
    x += md_dpp<0x142, 0xA>(x);     // row_bcast15 into rows 1 and 3: + the sum of the row before
    return x;
}

typedef _Float16 md_f16x8 __attribute__((ext_vector_type(8)));
// diagnostic builds (tools/ab_build.sh x "-DBG_MD_PROF=k", tools/md_prof.py): the work counter then holds the shader cycles
// the waves spent in phase k instead of the K-step count.  1 whole chunk loop, 2 rows -> masks, 3 unions, 4 image + map,
// 5 pieces (product, sigmoids, W2 dot), 6 output sigmoid + arg-max, 7 weight staging
#ifndef BG_MD_PROF
#define BG_MD_PROF 0
#endif
// timing experiments (results are WRONG): 1 no union atomics, 2 no image / map writes, 4 no transcendentals, 8 no lane reduction,
// 16 no B reads and MFMAs, 32 no per-game arg-max, 64 no pieces at all
#ifndef BG_MD_ABL
#define BG_MD_ABL 0
#endif
#define BG_MD_STAMP(K_BEGIN, K_END)                                                              \
    if (BG_MD_PROF == (K_BEGIN)) prof_t0 = __builtin_readcyclecounter();                          \
    if (BG_MD_PROF == (K_END)) n_work += (uint32_t)(__builtin_readcyclecounter() - prof_t0);
typedef __attribute__((address_space(3))) uint32_t md_lds_u32;

__global__ __launch_bounds__(MD_THREADS) void eval_rows_mdelta_kernel(
    const uint4 *__restrict__ rows, const unsigned long long *__restrict__ n_rows_ptr, long long n_rows_imm,
    unsigned long long *__restrict__ rows_eval_counter, const uint4 *__restrict__ wm, const float *__restrict__ w2,
    const float *__restrict__ b2p, const uint4 *__restrict__ root_rows, const float *__restrict__ root_hidden,
    float *__restrict__ values, const uint2 *__restrict__ info, unsigned long long *__restrict__ best,
    unsigned long long *__restrict__ work_counter, unsigned long long *__restrict__ zero_words, int n_zero_words,
    unsigned long long *__restrict__ err_word, unsigned long long err_bit)
{
    uint32_t n_work = 0;                                  // K-steps x tiles executed by this wave (x 4 = MFMAs)
    unsigned long long prof_t0 = 0;
    (void)prof_t0;
    BG_MD_STAMP(7, -1);
    // multi-step runs: the OTHER set of list counters is cleared here (see eval_rows_delta_kernel)
    if (zero_words && blockIdx.x == 0 && (int)threadIdx.x < n_zero_words) zero_words[threadIdx.x] = 0ull;
    extern __shared__ uint4 sMD[];
    uint32_t *sW = reinterpret_cast<uint32_t *>(sMD);                       // [198][4][32]
    const md_lds_u32 *sWl = (const md_lds_u32 *)sW;                         // the same, known to the compiler as LDS
    float *sW2 = reinterpret_cast<float *>(sW + MD_W_DWORDS);
    uint8_t *wave_lds = reinterpret_cast<uint8_t *>(sW2 + N_HID) + (threadIdx.x >> 6) * MD_WAVE_BYTES;
    uint8_t *img = wave_lds;                                                // [64][36]
    uint8_t *fmap = wave_lds + MD_IMG_BYTES;                                // [64][16]
    float *vals = reinterpret_cast<float *>(wave_lds + MD_IMG_BYTES + MD_MAP_BYTES);
    for (int i = threadIdx.x; i < MD_W_BYTES / 16; i += MD_THREADS) sMD[i] = wm[i];
    if (threadIdx.x < N_HID) sW2[threadIdx.x] = w2[threadIdx.x];
    __shared__ unsigned int s_ticket;
    if (threadIdx.x == 0) s_ticket = 0;
    __syncthreads();
    BG_MD_STAMP(-1, 7);

    long long n_rows = n_rows_imm;
    if (n_rows_ptr) { const long long c = (long long)*n_rows_ptr; n_rows = c < n_rows_imm ? c : n_rows_imm; }
    if (rows_eval_counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(rows_eval_counter, (unsigned long long)n_rows);
    const long long n_tiles = (n_rows + 63) >> 6;
    const int lane = threadIdx.x & 63;
    const int n = lane & 31, h = lane >> 5;
    // piece row 2j + h <-> accumulator register j of lane half h <-> MFMA row (j&3) + 8 (j>>2) + 4 h; the A operand of MFMA row
    // r = lane & 31 therefore comes from piece row rho
    const int rho = 2 * ((n & 3) + 4 * (n >> 3)) + ((n >> 2) & 1);
    auto grab = [&]() -> long long {
        unsigned int t = 0;
        if (lane == 0) t = atomicAdd(&s_ticket, 1u);
        return (long long)blockIdx.x + (long long)__builtin_amdgcn_readfirstlane(t) * (long long)gridDim.x;
    };
    const float b2 = *b2p;
    f32x2_t w2a = {sW2[n], sW2[32 + n]}, w2b = {sW2[64 + n], sW2[96 + n]};
    bool bad_row = false;

    uint4 nx0 = make_uint4(0, 0, 0, 0), nx1 = make_uint4(0, 0, 0, 0);
    uint2 nxi = make_uint2(0u, 0u);
    long long tile = grab();
    if (tile < n_tiles && tile * 64 + lane < n_rows) {
        const long long r0 = tile * 64 + lane;
        nx0 = rows[2 * r0]; nx1 = rows[2 * r0 + 1]; nxi = info[r0];
    }
    uint4 nr0 = root_rows[2 * (long long)nxi.x], nr1 = root_rows[2 * (long long)nxi.x + 1];

    BG_MD_STAMP(1, -1);
    while (tile < n_tiles) {
        BG_MD_STAMP(2, -1);
        const long long next_tile = grab();
        const long long row = tile * 64 + lane;
        const bool valid = row < n_rows;
        const uint2 inf = nxi;
        const uint32_t game = valid ? inf.x : 0xFFFFFFFFu;
        uint32_t M[8], G[8];                               // difference masks; G = the new row's thermometer words (signs)
        uint32_t pl[8], ql[8];
        {
            const uint4 r0 = nr0, r1 = nr1;
            const uint32_t p[8] = {nx0.x & ~TURN_BIT, nx0.y, nx0.z, nx0.w, nx1.x, nx1.y, nx1.z, nx1.w};
            const uint32_t q[8] = {r0.x & ~TURN_BIT, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) { pl[i] = valid ? p[i] : 0u; ql[i] = valid ? q[i] : 0u; }
        }
        uint32_t cnt = 0;
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
            const uint32_t b0 = pl[4 * sd], b1 = pl[4 * sd + 1], b2_ = pl[4 * sd + 2], b3 = pl[4 * sd + 3];
            const uint32_t c0 = ql[4 * sd], c1 = ql[4 * sd + 1], c2 = ql[4 * sd + 2], c3 = ql[4 * sd + 3];
            const uint32_t diff = (b0 ^ c0) | (b1 ^ c1) | (b2_ ^ c2) | (b3 ^ c3);
            G[4 * sd] = b0 | b1 | b2_ | b3; G[4 * sd + 1] = b1 | b2_ | b3; G[4 * sd + 2] = (b0 & b1) | b2_ | b3; G[4 * sd + 3] = 0;
            M[4 * sd] = (G[4 * sd] ^ (c0 | c1 | c2 | c3)) & PTS;
            M[4 * sd + 1] = (G[4 * sd + 1] ^ (c1 | c2 | c3)) & PTS;
            M[4 * sd + 2] = (G[4 * sd + 2] ^ ((c0 & c1) | c2 | c3)) & PTS;
            // (n-3)/2 moves wherever the count moves and is >= 4 before or after; bar / borne-off counters: bits 0 and 25
            M[4 * sd + 3] = diff & ((((b2_ | b3) | (c2 | c3)) & PTS) | 1u | (1u << 25));
#pragma unroll
            for (int k = 0; k < 4; ++k) cnt += (uint32_t)__popc(M[4 * sd + k]);
        }
        // more entries than any legal turn has: not an afterstate of its root -- flagged (BGAMD_E_DELTA), evaluated as the root
        if (cnt > (uint32_t)MD_MAX_ENTRIES) {
            bad_row = true;
#pragma unroll
            for (int i = 0; i < 8; ++i) M[i] = 0;
        }

        BG_MD_STAMP(3, 2);
        // ---- pieces: runs of equal games.  Hmask bit i = lane i starts a run.
        const uint32_t prev_game = __shfl_up(game, 1, 64);
        unsigned long long hmask = __ballot(lane == 0 || prev_game != game);
        uint32_t rs, U[8], pre[8], R[8], kc_total;      // R = the union bits this lane enters into the slot -> feature map
        uint32_t *us = reinterpret_cast<uint32_t *>(img);                   // union words [64 run starts][8] alias the image
        uint4 *img4 = reinterpret_cast<uint4 *>(img);
        for (;;) {
            const unsigned long long below = hmask & (~0ull >> (63 - lane));
            rs = 63u - (uint32_t)__clzll(below);
            const uint32_t dist = (uint32_t)lane - rs;
            img4[2 * lane] = make_uint4(0, 0, 0, 0); img4[2 * lane + 1] = make_uint4(0, 0, 0, 0);
            // OR over the rows of (run, 16-lane row) up to this lane by four DPP steps -- same-address LDS atomics from
            // every lane cost 15 us of the launch, one atomic per segment costs nothing
            uint32_t X[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) X[w] = M[w];
#define BG_MD_SCAN(D, CTRL)                                                                            \
            {                                                                                            \
                const uint32_t take = dist >= (uint32_t)(D) ? 0xFFFFFFFFu : 0u;                           \
                _Pragma("unroll") for (int w = 0; w < 8; ++w)                                            \
                    X[w] = ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)X[w], CTRL, 0xF, 0xF, true) & take) | X[w]; \
            }
            BG_MD_SCAN(1, 0x111) BG_MD_SCAN(2, 0x112) BG_MD_SCAN(4, 0x114) BG_MD_SCAN(8, 0x118)
#undef BG_MD_SCAN
            // a feature is entered into the map by the first lane of the segment that has it (every row of a piece writing
            // its features there: 13 lanes on one LDS byte)
            {
                const uint32_t take = dist >= 1u ? 0xFFFFFFFFu : 0u;
#pragma unroll
                for (int w = 0; w < 8; ++w)
                    R[w] = M[w] & ~((uint32_t)__builtin_amdgcn_update_dpp(0, (int)X[w], 0x111, 0xF, 0xF, true) & take);
            }
            const bool seg_last = (lane & 15) == 15 || ((hmask >> 1) >> lane) & 1ull;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (BG_MD_ABL & 1) { for (int w = 0; w < 8; ++w) us[lane * 8 + w] = M[w]; }
            else if (seg_last) {
#pragma unroll
                for (int w = 0; w < 8; ++w) atomicOr(&us[rs * 8 + w], X[w]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            {
                const uint4 u0 = img4[2 * rs], u1 = img4[2 * rs + 1];
                U[0] = u0.x; U[1] = u0.y; U[2] = u0.z; U[3] = u0.w; U[4] = u1.x; U[5] = u1.y; U[6] = u1.z; U[7] = u1.w;
            }
            uint32_t acc_k = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) { pre[w] = acc_k; acc_k += (uint32_t)__popc(U[w]); }
            kc_total = acc_k;
            __builtin_amdgcn_wave_barrier();
            // a union that does not fit the slots: cut the run in two and take the unions again (a single row has <= 16)
            const bool over = kc_total > (uint32_t)MD_SLOTS;
            if (__ballot(over) == 0ull) break;
            const unsigned long long above = (rs >= 63u) ? 0ull : (hmask >> (rs + 1));
            const uint32_t re = above ? rs + 1u + (uint32_t)__builtin_ctzll(above) : 64u;
            const uint32_t mid = rs + ((re - rs) >> 1);
            const unsigned long long cut = __ballot(over && (uint32_t)lane == mid && mid > rs);
            if (cut == 0ull) { bad_row = true; break; }                     // cannot happen (a row holds <= 16 entries): never spin
            hmask |= cut;
        }
        BG_MD_STAMP(4, 3);
        // ---- the image (zeroed: every slot a row does not touch multiplies by 0) and the slot -> feature map (zeroed: a K-step
        //      reads 8 slots whatever the union holds, and feature 0 is a finite W row)
        img4[2 * lane] = make_uint4(0, 0, 0, 0); img4[2 * lane + 1] = make_uint4(0, 0, 0, 0);
        if (lane < MD_IMG_BYTES / 16 - 128) img4[128 + lane] = make_uint4(0, 0, 0, 0);
        reinterpret_cast<uint4 *>(fmap)[lane] = make_uint4(0, 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
            // (kc < MD_SLOTS and rs * 16 + kc < MD_MAP_BYTES hold by construction: a union has <= MD_SLOTS features, a row <=
            //  MD_MAX_ENTRIES, a piece of R rows owns 16 R map bytes; a cut that failed above is the one exception, masked here)
            uint8_t *irow = img + lane * MD_IMG_STRIDE;
            uint8_t *mrow = fmap + rs * 16;
            uint8_t *dump = reinterpret_cast<uint8_t *>(vals + lane);      // (vals is written by the pieces below, read after them)
            if (kc_total > (uint32_t)MD_SLOTS) {
#pragma unroll
                for (int i = 0; i < 8; ++i) M[i] = 0;
            }
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const int sd = w >> 2, k = w & 3;
                uint32_t x = M[w];
                const uint32_t ng = ~G[w];
                while (x) {
                    const int pos = __ffs(x) - 1;
                    x &= x - 1;
                    const uint32_t kc = pre[w] + (uint32_t)__popc(U[w] & ((1u << pos) - 1u));
                    uint32_t f, byte;
                    if (k < 3) {
                        f = (uint32_t)(8 * pos + (4 * sd + k - 8));
                        byte = 0x3Cu | (((ng >> pos) & 1u) << 7);           // high byte of f16 +1 / -1
                    } else {
                        const Side sn{{pl[4 * sd], pl[4 * sd + 1], pl[4 * sd + 2], pl[4 * sd + 3]}};
                        const Side so{{ql[4 * sd], ql[4 * sd + 1], ql[4 * sd + 2], ql[4 * sd + 3]}};
                        const int n1 = count_at(sn, pos), n0 = count_at(so, pos);
                        int d;
                        if (pos == 0) { f = sd == 0 ? 194u : 197u; d = n1 - n0; }
                        else if (pos == 25) { f = sd == 0 ? 196u : 195u; d = n1 - n0; }
                        else { f = (uint32_t)(8 * (pos - 1) + 4 * sd + 3); d = (n1 > 3 ? n1 - 3 : 0) - (n0 > 3 ? n0 - 3 : 0); }
                        if (d > 8 || d < -8) { bad_row = true; d = 0; }     // no legal turn: the high byte alone would not hold it
                        byte = (uint32_t)f16_bits((_Float16)(float)d) >> 8;
                    }
                    // the map byte goes to the map only from the lane that answers for the feature, else to a byte of its own
                    uint8_t *mdst = ((R[w] >> pos) & 1u) ? mrow + kc : dump;
                    if (BG_MD_ABL & 256) reinterpret_cast<volatile uint32_t *>(vals)[lane] = byte;     // a conflict-free dword instead
                    else if (BG_MD_ABL & 512) irow[kc & 3] = (uint8_t)byte;                            // bytes, but the lane's own bank
                    else if (!(BG_MD_ABL & 2)) irow[kc] = (uint8_t)byte;
                    if (!(BG_MD_ABL & 128)) *mdst = (uint8_t)f;
                    if (BG_MD_ABL & (2 | 128)) asm volatile("" :: "v"(byte), "v"(f), "v"(kc), "v"(mdst));
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        BG_MD_STAMP(5, 4);
        // ---- piece by piece (wave-uniform): K-compacted product, hidden sigmoids, W2 dot
        unsigned long long hm = (BG_MD_ABL & 64) ? 0ull : hmask;
        // the first piece's root term is requested here, every later one while the piece before it is computed
        float rq0, rq1, rq2, rq3;
        {
            const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)game);
            const float *rh = root_hidden + (long long)(g0 == 0xFFFFFFFFu ? 0u : g0) * N_HID + n;
            rq0 = rh[0]; rq1 = rh[32]; rq2 = rh[64]; rq3 = rh[96];
        }
        while (hm) {
            const int L0 = __builtin_ctzll(hm);
            hm &= hm - 1;
            const int Lend = hm ? __builtin_ctzll(hm) : 64;
            const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)game, L0);
            const float r0 = rq0, r1 = rq1, r2 = rq2, r3 = rq3;
            if (hm) {                                                       // the next piece's root term
                const uint32_t gn = (uint32_t)__builtin_amdgcn_readlane((int)game, Lend);
                const float *rh = root_hidden + (long long)(gn == 0xFFFFFFFFu ? 0u : gn) * N_HID + n;
                rq0 = rh[0]; rq1 = rh[32]; rq2 = rh[64]; rq3 = rh[96];
            }
            if (g == 0xFFFFFFFFu) continue;                                 // the padding rows behind n_rows
            const int kcp = __builtin_amdgcn_readlane((int)kc_total, L0);
            const int nst = kcp > 8 ? (kcp > MD_SLOTS ? MD_SLOTS / 8 : (kcp + 7) >> 3) : 1;
            // 2^(root term) of this lane's four hidden units (the root term already carries -log2 e); clamped so that the
            // product with 2^acc below can never be inf x 0
            const f32x2_t ea = {__builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(r0, -80.0f, 80.0f)),
                                __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(r1, -80.0f, 80.0f))};
            const f32x2_t eb = {__builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(r2, -80.0f, 80.0f)),
                                __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(r3, -80.0f, 80.0f))};
            const uint32_t *mp = reinterpret_cast<const uint32_t *>(fmap + L0 * 16 + 4 * h);
            for (int t0 = L0; t0 < Lend; t0 += 32) {
                const int rt = Lend - t0 < 32 ? Lend - t0 : 32;             // rows of this tile
                const int nj = (rt + 1) >> 1;
                int arow = t0 + rho;
                arow = arow > 63 ? 63 : arow;                               // rows behind the piece: finite, never used
                const uint32_t *ap = reinterpret_cast<const uint32_t *>(img + arow * MD_IMG_STRIDE + 4 * h);
                floatx16 acc0, acc1, acc2, acc3;
                // one K-step: 8 features = 16 k-slots.  The B reads are volatile so that they stay sixteen ds_read_b32 into
                // the operand registers themselves (merged into ds_read2_b32 they need a v_mov each to get there)
#define BG_MD_KSTEP(S, C0, C1, C2, C3)                                                                   \
                {                                                                                          \
                    const uint32_t aw = ap[2 * (S)], mw = mp[2 * (S)];                                     \
                    union { uint32_t u[4]; md_f16x8 v; } a, bq0, bq1, bq2, bq3;                              \
                    a.u[0] = __builtin_amdgcn_perm(aw, aw, 0x000C000Cu);                                    \
                    a.u[1] = __builtin_amdgcn_perm(aw, aw, 0x010C010Cu);                                    \
                    a.u[2] = __builtin_amdgcn_perm(aw, aw, 0x020C020Cu);                                    \
                    a.u[3] = __builtin_amdgcn_perm(aw, aw, 0x030C030Cu);                                    \
                    _Pragma("unroll") for (int m = 0; m < 4; ++m) {                                        \
                        const volatile md_lds_u32 *wr = sWl + ((mw >> (8 * m)) & 255u) * N_HID + n;         \
                        if (BG_MD_ABL & 16) { bq0.u[m] = bq1.u[m] = bq2.u[m] = bq3.u[m] = (uint32_t)(uintptr_t)wr; } else {   \
                        bq0.u[m] = wr[0]; bq1.u[m] = wr[32]; bq2.u[m] = wr[64]; bq3.u[m] = wr[96]; }        \
                    }                                                                                      \
                    if (BG_MD_ABL & 16) {                                                                   \
                        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                    \
                            acc0[q] = C0[q] + __builtin_bit_cast(float, a.u[q & 3] ^ bq0.u[q & 3]); acc1[q] = C1[q] + __builtin_bit_cast(float, bq1.u[q & 3]); \
                            acc2[q] = C2[q] + __builtin_bit_cast(float, bq2.u[q & 3]); acc3[q] = C3[q] + __builtin_bit_cast(float, bq3.u[q & 3]); }            \
                    } else {                                                                                \
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, bq0.v, C0, 0, 0, 0);                 \
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, bq1.v, C1, 0, 0, 0);                 \
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, bq2.v, C2, 0, 0, 0);                 \
                    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, bq3.v, C3, 0, 0, 0); }               \
                }
                {
                    const floatx16 zero = {0};
                    BG_MD_KSTEP(0, zero, zero, zero, zero);                 // the first K-step starts from the constant 0
                }
                for (int s = 1; s < nst; ++s) BG_MD_KSTEP(s, acc0, acc1, acc2, acc3);
#undef BG_MD_KSTEP
                if (BG_MD_PROF == 0) n_work += (uint32_t)nst;
                // hidden sigmoids and the W2 dot of four accumulator registers (= eight rows) at a time; the sums over the 32
                // lanes of a half are taken by a reduce-scatter: lane l ends with the total of register 4 G + (l & 3)
                float outv = 0.0f;
#if BG_MD_ABL & 4
#define BG_MD_EXP(X) ((X) * 0.75f)
#define BG_MD_RCP(X) ((X) * 0.5f)
#else
#define BG_MD_EXP(X) __builtin_amdgcn_exp2f(X)
#define BG_MD_RCP(X) __builtin_amdgcn_rcpf(X)
#endif
#define BG_MD_ROWSUM(J)                                                                                    \
                ({                                                                                          \
                    f32x2_t e0 = {BG_MD_EXP(acc0[J]), BG_MD_EXP(acc1[J])};                                   \
                    f32x2_t e1 = {BG_MD_EXP(acc2[J]), BG_MD_EXP(acc3[J])};                                   \
                    e0 = __builtin_elementwise_fma(e0, ea, (f32x2_t){1.0f, 1.0f});                           \
                    e1 = __builtin_elementwise_fma(e1, eb, (f32x2_t){1.0f, 1.0f});                           \
                    const f32x2_t q0 = {BG_MD_RCP(e0.x), BG_MD_RCP(e0.y)};                                   \
                    const f32x2_t q1 = {BG_MD_RCP(e1.x), BG_MD_RCP(e1.y)};                                   \
                    f32x2_t ps = q0 * w2a;                                                                  \
                    ps = __builtin_elementwise_fma(q1, w2b, ps);                                            \
                    ps.x + ps.y;                                                                            \
                })
#pragma unroll
                for (int G4 = 0; G4 < 4; ++G4) {
                    if (4 * G4 < nj) {                                      // wave-uniform, like the four tests below
                        // two registers at a time: their transcendental chains fill each other's wait states (a register
                        // behind the piece's last holds finite values of rows that are never stored)
                        const float p0 = BG_MD_ROWSUM(4 * G4), p1 = BG_MD_ROWSUM(4 * G4 + 1);
                        float p2 = 0.0f, p3 = 0.0f;
                        if (4 * G4 + 2 < nj) { p2 = BG_MD_ROWSUM(4 * G4 + 2); p3 = BG_MD_ROWSUM(4 * G4 + 3); }
                        if (BG_MD_ABL & 8) { outv = ((n >> 2) == G4) ? (p0 + p1) + (p2 + p3) : outv; continue; }
                        const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
                        const float x = (b0 ? p1 : p0) + md_dpp<0xB1, 0xF>(b0 ? p0 : p1);      // + lane ^ 1
                        const float y = (b0 ? p3 : p2) + md_dpp<0xB1, 0xF>(b0 ? p2 : p3);
                        float z = (b1 ? y : x) + md_dpp<0x4E, 0xF>(b1 ? x : y);                // + lane ^ 2: the quad's sum of register l & 3
                        // the association must not depend on the lane (a row's value may not depend on the register it
                        // sits in): row_ror:8 first -- (Q0+Q2), (Q1+Q3) -- then row_ror:4 gives ((Q0+Q2)+(Q1+Q3)) on every lane
                        z += md_dpp<0x128, 0xF>(z);                                           // row_ror:8 = lane ^ 8
                        z += md_dpp<0x124, 0xF>(z);                                           // row_ror:4: the 16-lane row's sum
                        z += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, z), 0x401F));   // + lane ^ 16
                        outv = ((n >> 2) == G4) ? z : outv;                 // lane n < 16 keeps register n
                    }
                }
#undef BG_MD_ROWSUM
                if (n < 16 && 2 * n + h < rt) vals[t0 + 2 * n + h] = outv;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        BG_MD_STAMP(6, 5);
        // the next chunk's rows are requested only now: held across the pieces above they cost the 18 registers that decide
        // between three and four waves per SIMD
        {
            const long long nrow = next_tile * 64 + lane;
            nx0 = make_uint4(0, 0, 0, 0); nx1 = make_uint4(0, 0, 0, 0); nxi = make_uint2(0u, 0u);
            if (next_tile < n_tiles && nrow < n_rows) { nx0 = rows[2 * nrow]; nx1 = rows[2 * nrow + 1]; nxi = info[nrow]; }
        }
        {
            const float sum = valid ? vals[lane] : 0.0f;
            const float v = fast_sigmoid(sum + b2);
            if (valid) values[row] = v;
            if (!(BG_MD_ABL & 32)) best_atomic_max(best, inf.x, v, inf.y, valid, 64);
        }
        nr0 = root_rows[2 * (long long)nxi.x]; nr1 = root_rows[2 * (long long)nxi.x + 1];
        __builtin_amdgcn_wave_barrier();
        BG_MD_STAMP(-1, 6);
        tile = next_tile;
    }
    BG_MD_STAMP(-1, 1);
    __shared__ unsigned int s_nw;
    if (threadIdx.x == 0) s_nw = 0;
    __syncthreads();
    if (lane == 0 && n_work) atomicAdd(&s_nw, n_work);
    __syncthreads();
    if (work_counter && threadIdx.x == 0 && s_nw) atomicAdd(work_counter, (unsigned long long)s_nw);
    if (err_word && __ballot(bad_row) != 0ull && lane == 0) atomicOr(err_word, err_bit);
}

}  // namespace bg
