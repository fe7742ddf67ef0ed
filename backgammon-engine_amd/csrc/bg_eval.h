// bg_eval.h -- 198-feature encoder + 198->128->1 sigmoid value net over packed candidate rows.
//
// Restates TDLGammonModel._encode_states_np (pysrc/TD(λ) model/model.py:111-144) and
// TDLGammonModel.forward (model.py:63-67).  The 198 features are never materialised: each
// MFMA A-operand element is decoded from the row's bit planes right before it is consumed.
//
// fp32 kernel: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, K = 198 = 99 steps of 2).
//   A[row][k]  : lane l supplies row (l & 31), feature k = 2*s + (l >> 5)
//   B[k][n]    : W1[n][k]; lane l supplies n = 32*c + (l & 31), k = 2*s + (l >> 5) for the four
//                32-column tiles c -- staged ONCE per workgroup in LDS as float4[99][64]
//   D[row][n]  : 4 tiles x 16 accumulators; row = (reg&3) + 8*(reg>>2) + 4*(l>>5), n = 32*c + (l&31)
// Epilogue fused: + b1, sigmoid, dot with W2 over the 128 hidden units (4 per lane, then a
// 32-lane butterfly), + b2, sigmoid -> one fp32 value per row.
#pragma once
#include "bg_board.h"

namespace bg {

constexpr int N_IN = 198, N_HID = 128;
constexpr int N_PARAMS = N_HID * N_IN + N_HID + N_HID + 1;
constexpr int K_STEPS = N_IN / 2;                  // 99
constexpr int EVAL_LDS_BYTES = K_STEPS * 64 * 16;  // 101 376
constexpr int EVAL_THREADS = 768;                  // 12 waves per CU (3 per SIMD): block-sparse tiles leave the MFMA pipe ~50 % busy, more waves fill it

typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float fast_sigmoid(float x)
{
    // 1 / (1 + 2^(-x*log2 e)); v_exp_f32 + v_rcp_f32, |err| ~ 2e-7 absolute on (0,1)
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}

// Per-game arg-max (P1) / arg-min (P2) with "smallest reference key wins ties" (model.py:212-213): one 64-bit
// atomicMax of (ordered value bits, ~key) per game.  The rows of a game sit in neighbouring lanes, so the maximum is
// first taken over each run of equal games among the first `width` lanes and only the run's first lane goes to memory
// (a few atomics per wave instead of one per row, most of them on the same address).  All 64 lanes must call it.
__device__ __forceinline__ void best_atomic_max(unsigned long long *__restrict__ best, uint32_t game, float v, uint32_t key_turn,
                                                bool valid, int width)
{
    const int lane = threadIdx.x & 63;
    uint32_t bits = __float_as_uint(v);
    bits = (key_turn >> 31) ? ~bits : bits;
    unsigned long long pack = valid ? (((unsigned long long)bits << 32) | (uint32_t)~(key_turn & 0x7FFFFFFFu)) : 0ull;
    if (!valid) game = 0xFFFFFFFFu;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t og = __shfl_down(game, d, 64);
        const unsigned long long op = __shfl_down(pack, d, 64);
        if (d < width && lane + d < width && og == game && op > pack) pack = op;
    }
    const uint32_t pg = __shfl_up(game, 1, 64);
    if (valid && (lane == 0 || pg != game)) atomicMax(&best[game], pack);
}

// per-row decode state: for each side q (0 = PLAYER1, 1 = PLAYER2)
//   ev[q]   : mask whose bit (i+1) is this lane's EVEN-step feature of point i  (n>=1 | n>=2)
//   od[q][k]: planes of the odd-step code; feature value = 0.5 * code            (n>=3 | (n-3)/2)
struct RowDecode {
    uint32_t ev[2];
    uint32_t od[2][4];
    float tail[3];
};

__device__ __forceinline__ void decode_setup(const uint32_t (&p)[8], int h, RowDecode &r)
{
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint32_t b0 = p[4 * q] & PTS, b1 = p[4 * q + 1] & PTS, b2 = p[4 * q + 2] & PTS, b3 = p[4 * q + 3] & PTS;
        const uint32_t ge1 = b0 | b1 | b2 | b3;
        const uint32_t ge2 = b1 | b2 | b3;
        const uint32_t ge3 = (b0 & b1) | b2 | b3;
        const uint32_t ge4 = b2 | b3;
        // bit-sliced (n + 13) mod 16 == n - 3 for n >= 3, masked to n >= 4
        const uint32_t s0 = ~b0, c0 = b0;
        const uint32_t s1 = b1 ^ c0, c1 = b1 & c0;
        const uint32_t s2 = ~(b2 ^ c1), c2 = b2 | c1;
        const uint32_t s3 = ~(b3 ^ c2);
        r.ev[q] = h ? ge2 : ge1;
        r.od[q][0] = h ? (s0 & ge4) : 0u;
        r.od[q][1] = h ? (s1 & ge4) : ge3;          // code 2 -> 1.0 for the n>=3 flag
        r.od[q][2] = h ? (s2 & ge4) : 0u;
        r.od[q][3] = h ? (s3 & ge4) : 0u;
    }
    const Side a{{p[0], p[1], p[2], p[3]}}, b{{p[4], p[5], p[6], p[7]}};
    const int turn = (p[0] & TURN_BIT) ? 1 : 0;
    r.tail[0] = (h == turn) ? 1.0f : 0.0f;                               // 192 / 193
    r.tail[1] = 0.5f * (float)(h ? count_at(b, 25) : count_at(a, 0));    // 194 / 195: bar / 2
    r.tail[2] = (float)(h ? count_at(b, 0) : count_at(a, 25)) / 15.0f;   // 196 / 197: off / 15
}

__device__ __forceinline__ float decode_even(const RowDecode &r, int q, int i)
{
    return (float)((r.ev[q] >> (i + 1)) & 1u);
}
__device__ __forceinline__ float decode_odd(const RowDecode &r, int q, int i)
{
    const int sh = i + 1;
    const uint32_t code = ((r.od[q][0] >> sh) & 1u) | (((r.od[q][1] >> sh) & 1u) << 1) |
                          (((r.od[q][2] >> sh) & 1u) << 2) | (((r.od[q][3] >> sh) & 1u) << 3);
    return 0.5f * (float)code;
}

// Host-side weight re-layout for the fp32 kernel: Wl[s][l][c] = W1[32c + (l&31)][2s + (l>>5)]
inline void relayout_w1_f32(const float *w1 /*[128][198]*/, float *wl /*[99][64][4]*/)
{
    for (int s = 0; s < K_STEPS; ++s)
        for (int l = 0; l < 64; ++l)
            for (int c = 0; c < 4; ++c)
                wl[(s * 64 + l) * 4 + c] = w1[(32 * c + (l & 31)) * N_IN + 2 * s + (l >> 5)];
}

// rows: 2 x uint4 per candidate.  n_rows_ptr: device counter (rows emitted this step) capped by n_rows_imm, or null -> n_rows_imm.
constexpr int EVAL_RED_STRIDE = 33;                                   // floats per column in the reduction scratch
constexpr int EVAL_RED_FLOATS = 32 * EVAL_RED_STRIDE;                 // per wave
constexpr int EVAL_LDS_TOTAL = EVAL_LDS_BYTES + (EVAL_THREADS / 64) * EVAL_RED_FLOATS * 4;

// HIDDEN: instead of the value, store -log2(e) x the 128 hidden PRE-activations (W1 x + b1) of every row to values[row*128+n]
// (the per-game root term of the incremental evaluator below)
template <bool HIDDEN>
__global__ __launch_bounds__(EVAL_THREADS) void eval_rows_f32_kernel(
    const uint4 *__restrict__ rows, const unsigned long long *__restrict__ n_rows_ptr, long long n_rows_imm,
    unsigned long long *__restrict__ rows_eval_counter, const float4 *__restrict__ wl, const float *__restrict__ b1, const float *__restrict__ w2,
    const float *__restrict__ b2p, float *__restrict__ values, const uint2 *__restrict__ info,
    unsigned long long *__restrict__ best, unsigned long long *__restrict__ ksteps_counter)
{
    extern __shared__ float4 sW[];
    float *sRed = reinterpret_cast<float *>(sW + K_STEPS * 64) + (threadIdx.x >> 6) * EVAL_RED_FLOATS;
    for (int i = threadIdx.x; i < K_STEPS * 64; i += EVAL_THREADS) sW[i] = wl[i];
    __syncthreads();

    // device counter (clamped to the arena capacity passed in n_rows_imm: an overflowing step is flagged, never read
    // past the arena) or the immediate count
    long long n_rows = n_rows_imm;
    if (n_rows_ptr) { const long long c = (long long)*n_rows_ptr; n_rows = c < n_rows_imm ? c : n_rows_imm; }
    if (rows_eval_counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(rows_eval_counter, (unsigned long long)n_rows);
    const long long n_tiles = (n_rows + 31) >> 5;
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const long long wave = (long long)blockIdx.x * (EVAL_THREADS / 64) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (EVAL_THREADS / 64);

    // sigmoid(a + b1) = 1 / (1 + 2^(a*(-log2 e) + b1*(-log2 e)))
    constexpr float NL2E = -1.44269504088896340736f;
    float b1s[4], w2v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { b1s[c] = NL2E * b1[32 * c + r]; w2v[c] = w2[32 * c + r]; }
    const float b2 = *b2p;

    uint32_t ksteps = 0;                                  // executed k-steps of this wave (roofline report)
    // software pipeline: the next tile's row is in flight while this tile is in the MFMA loop
    uint4 nx0 = make_uint4(0, 0, 0, 0), nx1 = make_uint4(0, 0, 0, 0);
    if (wave < n_tiles && wave * 32 + r < n_rows) { nx0 = rows[2 * (wave * 32 + r)]; nx1 = rows[2 * (wave * 32 + r) + 1]; }

    for (long long tile = wave; tile < n_tiles; tile += n_waves) {
        const uint32_t p[8] = {nx0.x, nx0.y, nx0.z, nx0.w, nx1.x, nx1.y, nx1.z, nx1.w};
        {
            const long long nrow = (tile + n_waves) * 32 + r;
            nx0 = make_uint4(0, 0, 0, 0); nx1 = make_uint4(0, 0, 0, 0);
            if (nrow < n_rows) { nx0 = rows[2 * nrow]; nx1 = rows[2 * nrow + 1]; }
        }
        RowDecode rd;
        decode_setup(p, h, rd);

        // Block sparsity: a k-step (2 features x 32 rows) whose A column is zero in EVERY row of the tile adds
        // exactly nothing, so it is skipped.  Rows of a tile come from one or two games and share most empty
        // points: typically ~40 of the 96 board k-steps survive.  The executed steps stay in ascending k, so every
        // row sees its non-zero terms in the same order as the dense chain -- results are bit-identical.
        //   even step (n>=1 | n>=2) of (point i, side q) is live iff some row has a checker there;
        //   odd  step (n>=3 | (n-3)/2)               is live iff some row has >= 3 there.
        uint32_t live[4];
        {
            const uint32_t b0a = p[0] & PTS, b1a = p[1] & PTS, b2a = p[2] & PTS, b3a = p[3] & PTS;
            const uint32_t b0b = p[4] & PTS, b1b = p[5] & PTS, b2b = p[6] & PTS, b3b = p[7] & PTS;
            live[0] = b0a | b1a | b2a | b3a;            // side 0 even
            live[1] = (b0a & b1a) | b2a | b3a;          // side 0 odd
            live[2] = b0b | b1b | b2b | b3b;            // side 1 even
            live[3] = (b0b & b1b) | b2b | b3b;          // side 1 odd
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) live[q] |= __shfl_xor(live[q], m, 64);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) live[q] = __builtin_amdgcn_readfirstlane(live[q]) >> 1;   // bit i = point i
            ksteps += (uint32_t)(__popc(live[0]) + __popc(live[1]) + __popc(live[2]) + __popc(live[3]) + 3);
        }

        floatx16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
#define BG_MFMA4(aval, WV)                                                          \
    {                                                                                \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32((aval), (WV).x, acc0, 0, 0, 0);    \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32((aval), (WV).y, acc1, 0, 0, 0);    \
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32((aval), (WV).z, acc2, 0, 0, 0);    \
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32((aval), (WV).w, acc3, 0, 0, 0);    \
    }
        __builtin_amdgcn_s_setprio(2);
        // B operands (W1 from LDS) of the NEXT point are always fetched while this point's MFMAs run (LDS
        // bandwidth is cheap; a skipped step only wastes its 1 KB read)
        const float4 *wp = sW + lane;
        float4 wa = wp[0 * 64], wb = wp[1 * 64], wc = wp[2 * 64], wd = wp[3 * 64];
#pragma unroll 2                                   // ping-pong the prefetch registers instead of moving them
        for (int i = 0; i < 24; ++i) {
            const int nb = (4 * i + 4) * 64;
            const float4 na = wp[nb], nbb = wp[nb + 64], nc = wp[nb + 128], nd = wp[i < 23 ? nb + 192 : nb + 128];
            // the cheap even-step operands are formed up front; the 9-instruction odd-step decode only where that
            // step is live (it rarely is: >= 3 checkers).  (Decoding the even steps inside their branches too
            // was measured 50 % SLOWER: the MFMA group then waits on its own operand.)
            const float a0 = decode_even(rd, 0, i);
            const float a2 = decode_even(rd, 1, i);
            if ((live[0] >> i) & 1u) BG_MFMA4(a0, wa);
            if ((live[1] >> i) & 1u) { const float a1 = decode_odd(rd, 0, i); BG_MFMA4(a1, wb); }
            if ((live[2] >> i) & 1u) BG_MFMA4(a2, wc);
            if ((live[3] >> i) & 1u) { const float a3 = decode_odd(rd, 1, i); BG_MFMA4(a3, wd); }
            wa = na; wb = nbb; wc = nc; wd = nd;
        }
        BG_MFMA4(rd.tail[0], wa);
        BG_MFMA4(rd.tail[1], wb);
        BG_MFMA4(rd.tail[2], wc);
#undef BG_MFMA4

        // epilogue: hidden sigmoid and the dot with W2 over this lane's 4 columns ...
        __builtin_amdgcn_s_setprio(0);
        if (HIDDEN) {
            float bb[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) bb[c] = b1[32 * c + r];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const long long orow = tile * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
                if (orow < n_rows) {
                    float *o = values + orow * N_HID + r;
                    o[0] = NL2E * (acc0[j] + bb[0]); o[32] = NL2E * (acc1[j] + bb[1]);
                    o[64] = NL2E * (acc2[j] + bb[2]); o[96] = NL2E * (acc3[j] + bb[3]);
                }
            }
            continue;
        }
        float part[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float s0 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc0[j], NL2E, b1s[0])));
            const float s1 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc1[j], NL2E, b1s[1])));
            const float s2 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc2[j], NL2E, b1s[2])));
            const float s3 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc3[j], NL2E, b1s[3])));
            part[j] = w2v[0] * s0 + w2v[1] * s1 + w2v[2] * s2 + w2v[3] * s3;
        }
        // ... then the sum over the 32 columns through a per-wave LDS transpose: lane (col r, half h)
        // holds rows (j&3) + 8*(j>>2) + 4h; scratch is [col][row] with a 33-float column stride
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) sRed[r * EVAL_RED_STRIDE + 8 * q + 4 * h + j] = part[4 * q + j];
        __builtin_amdgcn_wave_barrier();
        // lane (row r, half h) adds columns 16h .. 16h+15 of its row, halves are combined by one swap
        float sum = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c) sum += sRed[(16 * h + c) * EVAL_RED_STRIDE + r];
        sum += __shfl_xor(sum, 32, 64);
        __builtin_amdgcn_wave_barrier();
        {   // lanes 0..31 (h == 0) hold the 32 rows of the tile
            const long long orow = tile * 32 + r;
            const bool vrow = h == 0 && orow < n_rows;
            const float v = fast_sigmoid(sum + b2);
            if (vrow) values[orow] = v;
            if (info) {                                        // wave-uniform
                const uint2 inf = vrow ? info[orow] : make_uint2(0u, 0u);
                best_atomic_max(best, inf.x, v, inf.y, vrow, 32);
            }
        }
    }
    // one global atomic per block: a per-tile atomic on one address would serialise the whole launch (~10 ns each)
    __shared__ unsigned int s_ksteps;
    if (threadIdx.x == 0) s_ksteps = 0;
    __syncthreads();
    if (lane == 0 && ksteps) atomicAdd(&s_ksteps, ksteps);
    __syncthreads();
    if (ksteps_counter && threadIdx.x == 0 && s_ksteps) atomicAdd(ksteps_counter, (unsigned long long)s_ksteps);
}

// ================================ incremental fp32 evaluator ======================================
// Every afterstate of a turn differs from the turn's root position in the few points its <= 4 moves touched, and
// the encoder (model.py:111-144) is a thermometer code: a count that changes by one flips exactly one feature.
// So with  a_root = W1 x_root + b1  (one dense pass per GAME on the MFMA pipe: root_hidden_bf16x3_kernel below),
//     a_row = a_root + Σ_{changed features f} Δx_f · W1[:, f]        (typically 4-8 columns instead of 198)
// in fp32 FMAs; hidden sigmoid, W2 dot, output sigmoid and the per-game arg-max are as in the dense kernel.
// Lane = row, 32 hidden units at a time in registers, W1^T in LDS (132-float row stride), each lane's (row, Δ) list in LDS.
// The sum is the same real number as the dense chain with a different association: values agree to ~1e-7.
#ifndef BG_DELTA_THREADS
#define BG_DELTA_THREADS 1024
#endif
constexpr int DELTA_THREADS = BG_DELTA_THREADS;         // 16 waves per CU, one workgroup per CU (LDS: W1^T + the lists).  -DBG_DELTA_THREADS=512/768: the round-5
                                                        // co-residency experiment (2 / 3 waves per SIMD at 128 VGPRs: room for another kernel's workgroup on the CU)
#ifndef BG_DW_STRIDE
#define BG_DW_STRIDE 132
#endif
constexpr int DW_STRIDE = BG_DW_STRIDE;                 // floats per feature row of W1^T in LDS
// Row of feature f in the LDS table: 9 * point + (4 * side + level) for the board features, 216.. for the tail.  The
// LDS bank class of a 16-byte read is (row + chunk) mod 8; candidates of one game mostly differ in WHICH point a
// checker left or reached at the same thermometer level, and with row = f (= 8 * point + level) all those lanes
// hit the same banks (measured: half of all LDS cycles were conflict cycles).  9 * point spreads them.
#if defined(BG_ROW_LAYOUT) && BG_ROW_LAYOUT == 1            // experiment: level-major rows
constexpr int DW_ROWS = 198;
__host__ __device__ constexpr int delta_row(int f) { return f < 192 ? (f >> 3) + 24 * (f & 7) : f; }
__host__ __device__ constexpr int delta_row_board(int point, int k8) { return point + 24 * k8; }
#else
constexpr int DW_ROWS = 9 * 24 + 6;                     // 222
__host__ __device__ constexpr int delta_row(int f) { return f < 192 ? 9 * (f >> 3) + (f & 7) : 216 + (f - 192); }
__host__ __device__ constexpr int delta_row_board(int point, int k8) { return 9 * point + k8; }     // board features: no range test
#endif
constexpr int DELTA_W_FLOATS = DW_ROWS * DW_STRIDE;     // 29 304
constexpr int DELTA_MAX = 16;                           // <= 4 moves x (origin, destination, hit point, bar)
#ifndef BG_CTR_STRIDE
#define BG_CTR_STRIDE 16                                // words between two list counters (bg_staged.h: one 128-byte line each)
#endif
constexpr int DELTA_LDS_TOTAL = (DELTA_W_FLOATS + N_HID) * 4 + (DELTA_THREADS / 64) * DELTA_MAX * 64 * 2;   // 16-bit list entries

// W1^T for the incremental kernel, pre-multiplied by -log2(e) (the hidden sigmoid is then rcp(1 + exp2(a))); the rows of
// the borne-off counters (196, 197: x = n/15) also carry the 1/15, so every list entry is a small multiple of 1/2 --
// and every row carries that 1/2 (exact), so the kernel multiplies by the integer 2Δ
inline void relayout_w1_delta(const float *w1 /*[128][198]*/, float *wt /*[DW_ROWS][132]*/)
{
    const float NL2E = -1.44269504088896340736f;
    for (int i = 0; i < DELTA_W_FLOATS; ++i) wt[i] = 0.0f;
    for (int f = 0; f < N_IN; ++f)
        for (int n = 0; n < N_HID; ++n) {
            float w = NL2E * w1[n * N_IN + f];
            if (f >= 196) w = w / 15.0f;
            wt[delta_row(f) * DW_STRIDE + n] = 0.5f * w;     // list multipliers are 2Δ (integers): the 1/2 lives here, exactly
        }
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// One list entry for 32 hidden units: a[0..15] += d * W1^T[row][32c .. 32c+31], the 8 ds_read_b128 software-pipelined by
// hand (four reads in flight, FMAs issued as each lands) and the packed FMAs pinned IN PLACE.  Left to itself the
// compiler renamed the loop-carried accumulators every iteration and copied them back, and, short of registers at 4
// waves per SIMD, serialised the reads with full lgkmcnt(0) waits; the temporaries are the fixed registers v[108:127].
// (OFF = byte offset of the 32 units inside the 128-float row, an immediate of the reads; the entry's multiplier is the
// LOW half of d2, broadcast to both halves of the packed FMA by op_sel_hi:[0,1,1])
#define BG_DA_FMA(ACC, W) "v_pk_fma_f32 " ACC ", %16, " W ", " ACC " op_sel_hi:[0,1,1]\n\t"
#define BG_DA_BODY(O0, O1, O2, O3, O4, O5, O6, O7)                                                          \
        "ds_read_b128 v[108:111], %17 offset:" #O0 "\n\t"                                                  \
        "ds_read_b128 v[112:115], %17 offset:" #O1 "\n\t"                                                  \
        "ds_read_b128 v[116:119], %17 offset:" #O2 "\n\t"                                                  \
        "ds_read_b128 v[120:123], %17 offset:" #O3 "\n\t"                                                  \
        "ds_read_b128 v[124:127], %17 offset:" #O4 "\n\t"                                                  \
        "s_waitcnt lgkmcnt(4)\n\t" BG_DA_FMA("%0", "v[108:109]") BG_DA_FMA("%1", "v[110:111]")              \
        "ds_read_b128 v[108:111], %17 offset:" #O5 "\n\t"                                                  \
        "s_waitcnt lgkmcnt(4)\n\t" BG_DA_FMA("%2", "v[112:113]") BG_DA_FMA("%3", "v[114:115]")              \
        "ds_read_b128 v[112:115], %17 offset:" #O6 "\n\t"                                                  \
        "s_waitcnt lgkmcnt(4)\n\t" BG_DA_FMA("%4", "v[116:117]") BG_DA_FMA("%5", "v[118:119]")              \
        "ds_read_b128 v[116:119], %17 offset:" #O7 "\n\t"                                                  \
        "s_waitcnt lgkmcnt(4)\n\t" BG_DA_FMA("%6", "v[120:121]") BG_DA_FMA("%7", "v[122:123]")              \
        "s_waitcnt lgkmcnt(3)\n\t" BG_DA_FMA("%8", "v[124:125]") BG_DA_FMA("%9", "v[126:127]")              \
        "s_waitcnt lgkmcnt(2)\n\t" BG_DA_FMA("%10", "v[108:109]") BG_DA_FMA("%11", "v[110:111]")            \
        "s_waitcnt lgkmcnt(1)\n\t" BG_DA_FMA("%12", "v[112:113]") BG_DA_FMA("%13", "v[114:115]")            \
        "s_waitcnt lgkmcnt(0)\n\t" BG_DA_FMA("%14", "v[116:117]") "v_pk_fma_f32 %15, %16, v[118:119], %15 op_sel_hi:[0,1,1]"
template <int OFF>
__device__ __forceinline__ void delta_apply_32(f32x2_t (&a)[16], f32x2_t d2, uint32_t lds_addr)
{
    static_assert(OFF == 0 || OFF == 128, "two halves of a 64-unit pass");
#define BG_DA_OPS                                                                                                  \
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]),  \
          "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15])                 \
        : "v"(d2), "v"(lds_addr)                                                                                   \
        : "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", \
          "v122", "v123", "v124", "v125", "v126", "v127", "memory"
    if (OFF == 0) asm volatile(BG_DA_BODY(0, 16, 32, 48, 64, 80, 96, 112) BG_DA_OPS);
    else asm volatile(BG_DA_BODY(128, 144, 160, 176, 192, 208, 224, 240) BG_DA_OPS);
#undef BG_DA_OPS
}
#undef BG_DA_BODY
#undef BG_DA_FMA

// list entry (feature row | multiplier m as int8 << 8) -> LDS address of the row's 64-unit half and m as a float in
// the low half of d2 (the high half is never read): three VALU operations
__device__ __forceinline__ void delta_entry(uint32_t ent, uint32_t base, f32x2_t &d2, uint32_t &addr)
{
    float d;
    asm("v_cvt_f32_i32_sdwa %0, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n\t"
        "v_mul_u32_u24_sdwa %1, %3, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_add_u32_e32 %1, %4, %1"
        : "=&v"(d), "=&v"(addr)
        : "v"(ent), "s"((uint32_t)(DW_STRIDE * 4)), "s"(base));
    d2.x = d;
}

// sum += w · 1 / (1 + 2^a) for two hidden units (a already carries the -log2 e)
__device__ __forceinline__ f32x2_t delta_sigmoid_pair(f32x2_t a, f32x2_t w, f32x2_t sum)
{
    f32x2_t e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    e = e + (f32x2_t){1.0f, 1.0f};
    const f32x2_t r = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
    return __builtin_elementwise_fma(w, r, sum);     // (not inline asm: a VALU read right after a transcendental write
                                                     //  needs a wait state on gfx950, which the compiler inserts)
}

#if BG_DELTA_THREADS == 1024
#define BG_DELTA_BOUNDS __launch_bounds__(DELTA_THREADS)
#else
#define BG_DELTA_BOUNDS __launch_bounds__(DELTA_THREADS, 4)       /* still 128 VGPRs: the freed half of the register file is for the neighbour */
#endif
__global__ BG_DELTA_BOUNDS void eval_rows_delta_kernel(
    const uint4 *__restrict__ rows, const unsigned long long *__restrict__ n_rows_ptr, long long n_rows_imm,
    unsigned long long *__restrict__ rows_eval_counter, const float4 *__restrict__ wt, const float *__restrict__ w2,
    const float *__restrict__ b2p, const uint4 *__restrict__ root_rows, const float *__restrict__ root_hidden,
    float *__restrict__ values, const uint2 *__restrict__ info, unsigned long long *__restrict__ best,
    unsigned long long *__restrict__ delta_counter, unsigned long long *__restrict__ zero_words, int n_zero_words,
    unsigned long long *__restrict__ err_word, unsigned long long err_bit, const unsigned long long *__restrict__ n_rows_b_ptr, long long b_base)
{
    // multi-step runs: the OTHER set of list counters is cleared here, while no kernel is using it, for the roots of
    // the next step (which share a launch with this step's apply)
    if (zero_words && blockIdx.x == 0 && (int)threadIdx.x < n_zero_words) zero_words[threadIdx.x] = 0ull;
#ifdef BG_EVAL_WGCLOCK
    const unsigned long long wg_t0 = wall_clock64();
#endif
    extern __shared__ float4 sW[];                       // [DW_ROWS][33] float4, then w2[128], then the lists
    float *sW2 = reinterpret_cast<float *>(sW) + DELTA_W_FLOATS;
    uint16_t *sList = reinterpret_cast<uint16_t *>(sW2 + N_HID) + (threadIdx.x >> 6) * (DELTA_MAX * 64);
    for (int i = threadIdx.x; i < DELTA_W_FLOATS / 4; i += DELTA_THREADS) sW[i] = wt[i];
    if (threadIdx.x < N_HID) sW2[threadIdx.x] = w2[threadIdx.x];
    __shared__ unsigned int s_ticket;                    // tiles of this workgroup's range are handed out on demand
    if (threadIdx.x == 0) s_ticket = 0;
    __syncthreads();

    // device counter (clamped to the arena capacity passed in n_rows_imm: an overflowing step is flagged, never read
    // past the arena) or the immediate count
    // Four arenas (n_rows_b_ptr, the greedy step): the leaf stage sorts its rows by the turn's kind (non-doubles / doubles) and by whether
    // their moves hit a blot -- what decides the length of a row's delta list -- into arena k = rows [k b_base, k b_base + n_k).  The tiles
    // of the arenas are numbered one arena after the other, and the strided share b, b + G, ... then gives every workgroup the same number
    // of tiles of EACH kind (+- 1): a tile costs what its longest list costs (4 gather passes for hit-free non-doubles rows, 10 and more
    // for doubles rows with hits), and the launch ends with its slowest workgroup (shares of one mixed arena: 4-7 us behind the mean of 68).
    long long n_rows = n_rows_imm, na1 = 0, na2 = 0, na3 = 0;
    if (n_rows_ptr) {
        const long long lim = n_rows_b_ptr ? b_base : n_rows_imm;
        const long long c = (long long)*n_rows_ptr;
        n_rows = c < lim ? c : lim;
    }
    if (n_rows_b_ptr) {
        const long long c1 = (long long)n_rows_b_ptr[0], c2 = (long long)n_rows_b_ptr[BG_CTR_STRIDE], c3 = (long long)n_rows_b_ptr[2 * BG_CTR_STRIDE];
        na1 = c1 < b_base ? c1 : b_base; na2 = c2 < b_base ? c2 : b_base; na3 = c3 < b_base ? c3 : b_base;
    }
    if (rows_eval_counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(rows_eval_counter, (unsigned long long)(n_rows + na1 + na2 + na3));
    // the order in which a workgroup meets the arenas (BG_ARENA_ORDER, four digits: 3210 = arena 3 first)
#ifndef BG_ARENA_ORDER
#define BG_ARENA_ORDER 123
#endif
    constexpr int AO0 = (BG_ARENA_ORDER / 1000) % 10, AO1 = (BG_ARENA_ORDER / 100) % 10, AO2 = (BG_ARENA_ORDER / 10) % 10, AO3 = BG_ARENA_ORDER % 10;
    static_assert(((1 << AO0) | (1 << AO1) | (1 << AO2) | (1 << AO3)) == 15, "a permutation of 0123");
    const long long cnt_of[4] = {n_rows, na1, na2, na3};
    const long long c0 = cnt_of[AO0], c1 = cnt_of[AO1], c2 = cnt_of[AO2], c3 = cnt_of[AO3];
    const long long t1 = (c0 + 63) >> 6, t2 = t1 + ((c1 + 63) >> 6), t3 = t2 + ((c2 + 63) >> 6);
    const long long n_tiles = t3 + ((c3 + 63) >> 6);
    // first row of a tile and the end of the rows of its arena
    auto tile_rows = [&](long long t, long long &end) -> long long {
        if (t < t1) { end = AO0 * b_base + c0; return AO0 * b_base + t * 64; }
        if (t < t2) { end = AO1 * b_base + c1; return AO1 * b_base + (t - t1) * 64; }
        if (t < t3) { end = AO2 * b_base + c2; return AO2 * b_base + (t - t2) * 64; }
        end = AO3 * b_base + c3;
        return AO3 * b_base + (t - t3) * 64;
    };
    const int lane = threadIdx.x & 63;
    // Row -> lane: 16 CONSECUTIVE rows of a tile sit on the 16 lanes the LDS serves together for a ds_read_b128 (the lane groups
    // are {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32: MI355X_MICROARCH.md).  Neighbouring rows are siblings of one
    // turn and ask for the same or for neighbouring W1^T rows, which broadcast or fall into different banks; rows of different
    // games collide at random.  rl = the lane's row inside its tile, inv = the lane that holds row `lane` (same-box A/B against
    // row = lane: value net 0.0808 -> 0.0796 ms).
    int rl, inv;
    {
        const int hl = lane & 31;
        const int g = (hl < 4 || (hl >= 12 && hl < 16) || (hl >= 20 && hl < 28)) ? 0 : 1;
        const int k = hl < 4 ? hl : hl < 12 ? hl - 4 : hl < 16 ? hl - 8 : hl < 20 ? hl - 8 : hl < 28 ? hl - 12 : hl - 16;
        rl = (lane & 32) + 16 * g + k;
        const int R = lane, gg = (R >> 4) & 1, kk = R & 15;
        inv = (R & 32) + (gg == 0 ? (kk < 4 ? kk : kk < 8 ? 8 + kk : 12 + kk) : (kk < 8 ? 4 + kk : kk < 12 ? 8 + kk : 16 + kk));
    }
    // A tile's cost follows its longest delta list (a doubles turn costs twice a plain one), so the 64-row tiles of a
    // workgroup's contiguous range go to whichever wave is free (LDS ticket) instead of a fixed stride per wave
    // (tiles of a workgroup: b, b + G, b + 2G, ... -- the arena holds runs of light rows (non-doubles turns) and runs of
    // heavy ones (doubles), a strided share gives every workgroup the same mix while each tile stays homogeneous)
    const long long t_hi = n_tiles;
    auto grab = [&]() -> long long {
        unsigned int t = 0;
        if (lane == 0) t = atomicAdd(&s_ticket, 1u);
        return (long long)blockIdx.x + (long long)__builtin_amdgcn_readfirstlane(t) * (long long)gridDim.x;
    };
    const float b2 = *b2p;
    constexpr float NL2E = -1.44269504088896340736f;
    uint32_t n_delta = 0;
    bool list_overflow = false;

    // software pipeline: the next tile's row + info are in flight while this tile is computed
    uint4 nx0 = make_uint4(0, 0, 0, 0), nx1 = make_uint4(0, 0, 0, 0);
    uint2 nxi = make_uint2(0u, 0u);
    long long tile = grab();
    long long row_end = 0, row_cur = tile_rows(tile, row_end) + rl;        // this tile's row of the lane, the end of its arena's rows
    if (tile < t_hi && row_cur < row_end) { nx0 = rows[2 * row_cur]; nx1 = rows[2 * row_cur + 1]; nxi = info[row_cur]; }
    // ... and so is its game's root row (issued during the last pass of the tile before)
    uint4 nr0 = root_rows[2 * (long long)nxi.x], nr1 = root_rows[2 * (long long)nxi.x + 1];
    while (tile < t_hi) {
        const long long next_tile = grab();
        const long long row = row_cur;
        const bool valid = row < row_end;
        const uint2 inf = nxi;                             // (0, 0) for a padding lane: game 0, harmless
        const uint4 r0 = nr0, r1 = nr1;
        const f32x4_t *ah = reinterpret_cast<const f32x4_t *>(root_hidden + (long long)inf.x * N_HID);
        const uint32_t p[8] = {nx0.x & ~TURN_BIT, nx0.y, nx0.z, nx0.w, nx1.x, nx1.y, nx1.z, nx1.w};
        const uint32_t mover = (nx0.x & TURN_BIT) ? 1u : 0u;       // the row carries the turn bit of the side that moved (model.py:209)
        const uint32_t q[8] = {valid ? r0.x & ~TURN_BIT : 0u, valid ? r0.y : 0u, valid ? r0.z : 0u, valid ? r0.w : 0u,
                               valid ? r1.x : 0u, valid ? r1.y : 0u, valid ? r1.z : 0u, valid ? r1.w : 0u};
        {
            const long long nrow = tile_rows(next_tile, row_end) + rl;
            row_cur = nrow;
            nx0 = make_uint4(0, 0, 0, 0); nx1 = make_uint4(0, 0, 0, 0); nxi = make_uint2(0u, 0u);
            if (next_tile < t_hi && nrow < row_end) { nx0 = rows[2 * nrow]; nx1 = rows[2 * nrow + 1]; nxi = info[nrow]; }
        }

        // ---- this lane's (feature, Δ) list: 16-bit entry = feature | (2Δ as int8) << 8; TYPE 0: Δ = m, 1: Δ = m/2,
        //      2: Δ = m (the W1^T row is pre-divided by 15)
        uint32_t cnt = 0;
        uint16_t *lst = sList + lane;
#pragma unroll
        for (int e = 0; e < DELTA_MAX; ++e) lst[e * 64] = 0;       // entry 0 = "add 0 x row 0": the apply loop reads blindly
#if defined(BG_ABL) && (BG_ABL & 1)
        if (p[0] == 0xFFFFFFFFu)                                    // ablation: no list construction (never true)
#endif
        {
#define BG_PUSH(F, TYPE, M)                                                                          \
    {                                                                                                \
        if (cnt < DELTA_MAX) lst[cnt * 64] = (uint16_t)((uint32_t)(F) | ((uint32_t)(((TYPE) == 1 ? 1 : 2) * (M)) & 255u) << 8); \
        ++cnt;                                                                                       \
    }
#if !defined(BG_DECODE_MOVER_FIRST) || BG_DECODE_MOVER_FIRST
        // The MOVER's side first, then the other one -- which differs from the root only where a blot was hit, i.e. in no lane of a tile of the
        // hit-free arenas (two thirds of the rows): the second pass is skipped there for the whole wave (one ballot).
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const uint32_t sd = mover ^ (uint32_t)pass;        // per lane: 0 = PLAYER1's planes, 1 = PLAYER2's
            const uint32_t b0 = sd ? p[4] : p[0], b1 = sd ? p[5] : p[1], b2 = sd ? p[6] : p[2], b3 = sd ? p[7] : p[3];
            const uint32_t c0 = sd ? q[4] : q[0], c1 = sd ? q[5] : q[1], c2 = sd ? q[6] : q[2], c3 = sd ? q[7] : q[3];
            const uint32_t diff = (b0 ^ c0) | (b1 ^ c1) | (b2 ^ c2) | (b3 ^ c3);
            if (pass == 1 && __ballot(diff != 0u) == 0ull) break;
            const uint32_t ge_new[3] = {b0 | b1 | b2 | b3, b1 | b2 | b3, (b0 & b1) | b2 | b3};
            const uint32_t ge_old[3] = {c0 | c1 | c2 | c3, c1 | c2 | c3, (c0 & c1) | c2 | c3};
            const uint32_t side4 = 4u * sd;
#pragma unroll
            for (int k = 0; k < 3; ++k) {               // thermometer features n>=1, n>=2, n>=3: ±1 where the bit flips
                uint32_t x = (ge_new[k] ^ ge_old[k]) & PTS;
                while (x) {
                    const int pos = __ffs(x) - 1; x &= x - 1;
                    BG_PUSH(delta_row_board(pos - 1, 0) + side4 + (uint32_t)k, 0, ((ge_new[k] >> pos) & 1u) ? 1 : -1);
                }
            }
            uint32_t x4 = diff & PTS & ((b2 | b3) | (c2 | c3)); // (n-3)/2 can only move where n >= 4 before or after
            const Side sn{{b0, b1, b2, b3}}, so{{c0, c1, c2, c3}};
            while (x4) {
                const int pos = __ffs(x4) - 1; x4 &= x4 - 1;
                const int n1 = count_at(sn, pos), n0 = count_at(so, pos);
                const int d = (n1 > 3 ? n1 - 3 : 0) - (n0 > 3 ? n0 - 3 : 0);
                if (d) BG_PUSH(delta_row_board(pos - 1, 0) + side4 + 3u, 1, d);
            }
            // bar and borne-off counters: PLAYER1 bar = pos 0 (feature 194, x = n/2), off = pos 25 (196, n/15); PLAYER2 bar = pos 25 (195), off = pos 0 (197)
            if (diff & 1u) {
                const int d = count_at(sn, 0) - count_at(so, 0);
                BG_PUSH(sd ? (uint32_t)delta_row(197) : (uint32_t)delta_row(194), 1, sd ? 2 * d : d);
            }
            if (diff & (1u << 25)) {
                const int d = count_at(sn, 25) - count_at(so, 25);
                BG_PUSH(sd ? (uint32_t)delta_row(195) : (uint32_t)delta_row(196), 1, sd ? d : 2 * d);
            }
        }
#else
#pragma unroll
        for (int sd = 0; sd < 2; ++sd) {
            const uint32_t b0 = p[4 * sd], b1 = p[4 * sd + 1], b2 = p[4 * sd + 2], b3 = p[4 * sd + 3];
            const uint32_t c0 = q[4 * sd], c1 = q[4 * sd + 1], c2 = q[4 * sd + 2], c3 = q[4 * sd + 3];
            const uint32_t diff = (b0 ^ c0) | (b1 ^ c1) | (b2 ^ c2) | (b3 ^ c3);
            const uint32_t ge_new[3] = {b0 | b1 | b2 | b3, b1 | b2 | b3, (b0 & b1) | b2 | b3};
            const uint32_t ge_old[3] = {c0 | c1 | c2 | c3, c1 | c2 | c3, (c0 & c1) | c2 | c3};
#pragma unroll
            for (int k = 0; k < 3; ++k) {               // thermometer features n>=1, n>=2, n>=3: ±1 where the bit flips
                uint32_t x = (ge_new[k] ^ ge_old[k]) & PTS;
                while (x) {
                    const int pos = __ffs(x) - 1; x &= x - 1;
                    BG_PUSH(delta_row_board(pos - 1, 4 * sd + k), 0, ((ge_new[k] >> pos) & 1u) ? 1 : -1);
                }
            }
            uint32_t x4 = diff & PTS & ((b2 | b3) | (c2 | c3)); // (n-3)/2 can only move where n >= 4 before or after
            const Side sn{{b0, b1, b2, b3}}, so{{c0, c1, c2, c3}};
            while (x4) {
                const int pos = __ffs(x4) - 1; x4 &= x4 - 1;
                const int n1 = count_at(sn, pos), n0 = count_at(so, pos);
                const int d = (n1 > 3 ? n1 - 3 : 0) - (n0 > 3 ? n0 - 3 : 0);
                if (d) BG_PUSH(delta_row_board(pos - 1, 4 * sd + 3), 1, d);
            }
            // bar and borne-off counters: PLAYER1 bar = pos 0, off = pos 25; PLAYER2 bar = pos 25, off = pos 0
            if (diff & 1u) {
                const int d = count_at(sn, 0) - count_at(so, 0);
                BG_PUSH(delta_row(sd == 0 ? 194 : 197), sd == 0 ? 1 : 2, d);
            }
            if (diff & (1u << 25)) {
                const int d = count_at(sn, 25) - count_at(so, 25);
                BG_PUSH(delta_row(sd == 0 ? 196 : 195), sd == 0 ? 2 : 1, d);
            }
        }
#endif
        }
#undef BG_PUSH
        // A legal turn changes at most 13 features (4 origins + 4 landing points of the mover, 4 hit points + the bar
        // counter of the opponent; test_delta_list_worst_case builds that turn).  More than DELTA_MAX means the row is
        // not an afterstate of its root: flagged (BGAMD_E_DELTA), and the list is cut so that the reads stay in bounds.
        if (cnt > DELTA_MAX) { cnt = DELTA_MAX; list_overflow = true; }
        n_delta += cnt;
        uint32_t maxcnt = 0;                               // wave-uniform trip count of the apply loops: max over the lanes,
#pragma unroll                                             // built bit by bit from ballots (no cross-lane data movement)
        for (int b = 4; b >= 0; --b) {
            const uint32_t t = maxcnt | (1u << b);
            if (__ballot(cnt >= t) != 0ull) maxcnt = t;
        }
#if defined(BG_ABL) && (BG_ABL & 2)
        maxcnt = maxcnt > 100 ? 1 : 0;                             // ablation: no gathers
#endif
        __builtin_amdgcn_wave_barrier();

        // ---- 64 hidden units at a time (two halves of 32 pairs): root term, a += Δ · W1[:, f] over the lane's entries (a
        //      lane that has run out reads a zero entry = adds 0 x row 0: the FMAs stay unconditional and in place),
        //      hidden sigmoid, partial dot with W2.  (Four passes of 32 units decoded every list entry four times.)
        f32x2_t sum2 = {0.0f, 0.0f};
        const f32x4_t *w2v = reinterpret_cast<const f32x4_t *>(sW2);
        const uint32_t sW_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float4 *)sW;   // LDS byte address
#pragma unroll 1
        for (int c = 0; c < 2; ++c) {
            f32x2_t a[16], a2[16];
#if defined(BG_ABL) && (BG_ABL & 8)
#pragma unroll
            for (int j = 0; j < 16; ++j) { a[j] = (f32x2_t){0.25f * (float)j, 0.5f}; a2[j] = (f32x2_t){0.125f, -0.25f * (float)j}; }   // ablation: no root-term loads
            (void)ah;
#else
#pragma unroll
            for (int j = 0; j < 8; ++j) { const f32x4_t t = ah[16 * c + j]; a[2 * j] = t.lo; a[2 * j + 1] = t.hi; }
#pragma unroll
            for (int j = 0; j < 8; ++j) { const f32x4_t t = ah[16 * c + 8 + j]; a2[2 * j] = t.lo; a2[2 * j + 1] = t.hi; }
#endif
            if (c == 1) { nr0 = root_rows[2 * (long long)nxi.x]; nr1 = root_rows[2 * (long long)nxi.x + 1]; }   // next tile's root row
            uint32_t ent = lst[0];
            for (uint32_t e = 0; e < maxcnt; ++e) {
                const uint32_t nent = lst[(e + 1 < DELTA_MAX ? e + 1 : e) * 64];      // next entry: its LDS latency hides here
                f32x2_t d2;
                uint32_t addr;
                delta_entry(ent, sW_lds + 256u * (uint32_t)c, d2, addr);
                delta_apply_32<0>(a, d2, addr);
                delta_apply_32<128>(a2, d2, addr);
                ent = nent;
            }
            // hidden sigmoids two units at a time: the "1 +" and the "· W2, +=" are packed fp32 operations
#if defined(BG_ABL) && (BG_ABL & 4)
#define BG_SIGP(A, W, S) __builtin_elementwise_fma(W, A, S)       /* ablation: no transcendentals */
#else
#define BG_SIGP(A, W, S) delta_sigmoid_pair(A, W, S)
#endif
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4_t w = w2v[16 * c + j];
                sum2 = BG_SIGP(a[2 * j], w.lo, sum2);
                sum2 = BG_SIGP(a[2 * j + 1], w.hi, sum2);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4_t w = w2v[16 * c + 8 + j];
                sum2 = BG_SIGP(a2[2 * j], w.lo, sum2);
                sum2 = BG_SIGP(a2[2 * j + 1], w.hi, sum2);
            }
#undef BG_SIGP
        }
        const float sum = sum2.x + sum2.y;
        __builtin_amdgcn_wave_barrier();
        {
            const float v = fast_sigmoid(sum + b2);
            if (valid) values[row] = v;
            // the per-game maximum runs over neighbouring ROWS: back to row order first
            best_atomic_max(best, __shfl(inf.x, inv, 64), __shfl(v, inv, 64), __shfl(inf.y, inv, 64), __shfl((int)valid, inv, 64) != 0, 64);
        }
        tile = next_tile;
    }
    __shared__ unsigned int s_nd;
    if (threadIdx.x == 0) s_nd = 0;
    __syncthreads();
    {
        uint32_t t = n_delta;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) t += __shfl_xor(t, m, 64);
        if (lane == 0 && t) atomicAdd(&s_nd, t);
    }
    __syncthreads();
    if (delta_counter && threadIdx.x == 0 && s_nd) atomicAdd(delta_counter, (unsigned long long)s_nd);
    if (err_word && __ballot(list_overflow) != 0ull && lane == 0) atomicOr(err_word, err_bit);
#ifdef BG_EVAL_WGCLOCK                                         // diagnostic build (tools/eval_wg_clock.py): when each workgroup ended, in us after it started,
    if (n_rows_ptr && threadIdx.x == 0) {                      // and how many tiles it worked, at the top end of the value array
        values[n_rows_imm - 1 - (long long)blockIdx.x] = (float)(wall_clock64() - wg_t0) * 0.01f;
        values[n_rows_imm - 1 - (long long)gridDim.x - (long long)blockIdx.x] = (float)s_ticket;
    }
#endif
}

// ================================ bf16 speed mode ================================================
// v_mfma_f32_32x32x16_bf16, fp32 accumulate.  K is laid out so that one K-step (16 features) is two
// board points: lane (row r, half h) supplies the 8 features of point 2s+h (PLAYER1's 4, PLAYER2's 4),
// fetched as two 64-bit LUT entries indexed by the checker counts; step 12 carries features 192..197.
// Every feature value except off/15 is exact in bf16; W1 is rounded to bf16 (RNE) on the host.
// NOT a parity mode: |Δvalue| vs fp32 is ~3e-4 (SURVEY hard part 3); judged by agreement rate.
constexpr int K16_STEPS = 13;
constexpr int EVAL16_W_BYTES = K16_STEPS * 4 * 64 * 16;             // 53 248
constexpr int EVAL16_LUT_BYTES = 16 * 8;
constexpr int EVAL16_LDS_TOTAL = EVAL16_W_BYTES + EVAL16_LUT_BYTES + (EVAL_THREADS / 64) * EVAL_RED_FLOATS * 4;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ inline uint16_t f32_to_bf16_rne(float f)
{
    union { float f; uint32_t u; } c{f};
    return (uint16_t)((c.u + 0x7FFFu + ((c.u >> 16) & 1u)) >> 16);
}

// Wl16[s][c][l][j] = bf16(W1[32c + (l&31)][16s + 8(l>>5) + j]), zero beyond feature 197
inline void relayout_w1_bf16(const float *w1, uint16_t *wl)
{
    for (int s = 0; s < K16_STEPS; ++s)
        for (int c = 0; c < 4; ++c)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int f = 16 * s + 8 * (l >> 5) + j;
                    wl[((s * 4 + c) * 64 + l) * 8 + j] = f < N_IN ? f32_to_bf16_rne(w1[(32 * c + (l & 31)) * N_IN + f]) : 0;
                }
}
// LUT[n] = bf16 x4 {n>=1, n>=2, n>=3, (n-3)/2 if n>=4}
inline void make_count_lut(uint32_t *lut /*[16][2]*/)
{
    for (int n = 0; n < 16; ++n) {
        const uint16_t e0 = n >= 1 ? 0x3F80 : 0, e1 = n >= 2 ? 0x3F80 : 0, e2 = n >= 3 ? 0x3F80 : 0;
        const uint16_t e3 = n >= 4 ? f32_to_bf16_rne(0.5f * (float)(n - 3)) : 0;
        lut[2 * n] = (uint32_t)e0 | ((uint32_t)e1 << 16);
        lut[2 * n + 1] = (uint32_t)e2 | ((uint32_t)e3 << 16);
    }
}

__global__ __launch_bounds__(EVAL_THREADS) void eval_rows_bf16_kernel(
    const uint4 *__restrict__ rows, const unsigned long long *__restrict__ n_rows_ptr, long long n_rows_imm,
    unsigned long long *__restrict__ rows_eval_counter, const uint4 *__restrict__ wl16, const uint2 *__restrict__ lut,
    const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ b2p, float *__restrict__ values,
    const uint2 *__restrict__ info, unsigned long long *__restrict__ best)
{
    extern __shared__ uint4 sW16[];
    uint2 *sLut = reinterpret_cast<uint2 *>(sW16 + K16_STEPS * 4 * 64);
    float *sRed = reinterpret_cast<float *>(sLut + 16) + (threadIdx.x >> 6) * EVAL_RED_FLOATS;
    for (int i = threadIdx.x; i < K16_STEPS * 4 * 64; i += EVAL_THREADS) sW16[i] = wl16[i];
    if (threadIdx.x < 16) sLut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();

    // device counter (clamped to the arena capacity passed in n_rows_imm: an overflowing step is flagged, never read
    // past the arena) or the immediate count
    long long n_rows = n_rows_imm;
    if (n_rows_ptr) { const long long c = (long long)*n_rows_ptr; n_rows = c < n_rows_imm ? c : n_rows_imm; }
    if (rows_eval_counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(rows_eval_counter, (unsigned long long)n_rows);
    const long long n_tiles = (n_rows + 31) >> 5;
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const long long wave = (long long)blockIdx.x * (EVAL_THREADS / 64) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (EVAL_THREADS / 64);
    constexpr float NL2E = -1.44269504088896340736f;
    float b1s[4], w2v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { b1s[c] = NL2E * b1[32 * c + r]; w2v[c] = w2[32 * c + r]; }
    const float b2 = *b2p;

    uint4 nx0 = make_uint4(0, 0, 0, 0), nx1 = make_uint4(0, 0, 0, 0);
    if (wave < n_tiles && wave * 32 + r < n_rows) { nx0 = rows[2 * (wave * 32 + r)]; nx1 = rows[2 * (wave * 32 + r) + 1]; }

    for (long long tile = wave; tile < n_tiles; tile += n_waves) {
        const uint32_t p[8] = {nx0.x, nx0.y, nx0.z, nx0.w, nx1.x, nx1.y, nx1.z, nx1.w};
        {
            const long long nrow = (tile + n_waves) * 32 + r;
            nx0 = make_uint4(0, 0, 0, 0); nx1 = make_uint4(0, 0, 0, 0);
            if (nrow < n_rows) { nx0 = rows[2 * nrow]; nx1 = rows[2 * nrow + 1]; }
        }
        const Side sa{{p[0], p[1], p[2], p[3]}}, sb{{p[4], p[5], p[6], p[7]}};
        floatx16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
        const uint4 *wp = sW16 + lane;
#pragma unroll 4
        for (int s = 0; s < 12; ++s) {
            const int pos = 2 * s + h + 1;                       // board point 2s+h  ->  bit position +1
            const uint2 l0 = sLut[count_at(sa, pos)], l1 = sLut[count_at(sb, pos)];
            union { uint4 u; bf16x8 v; } a, w0, w1, w2r, w3;
            a.u = make_uint4(l0.x, l0.y, l1.x, l1.y);
            w0.u = wp[(s * 4 + 0) * 64]; w1.u = wp[(s * 4 + 1) * 64]; w2r.u = wp[(s * 4 + 2) * 64]; w3.u = wp[(s * 4 + 3) * 64];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, w0.v, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, w1.v, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, w2r.v, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, w3.v, acc3, 0, 0, 0);
        }
        {   // step 12: features 192..197 on the h == 0 lanes
            const int turn = (p[0] & TURN_BIT) ? 1 : 0;
            const uint32_t t0 = turn == 0 ? 0x3F80u : 0u, t1 = turn == 0 ? 0u : 0x3F80u;
            const uint32_t bar1 = f32_to_bf16_rne(0.5f * (float)count_at(sa, 0)), bar2 = f32_to_bf16_rne(0.5f * (float)count_at(sb, 25));
            const uint32_t off1 = f32_to_bf16_rne((float)count_at(sa, 25) / 15.0f), off2 = f32_to_bf16_rne((float)count_at(sb, 0) / 15.0f);
            union { uint4 u; bf16x8 v; } a, w0, w1, w2r, w3;
            a.u = h ? make_uint4(0, 0, 0, 0) : make_uint4(t0 | (t1 << 16), bar1 | (bar2 << 16), off1 | (off2 << 16), 0u);
            w0.u = wp[(12 * 4 + 0) * 64]; w1.u = wp[(12 * 4 + 1) * 64]; w2r.u = wp[(12 * 4 + 2) * 64]; w3.u = wp[(12 * 4 + 3) * 64];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, w0.v, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, w1.v, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, w2r.v, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, w3.v, acc3, 0, 0, 0);
        }
        float part[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float s0 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc0[j], NL2E, b1s[0])));
            const float s1 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc1[j], NL2E, b1s[1])));
            const float s2 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc2[j], NL2E, b1s[2])));
            const float s3 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc3[j], NL2E, b1s[3])));
            part[j] = w2v[0] * s0 + w2v[1] * s1 + w2v[2] * s2 + w2v[3] * s3;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) sRed[r * EVAL_RED_STRIDE + 8 * q + 4 * h + j] = part[4 * q + j];
        __builtin_amdgcn_wave_barrier();
        float sum = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c) sum += sRed[(16 * h + c) * EVAL_RED_STRIDE + r];
        sum += __shfl_xor(sum, 32, 64);
        __builtin_amdgcn_wave_barrier();
        {   // lanes 0..31 (h == 0) hold the 32 rows of the tile
            const long long orow = tile * 32 + r;
            const bool vrow = h == 0 && orow < n_rows;
            const float v = fast_sigmoid(sum + b2);
            if (vrow) values[orow] = v;
            if (info) {                                        // wave-uniform
                const uint2 inf = vrow ? info[orow] : make_uint2(0u, 0u);
                best_atomic_max(best, inf.x, v, inf.y, vrow, 32);
            }
        }
    }
}

// ================================ bf16 x 3 split: the root term of the incremental evaluator =====================
// An fp32 weight is EXACTLY hi + mid + lo with three bf16 terms (8 + 8 + 8 significand bits, fp32's exponent range),
// and every encoder feature is exact in bf16 once the borne-off counters are fed as integers (their W1 rows carry
// the 1/15).  So  W1 x  is three v_mfma_f32_32x32x16_bf16 per (K-step, column tile) with exact products and fp32
// accumulation -- fp32-grade results at 3/16 of the fp32 MFMA's cost.  Used for the one dense pass per GAME
// (root positions); output = -log2(e) (W1 x + b1), what eval_rows_delta_kernel starts from.
// The three weight planes (160 KB) do not fit LDS together: K is staged in two phases (7 + 6 K-steps).
constexpr int ROOT3_THREADS = 512;
#ifndef BG_ROOT3_PHASE_STEPS
#define BG_ROOT3_PHASE_STEPS 7
#endif
constexpr int ROOT3_PHASE_STEPS = BG_ROOT3_PHASE_STEPS;     // K-steps staged in LDS at a time (13 = all three planes at once: 159 872 B)
constexpr int ROOT3_PHASES = (K16_STEPS + ROOT3_PHASE_STEPS - 1) / ROOT3_PHASE_STEPS;
constexpr int ROOT3_PART_U4 = K16_STEPS * 4 * 64;                                  // uint4 per weight plane
constexpr int ROOT3_LDS_TOTAL = 3 * ROOT3_PHASE_STEPS * 4 * 64 * 16 + EVAL16_LUT_BYTES;   // 86 144

// Wl[part][s][c][l][j]; tail step = [turn0, turn1, bar1/2, bar2/2, off1, off2, 0, 0] with rows 196/197 divided by 15
inline void relayout_w1_bf16x3(const float *w1, uint16_t *wl)
{
    static const int tail_map[8] = {192, 193, 194, 195, 196, 197, -1, -1};
    for (int s = 0; s < K16_STEPS; ++s)
        for (int c = 0; c < 4; ++c)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    int f = 16 * s + 8 * (l >> 5) + j;
                    if (s == 12) f = (l >> 5) == 0 ? tail_map[j] : -1;
                    float w = f >= 0 && f < N_IN ? w1[(32 * c + (l & 31)) * N_IN + f] : 0.0f;
                    if (f >= 196) w = w / 15.0f;
                    const size_t o = (((size_t)s * 4 + c) * 64 + l) * 8 + j;
                    for (int part = 0; part < 3; ++part) {
                        const uint16_t b = f32_to_bf16_rne(w);
                        union { uint32_t u; float f; } cv;
                        cv.u = (uint32_t)b << 16;
                        w -= cv.f;                                             // exact: the residual fits fp32
                        wl[(size_t)part * ROOT3_PART_U4 * 8 + o] = b;
                    }
                }
}

// where a tile row comes from: the env's root rows (row R = game R) ...
struct RootRowsFetch {
    const uint4 *rows;
    __device__ __forceinline__ bool get(long long R, uint4 &u0, uint4 &u1) const { u0 = rows[2 * R]; u1 = rows[2 * R + 1]; return true; }
};
// ... or the learner's trajectory log: row R = state s_{t + (R & 1)} of the game at order position R >> 1 (gmeta = lane,
// length); a state past the game's end is an all-zero row (its value is never used: train.py:165-166)
struct TrajRowsFetch {
    const uint4 *rows; const int4 *gmeta; long long t, n_lanes, T;
    __device__ __forceinline__ bool get(long long R, uint4 &u0, uint4 &u1) const
    {
        const int4 gm = gmeta[R >> 1];                            // (lane, length, p1_won | first log row << 1, the step the game started at)
        const long long tt = t - gm.w + (R & 1);
        if (tt >= gm.y || tt >= T) return false;
        long long lr = tt + (gm.z >> 1);                          // a ring log (streamed replay of continuous self-play): first row != 0
        if (lr >= T) lr -= T;
        const uint4 *src = rows + (lr * n_lanes + gm.x) * 2;
        u0 = src[0]; u1 = src[1];
        return true;
    }
};

// OUT_SCALE: the factor on (W1 x + b1) in the output: -log2(e) for the incremental evaluator, 1 for the learner
template <class Fetch, bool LOG2E_SCALE>
__device__ __forceinline__ void root3_body(const Fetch fetch, long long n_rows, const uint4 *__restrict__ wl3, const uint2 *__restrict__ lut,
                                           const float *__restrict__ b1, float *__restrict__ hidden)
{
    extern __shared__ uint4 sW3[];                          // [3][ROOT3_PHASE_STEPS][4][64]
    uint2 *sLut = reinterpret_cast<uint2 *>(sW3 + 3 * ROOT3_PHASE_STEPS * 4 * 64);
    if (threadIdx.x < 16) sLut[threadIdx.x] = lut[threadIdx.x];
    const long long n_tiles = (n_rows + 31) >> 5;
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const long long wave = (long long)blockIdx.x * (ROOT3_THREADS / 64) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (ROOT3_THREADS / 64);
    constexpr float NL2E = -1.44269504088896340736f;
    constexpr int PH = ROOT3_PHASE_STEPS * 4 * 64;          // uint4 per plane and phase

    for (long long base = 0; base < n_tiles; base += n_waves) {          // every wave of the block runs every round
        const long long tile = base + wave;
        const bool has = tile < n_tiles;
        uint32_t p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        bool row_ok = false;
        if (has && tile * 32 + r < n_rows) {
            uint4 u0, u1;
            if (fetch.get(tile * 32 + r, u0, u1)) {
                row_ok = true;
                p[0] = u0.x; p[1] = u0.y; p[2] = u0.z; p[3] = u0.w; p[4] = u1.x; p[5] = u1.y; p[6] = u1.z; p[7] = u1.w;
            }
        }
        const Side sa{{p[0], p[1], p[2], p[3]}}, sb{{p[4], p[5], p[6], p[7]}};
        floatx16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
#pragma unroll 1
        for (int phase = 0; phase < ROOT3_PHASES; ++phase) {
            const int s0 = phase * ROOT3_PHASE_STEPS;
            const int ns = K16_STEPS - s0 < ROOT3_PHASE_STEPS ? K16_STEPS - s0 : ROOT3_PHASE_STEPS;
            __syncthreads();                                 // the previous phase's readers are done
            for (int i = threadIdx.x; i < 3 * ns * 4 * 64; i += ROOT3_THREADS) {
                const int part = i / (ns * 4 * 64), o = i - part * (ns * 4 * 64);
                sW3[part * PH + o] = wl3[(size_t)part * ROOT3_PART_U4 + (size_t)s0 * 4 * 64 + o];
            }
            __syncthreads();
            if (has) {
                const uint4 *wp = sW3 + lane;
                for (int k = 0; k < ns; ++k) {
                    const int s = s0 + k;
                    union { uint4 u; bf16x8 v; } a;
                    if (s < 12) {
                        const int pos = 2 * s + h + 1;
                        const uint2 l0 = sLut[count_at(sa, pos)], l1 = sLut[count_at(sb, pos)];
                        a.u = make_uint4(l0.x, l0.y, l1.x, l1.y);
                    } else {                                 // features 192..197 on the h == 0 lanes; off counts as integers
                        const int turn = (p[0] & TURN_BIT) ? 1 : 0;
                        const uint32_t t0 = (turn == 0 && row_ok) ? 0x3F80u : 0u, t1 = (turn == 0 || !row_ok) ? 0u : 0x3F80u;
                        const uint32_t bar1 = f32_to_bf16_rne(0.5f * (float)count_at(sa, 0)), bar2 = f32_to_bf16_rne(0.5f * (float)count_at(sb, 25));
                        const uint32_t off1 = f32_to_bf16_rne((float)count_at(sa, 25)), off2 = f32_to_bf16_rne((float)count_at(sb, 0));
                        a.u = h ? make_uint4(0, 0, 0, 0) : make_uint4(t0 | (t1 << 16), bar1 | (bar2 << 16), off1 | (off2 << 16), 0u);
                    }
#pragma unroll
                    for (int part = 0; part < 3; ++part) {
                        union { uint4 u; bf16x8 v; } w0, w1, w2r, w3;
                        const uint4 *q = wp + part * PH + k * 4 * 64;
                        w0.u = q[0]; w1.u = q[64]; w2r.u = q[128]; w3.u = q[192];
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, w0.v, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, w1.v, acc1, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, w2r.v, acc2, 0, 0, 0);
                        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, w3.v, acc3, 0, 0, 0);
                    }
                }
            }
        }
        if (has) {
            float bb[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) bb[c] = b1[32 * c + r];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const long long orow = tile * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
                if (orow < n_rows) {
                    float *o = hidden + orow * N_HID + r;
                    constexpr float SC = LOG2E_SCALE ? NL2E : 1.0f;
                    o[0] = SC * (acc0[j] + bb[0]); o[32] = SC * (acc1[j] + bb[1]);
                    o[64] = SC * (acc2[j] + bb[2]); o[96] = SC * (acc3[j] + bb[3]);
                }
            }
        }
    }
}

#ifdef BGAMD_EXPERIMENTAL       // (BGAMD_ROOT_RESIDENT=0: the env's LDS-staged root pass of rounds 1-2; root_hidden_resident_kernel gives the same bits)
__global__ __launch_bounds__(ROOT3_THREADS) void root_hidden_bf16x3_kernel(
    const uint4 *__restrict__ rows, long long n_rows, const uint4 *__restrict__ wl3, const uint2 *__restrict__ lut,
    const float *__restrict__ b1, float *__restrict__ hidden)
{
    root3_body<RootRowsFetch, true>(RootRowsFetch{rows}, n_rows, wl3, lut, b1, hidden);
}
#endif

// the learner's forward pass: plain W1 x + b1 for s_t and s_{t+1} of every running game (row 2 i + s), bg_learner.h
__global__ __launch_bounds__(ROOT3_THREADS) void traj_hidden_bf16x3_kernel(
    const uint4 *__restrict__ rows, const int4 *__restrict__ gmeta, long long t, long long n_lanes, long long T, long long n_rows,
    const uint4 *__restrict__ wl3, const uint2 *__restrict__ lut, const float *__restrict__ b1, float *__restrict__ hidden)
{
    root3_body<TrajRowsFetch, false>(TrajRowsFetch{rows, gmeta, t, n_lanes, T}, n_rows, wl3, lut, b1, hidden);
}

// The same product for a FEW THOUSAND rows (the learner's mid-sized rounds: 1 024 .. 3 072 running slots = 2 048 .. 6 144 rows): a
// workgroup of four waves per 32-row tile, wave c = hidden units 32 c .. 32 c + 31, the weight planes read straight from the L2
// (39 coalesced 1-KB loads per wave, no LDS staging, no block barrier but the one behind the count LUT) -- the staged kernel above
// gives such a step only 16 workgroups and each of them 86 KB of staging; the VALU forward kernel re-reads W1 per workgroup
// (104 MB per step at 2 048 slots).  The MFMAs run in the staged kernel's order (K-step outer, plane inner): the same bits.
constexpr int ROOT3D_THREADS = 256;
#ifdef BGAMD_EXPERIMENTAL       // (BGAMD_TD_FUSED=0: the unfused forward pass; td_forward_mfma_kernel / td_step_fused_kernel carry the same product)
__global__ __launch_bounds__(ROOT3D_THREADS) void traj_hidden_direct_kernel(
    const uint4 *__restrict__ rows, const int4 *__restrict__ gmeta, long long t, long long n_lanes, long long T, long long n_rows,
    const uint4 *__restrict__ wl3, const uint2 *__restrict__ lut, const float *__restrict__ b1, float *__restrict__ hidden)
{
    __shared__ uint2 sLut[16];
    if (threadIdx.x < 16) sLut[threadIdx.x] = lut[threadIdx.x];
    const int lane = threadIdx.x & 63, c = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long long tile = blockIdx.x;
    const TrajRowsFetch fetch{rows, gmeta, t, n_lanes, T};
    uint32_t p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool row_ok = false;
    if (tile * 32 + r < n_rows) {
        uint4 u0, u1;
        if (fetch.get(tile * 32 + r, u0, u1)) {
            row_ok = true;
            p[0] = u0.x; p[1] = u0.y; p[2] = u0.z; p[3] = u0.w; p[4] = u1.x; p[5] = u1.y; p[6] = u1.z; p[7] = u1.w;
        }
    }
    const float bb = b1[32 * c + r];
    const uint4 *wp = wl3 + (size_t)c * 64 + lane;
    uint4 w[K16_STEPS][3];                                   // all 39 loads go out before the first MFMA needs one
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
        } else {                                             // the tail K-step exactly as in root3_body
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
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const long long orow = tile * 32 + (j & 3) + 8 * (j >> 2) + 4 * h;
        if (orow < n_rows) hidden[orow * N_HID + 32 * c + r] = acc[j] + bb;
    }
}
#endif

// position of W1[n][f] in ONE bf16 plane of the root3 layout (relayout_w1_bf16x3), in 16-bit units
__host__ __device__ __forceinline__ int root3_plane_index(int n, int f)
{
    const int s = f < 192 ? (f >> 4) : 12, lh = f < 192 ? ((f >> 3) & 1) : 0, jj = f < 192 ? (f & 7) : f - 192;
    return (((s * 4 + (n >> 5)) * 64) + (n & 31) + 32 * lh) * 8 + jj;
}
// W1[n][f] -> its three bf16 terms, stored into the three planes (the learner refreshes them after every update)
__device__ __forceinline__ void root3_store_weight(uint16_t *__restrict__ wl3, int n, int f, float w)
{
    if (f >= 196) w = w / 15.0f;
    const int o = root3_plane_index(n, f);
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        const uint16_t b = f32_to_bf16_rne(w);
        w -= __uint_as_float((uint32_t)b << 16);                  // exact: the residual fits fp32
        wl3[(size_t)part * ROOT3_PART_U4 * 8 + o] = b;
    }
}

// ================================ f16 x 2 split: fp32-grade values on the fast matrix pipe ================
// W1 = hi + lo with hi = f16(W1), lo = f16(W1 - hi): 22 mantissa bits, |error| <= 2^-22 |w| + 3e-8.  Every
// feature is exact in f16 except off/15, which is fed as its own hi + lo pair, so each product is exact in
// fp32 and the only rounding is the fp32 accumulation -- the same error class as the f32 MFMA kernel, at two
// v_mfma_f32_32x32x16_f16 per (K-step, column tile) instead of sixteen-times-slower f32 MFMAs, and on a pipe
// that runs beside the VALU.  Same K layout as the bf16 kernel (one K-step = two board points); the tail step
// is [turn0, turn1, bar1/2, bar2/2, off1.hi, off1.lo, off2.hi, off2.lo].
constexpr int EVAL16X2_W_BYTES = 2 * EVAL16_W_BYTES;               // 106 496
constexpr int EVAL16X2_LDS_TOTAL = EVAL16X2_W_BYTES + EVAL16_LUT_BYTES + (EVAL_THREADS / 64) * EVAL_RED_FLOATS * 4;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ inline uint16_t f16_bits(_Float16 h)
{
    union { _Float16 h; uint16_t u; } c{h};
    return c.u;
}

// Wl[part][s][c][l][j], part 0 = hi, 1 = lo; tail feature map above
inline void relayout_w1_f16x2(const float *w1, uint16_t *wl)
{
    static const int tail_map[8] = {192, 193, 194, 195, 196, 196, 197, 197};
    for (int s = 0; s < K16_STEPS; ++s)
        for (int c = 0; c < 4; ++c)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    int f = 16 * s + 8 * (l >> 5) + j;
                    if (s == 12) f = (l >> 5) == 0 ? tail_map[j] : -1;
                    float w = f >= 0 && f < N_IN ? w1[(32 * c + (l & 31)) * N_IN + f] : 0.0f;
                    const _Float16 hi = (_Float16)w;
                    const _Float16 lo = (_Float16)(w - (float)hi);
                    const size_t o = (((size_t)s * 4 + c) * 64 + l) * 8 + j;
                    wl[o] = f16_bits(hi);
                    wl[(size_t)K16_STEPS * 4 * 64 * 8 + o] = f16_bits(lo);
                }
}
// LUT[n] = f16 x4 {n>=1, n>=2, n>=3, (n-3)/2 if n>=4}
inline void make_count_lut_f16(uint32_t *lut /*[16][2]*/)
{
    for (int n = 0; n < 16; ++n) {
        const uint16_t one = f16_bits((_Float16)1.0f);
        const uint16_t e0 = n >= 1 ? one : 0, e1 = n >= 2 ? one : 0, e2 = n >= 3 ? one : 0;
        const uint16_t e3 = n >= 4 ? f16_bits((_Float16)(0.5f * (float)(n - 3))) : 0;
        lut[2 * n] = (uint32_t)e0 | ((uint32_t)e1 << 16);
        lut[2 * n + 1] = (uint32_t)e2 | ((uint32_t)e3 << 16);
    }
}

__global__ __launch_bounds__(EVAL_THREADS) void eval_rows_f16x2_kernel(
    const uint4 *__restrict__ rows, const unsigned long long *__restrict__ n_rows_ptr, long long n_rows_imm,
    unsigned long long *__restrict__ rows_eval_counter, const uint4 *__restrict__ wl16, const uint2 *__restrict__ lut,
    const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ b2p, float *__restrict__ values,
    const uint2 *__restrict__ info, unsigned long long *__restrict__ best)
{
    extern __shared__ uint4 sW16[];
    constexpr int PART = K16_STEPS * 4 * 64;                     // uint4 entries per weight part
    uint2 *sLut = reinterpret_cast<uint2 *>(sW16 + 2 * PART);
    float *sRed = reinterpret_cast<float *>(sLut + 16) + (threadIdx.x >> 6) * EVAL_RED_FLOATS;
    for (int i = threadIdx.x; i < 2 * PART; i += EVAL_THREADS) sW16[i] = wl16[i];
    if (threadIdx.x < 16) sLut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();

    // device counter (clamped to the arena capacity passed in n_rows_imm: an overflowing step is flagged, never read
    // past the arena) or the immediate count
    long long n_rows = n_rows_imm;
    if (n_rows_ptr) { const long long c = (long long)*n_rows_ptr; n_rows = c < n_rows_imm ? c : n_rows_imm; }
    if (rows_eval_counter && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(rows_eval_counter, (unsigned long long)n_rows);
    const long long n_tiles = (n_rows + 31) >> 5;
    const int lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const long long wave = (long long)blockIdx.x * (EVAL_THREADS / 64) + (threadIdx.x >> 6);
    const long long n_waves = (long long)gridDim.x * (EVAL_THREADS / 64);
    constexpr float NL2E = -1.44269504088896340736f;
    float b1s[4], w2v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { b1s[c] = NL2E * b1[32 * c + r]; w2v[c] = w2[32 * c + r]; }
    const float b2 = *b2p;

    uint4 nx0 = make_uint4(0, 0, 0, 0), nx1 = make_uint4(0, 0, 0, 0);
    if (wave < n_tiles && wave * 32 + r < n_rows) { nx0 = rows[2 * (wave * 32 + r)]; nx1 = rows[2 * (wave * 32 + r) + 1]; }

    for (long long tile = wave; tile < n_tiles; tile += n_waves) {
        const uint32_t p[8] = {nx0.x, nx0.y, nx0.z, nx0.w, nx1.x, nx1.y, nx1.z, nx1.w};
        {
            const long long nrow = (tile + n_waves) * 32 + r;
            nx0 = make_uint4(0, 0, 0, 0); nx1 = make_uint4(0, 0, 0, 0);
            if (nrow < n_rows) { nx0 = rows[2 * nrow]; nx1 = rows[2 * nrow + 1]; }
        }
        const Side sa{{p[0], p[1], p[2], p[3]}}, sb{{p[4], p[5], p[6], p[7]}};
        floatx16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
        const uint4 *wp = sW16 + lane;
#define BG_F16X2_STEP(AFRAG, S)                                                                         \
    {                                                                                                   \
        union { uint4 u; f16x8 v; } w0, w1, w2r, w3, l0, l1, l2, l3;                                     \
        w0.u = wp[((S) * 4 + 0) * 64]; w1.u = wp[((S) * 4 + 1) * 64];                                     \
        w2r.u = wp[((S) * 4 + 2) * 64]; w3.u = wp[((S) * 4 + 3) * 64];                                    \
        l0.u = wp[PART + ((S) * 4 + 0) * 64]; l1.u = wp[PART + ((S) * 4 + 1) * 64];                       \
        l2.u = wp[PART + ((S) * 4 + 2) * 64]; l3.u = wp[PART + ((S) * 4 + 3) * 64];                       \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16((AFRAG), l0.v, acc0, 0, 0, 0);                      \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16((AFRAG), l1.v, acc1, 0, 0, 0);                      \
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16((AFRAG), l2.v, acc2, 0, 0, 0);                      \
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16((AFRAG), l3.v, acc3, 0, 0, 0);                      \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16((AFRAG), w0.v, acc0, 0, 0, 0);                      \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16((AFRAG), w1.v, acc1, 0, 0, 0);                      \
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16((AFRAG), w2r.v, acc2, 0, 0, 0);                     \
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16((AFRAG), w3.v, acc3, 0, 0, 0);                      \
    }
#pragma unroll 4
        for (int s = 0; s < 12; ++s) {
            const int pos = 2 * s + h + 1;                       // board point 2s+h  ->  bit position +1
            const uint2 e0 = sLut[count_at(sa, pos)], e1 = sLut[count_at(sb, pos)];
            union { uint4 u; f16x8 v; } a;
            a.u = make_uint4(e0.x, e0.y, e1.x, e1.y);
            BG_F16X2_STEP(a.v, s);
        }
        {   // step 12: turn, bar/2, and off/15 as hi + lo halves, on the h == 0 lanes
            const int turn = (p[0] & TURN_BIT) ? 1 : 0;
            const uint32_t one = 0x3C00u;
            const uint32_t t0 = turn == 0 ? one : 0u, t1 = turn == 0 ? 0u : one;
            const uint32_t bar1 = f16_bits((_Float16)(0.5f * (float)count_at(sa, 0))), bar2 = f16_bits((_Float16)(0.5f * (float)count_at(sb, 25)));
            const float o1 = (float)count_at(sa, 25) / 15.0f, o2 = (float)count_at(sb, 0) / 15.0f;
            const _Float16 o1h = (_Float16)o1, o2h = (_Float16)o2;
            const uint32_t o1l = f16_bits((_Float16)(o1 - (float)o1h)), o2l = f16_bits((_Float16)(o2 - (float)o2h));
            union { uint4 u; f16x8 v; } a;
            a.u = h ? make_uint4(0, 0, 0, 0)
                    : make_uint4(t0 | (t1 << 16), bar1 | (bar2 << 16), (uint32_t)f16_bits(o1h) | (o1l << 16), (uint32_t)f16_bits(o2h) | (o2l << 16));
            BG_F16X2_STEP(a.v, 12);
        }
#undef BG_F16X2_STEP
        float part[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float s0 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc0[j], NL2E, b1s[0])));
            const float s1 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc1[j], NL2E, b1s[1])));
            const float s2 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc2[j], NL2E, b1s[2])));
            const float s3 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc3[j], NL2E, b1s[3])));
            part[j] = w2v[0] * s0 + w2v[1] * s1 + w2v[2] * s2 + w2v[3] * s3;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) sRed[r * EVAL_RED_STRIDE + 8 * q + 4 * h + j] = part[4 * q + j];
        __builtin_amdgcn_wave_barrier();
        float sum = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c) sum += sRed[(16 * h + c) * EVAL_RED_STRIDE + r];
        sum += __shfl_xor(sum, 32, 64);
        __builtin_amdgcn_wave_barrier();
        {   // lanes 0..31 (h == 0) hold the 32 rows of the tile
            const long long orow = tile * 32 + r;
            const bool vrow = h == 0 && orow < n_rows;
            const float v = fast_sigmoid(sum + b2);
            if (vrow) values[orow] = v;
            if (info) {                                        // wave-uniform
                const uint2 inf = vrow ? info[orow] : make_uint2(0u, 0u);
                best_atomic_max(best, inf.x, v, inf.y, vrow, 32);
            }
        }
    }
}

// Stand-alone encoder (the reference's _encode_states_np surface): rows -> float[n][198]
__global__ void encode_rows_kernel(const uint4 *__restrict__ rows, long long n, float *__restrict__ out)
{
    const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const uint4 u0 = rows[2 * row], u1 = rows[2 * row + 1];
    const uint32_t p[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
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
    const int turn = (p[0] & TURN_BIT) ? 1 : 0;
    x[192] = turn == 0 ? 1.0f : 0.0f;
    x[193] = turn == 0 ? 0.0f : 1.0f;
    x[194] = 0.5f * (float)count_at(sd[0], 0);
    x[195] = 0.5f * (float)count_at(sd[1], 25);
    x[196] = (float)count_at(sd[0], 25) / 15.0f;
    x[197] = (float)count_at(sd[1], 0) / 15.0f;
}

}  // namespace bg
