// bg_staged_kernels.h -- kernels of the staged greedy step (see bg_staged.h).  Included by
// bgamd.hip inside its anonymous namespace, after EnvView / LaneCtx / finish_turn are defined.
#pragma once

// state of a node: the lane's game after replaying the key's prefix moves
struct NodeState {
    Side own, opp;
    int pl, dA, dB, len;
    bool dbl;
};

__device__ __forceinline__ void node_state(const EnvView &e, const Node &nd, NodeState &s)
{
    uint32_t p[8];
    load_planes(e, (long long)nd.game, p);
    const uint32_t meta = e.meta[nd.game];
    s.pl = meta & 1;
    const int d1 = (meta >> 4) & 7, d2 = (meta >> 8) & 7;
    const int pass = key_pass(nd.key);
    s.dA = pass ? d2 : d1;
    s.dB = pass ? d1 : d2;
    s.dbl = d1 == d2;
    s.len = key_len(nd.key);
    split_sides(p, s.pl, s.own, s.opp);
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (k < s.len) apply_move(s.own, s.opp, s.pl, key_origin(nd.key, k), (k & 1) ? s.dB : s.dA);
}

__device__ __forceinline__ void flag_overflow(const EnvView &e)
{
    if (threadIdx.x == 0) atomicOr(&e.counters[C_ERR], (unsigned long long)ERRF_ARENA);
}

// ---- ply 1: lane per game -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void roots_kernel(EnvView e, StagedView sv, int flags)
{
    __shared__ uint32_t s_wave[4];
    __shared__ unsigned long long s_slot;
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    LaneCtx c;
    lane_begin(e, g, flags, c);
    Side own, opp;
    split_sides(c.p, c.turn, own, opp);
    const bool dbl = c.d1 == c.d2;
    uint32_t ma = 0, mb = 0;
    if (c.live) {
        ma = legal_origins(own, opp, c.turn, c.d1);
        if (!dbl) mb = legal_origins(own, opp, c.turn, c.d2);
    }
    // doubles: first-ply nodes go on to ply 2 (D1); a stuck root is ONE empty sequence.
    // non-doubles: first-ply nodes of both die orders are leaf parents (F); a stuck root yields nothing.
    const uint32_t nF = !c.live ? 0u : (dbl ? (ma == 0 ? 1u : 0u) : (uint32_t)(__popc(ma) + __popc(mb)));
    const uint32_t nD = (c.live && dbl) ? (uint32_t)__popc(ma) : 0u;
    uint32_t totF, totD;
    uint32_t offF = block_scan_256(nF, &totF, s_wave);
    uint32_t offD = block_scan_256(nD, &totD, s_wave);
    const unsigned long long baseF = block_alloc(&sv.tops[T_F], totF, &s_slot);
    const unsigned long long baseD = block_alloc(&sv.tops[T_D1], totD, &s_slot);
    const bool okF = baseF + totF <= (unsigned long long)sv.cap_f, okD = baseD + totD <= (unsigned long long)sv.cap_d1;
    if (!okF || !okD) flag_overflow(e);
    if (c.live) {
        const uint32_t gg = (uint32_t)g;
        if (dbl) {
            if (ma == 0) { if (okF) sv.f[baseF + offF] = Node{gg, 0u}; }
            else if (okD) {
                uint32_t m = ma;
                while (m) { const int o = __ffs(m) - 1; m &= m - 1; sv.d1[baseD + offD++] = Node{gg, key_child(0u, o)}; }
            }
        } else if (okF) {
            uint32_t m = ma;
            while (m) { const int o = __ffs(m) - 1; m &= m - 1; sv.f[baseF + offF++] = Node{gg, key_child(0u, o)}; }
            m = mb;
            while (m) { const int o = __ffs(m) - 1; m &= m - 1; sv.f[baseF + offF++] = Node{gg, key_child(1u << KEY_PASS_SHIFT, o)}; }
        }
        if (flags & BGAMD_ROLL) e.meta[g] = meta_pack(c.turn, c.d1, c.d2, false);
        if (e.traj && (long long)c.ply < e.traj_plies) {        // trajectory log: 32 B per turn instead of 792 B
            const unsigned long long t = ((unsigned long long)c.ply * (unsigned long long)e.n + (unsigned long long)g) * 2;
            e.traj[t] = make_uint4(c.p[0] | (c.turn ? TURN_BIT : 0u), c.p[1], c.p[2], c.p[3]);
            e.traj[t + 1] = make_uint4(c.p[4], c.p[5], c.p[6], c.p[7]);
        }
    }
    if (g < e.n) sv.best[g] = 0ull;
}

// ---- doubles ply 2 (LEVEL 1: D1 -> D2/F) and ply 3 (LEVEL 2: D2 -> F) ----------------------------
template <int LEVEL>
__global__ __launch_bounds__(256) void expand_kernel(EnvView e, StagedView sv)
{
    __shared__ uint32_t s_wave[4];
    __shared__ unsigned long long s_slot;
    const Node *in = LEVEL == 1 ? sv.d1 : sv.d2;
    const unsigned long long cap_in = (unsigned long long)(LEVEL == 1 ? sv.cap_d1 : sv.cap_d2);
    unsigned long long n_in = sv.tops[LEVEL == 1 ? T_D1 : T_D2];
    if (n_in > cap_in) n_in = cap_in;
    for (unsigned long long blk = blockIdx.x; blk * 256 < n_in; blk += gridDim.x) {
        const unsigned long long i = blk * 256 + threadIdx.x;
        const bool valid = i < n_in;
        Node nd{0u, 0u};
        NodeState s;
        uint32_t m = 0;
        if (valid) {
            nd = in[i];
            node_state(e, nd, s);
            m = legal_origins(s.own, s.opp, s.pl, s.dA);
        }
        // stuck here -> the node itself is a leaf (goes to F unchanged); otherwise its children go on
        const uint32_t nkids = (uint32_t)__popc(m);
        const uint32_t nF = !valid ? 0u : (m == 0 ? 1u : (LEVEL == 2 ? nkids : 0u));
        const uint32_t nD = (valid && LEVEL == 1) ? nkids : 0u;
        uint32_t totF, totD = 0;
        uint32_t offF = block_scan_256(nF, &totF, s_wave);
        const unsigned long long baseF = block_alloc(&sv.tops[T_F], totF, &s_slot);
        const bool okF = baseF + totF <= (unsigned long long)sv.cap_f;
        uint32_t offD = 0;
        unsigned long long baseD = 0;
        bool okD = true;
        if (LEVEL == 1) {
            offD = block_scan_256(nD, &totD, s_wave);
            baseD = block_alloc(&sv.tops[T_D2], totD, &s_slot);
            okD = baseD + totD <= (unsigned long long)sv.cap_d2;
        }
        if (!okF || !okD) flag_overflow(e);
        if (valid) {
            if (m == 0) { if (okF) sv.f[baseF + offF] = nd; }
            else {
                while (m) {
                    const int o = __ffs(m) - 1; m &= m - 1;
                    const Node ch{nd.game, key_child(nd.key, o)};
                    if (LEVEL == 1) { if (okD) sv.d2[baseD + offD++] = ch; }
                    else { if (okF) sv.f[baseF + offF++] = ch; }
                }
            }
        }
    }
}

// ---- leaves: lane per leaf-parent, per-workgroup LDS staging + exact de-duplication ------------------
// A block takes 256 leaf parents (<= 15 afterstates each).  Leaves are staged in LDS in rounds of at
// most LEAF_CAP rows, de-duplicated through an LDS hash (slot = SMALLEST staging index among identical
// (game, 256-bit row); inside a block staging order == reference order for rows of one game), and only
// the representatives are written to HBM (u_rows / u_info).  Nothing else leaves the CU.
constexpr int LEAF_THREADS = 512;          // leaf parents per block iteration
constexpr int LEAF_CAP = 2048;             // staged rows per round (40 B each, SoA)
constexpr int LEAF_T = 4096;               // hash slots
constexpr uint32_t LEAF_EMPTY = 0xFFFFFFFFu;

__global__ __launch_bounds__(LEAF_THREADS) void leaves_kernel(EnvView e, StagedView sv)
{
    __shared__ uint32_t s_row[10][LEAF_CAP];   // p0..p7, game, key|turn<<31
    __shared__ uint32_t s_tab[LEAF_T];
    __shared__ uint16_t s_pos[LEAF_CAP];
    __shared__ uint32_t s_wave[LEAF_THREADS / 64];
    __shared__ unsigned long long s_slot;
    unsigned long long n_in = sv.tops[T_F];
    if (n_in > (unsigned long long)sv.cap_f) n_in = (unsigned long long)sv.cap_f;
    unsigned long long raw_total = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&e.counters[C_FNODES], n_in);
        atomicAdd(&e.counters[C_DNODES], sv.tops[T_D1] + sv.tops[T_D2]);
    }
    for (unsigned long long blk = blockIdx.x; blk * LEAF_THREADS < n_in; blk += gridDim.x) {
        const unsigned long long ni = blk * LEAF_THREADS + threadIdx.x;
        const bool valid = ni < n_in;
        Node nd{0u, 0u};
        NodeState s;
        uint32_t m0 = 0;
        int die = 1;
        if (valid) {
            nd = sv.f[ni];
            node_state(e, nd, s);
            die = (s.len & 1) ? s.dB : s.dA;
            if (s.len < (s.dbl ? 4 : 2)) m0 = legal_origins(s.own, s.opp, s.pl, die);
        }
        const uint32_t cnt = valid ? (m0 ? (uint32_t)__popc(m0) : 1u) : 0u;
        uint32_t total;
        const uint32_t off = block_scan_256<LEAF_THREADS / 64>(cnt, &total, s_wave);
        raw_total += total;
        const uint32_t info_turn = valid && s.pl ? 0x80000000u : 0u;
        for (uint32_t r0 = 0; r0 < total; r0 += LEAF_CAP) {
            const uint32_t nrow = total - r0 < (uint32_t)LEAF_CAP ? total - r0 : (uint32_t)LEAF_CAP;
            for (int i = threadIdx.x; i < LEAF_T; i += LEAF_THREADS) s_tab[i] = LEAF_EMPTY;
            // 1. stage this round's window [r0, r0 + nrow) of the block's leaves
            if (cnt && off < r0 + nrow && off + cnt > r0) {
                uint32_t m = m0, j = off;
                do {
                    int o = -1;
                    if (m) { o = __ffs(m) - 1; m &= m - 1; }
                    if (j >= r0 && j < r0 + nrow) {
                        Side a = s.own, b = s.opp;
                        uint32_t key = nd.key;
                        if (o >= 0) { apply_move(a, b, s.pl, o, die); key = key_child(nd.key, o); }
                        const Side &s1 = s.pl ? b : a, &s2 = s.pl ? a : b;
                        const uint32_t q = j - r0;
                        s_row[0][q] = s1.b[0] | (s.pl ? TURN_BIT : 0u); s_row[1][q] = s1.b[1];
                        s_row[2][q] = s1.b[2]; s_row[3][q] = s1.b[3];
                        s_row[4][q] = s2.b[0]; s_row[5][q] = s2.b[1]; s_row[6][q] = s2.b[2]; s_row[7][q] = s2.b[3];
                        s_row[8][q] = nd.game; s_row[9][q] = key | info_turn;
                    }
                    ++j;
                } while (m);
            }
            __syncthreads();
            // 2. hash insert
            for (uint32_t i = threadIdx.x; i < nrow; i += LEAF_THREADS) {
                uint32_t p[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) p[k] = s_row[k][i];
                const uint32_t game = s_row[8][i];
                uint32_t h = hash_row(p, game) & (LEAF_T - 1);
                for (;;) {
                    const uint32_t cur = atomicCAS(&s_tab[h], LEAF_EMPTY, i);
                    if (cur == LEAF_EMPTY) break;
                    bool same = s_row[8][cur] == game;
#pragma unroll
                    for (int k = 0; k < 8; ++k) same = same && (s_row[k][cur] == p[k]);
                    if (same) { atomicMin(&s_tab[h], i); break; }
                    h = (h + 1) & (LEAF_T - 1);
                }
                s_pos[i] = (uint16_t)h;
            }
            __syncthreads();
            // 3. representatives -> unique arena (order is irrelevant: the key carries it)
            uint32_t mine = 0;
            for (uint32_t i = threadIdx.x; i < nrow; i += LEAF_THREADS) mine += (s_tab[s_pos[i]] == i) ? 1u : 0u;
            uint32_t tot2;
            uint32_t off2 = block_scan_256<LEAF_THREADS / 64>(mine, &tot2, s_wave);
            const unsigned long long base2 = block_alloc(&sv.tops[T_U], tot2, &s_slot);
            const bool ok = base2 + tot2 <= (unsigned long long)sv.cap_rows;
            if (!ok) flag_overflow(e);
            if (ok) {
                for (uint32_t i = threadIdx.x; i < nrow; i += LEAF_THREADS) {
                    if (s_tab[s_pos[i]] != i) continue;
                    const unsigned long long d = base2 + off2++;
                    sv.u_rows[2 * d] = make_uint4(s_row[0][i], s_row[1][i], s_row[2][i], s_row[3][i]);
                    sv.u_rows[2 * d + 1] = make_uint4(s_row[4][i], s_row[5][i], s_row[6][i], s_row[7][i]);
                    sv.u_info[d] = make_uint2(s_row[8][i], s_row[9][i]);
                }
            }
            __syncthreads();                   // s_row / s_tab / s_pos are rewritten by the next round
        }
    }
    if (threadIdx.x == 0 && raw_total) atomicAdd(&e.counters[C_CAND_RAW], raw_total);
}

// finds the reference-order index of a given sequence and the list length (BGAMD_WANT_INDEX)
struct IndexVisitor {
    uint32_t n = 0, idx = 0xFFFFFFFFu, t_orig, t_len, t_dA, t_dB;
    __device__ __forceinline__ bool leaf(const Side &, const Side &, uint32_t origins, int len, int dA, int dB)
    {
        if (origins == t_orig && (uint32_t)len == t_len && (uint32_t)dA == t_dA && (uint32_t)dB == t_dB) idx = n;
        ++n;
        return false;
    }
};

// ---- apply: lane per game ------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void apply_kernel(EnvView e, StagedView sv, int flags, float epsilon)
{
    const long long g = (long long)blockIdx.x * 64 + threadIdx.x;
    LaneCtx c;
    lane_begin(e, g, flags & ~BGAMD_ROLL, c);            // dice were stored by roots_kernel
    const unsigned long long pack = (g < e.n && c.live) ? sv.best[g] : 0ull;
    const bool has = pack != 0ull;
    int32_t chosen = -1;
    uint32_t cseq = 0, ccount = has ? 1u : 0u;
    float cval = 0.0f;
    if (has) {
        Side own, opp;
        split_sides(c.p, c.turn, own, opp);
        const bool explore = epsilon > 0.0f && (float)(c.x.w >> 8) * (1.0f / 16777216.0f) < epsilon;
        uint32_t key = ~(uint32_t)pack & 0x7FFFFFFFu;
        const uint32_t vb = (uint32_t)(pack >> 32);
        cval = __uint_as_float(c.turn ? ~vb : vb);
        if (explore) {                                     // model.py:205-206, index from the TURN stream
            CountVisitor cv;
            walk_sequences(own, opp, c.turn, c.d1, c.d2, cv);
            const uint32_t k = (uint32_t)(((unsigned long long)c.x.z * cv.n) >> 32);
            SelectVisitor sel(k);
            walk_sequences(own, opp, c.turn, c.d1, c.d2, sel);
            own = sel.own; opp = sel.opp;
            cseq = sel.seq; chosen = (int32_t)k; ccount = cv.n; cval = 0.0f;
        } else {
            const int len = key_len(key), pass = key_pass(key);
            const int dA = pass ? c.d2 : c.d1, dB = pass ? c.d1 : c.d2;
            uint32_t origins = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < len) origins |= (uint32_t)key_origin(key, k) << (5 * k);
            if (flags & BGAMD_WANT_INDEX) {                // reference-order index + list length (slow path)
                IndexVisitor iv;
                iv.t_orig = origins; iv.t_len = (uint32_t)len; iv.t_dA = (uint32_t)dA; iv.t_dB = (uint32_t)dB;
                walk_sequences(own, opp, c.turn, c.d1, c.d2, iv);
                chosen = (int32_t)iv.idx; ccount = iv.n;
            } else chosen = 0;                             // "a move was made"; exact index only on request
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < len) apply_move(own, opp, c.turn, key_origin(key, k), (k & 1) ? dB : dA);
            cseq = seq_pack(origins, len, dA, dB);
        }
        join_sides(own, opp, c.turn, c.p);
    }
    if (c.live) { e.chosen[g] = chosen; e.chosen_seq[g] = cseq; e.chosen_val[g] = cval; e.cand_cnt[g] = ccount; }
    finish_turn(e, g, c.p, c.turn, c.d1, c.d2, c.ply, c.epi, flags, c.live);
}
