// bg_random_kernels.h -- random-policy step (benchmark.py:54-61: uniform over the reference-order list of
// legalTurnSequences) with BOUNDED work per lane.  Included by bgamd.hip inside its anonymous namespace.
//
//   rnd_tasks_kernel   lane per game : roll; the game's tree is cut into tasks in reference order --
//                                      non-doubles: one per (die order, first move); doubles: one per
//                                      (first, second) move pair (or per stuck prefix)
//   rnd_count_kernel   lane per task : number of sequences below the task (<= 15 x 15)
//   rnd_select_kernel  lane per game : C = sum over the game's tasks, k = (u32 * C) >> 32, locate the task that
//                                      holds sequence k, replay it, terminal check, reset / flip
// No lane ever walks more than 15 x 15 positions (the whole-tree walk of step_random_kernel reaches 10 063
// sequences in one lane and makes the kernel wait for it).
#pragma once

struct RandomView {
    Node *tasks;               // [cap]  (shares the leaf-parent list of the greedy step)
    uint32_t *task_count;      // [cap]
    uint32_t *task_off;        // [n] first task of the game
    uint32_t *task_n;          // [n]
    long long cap;
    unsigned long long *top;   // bump pointer
};

// ε-greedy (model.py:205-206): the lanes whose TURN-stream draw falls below epsilon pick uniformly
__device__ __forceinline__ bool explore_draw(const LaneCtx &c, float epsilon)
{
    return epsilon > 0.0f && (float)(c.x.w >> 8) * (1.0f / 16777216.0f) < epsilon;
}

// explore_eps < 0: every live lane gets tasks (the random-policy step); otherwise only the exploring lanes of an
// ε-greedy step do (dice already stored, flags without BGAMD_ROLL)
__global__ __launch_bounds__(256) void rnd_tasks_kernel(EnvView e, RandomView rv, int flags, float explore_eps)
{
    __shared__ uint32_t s_wave[4];
    __shared__ unsigned long long s_slot;
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    LaneCtx c;
    lane_begin(e, g, flags, c);
    Side own, opp;
    split_sides(c.p, c.turn, own, opp);
    const bool dbl = c.d1 == c.d2;
    uint32_t ma = 0, mb = 0, nT = 0;
    const bool want = c.live && (explore_eps < 0.0f || explore_draw(c, explore_eps));
    if (want) {
        ma = legal_origins(own, opp, c.turn, c.d1);
        if (!dbl) { mb = legal_origins(own, opp, c.turn, c.d2); nT = (uint32_t)(__popc(ma) + __popc(mb)); }
        else if (ma == 0) nT = 1;                                   // ONE empty sequence
        else {
            uint32_t m = ma;
            while (m) {
                const int o = __ffs(m) - 1; m &= m - 1;
                Side a = own, b = opp;
                apply_move(a, b, c.turn, o, c.d1);
                const uint32_t m1 = legal_origins(a, b, c.turn, c.d1);
                nT += m1 ? (uint32_t)__popc(m1) : 1u;
            }
        }
    }
    uint32_t tot;
    uint32_t off = block_scan_256(nT, &tot, s_wave);
    const unsigned long long base = block_alloc(rv.top, tot, &s_slot);
    const bool ok = base + tot <= (unsigned long long)rv.cap;
    if (!ok) flag_overflow(e);
    if (want) {
        const uint32_t gg = (uint32_t)g;
        Node *out = rv.tasks + base;
        if (ok) {
            if (!dbl) {
                uint32_t m = ma;
                while (m) { const int o = __ffs(m) - 1; m &= m - 1; out[off++] = Node{gg, key_child(0u, o)}; }
                m = mb;
                while (m) { const int o = __ffs(m) - 1; m &= m - 1; out[off++] = Node{gg, key_child(1u << KEY_PASS_SHIFT, o)}; }
            } else if (ma == 0) out[off++] = Node{gg, 0u};
            else {
                uint32_t m = ma;
                while (m) {
                    const int o = __ffs(m) - 1; m &= m - 1;
                    Side a = own, b = opp;
                    apply_move(a, b, c.turn, o, c.d1);
                    uint32_t m1 = legal_origins(a, b, c.turn, c.d1);
                    const uint32_t k1 = key_child(0u, o);
                    if (m1 == 0) out[off++] = Node{gg, k1};
                    while (m1) { const int o1 = __ffs(m1) - 1; m1 &= m1 - 1; out[off++] = Node{gg, key_child(k1, o1)}; }
                }
            }
        }
        rv.task_off[g] = (uint32_t)(base + off - nT);
        rv.task_n[g] = ok ? nT : 0u;
    } else if (g < e.n) rv.task_n[g] = 0u;
    if (c.live && (flags & BGAMD_ROLL)) e.meta[g] = meta_pack(c.turn, c.d1, c.d2, false);
}

// sequences below a task, and (SELECT) the position + packed origins of the `want`-th one
template <bool SELECT>
__device__ __forceinline__ uint32_t task_walk(const NodeState &s, uint32_t key, uint32_t want, Side &ro, Side &rp, uint32_t &rkey)
{
    const int maxd = s.dbl ? 4 : 2;
    uint32_t n = 0;
    const int d0 = (s.len & 1) ? s.dB : s.dA;
    uint32_t m = s.len < maxd ? legal_origins(s.own, s.opp, s.pl, d0) : 0u;
    if (m == 0) { if (SELECT && want == 0) { ro = s.own; rp = s.opp; rkey = key; } return 1; }
    while (m) {
        const int o = __ffs(m) - 1; m &= m - 1;
        Side a = s.own, b = s.opp;
        apply_move(a, b, s.pl, o, d0);
        const uint32_t k1 = key_child(key, o);
        const int len1 = s.len + 1;
        uint32_t m1 = len1 < maxd ? legal_origins(a, b, s.pl, (len1 & 1) ? s.dB : s.dA) : 0u;
        if (m1 == 0) {
            if (SELECT && n == want) { ro = a; rp = b; rkey = k1; }
            ++n;
            continue;
        }
        if (!SELECT) { n += (uint32_t)__popc(m1); continue; }   // tasks start at most 2 plies above the leaves
        while (m1) {
            const int o1 = __ffs(m1) - 1; m1 &= m1 - 1;
            if (n == want) {
                Side a2 = a, b2 = b;
                apply_move(a2, b2, s.pl, o1, (len1 & 1) ? s.dB : s.dA);
                ro = a2; rp = b2; rkey = key_child(k1, o1);
            }
            ++n;
        }
    }
    return n;
}

__global__ __launch_bounds__(256) void rnd_count_kernel(EnvView e, RandomView rv)
{
    unsigned long long n_in = *rv.top;
    if (n_in > (unsigned long long)rv.cap) n_in = (unsigned long long)rv.cap;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n_in; i += (unsigned long long)gridDim.x * 256) {
        const Node nd = rv.tasks[i];
        NodeState s;
        node_state(e, nd, s);
        Side ro, rp;
        uint32_t rk;
        rv.task_count[i] = task_walk<false>(s, nd.key, 0u, ro, rp, rk);
    }
}

// the k-th sequence (reference order) of game g out of its counted tasks: afterstate into (ro, rp), key into rk
__device__ __forceinline__ void task_select(const EnvView &e, const RandomView &rv, uint32_t t0, uint32_t nT, uint32_t k,
                                            Side &ro, Side &rp, uint32_t &rk)
{
    uint32_t acc = 0, t = 0;
    for (; t + 1 < nT; ++t) {
        const uint32_t cnt = rv.task_count[t0 + t];
        if (k < acc + cnt) break;
        acc += cnt;
    }
    const Node nd = rv.tasks[t0 + t];
    NodeState s;
    node_state(e, nd, s);
    ro = s.own; rp = s.opp; rk = nd.key;
    task_walk<true>(s, nd.key, k - acc, ro, rp, rk);
}

__device__ __forceinline__ bool explore_pick(const EnvView &e, const ExploreView &xv, long long g, uint32_t u, Side &own,
                                             Side &opp, uint32_t &key, uint32_t &k, uint32_t &C)
{
    const uint32_t nT = xv.task_n[g];
    if (nT == 0) return false;                             // arena overflow (flagged): the lane stays greedy
    const uint32_t t0 = xv.task_off[g];
    C = 0;
    for (uint32_t t = 0; t < nT; ++t) C += xv.task_count[t0 + t];
    if (C == 0) return false;
    k = (uint32_t)(((unsigned long long)u * C) >> 32);
    RandomView rv{};
    rv.tasks = const_cast<Node *>(xv.tasks); rv.task_count = const_cast<uint32_t *>(xv.task_count);
    task_select(e, rv, t0, nT, k, own, opp, key);
    return true;
}

__global__ __launch_bounds__(256) void rnd_select_kernel(EnvView e, RandomView rv, int flags, const uint32_t *__restrict__ choice)
{
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    LaneCtx c;
    lane_begin(e, g, flags & ~BGAMD_ROLL, c);            // dice were stored by rnd_tasks_kernel
    const uint32_t nT = (g < e.n && c.live) ? rv.task_n[g] : 0u;
    const uint32_t t0 = nT ? rv.task_off[g] : 0u;
    uint32_t C = 0;
    for (uint32_t t = 0; t < nT; ++t) C += rv.task_count[t0 + t];
    int32_t chosen = -1;
    uint32_t cseq = 0;
    if (C > 0) {
        const uint32_t u = choice ? choice[g] : c.x.z;
        const uint32_t k = (uint32_t)(((unsigned long long)u * C) >> 32);
        Side ro, rp;
        uint32_t rk;
        task_select(e, rv, t0, nT, k, ro, rp, rk);
        join_sides(ro, rp, c.turn, c.p);
        const int len = key_len(rk), pass = key_pass(rk);
        uint32_t origins = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < len) origins |= (uint32_t)key_origin(rk, q) << (5 * q);
        cseq = seq_pack(origins, len, pass ? c.d2 : c.d1, pass ? c.d1 : c.d2) | (c.turn ? (1u << 29) : 0u);
        chosen = (int32_t)k;
    }
    if (c.live) { e.chosen[g] = chosen; e.chosen_seq[g] = cseq; e.cand_cnt[g] = C; e.chosen_val[g] = 0.0f; }
    const unsigned long long tot = wave_sum_u32(C);
    if ((threadIdx.x & 63) == 0 && tot) atomicAdd(&e.counters[C_CAND_RAW], tot);
    finish_turn(e, g, c.p, c.turn, c.d1, c.d2, c.ply, c.epi, flags, c.live);
}
