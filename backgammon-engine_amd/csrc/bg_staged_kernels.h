// bg_staged_kernels.h -- kernels of the staged greedy step (see bg_staged.h).  Included by
// bgamd.hip inside its anonymous namespace, after EnvView / LaneCtx / finish_turn are defined.
#pragma once

// state of a node: the lane's game after replaying the key's prefix moves
struct NodeState {
    Side own, opp;
    int pl, dA, dB, len;
    bool dbl;
};

// node_build: the position of a node from its game's planes + meta (already in registers); before_last (optional)
// receives the position before the node's last move
__device__ __forceinline__ void node_build(const Node &nd, const uint32_t (&p)[8], uint32_t meta, NodeState &s,
                                           Side *before_last_own = nullptr, Side *before_last_opp = nullptr)
{
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
        if (k < s.len) {
            if (before_last_own && k == s.len - 1) { *before_last_own = s.own; *before_last_opp = s.opp; }
            apply_move(s.own, s.opp, s.pl, key_origin(nd.key, k), (k & 1) ? s.dB : s.dA);
        }
}

__device__ __forceinline__ void node_state(const EnvView &e, const Node &nd, NodeState &s)
{
    uint32_t p[8];
    load_planes(e, (long long)nd.game, p);
    node_build(nd, p, e.meta[nd.game], s);
}

__device__ __forceinline__ void flag_overflow(const EnvView &e)
{
    if (threadIdx.x == 0) atomicOr(&e.counters[C_ERR], (unsigned long long)ERRF_ARENA);
}

// ---- ply 1: lane per game -----------------------------------------------------------------------
// lane-per-game kernels of the step boundary: threads per workgroup (measured 64 .. 1 024: 256 and 512 tie, smaller is
// slower -- every workgroup pays two allocation atomics in the roots)
#ifndef BG_LANE_NT
#define BG_LANE_NT 256
#endif
constexpr int LANE_NT = BG_LANE_NT;

// The roots of a step in two halves (round 5): roots_issue does everything that needs no list position -- the roll, the ply-1 moves, the scan, the
// lane's root row, turn log and dice -- and ISSUES the workgroup's two list allocations (returning atomics, thread 0); roots_collect waits for
// them and writes the nodes.  A caller with work of its own in between (the boundary launch: the staging of the root pass) hides the
// allocation's round trip under it; roots_body is the two back to back.
struct RootsPending {
    uint32_t ma, mb, offF, offD, totF, totD;
    uint32_t game;
    int sh;
    bool live, dbl;
    unsigned long long r0, r1;                             // thread 0: the two bases on their way
};

// pre (optional): the lane's state handed over in registers by the apply of the turn before (boundary_kernel)
// row_out (optional, [2]): the lane's root row as written to sv.root_rows (zero for a lane past the env), for a caller that goes on with it
__device__ __forceinline__ void roots_issue(const EnvView &e, const StagedView &sv, int flags, long long g, const LaneCtx *pre, uint4 *row_out,
                                            RootsPending &P)
{
    __shared__ uint32_t s_wave[LANE_NT / 64];
    LaneCtx c;
    if (pre) {
#pragma unroll
        for (int k = 0; k < 8; ++k) c.p[k] = pre->p[k];
        c.meta = pre->meta; c.ply = pre->ply; c.epi = pre->epi;
        lane_derive(e, g, flags, c);
    } else lane_begin(e, g, flags, c);
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
    // one scan for both counts (nF <= 30, nD <= 15 per lane: 16 bits each hold a workgroup's totals), both allocations
    // in flight together
    uint32_t totFD;
    const uint32_t offFD = block_scan_256<LANE_NT / 64>(nF | (nD << 16), &totFD, s_wave);
    P.totF = totFD & 0xFFFFu; P.totD = totFD >> 16;
    P.offF = offFD & 0xFFFFu; P.offD = offFD >> 16;
    // (sharded lists: this workgroup's lists are number blockIdx.x % shards)
    P.sh = sv.shards > 1 ? (int)(blockIdx.x % (unsigned)sv.shards) : 0;
    P.r0 = P.r1 = 0;
    if (threadIdx.x == 0) {                                 // both issued unconditionally: the totals are almost never zero
        P.r0 = atomic_add_deferred(&sv.tops[f_counter(P.sh)], (unsigned long long)P.totF);
        P.r1 = atomic_add_deferred(&sv.tops[d1_counter(P.sh)], (unsigned long long)P.totD);
    }
    P.ma = ma; P.mb = mb; P.live = c.live; P.dbl = dbl; P.game = (uint32_t)g;
    if (c.live) {
        if (flags & BGAMD_ROLL) e.meta[g] = meta_pack(c.turn, c.d1, c.d2, false);
        const long long lrow = e.traj_ring ? e.log_slot : (long long)c.ply;      // ring log: by env step; else by the lane's ply
        if (e.traj && lrow < e.traj_plies) {                    // trajectory log: 32 B per turn instead of 792 B
            const unsigned long long t = ((unsigned long long)lrow * (unsigned long long)e.n + (unsigned long long)g) * 2;
            e.traj[t] = make_uint4(c.p[0] | (c.turn ? TURN_BIT : 0u), c.p[1], c.p[2], c.p[3]);
            e.traj[t + 1] = make_uint4(c.p[4], c.p[5], c.p[6], c.p[7]);
        }
    }
    if (row_out) { row_out[0] = make_uint4(0u, 0u, 0u, 0u); row_out[1] = make_uint4(0u, 0u, 0u, 0u); }
    if (g < e.n) {
        sv.best[g] = 0ull;
        const uint4 r0 = make_uint4(c.p[0] | (c.turn ? TURN_BIT : 0u), c.p[1], c.p[2], c.p[3]), r1 = make_uint4(c.p[4], c.p[5], c.p[6], c.p[7]);
        sv.root_rows[2 * g] = r0;
        sv.root_rows[2 * g + 1] = r1;
        if (row_out) { row_out[0] = r0; row_out[1] = r1; }
    }
}

__device__ __forceinline__ void roots_collect(const EnvView &e, const StagedView &sv, RootsPending &P)
{
    __shared__ unsigned long long s_slot[2];
    if (threadIdx.x == 0) { s_slot[0] = P.r0; s_slot[1] = P.r1; }      // (the compiler's wait for the two results sits here)
    __syncthreads();
    unsigned long long baseF = uniform_u64(s_slot[0]), baseD = uniform_u64(s_slot[1]);
    const unsigned long long capF = (unsigned long long)(sv.cap_f / sv.shards), capD = (unsigned long long)(sv.cap_d1 / sv.shards);
    const bool okF = baseF + P.totF <= capF, okD = baseD + P.totD <= capD;
    baseF += (unsigned long long)P.sh * capF;               // absolute positions in sv.f / sv.d1
    baseD += (unsigned long long)P.sh * capD;
    if (!okF || !okD) flag_overflow(e);
    if (P.live) {
        const uint32_t gg = P.game;
        uint32_t offF = P.offF, offD = P.offD;
        if (P.dbl) {
            if (P.ma == 0) { if (okF) sv.f[baseF + offF] = Node{gg, 0u}; }
            else if (okD) {
                uint32_t m = P.ma;
                while (m) { const int o = __ffs(m) - 1; m &= m - 1; sv.d1[baseD + offD++] = Node{gg, key_child(0u, o)}; }
            }
        } else if (okF) {
            uint32_t m = P.ma;
            while (m) { const int o = __ffs(m) - 1; m &= m - 1; sv.f[baseF + offF++] = Node{gg, key_child(0u, o)}; }
            m = P.mb;
            while (m) { const int o = __ffs(m) - 1; m &= m - 1; sv.f[baseF + offF++] = Node{gg, key_child(1u << KEY_PASS_SHIFT, o)}; }
        }
    }
}

__device__ __forceinline__ void roots_body(const EnvView &e, const StagedView &sv, int flags, long long g, const LaneCtx *pre = nullptr,
                                           uint4 *row_out = nullptr)
{
    RootsPending P;
    roots_issue(e, sv, flags, g, pre, row_out, P);
    roots_collect(e, sv, P);
}

__global__ __launch_bounds__(LANE_NT) void roots_kernel(EnvView e, StagedView sv, int flags)
{
    roots_body(e, sv, flags, (long long)blockIdx.x * LANE_NT + threadIdx.x);
}

// ---- one ply per launch ---------------------------------------------------------------------------------------
// expand_kernel<MODE>: a block takes up to expand_threads(MODE) nodes; every node contributes its successor positions
// (<= 15: one per legal origin; a node with no legal move, or at full depth, contributes itself):
//   MODE_PLY2  (in: D1)  children -> D2, stuck nodes -> F      doubles after 2 moves
//   MODE_PLY3  (in: D2)  everything -> F                       doubles after 3 moves (leaf parents)
//   MODE_LEAF  (in: F)   everything -> u_rows / u_info         afterstates handed to the value net
// Duplicates are not generated in the first place: of two orders of the same two commuting checker moves only the
// one with the smaller reference key is expanded (rules below, in the node phase).  What the rules cannot prove equal
// (about 3 % of the rows: e.g. two different pairs of moves that happen to build the same position) is simply
// evaluated twice -- identical rows get bit-identical values and the arg-max keeps the smaller key, so the result is
// the same.  An earlier version staged every successor in LDS and removed duplicates exactly through an LDS hash
// (40 B per row, CAS chains, five barriers per 1 024 rows); with the rules in place that machinery removed 3 % of the
// rows and cost a third of the step.
//
// One LANE PER SUCCESSOR: the node threads only publish their position (11 dwords in LDS) and their successor count;
// after the scan and ONE allocation per block every successor gets its own lane, which reads (parent, origin) from the
// table the node threads filled, rebuilds the position and writes it to its final place (lane q -> row base + q: coalesced,
// and the rows leave in reference order of the nodes).
enum { MODE_PLY2 = 1, MODE_PLY3 = 2, MODE_LEAF = 3 };
// threads per workgroup: every block iteration costs one same-address allocation atomic, so the leaf stage (2 000
// block iterations of 256) takes big workgroups; the doubles plies are small and latency-bound and take smaller ones
#ifndef BG_EXPAND_NT_PLY
#define BG_EXPAND_NT_PLY 512
#endif
#ifndef BG_EXPAND_NT_LEAF
#define BG_EXPAND_NT_LEAF 1024
#endif
__host__ __device__ constexpr int expand_threads(int mode) { return mode == 3 ? BG_EXPAND_NT_LEAF : BG_EXPAND_NT_PLY; }
constexpr uint32_t KEY_MASK = 0x00FFFFFFu;    // pass | origins | len

// a thread's node with its game's planes and meta word, loaded ahead of use
struct NodeIn { Node nd; uint32_t pl[8]; uint32_t meta; bool valid; };
__device__ __forceinline__ void node_fetch(const EnvView &e, const Node *__restrict__ in, long long node_idx, NodeIn &x)
{
    x.valid = node_idx >= 0;
    x.nd = Node{0u, 0u}; x.meta = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) x.pl[k] = 0;
    if (x.valid) {
        x.nd = in[node_idx];
        load_planes(e, (long long)x.nd.game, x.pl);
        x.meta = e.meta[x.nd.game];
    }
}

// where the doubles plies put their leaf parents: the shared F list (one ply per launch / doubles_kernel) or the doubles
// workgroups' own F2 list (expand_all_kernel, whose other workgroups are reading F at the same time)
struct LeafParents { Node *list; unsigned long long *top; unsigned long long cap; };
__device__ __forceinline__ LeafParents leaf_parents_shared(const StagedView &sv)
{
    return LeafParents{sv.f, &sv.tops[T_F], (unsigned long long)sv.cap_f};
}

// One block iteration: every thread brings its node (threads >= np never have one).  Returns (block-uniform) where the
// children went: *out_base / *out_total in the MODE's output list.  All threads of the block must call it.
// STUCK_ROWS (MODE_PLY2): a stuck ply-2 node -- a leaf parent whose only "successor" is itself -- is written as its row right
// here instead of going through the leaf-parent list (*out_stuck = how many: they count as leaf parents and as rows).
// the phase's LDS records: ONE instance per kernel, declared by the kernel and shared by all its phases (every phase ends with a
// block barrier); as function-local statics every instantiation of expand_phase inlined into a kernel took its own 70 B per thread
template <int NT>
struct ExpandLds {
    uint32_t par_plane[8][NT];         // parent position: mover's planes 0-3, opponent's 4-7
    uint32_t par_game[NT], par_key[NT];                    // key | die<<27 | turn<<31
    uint16_t child[NT * 15];           // successor q of the workgroup: parent << 5 | origin (31: the parent itself)
    uint32_t wave[NT / 64];
    unsigned long long slot;
    unsigned long long slot2[2];
};
// A node handed from one phase to the next IN REGISTERS (expand_all_kernel's doubles workgroups): the lane that built successor q
// of one phase is the node thread of the next -- no node list, no allocation, no reload of the game's planes, no replay of the key
struct ChainNode {
    Side own, opp, pown, popp;         // the node's position (mover's side, opponent's) and the position before its last move
    uint32_t game, key;
    int pl, die;
    bool valid;
    bool hit;                          // a move of the key has hit a blot (the row's delta list is longer by two entries per hit)
};

// The phase proper.  The node's position comes in registers: s (after the key's moves), prev_* (before the last of them), root_pl (the
// game's planes as stored: the non-doubles rule of the leaf stage asks the root position).
// CHAIN (MODE_PLY2 / MODE_PLY3): successors q < NT are handed to the caller in *chain (lane q gets successor q), only the successors
// from NT on go through the MODE's output list: *out_base / *out_total describe THAT part, *out_all counts all of them.
template <int MODE, int NT, bool STUCK_ROWS, bool CHAIN>
__device__ __forceinline__ void expand_phase_core(const EnvView &e, const StagedView &sv, bool valid, const Node nd, const NodeState &s,
                                                  const Side &prev_own, const Side &prev_opp, const uint32_t (&root_pl)[8], bool prefix_hit,
                                                  unsigned long long *out_base, uint32_t *out_total, const LeafParents &lp,
                                                  ExpandLds<NT> &L, uint32_t *out_stuck, ChainNode *chain, uint32_t *out_all, bool s_kind_dbl = false)
{
    static_assert(!STUCK_ROWS || MODE == MODE_PLY2, "only the ply-2 phase has stuck nodes of its own");
    static_assert(!CHAIN || MODE != MODE_LEAF, "the leaf stage's successors are rows");
    uint32_t (&s_par_plane)[8][NT] = L.par_plane;
    uint32_t (&s_par_game)[NT] = L.par_game, (&s_par_key)[NT] = L.par_key;
    uint16_t (&s_child)[NT * 15] = L.child;
    uint32_t (&s_wave)[NT / 64] = L.wave;
    unsigned long long &s_slot = L.slot;
    constexpr int NW = NT / 64;
#if defined(BG_PHASE_FULL_BARRIERS) && BG_PHASE_FULL_BARRIERS
    constexpr bool PH_LDS = false;
#else
    constexpr bool PH_LDS = true;
#endif
    // Leaf stage: the rows of a phase leave in two runs -- first the successors whose moves hit no blot, then the others.  The value net
    // works 64 consecutive rows per wave for as many gather passes as the LONGEST delta list among them, and a hit is what makes a list
    // longer (two entries: the blot's point and the bar); the class costs one mask per node here, and the two counts share one scan.
#if defined(BG_LEAF_CLASSES) && !BG_LEAF_CLASSES
    constexpr bool CLASSES = false;
#else
    constexpr bool CLASSES = MODE == MODE_LEAF;
#endif
#ifdef BG_EVAL_WGCLOCK                                         // diagnostic build: thread 0's time in the phase's stages -> the top end of the value array
    const unsigned long long pt0 = wall_clock64();
#define BG_PSTAMP(K) do { if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); \
        e.values[e.cap - 1 - 4096 - (long long)blockIdx.x * 8 - (K)] += (float)(now_ - pt_) * 0.01f; pt_ = now_; } } while (0)
    unsigned long long pt_ = pt0;
#else
#define BG_PSTAMP(K) do { } while (0)
#endif
    uint32_t cnt = 0, cntB = 0;            // successors of this node; (PLY2) 1 if the node is stuck -> F; (leaf stage) successors of the second run
    uint32_t succ = 0, succ_hit = 0;       // their origins; (leaf stage) those of the second run
    if (valid) {
        uint32_t m0 = 0;
        {
            const int die = (s.len & 1) ? s.dB : s.dA;
            if (s.len < (s.dbl ? 4 : 2)) m0 = legal_origins(s.own, s.opp, s.pl, die);
                bool pruned_all = false;
                if (MODE == MODE_LEAF && !s.dbl && s.len == 1 && key_pass(nd.key) && m0) {
                    // Non-doubles, second die order: this node is "x with d2", its successors are "then y with d1".
                    // If y was a legal FIRST move with d1 and x is then legal with d2, the first die order already
                    // produced (y, x) -- the same two checker moves, the same landing points, hence the same
                    // afterstate under a smaller key.  Such successors are not generated.
                    Side rown, ropp;
                    split_sides(root_pl, s.pl, rown, ropp);
                    const int x = key_origin(nd.key, 0);
                    const uint32_t ma_root = legal_origins(rown, ropp, s.pl, s.dB);
                    uint32_t cand = m0 & ma_root, dup = 0;
                    const int zx = x + (s.pl ? -s.dA : s.dA);
                    if (zx >= 1 && zx <= 24) {
                        // x lands on the board: its legality after y needs a checker left at x (y == x with a single
                        // checker there -- the bar included -- is the only way y takes it), nothing else: y moves no
                        // opposing checker onto x's landing point and leaves no checker on the bar that was not there
                        dup = cand;
                        if (((rown.b[0] & ~(rown.b[1] | rown.b[2] | rown.b[3])) >> x) & 1u) dup &= ~(1u << x);
                    } else
                        while (cand) {                         // x bears off: y can change that either way, ask
                            const int y = __ffs(cand) - 1; cand &= cand - 1;
                            Side a = rown, b = ropp;
                            apply_move(a, b, s.pl, y, s.dB);
                            if ((legal_origins(a, b, s.pl, s.dA) >> x) & 1u) dup |= 1u << y;
                        }
                    // ... and the one checker that takes both dice: "x with d2, on with d1" lands where "x with d1, on
                    // with d2" does.  Same afterstate iff neither stop-over point holds an opposing checker (a hit on
                    // the way is the only thing the order can change) and the first order is legal.
                    {
                        const int dir = s.pl ? -1 : 1;
                        const int y2 = x + dir * s.dA, z = x + dir * s.dB;
                        if (y2 >= 1 && y2 <= 24 && z >= 1 && z <= 24 && ((m0 >> y2) & 1u) && ((ma_root >> x) & 1u)) {
                            const uint32_t own_any = rown.b[0] | rown.b[1] | rown.b[2] | rown.b[3];
                            const uint32_t opp_any = ropp.b[0] | ropp.b[1] | ropp.b[2] | ropp.b[3];
                            if (!((own_any >> y2) & 1u) && !((opp_any >> y2) & 1u) && !((opp_any >> z) & 1u)) {
                                Side a = rown, b = ropp;
                                apply_move(a, b, s.pl, x, s.dB);
                                if ((legal_origins(a, b, s.pl, s.dA) >> z) & 1u) dup |= 1u << y2;
                            }
                        }
                    }
                    m0 &= ~dup;
                    pruned_all = m0 == 0;
                } else if (s.dbl && s.len >= 1 && m0) {
                    // Doubles: the node's last move was "x", a successor "then y" with y < x has a twin "... y, x" under
                    // a smaller key whenever y was legal before x and x is still legal after y -- the same moves of the
                    // same die, the same landing points.  The twin is the one that is expanded.
                    const int x = key_origin(nd.key, s.len - 1);
                    uint32_t cand = m0 & ((1u << x) - 1u) & legal_origins(prev_own, prev_opp, s.pl, die), dup = 0;
                    const int zx = x + (s.pl ? -die : die);
                    if (zx >= 1 && zx <= 24) dup = cand;       // x lands on the board and y != x: still legal after y (as above)
                    else
                        while (cand) {
                            const int y = __ffs(cand) - 1; cand &= cand - 1;
                            Side a = prev_own, b = prev_opp;
                            apply_move(a, b, s.pl, y, die);
                            if ((legal_origins(a, b, s.pl, die) >> x) & 1u) dup |= 1u << y;
                        }
                    m0 &= ~dup;
                    pruned_all = m0 == 0;
                }
                const bool stuck = m0 == 0 && !pruned_all;     // no legal move (or full depth): the node itself goes on
                if (MODE == MODE_PLY2) { cnt = (uint32_t)__popc(m0); cntB = stuck ? 1u : 0u; }
                else cnt = m0 ? (uint32_t)__popc(m0) : (stuck ? 1u : 0u);
#pragma unroll
                for (int k = 0; k < 4; ++k) { s_par_plane[k][threadIdx.x] = s.own.b[k]; s_par_plane[4 + k][threadIdx.x] = s.opp.b[k]; }
                succ = m0;
                if (CLASSES) {                                 // origins whose move lands on a blot: exactly one opposing checker on the landing point
                    const uint32_t blots = s.opp.b[0] & ~(s.opp.b[1] | s.opp.b[2] | s.opp.b[3]) & PTS;
                    succ_hit = m0 & (s.pl ? (blots << die) : (blots >> die));
                    if (prefix_hit) succ_hit = m0;
                    cntB = m0 ? (uint32_t)__popc(succ_hit) : ((stuck && prefix_hit) ? 1u : 0u);
                }
                s_par_game[threadIdx.x] = nd.game;
                s_par_key[threadIdx.x] = nd.key | (prefix_hit ? 0x01000000u : 0u) | ((uint32_t)die << 27) | (s.pl ? 0x80000000u : 0u);
        }
    }
    BG_PSTAMP(0);                          // node logic
    uint32_t total, totB = 0;
    // (the phase's barriers order LDS only -- barrier_lds(), BG_PHASE_FULL_BARRIERS=1: __syncthreads() everywhere -- except the last
    //  one of a phase whose successors the workgroup's next phase fetches from the list)
    uint32_t off, totA = 0;
    if (CLASSES) {                         // one scan for both runs: first-run count in the low half, second-run count in the high half (<= 7 680 each)
        uint32_t tot2;
        const uint32_t off2 = block_scan_256<NW, true, PH_LDS>((cnt - cntB) | (cntB << 16), &tot2, s_wave);
        totA = tot2 & 0xFFFFu;
        total = totA + (tot2 >> 16);
        uint16_t *ca = s_child + (off2 & 0xFFFFu), *cb = s_child + totA + (off2 >> 16);
        const uint32_t tag = threadIdx.x << 5;
        if (succ == 0 && cnt) *(cntB ? cb : ca) = (uint16_t)(tag | 31u);
        uint32_t sa = succ & ~succ_hit, sb = succ & succ_hit;
        while (sa) { const int o = __ffs(sa) - 1; sa &= sa - 1; *ca++ = (uint16_t)(tag | (uint32_t)o); }
        while (sb) { const int o = __ffs(sb) - 1; sb &= sb - 1; *cb++ = (uint16_t)(tag | (uint32_t)o); }
        off = 0;
        cntB = 0;
    } else {
        off = block_scan_256<NW, true, PH_LDS>(cnt, &total, s_wave);
        // the node names its successors: the lane that builds successor q reads (parent, origin) in one access instead
        // of searching the offsets and stepping through the mask
        uint16_t *c = s_child + off;
        const uint32_t tag = threadIdx.x << 5;
        if (succ == 0 && cnt) *c = (uint16_t)(tag | 31u);
        while (succ) { const int o = __ffs(succ) - 1; succ &= succ - 1; *c++ = (uint16_t)(tag | (uint32_t)o); }
    }
    (void)off;
    BG_PSTAMP(1);                          // scan + successor table
    // (Round 5, measured and not kept: the phase's allocations issued right after the scan and collected after every lane has built its FIRST successor,
    //  the round trip of thread 0's returning atomics under that work -- atomic_add_deferred, bg_staged.h.  The successor held across the barrier costs
    //  registers the kernel does not have: 46 spilled VGPRs and 33.3 us against 28.5; with the block-uniform values in SGPRs 19 and 27.2 against 27.9 us at
    //  65 536 lanes, 21.9 against 21.0 us at 32 768: profiles/r05_ab_expansion_registers.txt.)
    // what goes through the output list: everything, or (CHAIN) the successors past the first NT
    const uint32_t listed = CHAIN ? (total > (uint32_t)NT ? total - (uint32_t)NT : 0u) : total;
    const uint32_t q0 = CHAIN ? (uint32_t)NT : 0u;             // first successor that does
    unsigned long long *topA = MODE == MODE_PLY2 ? &sv.tops[T_D2] : (MODE == MODE_PLY3 ? lp.top : &sv.tops[T_U]);
    const unsigned long long capA = MODE == MODE_PLY2 ? (unsigned long long)sv.cap_d2 : (MODE == MODE_PLY3 ? lp.cap : (unsigned long long)sv.cap_rows);
    // (leaf stage, two arenas: the two runs of the phase are allocated from their own counters, both atomics in flight together)
    const bool two = CLASSES && sv.b_base > 0;
    const int ar0 = s_kind_dbl ? 2 : 0;                        // (block-uniform) the phase's pair of arenas: non-doubles turns 0 / 1, doubles 2 / 3
    unsigned long long baseA, baseB2 = 0;
    bool ok;
    if (two) {
        block_alloc2(&sv.tops[arena_counter(ar0)], totA, &sv.tops[arena_counter(ar0 + 1)], total - totA, L.slot2, baseA, baseB2);
        ok = baseA + totA <= (unsigned long long)sv.b_base && baseB2 + (total - totA) <= (unsigned long long)sv.b_base;
    } else {
        baseA = block_alloc<false, PH_LDS>(topA, listed, &s_slot);     // the scan just synchronised
        ok = baseA + listed <= capA;
    }
    if (MODE == MODE_PLY2) {                               // stuck doubles nodes are leaf parents as they are
        const uint32_t offB = block_scan_256<NW, true, PH_LDS>(cntB, &totB, s_wave);
        if (STUCK_ROWS) {
            const bool four = sv.b_base > 0;                   // (four arenas: these are rows of doubles turns; the class does not matter for so few)
            const unsigned long long baseB = block_alloc<true, PH_LDS>(&sv.tops[arena_counter(four ? 2 : 0)], totB, &s_slot);
            ok = ok && baseB + totB <= (unsigned long long)(four ? sv.b_base : sv.cap_rows);
            if (ok && cntB) {                              // the row of the node's own position (what the leaf stage writes for o == 31)
                const unsigned long long d = (four ? 2ull * (unsigned long long)sv.b_base : 0ull) + baseB + offB;
                const uint32_t pk = s_par_key[threadIdx.x];
                const int pl = (int)(pk >> 31);
                const int a0 = pl ? 4 : 0, b0 = pl ? 0 : 4;
                sv.u_rows[2 * d] = make_uint4(s_par_plane[a0][threadIdx.x] | (pl ? TURN_BIT : 0u), s_par_plane[a0 + 1][threadIdx.x],
                                              s_par_plane[a0 + 2][threadIdx.x], s_par_plane[a0 + 3][threadIdx.x]);
                sv.u_rows[2 * d + 1] = make_uint4(s_par_plane[b0][threadIdx.x], s_par_plane[b0 + 1][threadIdx.x], s_par_plane[b0 + 2][threadIdx.x],
                                                  s_par_plane[b0 + 3][threadIdx.x]);
                sv.u_info[d] = make_uint2(nd.game, nd.key | (pl ? 0x80000000u : 0u));
            }
        } else {
            const unsigned long long baseB = block_alloc<true, PH_LDS>(lp.top, totB, &s_slot);
            ok = ok && baseB + totB <= lp.cap;
            if (ok && cntB) lp.list[baseB + offB] = Node{nd.game, nd.key};
        }
        if (out_stuck) *out_stuck = ok ? totB : 0u;
    }
    if (!ok) flag_overflow(e);
    block_barrier<PH_LDS>();                               // parent records and offsets are in place
    BG_PSTAMP(2);                          // allocation(s)
    if (CHAIN) chain->valid = false;
    if (ok) {
        for (uint32_t q = CHAIN ? threadIdx.x : threadIdx.x + q0; q < total; q += NT) {
            const uint32_t ce = s_child[q];
            const int par = (int)(ce >> 5), o = (int)(ce & 31u);
            const uint32_t pk = s_par_key[par];
            const int pl = (int)(pk >> 31), die = (int)((pk >> 27) & 7u);
            Side a{{s_par_plane[0][par], s_par_plane[1][par], s_par_plane[2][par], s_par_plane[3][par]}};
            Side b{{s_par_plane[4][par], s_par_plane[5][par], s_par_plane[6][par], s_par_plane[7][par]}};
            uint32_t key = pk & KEY_MASK;
            const uint32_t game = s_par_game[par];
            if (CHAIN && q < (uint32_t)NT) {                   // (q == threadIdx.x) this lane's node of the next phase
                chain->pown = a; chain->popp = b;
                bool hit = (pk & 0x01000000u) != 0;
                if (o != 31) {
                    const int dest = pl ? o - die : o + die;           // a blot on the landing point (points 1 .. 24 only)
                    if (dest >= 1 && dest <= 24) hit = hit || (((b.b[0] & ~(b.b[1] | b.b[2] | b.b[3])) >> dest) & 1u);
                    apply_move(a, b, pl, o, die);
                    key = key_child(key, o);
                }
                chain->hit = hit;
                chain->own = a; chain->opp = b; chain->game = game; chain->key = key; chain->pl = pl; chain->die = die;
                chain->valid = true;
                continue;
            }
            if (o != 31) {
                apply_move(a, b, pl, o, die);
                key = key_child(key, o);
            }
            const unsigned long long d = two ? (q >= totA ? (unsigned long long)(ar0 + 1) * sv.b_base + baseB2 + (q - totA)
                                                           : (unsigned long long)ar0 * sv.b_base + baseA + q)
                                             : baseA + (q - q0);
            if (MODE == MODE_LEAF) {
                const Side &s1 = pl ? b : a, &s2 = pl ? a : b;
                sv.u_rows[2 * d] = make_uint4(s1.b[0] | (pl ? TURN_BIT : 0u), s1.b[1], s1.b[2], s1.b[3]);
                sv.u_rows[2 * d + 1] = make_uint4(s2.b[0], s2.b[1], s2.b[2], s2.b[3]);
                sv.u_info[d] = make_uint2(game, key | (pl ? 0x80000000u : 0u));
            } else if (MODE == MODE_PLY2) sv.d2[d] = Node{game, key};
            else lp.list[d] = Node{game, key};
        }
    }
    // the records are reused by the next phase; where that phase fetches what this one wrote to a list (one ply per phase: not the
    // rows, not the register hand-off) its reads come after a barrier that has waited for the stores
    block_barrier<PH_LDS && (MODE == MODE_LEAF || CHAIN)>();
    BG_PSTAMP(3);                          // successors built and written
#undef BG_PSTAMP
    *out_base = baseA;
    *out_total = ok ? listed : 0u;
    if (out_all) *out_all = ok ? total : 0u;
}

// ... with the node fetched from a list: its position is rebuilt from the game's planes by replaying the key
template <int MODE, int NT, bool STUCK_ROWS = false, bool CHAIN = false>
__device__ __forceinline__ void expand_phase(const EnvView &e, const StagedView &sv, const NodeIn &x, int np,
                                             unsigned long long *out_base, uint32_t *out_total, const LeafParents &lp,
                                             ExpandLds<NT> &L, uint32_t *out_stuck = nullptr, ChainNode *chain = nullptr,
                                             uint32_t *out_all = nullptr, bool kind_dbl = false)
{
    (void)np;
    NodeState s;
    s.own = s.opp = Side{{0, 0, 0, 0}};
    s.pl = 0; s.dA = s.dB = 1; s.len = 0; s.dbl = false;
    Side prev_own{{0, 0, 0, 0}}, prev_opp{{0, 0, 0, 0}};
    bool prefix_hit = false;
    if (x.valid) {
        node_build(x.nd, x.pl, x.meta, s, &prev_own, &prev_opp);
        // has a move of the key hit a blot?  Then the opponent's bar counter is not the root's.
        Side rown, ropp;
        split_sides(x.pl, s.pl, rown, ropp);
        const int bar = s.pl ? 0 : 25;
        prefix_hit = ((((s.opp.b[0] ^ ropp.b[0]) | (s.opp.b[1] ^ ropp.b[1]) | (s.opp.b[2] ^ ropp.b[2]) | (s.opp.b[3] ^ ropp.b[3])) >> bar) & 1u) != 0;
    }
    expand_phase_core<MODE, NT, STUCK_ROWS, CHAIN>(e, sv, x.valid, x.nd, s, prev_own, prev_opp, x.pl, prefix_hit, out_base, out_total, lp, L,
                                                   out_stuck, chain, out_all, kind_dbl);
}

// ... with the node handed over in registers by the phase before (doubles turns only: the die is the same at every ply)
template <int MODE, int NT>
__device__ __forceinline__ void expand_phase_chained(const EnvView &e, const StagedView &sv, const ChainNode &n,
                                                     unsigned long long *out_base, uint32_t *out_total, const LeafParents &lp,
                                                     ExpandLds<NT> &L, ChainNode *chain, uint32_t *out_all)
{
    NodeState s;
    s.own = n.own; s.opp = n.opp; s.pl = n.pl; s.dA = s.dB = n.die; s.dbl = true; s.len = key_len(n.key);
    const uint32_t no_root[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // (only the non-doubles rule reads the root)
    expand_phase_core<MODE, NT, false, MODE != MODE_LEAF>(e, sv, n.valid, Node{n.game, n.key}, s, n.pown, n.popp, no_root, n.valid && n.hit, out_base,
                                                          out_total, lp, L, nullptr, chain, out_all, true);
}

// nodes per block iteration: a launch with few nodes (the doubles plies, small envs) spreads them over the whole grid,
// 64 per workgroup at least
__device__ __forceinline__ unsigned long long nodes_per_block(unsigned long long n_in, int nt)
{
    unsigned long long npb = (n_in + gridDim.x - 1) / gridDim.x;
    npb = (npb + 63) & ~63ull;
    return npb < 64 ? 64 : (npb > (unsigned long long)nt ? (unsigned long long)nt : npb);
}

#ifdef BGAMD_EXPERIMENTAL          // rounds 1-4's two launches below the roots: bit-identity references of expand_all_kernel (round 5: experimental build only)
// leaf stage (and a single doubles ply, kept for tests and experiments)
template <int MODE>
__global__ __launch_bounds__(expand_threads(MODE)) void expand_kernel(EnvView e, StagedView sv)
{
    constexpr int NT = expand_threads(MODE);
    __shared__ ExpandLds<NT> L;
    const Node *in = MODE == MODE_PLY2 ? sv.d1 : (MODE == MODE_PLY3 ? sv.d2 : sv.f);
    const unsigned long long cap_in = (unsigned long long)(MODE == MODE_PLY2 ? sv.cap_d1 : (MODE == MODE_PLY3 ? sv.cap_d2 : sv.cap_f));
    unsigned long long n_in = sv.tops[MODE == MODE_PLY2 ? T_D1 : (MODE == MODE_PLY3 ? T_D2 : T_F)];
    if (n_in > cap_in) n_in = cap_in;
    unsigned long long staged_total = 0;
    if (MODE == MODE_LEAF && blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&e.counters[C_FNODES], n_in);
        atomicAdd(&e.counters[C_DNODES], sv.tops[T_D1] + sv.tops[T_D2]);
    }
    // contiguous shares: a workgroup's nodes are all of one kind (the list holds the non-doubles leaf parents first, the
    // doubles ones at the end), so its rows form tiles of similar delta-list lengths for the value net
    const unsigned long long NPB = nodes_per_block(n_in, NT);
    auto idx_of = [&](unsigned long long blk) -> long long {
        const unsigned long long node = blk * NPB + threadIdx.x;
        return (threadIdx.x < NPB && node < n_in) ? (long long)node : -1ll;
    };
    // software pipeline over block iterations: the next iteration's node, planes and meta are in flight while this
    // iteration scans, allocates and writes its successors
    NodeIn cur, nxt;
    node_fetch(e, in, blockIdx.x * NPB < n_in ? idx_of(blockIdx.x) : -1ll, cur);
    for (unsigned long long blk = blockIdx.x; blk * NPB < n_in; blk += gridDim.x) {
        const unsigned long long nb = blk + gridDim.x;
        node_fetch(e, in, nb * NPB < n_in ? idx_of(nb) : -1ll, nxt);
        unsigned long long base;
        uint32_t total;
        expand_phase<MODE, NT>(e, sv, cur, (int)NPB, &base, &total, leaf_parents_shared(sv), L);
        staged_total += total;
        cur = nxt;
    }
    if (MODE == MODE_LEAF && threadIdx.x == 0 && staged_total) atomicAdd(&e.counters[C_CAND_RAW], staged_total);
}

// the doubles turns' plies 2 AND 3 in one launch: a workgroup expands its share of the ply-1 nodes and goes straight on
// with the ply-2 nodes it just wrote (one contiguous allocation), so the second ply costs neither a launch nor a
// trip through the list counter
__global__ __launch_bounds__(expand_threads(MODE_PLY2)) void doubles_kernel(EnvView e, StagedView sv)
{
    constexpr int NT = expand_threads(MODE_PLY2);
    __shared__ ExpandLds<NT> L;
    unsigned long long n_in = sv.tops[T_D1];
    if (n_in > (unsigned long long)sv.cap_d1) n_in = (unsigned long long)sv.cap_d1;
    const unsigned long long NPB = nodes_per_block(n_in, NT / 4);
    for (unsigned long long blk = blockIdx.x; blk * NPB < n_in; blk += gridDim.x) {
        const unsigned long long first = blk * NPB;
        const unsigned long long cnt = n_in - first < NPB ? n_in - first : NPB;
        unsigned long long base2;
        uint32_t total2;
        NodeIn x2;
        node_fetch(e, sv.d1, threadIdx.x < cnt ? (long long)(first + threadIdx.x) : -1ll, x2);
        expand_phase<MODE_PLY2, NT>(e, sv, x2, (int)NPB, &base2, &total2, leaf_parents_shared(sv), L);
        for (uint32_t c = 0; c < total2; c += NT) {
            const unsigned long long n3 = total2 - c < (uint32_t)NT ? total2 - c : (uint32_t)NT;
            unsigned long long base3;
            uint32_t total3;
            NodeIn x3;
            node_fetch(e, sv.d2, threadIdx.x < n3 ? (long long)(base2 + c + threadIdx.x) : -1ll, x3);
            expand_phase<MODE_PLY3, NT>(e, sv, x3, NT, &base3, &total3, leaf_parents_shared(sv), L);
        }
    }
}
#endif  // BGAMD_EXPERIMENTAL

// The whole expansion below the roots in ONE launch (round 4; the experimental build's BGAMD_EXPAND_MERGED=0 brings the two launches back).
// The doubles plies (28 k + 45 k nodes at 65 536 lanes) are a chain of latencies that leaves most of the chip idle, and the leaf
// stage of the NON-doubles turns -- three fifths of the leaf stage -- does not depend on them: its parents are written by the roots.
//   workgroups [0, n_dbl)       a share of the doubles turns' ply-1 nodes through ply 2, ply 3 AND their leaf stage, one phase
//                               after the other on the workgroup's own allocations (leaf parents in F2, rows in the common arena);
//   workgroups [n_dbl, grid)    the leaf stage over F, which in this mode holds nothing but the roots' output.
// No workgroup waits for another.  The rows are the same rows, in another order of the arena -- nothing depends on that order
// (a row's value is a function of the row; the arg-max keeps the smallest key).
#ifndef BG_XALL_NT
#define BG_XALL_NT 512
#endif
constexpr int XALL_NT = BG_XALL_NT;
__global__ __launch_bounds__(XALL_NT, 4) void expand_all_kernel(EnvView e, StagedView sv, unsigned n_dbl, unsigned dbl_npb, unsigned parts,
                                                                  unsigned long long *__restrict__ zero_words, int n_zero_words)
{
    constexpr int NT = XALL_NT;
    // multi-step runs: the OTHER set of list counters is cleared here, while no kernel is using it, for the roots of the next step (which share
    // a launch with this step's apply)
    if (zero_words && blockIdx.x == gridDim.x - 1 && (int)threadIdx.x < n_zero_words) zero_words[threadIdx.x] = 0ull;
    __shared__ ExpandLds<NT> L;
#ifdef BG_EVAL_WGCLOCK
    const unsigned long long xall_t0 = wall_clock64();
    if (threadIdx.x == 0) for (int k = 0; k < 8; ++k) e.values[e.cap - 1 - 4096 - (long long)blockIdx.x * 8 - k] = 0.0f;
#endif
    // (the launch's three statistics as 32-bit per-thread accumulators: as 64-bit ones they held six VGPRs for the whole launch in a kernel built at its
    //  register cap; a workgroup stages a few thousand rows at most)
    uint32_t acc_staged = 0, acc_fnodes = 0, acc_dnodes = 0;
    auto add_staged = [&](unsigned long long v) { acc_staged += (uint32_t)v; };
    auto add_fnodes = [&](unsigned long long v) { acc_fnodes += (uint32_t)v; };
    auto add_dnodes = [&](unsigned long long v) { acc_dnodes += (uint32_t)v; };
    // the first n_dbl workgroups take the doubles turns, the others the non-doubles leaf stage.  (Every workgroup taking a share of both
    // kinds -- the same mix everywhere -- was measured too: 32-37 us against 29.4: the two kinds overlap when they share a CU as
    // different workgroups, and follow one another inside one.)
    // (parts: timing experiments only -- bit 0: the doubles turns, bit 1: the non-doubles leaf stage; 3 = the whole step)
    const bool do_dbl = blockIdx.x < n_dbl && (parts & 1u), do_leaf = blockIdx.x >= n_dbl && (parts & 2u);
    if (do_dbl) {
        const LeafParents lp{sv.f2, &sv.tops[T_F2], (unsigned long long)sv.cap_f2};
        // (sharded lists: this workgroup takes list d_first % shards, as the d_first / shards-th of d_stride / shards workgroups)
        const int shd = sv.shards > 1 ? (int)(blockIdx.x % (unsigned)sv.shards) : 0;
        const unsigned long long capD = (unsigned long long)(sv.cap_d1 / sv.shards);
        const Node *d1_list = sv.d1 + (unsigned long long)shd * capD;
        unsigned long long n_in = sv.tops[d1_counter(shd)];
        if (n_in > capD) n_in = capD;
        const unsigned long long d_first = (unsigned long long)blockIdx.x / (unsigned)sv.shards, d_stride = (unsigned long long)n_dbl / (unsigned)sv.shards;
        if (d_first == 0) add_dnodes(n_in);
        unsigned long long NPB = (n_in + d_stride - 1) / d_stride;
        NPB = NPB < 1 ? 1 : (NPB > (unsigned long long)dbl_npb ? (unsigned long long)dbl_npb : NPB);           // (dbl_npb <= NT)
        // leaf stage over a range of the workgroup's own F2 entries (successors that did not fit the register hand-off)
        auto leaf_range = [&](unsigned long long base, uint32_t n) {
            for (uint32_t c4 = 0; c4 < n; c4 += NT) {
                const unsigned long long n4 = n - c4 < (uint32_t)NT ? n - c4 : (uint32_t)NT;
                unsigned long long base4;
                uint32_t total4;
                NodeIn x4;
                node_fetch(e, sv.f2, threadIdx.x < n4 ? (long long)(base + c4 + threadIdx.x) : -1ll, x4);
                expand_phase<MODE_LEAF, NT>(e, sv, x4, NT, &base4, &total4, lp, L, nullptr, nullptr, nullptr, true);
                add_staged(total4);
            }
        };
        // (a workgroup takes a CONTIGUOUS run of the list: every W-th node instead -- a game's ply-1 nodes on different workgroups, whose
        // rows then vary less than the 600 ... 3 200 of today -- was measured: expansion 29.1 -> 30.7 us, and the value net 74.5 -> 76.0 us
        // because a game's rows then lie all over the arenas)
        for (unsigned long long blk = d_first; blk * NPB < n_in; blk += d_stride) {
            const unsigned long long first = blk * NPB;
            const unsigned long long cnt = n_in - first < NPB ? n_in - first : NPB;
            // ply 2 from the roots' list; its first NT successors go on in registers, lane q with successor q ...
            unsigned long long over2_base, over3_base, base4;
            uint32_t over2, over3, all2, all3, total4, stuck2;
            NodeIn x2;
            ChainNode n3, n4;
            node_fetch(e, d1_list, threadIdx.x < cnt ? (long long)(first + threadIdx.x) : -1ll, x2);
            expand_phase<MODE_PLY2, NT, true, true>(e, sv, x2, (int)NPB, &over2_base, &over2, lp, L, &stuck2, &n3, &all2);
            add_dnodes(all2);
            add_fnodes(stuck2);
            add_staged(stuck2);
            // ... ply 3 on those, the same way, then their leaf stage
            expand_phase_chained<MODE_PLY3, NT>(e, sv, n3, &over3_base, &over3, lp, L, &n4, &all3);
            add_fnodes(all3);
            expand_phase_chained<MODE_LEAF, NT>(e, sv, n4, &base4, &total4, lp, L, nullptr, nullptr);
            add_staged(total4);
            if (over2 | over3) __syncthreads();                // (block-uniform) listed successors: their stores have to have landed
            leaf_range(over3_base, over3);                     // ply-3 successors past the first NT: through F2
            for (uint32_t c = 0; c < over2; c += NT) {          // ply-2 successors past the first NT: through D2, one ply per phase
                const unsigned long long n3c = over2 - c < (uint32_t)NT ? over2 - c : (uint32_t)NT;
                unsigned long long base3;
                uint32_t total3;
                NodeIn x3;
                node_fetch(e, sv.d2, threadIdx.x < n3c ? (long long)(over2_base + c + threadIdx.x) : -1ll, x3);
                expand_phase<MODE_PLY3, NT>(e, sv, x3, NT, &base3, &total3, lp, L);
                add_fnodes(total3);
                leaf_range(base3, total3);
            }
        }
    }
    if (do_leaf) {
        const unsigned int lw = blockIdx.x - n_dbl;
        const int shf = sv.shards > 1 ? (int)(lw % (unsigned)sv.shards) : 0;
        const unsigned long long capF = (unsigned long long)(sv.cap_f / sv.shards);
        const Node *f_list = sv.f + (unsigned long long)shf * capF;
        unsigned long long n_in = sv.tops[f_counter(shf)];
        if (n_in > capF) n_in = capF;
        const unsigned long long l_first = lw / (unsigned)sv.shards, l_stride = (unsigned long long)(gridDim.x - n_dbl) / (unsigned)sv.shards;
        if (l_first == 0) add_fnodes(n_in);
        // a contiguous share per workgroup, cut into equal phases of at most NT parents (645 parents are 2 x 323, not 512 + 133: a phase
        // costs its latencies whatever it holds, and equal shares end together)
        const unsigned long long share = (n_in + l_stride - 1) / l_stride;
        const unsigned long long lo = l_first * share < n_in ? l_first * share : n_in;
        const unsigned long long hi = lo + share < n_in ? lo + share : n_in;
        const unsigned long long chunks = (hi - lo + NT - 1) / NT;
        const unsigned long long per = chunks ? (hi - lo + chunks - 1) / chunks : 0;
        auto idx_of = [&](unsigned long long c) -> long long {
            const unsigned long long node = lo + c * per + threadIdx.x;
            return (c < chunks && threadIdx.x < per && node < hi) ? (long long)node : -1ll;
        };
        NodeIn cur, nxt;
        node_fetch(e, f_list, idx_of(0), cur);
        for (unsigned long long c = 0; c < chunks; ++c) {
            node_fetch(e, f_list, idx_of(c + 1), nxt);
            unsigned long long base;
            uint32_t total;
            expand_phase<MODE_LEAF, NT>(e, sv, cur, (int)per, &base, &total, leaf_parents_shared(sv), L);
            add_staged(total);
            cur = nxt;
        }
    }
    if (threadIdx.x == 0) {
        const unsigned long long staged_total = acc_staged, fnodes = acc_fnodes, dnodes = acc_dnodes;
        if (staged_total) atomicAdd(&e.counters[C_CAND_RAW], staged_total);
        if (fnodes) atomicAdd(&e.counters[C_FNODES], fnodes);
        if (dnodes) atomicAdd(&e.counters[C_DNODES], dnodes);
#ifdef BG_EVAL_WGCLOCK                                         // diagnostic build (tools/eval_wg_clock.py): when each workgroup ended, us after it started, and its rows
        e.values[e.cap - 1 - 1024 - (long long)blockIdx.x] = (float)(wall_clock64() - xall_t0) * 0.01f;
        e.values[e.cap - 1 - 2048 - (long long)blockIdx.x] = (float)staged_total;
#endif
    }
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
struct ExploreView {                                       // counted tasks of the exploring lanes (bg_random_kernels.h)
    const Node *tasks; const uint32_t *task_count, *task_off, *task_n;
};
__device__ __forceinline__ bool explore_pick(const EnvView &e, const ExploreView &xv, long long g, uint32_t u, Side &own,
                                             Side &opp, uint32_t &key, uint32_t &k, uint32_t &C);

__device__ __forceinline__ void apply_body(const EnvView &e, const StagedView &sv, const ExploreView &xv, int flags, float epsilon,
                                           long long g, LaneCtx *fwd = nullptr)
{
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
        // model.py:205-206: an exploring lane takes the k-th sequence of the reference-order list, k from the TURN
        // stream, located through the lane's counted tasks (bounded work; no lane walks a whole tree)
        const bool explore = epsilon > 0.0f && (float)(c.x.w >> 8) * (1.0f / 16777216.0f) < epsilon;
        uint32_t key = ~(uint32_t)pack & 0x7FFFFFFFu;
        const uint32_t vb = (uint32_t)(pack >> 32);
        cval = __uint_as_float(c.turn ? ~vb : vb);
        uint32_t xk = 0, xC = 0;
        const bool explored = explore && explore_pick(e, xv, g, c.x.z, own, opp, key, xk, xC);
        {
            const int len = key_len(key), pass = key_pass(key);
            const int dA = pass ? c.d2 : c.d1, dB = pass ? c.d1 : c.d2;
            uint32_t origins = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < len) origins |= (uint32_t)key_origin(key, k) << (5 * k);
            if (explored) { chosen = (int32_t)xk; ccount = xC; cval = 0.0f; }
            else if (flags & BGAMD_WANT_INDEX) {                // reference-order index + list length (slow path)
                IndexVisitor iv;
                iv.t_orig = origins; iv.t_len = (uint32_t)len; iv.t_dA = (uint32_t)dA; iv.t_dB = (uint32_t)dB;
                walk_sequences(own, opp, c.turn, c.d1, c.d2, iv);
                chosen = (int32_t)iv.idx; ccount = iv.n;
            } else chosen = 0;                             // "a move was made"; exact index only on request
            if (!explored) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < len) apply_move(own, opp, c.turn, key_origin(key, k), (k & 1) ? dB : dA);
            }
            cseq = seq_pack(origins, len, dA, dB) | (c.turn ? (1u << 29) : 0u);   // bit 29: mover moves down
        }
        join_sides(own, opp, c.turn, c.p);
    }
    if (c.live) { e.chosen[g] = chosen; e.chosen_seq[g] = cseq; e.chosen_val[g] = cval; e.cand_cnt[g] = ccount; }
    uint32_t after[3] = {c.meta, c.ply, c.epi};            // a lane that does not take part keeps its state
    finish_turn(e, g, c.p, c.turn, c.d1, c.d2, c.ply, c.epi, flags, c.live, after);
    if (fwd) {
#pragma unroll
        for (int k = 0; k < 8; ++k) fwd->p[k] = c.p[k];
        fwd->meta = after[0]; fwd->ply = after[1]; fwd->epi = after[2];
    }
}

__global__ __launch_bounds__(LANE_NT) void apply_kernel(EnvView e, StagedView sv, ExploreView xv, int flags, float epsilon)
{
    apply_body(e, sv, xv, flags, epsilon, (long long)blockIdx.x * LANE_NT + threadIdx.x);
}

// Step boundary of a multi-step run: the apply of step t and the roots of step t+1 for the same lane in one launch
// (the lane's new position is read back by the thread that just stored it).  sv_next carries the other set of list
// counters: the value-net kernel of step t cleared it while nothing was using it.
// ROOT: the workgroup goes on with the value net's root pass of step t + 1 for its 256 games (bg_root_resident.h: the same bits as
// root_hidden_resident_kernel; needs LANE_NT == 256 and BROOT_LDS_BYTES of dynamic LDS) -- the step then has no root-pass launch
template <bool ROOT>
__global__ __launch_bounds__(LANE_NT, 2) void boundary_kernel(EnvView e, StagedView sv, StagedView sv_next, ExploreView xv, int flags,
                                                           float epsilon, const uint4 *__restrict__ wl3, const uint2 *__restrict__ lut,
                                                           const float *__restrict__ b1)
{
    static_assert(!ROOT || LANE_NT == BROOT_THREADS, "the in-launch root pass is written for 256-thread workgroups");
    // with the root pass a workgroup owns BROOT_GPW games (the threads past them idle through the lane-per-game halves: g = n is no lane of the env)
    constexpr int GPW = ROOT ? BROOT_GPW : LANE_NT;
    const long long g = (int)threadIdx.x < GPW ? (long long)blockIdx.x * GPW + threadIdx.x : e.n;
    LaneCtx next;                                       // the lane's new state goes on in registers: no read-back
    apply_body(e, sv, xv, flags, epsilon, g, &next);
    __syncthreads();                                    // finish_turn's statistics scratch is free again
    uint4 row[2];
    BRootWeights wf;
    if (ROOT) broot_load_weights(wl3, wf);
    RootsPending P;
    roots_issue(e, sv_next, flags, g, &next, ROOT ? row : nullptr, P);
    // (round 5) the roots' two list allocations are in flight while the root pass stages its A operands: their round trip used to stand
    // in front of it (-DBG_ALLOC_DEFERRED=0: collect first, as rounds 1-4 did)
#if defined(BG_ALLOC_DEFERRED) && !BG_ALLOC_DEFERRED
    roots_collect(e, sv_next, P);
#endif
    int n_tiles = 0;
    if (ROOT) n_tiles = boundary_root_stage(row[0], row[1], (long long)blockIdx.x * GPW, e.n, lut);
#if !defined(BG_ALLOC_DEFERRED) || BG_ALLOC_DEFERRED
    roots_collect(e, sv_next, P);
#endif
    if (ROOT) boundary_root_compute(wf, n_tiles, (long long)blockIdx.x * GPW, e.n, b1, sv_next.root_hidden);
}
