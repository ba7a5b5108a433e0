// azr_engine.hip — kernels and C-ABI (include/azr.h) of the batched Risk state-step + flattened MCTS.
// One wavefront per game; grid = G workgroups of 64 threads.  gfx950 only.
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>

#include "azr_internal.hpp"
#include "azr_players.hpp"

using namespace azr;

#define HIPCHK(h, call)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (void)hipGetLastError(); /* the runtime's last-error slot is sticky: clear it */        \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e__);                      \
            return AZR_E_HIP;                                                                   \
        }                                                                                       \
    } while (0)

// ================================================================================================
// device helpers
// ================================================================================================
__device__ __forceinline__ Tree tree_of(const Dev& E, int g)
{
    Tree t;
    t.C = E.C; t.H = E.H; t.DMAX = E.DMAX;
    t.nodes = E.nodes + (size_t)g * E.C * NODE_BYTES;
    t.touch = E.touch + (size_t)g * E.C;
    t.nhash = E.nhash + (size_t)g * E.C;
    t.table = E.table + (size_t)g * E.H;
    t.freel = E.freel + (size_t)g * E.C;
    t.path = E.path + (size_t)g * E.T * E.DMAX;  // thread 0's stack; thread_tree() selects thread k's
    return t;
}

// the opponent AlphaZero player's tree of game g (two-net arena)
__device__ __forceinline__ Tree tree2_of(const Dev& E, int g)
{
    Tree t = tree_of(E, g);
    t.nodes = E.nodes2 + (size_t)g * E.C * NODE_BYTES;
    t.touch = E.touch2 + (size_t)g * E.C;
    t.nhash = E.nhash2 + (size_t)g * E.C;
    t.table = E.table2 + (size_t)g * E.H;
    t.freel = E.freel2 + (size_t)g * E.C;
    return t;
}
// tree 2's allocator / trim state lives outside the (full) Ctl line; it is swapped into the Ctl fields the tree
// functions use while that tree is being worked on
struct TreeCtl { uint32_t search_id, nfree, hiwater; };
__device__ __forceinline__ void swap_tree_ctl(Ctl& c, TreeCtl& x)
{
    uint32_t a = c.search_id, b = c.nfree, d = c.hiwater;
    c.search_id = x.search_id; c.nfree = x.nfree; c.hiwater = x.hiwater;
    x.search_id = a; x.nfree = b; x.hiwater = d;
}

__device__ __forceinline__ void ctl_load(Ctl& c, const Ctl* src)
{
    const uint32_t* p = reinterpret_cast<const uint32_t*>(src);
    uint32_t w = p[lane_id() & 31u];
    c.mode = rdl(w, 0); c.search_id = rdl(w, 1); c.sims_done = rdl(w, 2); c.pending = rdl(w, 3);
    c.sims_started = rdl(w, 4); c.nfree = rdl(w, 5); c.hiwater = rdl(w, 6); c.search_done = rdl(w, 7);
    c.rng = rdl(w, 8); c.game_no = rdl(w, 9); c.nsamples = rdl(w, 10); c.status = (int32_t)rdl(w, 11);
    c.error = rdl(w, 12); c.last_move = rdl(w, 13); c.decisions = rdl(w, 14); c.seed = rdl(w, 15);
    c.arena_state = rdl(w, 16); c.player_start = rdl(w, 17); c.pair_phase = rdl(w, 18); c.turn_started = rdl(w, 19);
    c.search_active = rdl(w, 20); c.slot_games = rdl(w, 21); c.dup_dropped = rdl(w, 22);
#pragma unroll
    for (int k = 0; k < MAX_THREADS; k++) c.plen[k] = rdl(w, 23 + k);
    c.search_tree = rdl(w, 31);
}
// c.plen[k] with a wave-uniform runtime k, without indexing the register array
__device__ __forceinline__ uint32_t plen_get(const Ctl& c, int k)
{
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < MAX_THREADS; i++) v = k == i ? c.plen[i] : v;
    return v;
}
__device__ __forceinline__ void plen_set(Ctl& c, int k, uint32_t v)
{
#pragma unroll
    for (int i = 0; i < MAX_THREADS; i++) c.plen[i] = k == i ? v : c.plen[i];
}
__device__ __forceinline__ Tree thread_tree(const Tree& t, int k)
{
    Tree tk = t;
    tk.path = t.path + (size_t)k * t.DMAX;
    return tk;
}
__device__ __forceinline__ void ctl_store(const Ctl& c, Ctl* dst)
{
    uint32_t l = lane_id();
    uint32_t w = 0;
    w = l == 0 ? c.mode : w; w = l == 1 ? c.search_id : w; w = l == 2 ? c.sims_done : w; w = l == 3 ? c.pending : w;
    w = l == 4 ? c.sims_started : w; w = l == 5 ? c.nfree : w; w = l == 6 ? c.hiwater : w; w = l == 7 ? c.search_done : w;
    w = l == 8 ? c.rng : w; w = l == 9 ? c.game_no : w; w = l == 10 ? c.nsamples : w; w = l == 11 ? (uint32_t)c.status : w;
    w = l == 12 ? c.error : w; w = l == 13 ? c.last_move : w; w = l == 14 ? c.decisions : w; w = l == 15 ? c.seed : w;
    w = l == 16 ? c.arena_state : w; w = l == 17 ? c.player_start : w; w = l == 18 ? c.pair_phase : w;
    w = l == 19 ? c.turn_started : w; w = l == 20 ? c.search_active : w; w = l == 21 ? c.slot_games : w;
    w = l == 22 ? c.dup_dropped : w;
#pragma unroll
    for (int k = 0; k < MAX_THREADS; k++) w = l == 23u + k ? c.plen[k] : w;
    w = l == 31 ? c.search_tree : w;
    if (l < 32) reinterpret_cast<uint32_t*>(dst)[l] = w;
}

// ================================================================================================
// rules kernels (UtilityNN / State seams)
// ================================================================================================
__global__ __launch_bounds__(64) void k_new_games(Dev E, const uint32_t* seeds)
{
    const int g = blockIdx.x;
    WS s;
    ws_blank(s);
    s.rng = rng_seed(rfl(seeds[g]));
    new_game(s);
    ws_store(s, E.state + (size_t)g * GREC);
    if (lane_id() == 0) E.ctl[g].rng = s.rng;
}

__global__ __launch_bounds__(64) void k_valid_moves(Dev E, uint64_t* out)
{
    const int g = blockIdx.x;
    WS s;
    ws_load(s, E.state + (size_t)g * GREC);
    uint64_t vm = valid_moves(s, E.rules);
    if (lane_id() == 0) out[g] = vm;
}

__global__ __launch_bounds__(64) void k_make_moves(Dev E, const uint8_t* moves, uint8_t* rc)
{
    const int g = blockIdx.x;
    const uint32_t mv = rfl(moves[g]);
    if (mv == 255u) {
        if (lane_id() == 0) rc[g] = 0;
        return;
    }
    WS s;
    ws_load(s, E.state + (size_t)g * GREC);
    s.rng = rfl(E.ctl[g].rng);
    make_move(s, mv, E.rules);
    if (s.err == 0) {  // a throwing move leaves the stored game untouched
        ws_store(s, E.state + (size_t)g * GREC);
        if (lane_id() == 0) E.ctl[g].rng = s.rng;
    }
    if (lane_id() == 0) rc[g] = (uint8_t)s.err;
}

__global__ __launch_bounds__(64) void k_status(Dev E, int8_t* out)
{
    const int g = blockIdx.x;
    WS s;
    ws_load(s, E.state + (size_t)g * GREC);
    int st = game_status(s, E.rules);
    if (lane_id() == 0) out[g] = (int8_t)st;
}

__global__ __launch_bounds__(64) void k_encode(Dev E, uint8_t* out /*[G][88]*/)
{
    const int g = blockIdx.x;
    WS s;
    ws_load(s, E.state + (size_t)g * GREC);
    encode88(s, out + (size_t)g * 88);
}

// reference `Data` image (160 B) -> 64-B record.  The five masks / totalArmy of the image are derived data and
// are ignored on import (recomputed on export).
__global__ __launch_bounds__(64) void k_import160(Dev E, const uint8_t* data160)
{
    const int g = blockIdx.x;
    const uint8_t* d = data160 + (size_t)g * 160;
    const uint32_t l = lane_id();
    uint32_t b = 0;
    if (l < LANDS) b = d[l];
    b = l == GR_CUR ? d[146] : b;
    b = l == GR_CARD_SETS ? d[147] : b;
    b = l == GR_REINF ? d[148] : b;
    b = l == GR_PHASE ? d[149] : b;
    b = l == GR_MOB_FROM ? d[150] : b;
    b = l == GR_MOB_TO ? d[151] : b;
    b = l == GR_ALLOW_DRAW ? d[152] : b;
    b = l == GR_ATTACKS ? d[153] : b;
    b = l == GR_ROUND_LO ? d[144] : b;
    b = l == GR_ROUND_HI ? d[145] : b;
    b = l == GR_CARDS0 ? d[48 + 40] : b;
    b = l == GR_CARDS1 ? d[96 + 40] : b;
    E.state[(size_t)g * GREC + l] = (uint8_t)b;
}

__global__ __launch_bounds__(64) void k_export160(Dev E, const uint8_t* records, uint8_t* data160)
{
    const int g = blockIdx.x;
    uint8_t* d = data160 + (size_t)g * 160;
    const uint32_t l = lane_id();
    WS s;
    ws_load(s, records + (size_t)g * GREC);
    for (uint32_t i = l; i < 160; i += 64) d[i] = 0;
    wave_mem_sync();
    if (l < LANDS) d[l] = (uint8_t)s.la;
    for (uint32_t p = 0; p < 2; p++) {
        uint64_t m[5] = {m_owned(s, p), m_owned_army(s, p), m_owned_full(s, p), m_attack(s, p), m_attack_army(s, p)};
        int ta = total_army(s, p);
        uint8_t* q = d + 48 + 48 * p;
        if (l < 30) {  // 5 masks x 6 bytes
            uint32_t k = l / 6, by = l % 6;
            q[8 * k + by] = (uint8_t)(m[k] >> (8 * by));
        }
        if (l == 30) q[38] = (uint8_t)(ta & 0xff);
        if (l == 31) q[39] = (uint8_t)((ta >> 8) & 0xff);
        if (l == 32) q[40] = (uint8_t)cards_of(s, p);
    }
    if (l == 0) {
        d[144] = (uint8_t)(s.round & 0xff); d[145] = (uint8_t)(s.round >> 8); d[146] = (uint8_t)s.cur;
        d[147] = (uint8_t)s.card_sets; d[148] = (uint8_t)s.reinf; d[149] = (uint8_t)s.phase;
        d[150] = (uint8_t)s.mob_from; d[151] = (uint8_t)s.mob_to; d[152] = (uint8_t)s.allow_draw;
        d[153] = (uint8_t)s.attacks;
    }
}

// ================================================================================================
// search kernels
// ================================================================================================
__global__ __launch_bounds__(64) void k_tree_clear(Dev E)
{
    const int g = blockIdx.x;
    Ctl c;
    ctl_load(c, &E.ctl[g]);
    Tree t = tree_of(E, g);
    // hiwater may be stale at creation: clear everything the pool can hold
    c.hiwater = (uint32_t)E.C;
    tree_clear(t, c);
    c.pending = 0; c.sims_started = 0; c.sims_done = 0; c.search_done = 1;
    ctl_store(c, &E.ctl[g]);
}

__global__ __launch_bounds__(64) void k_tree_trim(Dev E)
{
    const int g = blockIdx.x;
    Ctl c;
    ctl_load(c, &E.ctl[g]);
    Tree t = tree_of(E, g);
    tree_trim(t, c);
    ctl_store(c, &E.ctl[g]);
}

// AlphaZeroMCTS::simulate prologue (setRootState's trimNodes) for host-stepped searches
__global__ __launch_bounds__(64) void k_search_begin(Dev E)
{
    const int g = blockIdx.x;
    Ctl c;
    ctl_load(c, &E.ctl[g]);
    Tree t = tree_of(E, g);
    WS s;
    ws_load(s, E.state + (size_t)g * GREC);
    tree_trim(t, c);
    c.mode = 1;
    c.sims_done = 0; c.sims_started = 0; c.pending = 0; c.error = 0;
    c.search_done = game_status(s, E.rules) != ST_NOT_ENDED ? 1u : 0u;
    ctl_store(c, &E.ctl[g]);
}

// packs one finished game's staged records into the 265-byte on-disk layout (alphazero_nn_data.cpp:123-130)
__device__ __forceinline__ void flush_samples(const Dev& E, int g, uint32_t n, int status, unsigned long long& dropped)
{
    if (n == 0) return;
    unsigned long long start = 0;
    if (lane_id() == 0) start = atomicAdd(E.ring_count, (unsigned long long)n);
    start = rfl64(start);
    const uint8_t* st = E.stage + (size_t)g * E.SCAP * STAGE_BYTES;
    for (uint32_t r = 0; r < n; r++) {
        unsigned long long slot = start + r;
        if (slot >= E.ring_cap) { dropped += 1; continue; }
        const uint8_t* src = st + (size_t)r * STAGE_BYTES;
        uint8_t* dst = E.ring + (size_t)slot * AZR_RECORD_BYTES;
        uint32_t player = rfl((uint32_t)src[260]);
        // NNTrainDataStorage::updateValues (alphazero_nn_data.cpp:51-65)
        float z = status == ST_DRAW ? 0.0f : ((int)player == status ? 1.0f : -1.0f);
        uint32_t zb = __float_as_uint(z);
        for (uint32_t j = lane_id(); j < AZR_RECORD_BYTES; j += 64) {
            uint8_t b;
            if (j == 0) b = (uint8_t)player;
            else if (j < 89) b = src[j - 1];
            else if (j < 93) b = (uint8_t)(zb >> (8 * (j - 89)));
            else b = src[88 + (j - 93)];
            dst[j] = b;
        }
    }
}

struct StepCount {
    unsigned long long sims = 0, evals = 0, levels = 0, dec = 0, games = 0, samples = 0, drop = 0, err = 0, ringdrop = 0;
};

// Counters are kept PER GAME (one 72-byte row each, written by the game's own wave: no atomics) and summed by the host
// when somebody asks (azr_selfplay_counters).  One shared row bumped with atomics made every pass end with G x 4..9
// same-address device atomics, which the L2 retires one by one (~12 ns each): 20 us of a 47-us tree step at 512 games.
// `count_active`: host-stepped search only (azr_mcts_leaves reads the number of games that wait for the net).
// The row is read-modify-write, and the read is issued when the wave STARTS (counters_begin): at the end of a step the wave's stores
// are still draining, and a load issued behind them waits for every one of them (memory operations of a wave retire in order) —
// 3.7 us of an average mid-game wave, 9 us of the slow ones the launch waits for (profiles/r03_tree_step_profile.txt).
__device__ __forceinline__ unsigned long long counters_begin(const Dev& E, int g)
{
    const uint32_t l = lane_id();
    return l < 9 ? reinterpret_cast<const unsigned long long*>(E.counters + g)[l] : 0ull;
}
__device__ __forceinline__ void flush_counters(const Dev& E, int g, const Ctl& c, const StepCount& k, bool count_active, unsigned long long base)
{
    if (count_active && lane_id() == 0 && c.pending) atomicAdd(E.active, 1u);
    const uint32_t l = lane_id();
    unsigned long long d = 0;
    d = l == 0 ? k.sims : d; d = l == 1 ? k.evals : d; d = l == 2 ? k.levels : d; d = l == 3 ? k.dec : d; d = l == 4 ? k.games : d;
    d = l == 5 ? k.samples : d; d = l == 6 ? k.drop : d; d = l == 7 ? k.err : d; d = l == 8 ? k.ringdrop : d;
    if (l < 9 && d) {
        unsigned long long* row = reinterpret_cast<unsigned long long*>(E.counters + g);
        row[l] = base + d;
    }
}

// AlphaZeroMCTS::search leaf branch, after the future resolved (alphazero_mcts.cpp:350-356): expand + backup, for every
// search thread with a pending leaf, in thread order.  A state another thread has added meanwhile is dropped
// (StateSimulationsStorage::add, alphazero_mcts.cpp:203-215) and its value still backed up.
__device__ __forceinline__ void consume_pending(const Dev& E, int g, const Tree& t, Ctl& c, StepCount& k)
{
    if (!c.pending) return;
    const uint32_t l = lane_id();
    for (int th = 0; th < E.T; th++) {
        if (!((c.pending >> th) & 1u)) continue;
        const size_t slot = (size_t)g * E.T + th;
        float pi = E.net_pi[slot * PI_STRIDE + (l < MOVES ? l : 0)];
        float v = rdlf(E.net_v[slot], 0);
        uint64_t valid = rfl64(E.leaf_valid[slot]);
        uint32_t kd = reinterpret_cast<const uint32_t*>(E.leaf_key + slot * GREC)[l & 15u];
        uint32_t h = rfl(E.leaf_hash[slot]);
        TP(1);
        if (E.T > 1 && tree_lookup(t, kd, h) != NO_NODE) c.dup_dropped++;
        else {
            TP(2);
            float prior = normalize_prior(pi, valid);
            TP(3);
            if (tree_expand(t, c, kd, h, valid, prior) == NO_NODE) k.drop++;
        }
        TP(4);
        k.evals++;
        const uint32_t plen = plen_get(c, th);
        if (plen > 0) {  // plen == 0: this was setRootState's root expansion (not a simulation)
            tree_backup(thread_tree(t, th), plen, v, false);
            c.sims_done++;
            k.sims++;
        }
        TP(5);
    }
    c.pending = 0;
}

enum : int { RD_DONE = 0, RD_LEAF = 1, RD_FAIL = 2 };

// AlphaZeroMCTS::threadSimulateJob + search (alphazero_mcts.cpp:310-377) for search thread `th`, iteratively: claim the
// next simulation from the counter and descend from the root, repeated until the counter is exhausted (RD_DONE), a leaf
// needs the net (RD_LEAF: leaf record written to slot g * T + th, pending bit set) or a rule error (RD_FAIL).
__device__ __forceinline__ int run_descents(const Dev& E, int g, const Tree& t0, int th, Ctl& c, const WS& root, int8_t* scratch,
                                            StepCount& k, uint32_t& err_out)
{
    const Rules R = E.rules;
    const Search S = E.search;
    const Tree t = thread_tree(t0, th);
    const size_t slot = (size_t)g * E.T + th;
    while ((int)c.sims_started < S.simulations) {
        c.sims_started++;  // Counter::hasNext
        WS s = root;
        s.rng = c.rng;
        s.err = 0;
        uint32_t plen = 0;
        bool leaf = false, fail = false;
        TP(6);
        for (;;) {
            int gs = game_status(s, R);
            TP(7);
            if (gs != ST_NOT_ENDED) {
                float v = gs == ST_DRAW ? 0.0f : (gs == (int)s.cur ? 1.0f : -1.0f);
                tree_backup(t, plen, v);
                c.sims_done++;
                k.sims++;
                TP(5);
                break;
            }
            uint64_t valid = valid_moves(s, R);
            TP(8);
            if (valid == 0) { fail = true; s.err = E_INVALID_ARGUMENT; break; }
            uint32_t kd = ws_record_dword(s);
            uint32_t h = key_hash(kd);
            TP(9);
            NodeRegs nr;
            uint32_t idx = tree_lookup_node(t, kd, h, nr);
            TP(10);
            if (idx == NO_NODE) {  // leaf: hand the position to the NN service
                encode88(s, E.leaf_in + slot * LEAF_STRIDE);
                const uint32_t l = lane_id();
                if (l < 16) reinterpret_cast<uint32_t*>(E.leaf_key + slot * GREC)[l] = kd;
                if (l == 0) { E.leaf_valid[slot] = valid; E.leaf_hash[slot] = h; }
                leaf = true;
#ifdef AZR_EXP_LATE_FENCE
                wave_mem_sync();   // (experiment: one fence per descent instead of one per level)
#endif
                TP(11);
                break;
            }
            k.levels++;
            uint32_t mv = tree_select(t, idx, nr, S, c.search_id, scratch);
            TP(12);
            if (mv == NONE) { fail = true; s.err = E_LOGIC; break; }
            uint32_t before = s.cur;
#ifdef AZR_TREE_PROF
            const uint32_t ph0 = s.phase;
#endif
            make_move(s, mv, R);
#ifdef AZR_TREE_PROF
            TP(ph0 == PH_FORTIFY ? 22 : ph0 == PH_ATTACK ? 23 : 13);
#endif
            if (s.err) { fail = true; break; }
            if ((int)plen >= t.DMAX) { fail = true; s.err = AZR_E_CAPACITY; break; }
            if (lane_id() == 0) t.path[plen] = idx | (mv << 16) | ((s.cur != before ? 1u : 0u) << 24);
            plen++;
        }
        c.rng = s.rng;
        if (fail) { err_out = s.err; return RD_FAIL; }
        if (leaf) {
            if (plen == 0) c.sims_started--;  // setRootState's root expansion is not one of the S simulations
            c.pending |= 1u << th;
            plen_set(c, th, plen);
            return RD_LEAF;
        }
    }
    return RD_DONE;
}

// One round of AlphaZeroMCTS::simulate for all T search threads of the game, in thread order: every thread without a
// pending leaf runs descents until it blocks on the net.  RD_LEAF = at least one leaf is waiting; RD_DONE = the counter
// is exhausted and every claimed simulation is backed up.
__device__ __forceinline__ int search_round(const Dev& E, int g, const Tree& t, Ctl& c, const WS& root, int8_t* scratch,
                                            StepCount& k, uint32_t& err_out)
{
    for (int th = 0; th < E.T; th++) {
        if ((c.pending >> th) & 1u) continue;
        int r = run_descents(E, g, t, th, c, root, scratch, k, err_out);
        if (r == RD_FAIL) { c.pending = 0; return RD_FAIL; }
        if (r == RD_LEAF && plen_get(c, th) == 0) break;  // root expansion: the threads start after setRootState
    }
    return c.pending ? RD_LEAF : RD_DONE;
}

// N[lane] and the legal mask of the node of `root` (NO_NODE if the root is not in the tree)
__device__ __forceinline__ uint32_t root_node(const Tree& t, const WS& root, uint32_t& N, uint64_t& valid)
{
    uint32_t rkd = ws_record_dword(root);
    uint32_t ridx = tree_lookup(t, rkd, key_hash(rkd));
    N = 0; valid = 0;
    if (ridx != NO_NODE) {
        const uint8_t* n = node_ptr(t, ridx);
        const uint32_t l = lane_id();
        N = reinterpret_cast<const uint32_t*>(n + ND_N)[l < MOVES ? l : 0] & N_MASK;
        valid = (uint64_t)rfl(*reinterpret_cast<const uint32_t*>(n + ND_VALID_LO)) |
                ((uint64_t)rfl(*reinterpret_cast<const uint32_t*>(n + ND_VALID_HI)) << 32);
    }
    return ridx;
}

// the slot's next self-play game.  Unlimited mode: seeds base + g, base + G + g, ...  Quota mode (azr_selfplay_start_games,
// Counter::hasNext of alphazero_trainer.cpp:83): the next game index is a ticket from one atomic counter — exactly
// sp_quota games are started, seeds base .. base + sp_quota - 1, each game a function of its seed alone; a slot that
// draws no ticket goes idle (mode 0).
__device__ __forceinline__ void selfplay_next_game(const Dev& E, int g, const Tree& t, Ctl& c, WS& root)
{
    c.game_no++;
    c.seed = E.base_seed + c.game_no * (uint32_t)E.G + (uint32_t)g;
    if (E.sp_quota) {
        unsigned long long ticket = 0;
        if (lane_id() == 0) ticket = atomicAdd(E.sp_started, 1ull);
        ticket = rfl64(ticket);
        if (ticket >= E.sp_quota) { c.mode = 0; c.pending = 0; c.nsamples = 0; return; }
        c.seed = E.base_seed + (uint32_t)ticket;
    }
    ws_blank(root);
    root.rng = rng_seed(c.seed);
    new_game(root);
    c.rng = root.rng;
    c.nsamples = 0; c.decisions = 0; c.sims_done = 0; c.sims_started = 0; c.pending = 0;
    tree_clear(t, c);
}

// One tree step for game g: consume the pending leaf's (pi, v) [expand + backup], then run searches — and in
// self-play mode decisions, moves and game restarts — until the next leaf that needs the net.
template <bool SELFPLAY>
__global__ __launch_bounds__(64) void k_tree_step(Dev E)
{
    __shared__ int8_t scratch[128];
    const int g = blockIdx.x;
    TP_BEGIN();
    Ctl c;
    ctl_load(c, &E.ctl[g]);
    if (c.mode == 0 || (!SELFPLAY && c.search_done)) return;
    const unsigned long long cnt0 = counters_begin(E, g);
    Tree t = tree_of(E, g);
    const Rules R = E.rules;
    const Search S = E.search;
    WS root;
    ws_load(root, E.state + (size_t)g * GREC);
    StepCount k;
    bool root_dirty = false;
    TP(0);
    consume_pending(E, g, t, c, k);
    for (;;) {
        TP(6);
        if ((int)c.sims_done >= S.simulations) {
            if (!SELFPLAY) { c.search_done = 1; break; }
            // ---- one decision of the trainer's move loop (alphazero_trainer.cpp:91-112) ----
            root.rng = c.rng;
            uint32_t N; uint64_t valid;
            uint32_t mv = NONE;
            if (root_node(t, root, N, valid) != NO_NODE) {
                const uint32_t l = lane_id();
                float pi = root_policy(N, valid);
                mv = (int)root.round > S.temperature_threshold ? pick_highest(pi) : pick_random(root, pi);
                if (c.nsamples < (uint32_t)E.SCAP) {
                    uint8_t* rec = E.stage + ((size_t)g * E.SCAP + c.nsamples) * STAGE_BYTES;
                    encode88(root, rec);
                    if (l < MOVES) reinterpret_cast<float*>(rec + 88)[l] = pi;
                    if (l == 0) rec[260] = (uint8_t)root.cur;
                    c.nsamples++;
                } else k.ringdrop++;
            }
            TP(16);
            if (mv != NONE) make_move(root, mv, R); else root.err = E_LOGIC;
            c.last_move = mv;
            c.decisions++;
            k.dec++;
            c.rng = root.rng;
            int st = root.err ? ST_NOT_ENDED : game_status(root, R);
            TP(17);
            if (root.err || st != ST_NOT_ENDED) {
                if (root.err) { k.err++; c.error = root.err; }
                else {
                    wave_mem_sync();
                    flush_samples(E, g, c.nsamples, st, k.ringdrop);
                    k.samples += c.nsamples;
                    k.games++;
                }
                c.status = st;
                selfplay_next_game(E, g, t, c, root);
            }
            root_dirty = true;
            TP(18);
            if (c.mode == 0) break;  // quota exhausted: the slot idles
            tree_trim(t, c);
            c.sims_done = 0; c.sims_started = 0;
            TP(14);
        }
        uint32_t err = 0;
        int r = search_round(E, g, t, c, root, scratch, k, err);
        if (r == RD_LEAF) break;
        if (r == RD_FAIL) {
            k.err++;
            c.error = err;
            if (SELFPLAY) {  // abandon the game (the reference would have thrown): restart the slot
                selfplay_next_game(E, g, t, c, root);
                root_dirty = true;
                if (c.mode == 0) break;
                tree_trim(t, c);
                continue;
            }
            c.search_done = 1;
            break;
        }
    }
    TP(6);
    if (root_dirty) ws_store(root, E.state + (size_t)g * GREC);
    TP(19);
    ctl_store(c, &E.ctl[g]);
    TP(20);
    flush_counters(E, g, c, k, !SELFPLAY, cnt0);
    TP(21);
    // self-play tail (quota mode, slots going idle): the net of this pass runs on the waiting leaf slots only
    if (SELFPLAY && E.sp_compact && c.pending) {
        int base = 0;
        if (lane_id() == 0) base = atomicAdd(&E.leaf_count[0], (int)__builtin_popcount(c.pending));   // one atomic per game
        base = (int)rfl((uint32_t)base);
        const uint32_t l = lane_id();
        if (l < (uint32_t)E.T && ((c.pending >> l) & 1u)) E.leaf_list[base + (int)__builtin_popcount(c.pending & ((1u << l) - 1u))] = g * E.T + (int)l;
    }
    TP(15);
    TP_END();
}

// ================================================================================================
// arena: GameGroup::playGames on the device (game/game.cpp:101-312).  One slot = one player pair = one "thread" of
// the reference: games in mirrored pairs with alternating starts, AlphaZeroPlayer::takeTurn (alphazero_player.cpp:3-21)
// through the search above, ScriptPlayer / RandomPlayer as wave-resident code (azr_players.hpp).
// ================================================================================================
__global__ __launch_bounds__(64) void k_arena_step(Dev E)
{
    __shared__ int8_t scratch[128];
    const int g = blockIdx.x;
    if (E.lc_zero >= 0 && g == 0 && threadIdx.x < 2) E.leaf_count[E.lc_zero + threadIdx.x] = 0;   // the next pass's counts (the last readers are done)
    Ctl c;
    ctl_load(c, &E.ctl[g]);
    if (c.mode != 3 || c.arena_state == 2) return;
    const unsigned long long cnt0 = counters_begin(E, g);
    Tree t = tree_of(E, g);
    const Rules R = E.rules;
    const Search S = E.search;
    WS root;
    ws_load(root, E.state + (size_t)g * GREC);
    root.rng = c.rng;
    ScriptW sp[2];
    {
        const ScriptW* src = reinterpret_cast<const ScriptW*>(E.script) + (size_t)g * 2;
        sp[0] = src[0]; sp[1] = src[1];
        sp[0].order = rfl(sp[0].order); sp[0].attacking_set = rfl(sp[0].attacking_set); sp[0].land_to = rfl(sp[0].land_to);
        sp[0].land_from = rfl(sp[0].land_from); sp[0].attack_from_army = rfl(sp[0].attack_from_army);
        sp[1].order = rfl(sp[1].order); sp[1].attacking_set = rfl(sp[1].attacking_set); sp[1].land_to = rfl(sp[1].land_to);
        sp[1].land_from = rfl(sp[1].land_from); sp[1].attack_from_army = rfl(sp[1].attack_from_army);
    }
    StepCount k;
    // two-net arena: the opponent AlphaZero player's own tree and its allocator state
    const bool two = E.nodes2 != nullptr;
    const Tree t2 = two ? tree2_of(E, g) : t;
    TreeCtl x2 = {0, 0, 0};
    if (two) {
        const uint32_t w = E.tctl2[(size_t)g * 4 + (lane_id() & 3u)];
        x2.search_id = rdl(w, 0); x2.nfree = rdl(w, 1); x2.hiwater = rdl(w, 2);
    }
    if (c.search_tree) { swap_tree_ctl(c, x2); consume_pending(E, g, t2, c, k); swap_tree_ctl(c, x2); }
    else consume_pending(E, g, t, c, k);
    for (;;) {
        if (c.arena_state == 0 && E.arena_mirror == AZR_MIRROR_CONCURRENT) {
            // Both games of a mirrored pair at the same time: slot 2j plays half 0 of slot pair j's k-th pair, slot 2j + 1 the
            // mirrored half (include/azr.h).  Nothing in Game orders the two games (game.cpp:238-254); what they share is the
            // deal (game.cpp:170-191), and the deal is a function of the pair's seed, so each half deals for itself.
            const int L = E.G >> 1, lane = g >> 1;
            const uint32_t half = (uint32_t)g & 1u;
            const long long pr = (long long)lane + (long long)c.slot_games * L;   // pairs are assigned statically
            const bool capped = E.arena_slot_cap > 0 && (int)c.slot_games >= E.arena_slot_cap;
            if (capped || g >= 2 * L || pr >= (long long)(E.arena_total / 2)) { c.arena_state = 2; break; }
            const uint32_t pseed = E.base_seed + (uint32_t)pr;
            root.rng = rng_seed(pseed);
            new_game(root);
            if (half) {   // Game::newGame's mirrored branch: invertPlayers of the pair's deal, player 1 starts, own dice stream
                invert_players(root);
                root.rng = rng_seed(pseed + (1u << 30));
            }
            root.cur = half;
            c.player_start = half;
            c.seed = pseed;
            tree_clear(t, c);  // AlphaZeroPlayer::newGame
            if (two) { swap_tree_ctl(c, x2); tree_clear(t2, c); swap_tree_ctl(c, x2); }
            c.nsamples = 0;
            c.sims_done = 0; c.sims_started = 0; c.search_active = 0; c.turn_started = 0; c.pending = 0;
            c.arena_state = 1;
        }
        if (c.arena_state == 0) {  // Game::newGame (game.cpp:170-191) for the next Game::playGames(1)
            if (c.pair_phase == 0) {  // Counter::hasNext(2) (game.cpp:14-26)
                int taken = 0;
                const bool capped = E.arena_slot_cap > 0 && (int)c.slot_games >= E.arena_slot_cap;
                if (!capped && lane_id() == 0) taken = atomicAdd(E.arena_taken, 2);
                taken = (int)rfl((uint32_t)taken);
                if (capped || taken + 2 > E.arena_total) { c.arena_state = 2; break; }
            }
            if (E.arena_mirror && c.player_start != 0) {
                uint32_t keep = root.rng;
                ws_load(root, E.prev_start + (size_t)g * GREC);
                root.rng = keep;
                invert_players(root);
                root.cur = c.player_start;
            } else {
                new_game(root);
                root.cur = c.player_start;
                ws_store(root, E.prev_start + (size_t)g * GREC);
            }
            tree_clear(t, c);  // AlphaZeroPlayer::newGame
            if (two) { swap_tree_ctl(c, x2); tree_clear(t2, c); swap_tree_ctl(c, x2); }
            c.nsamples = 0;
            c.sims_done = 0; c.sims_started = 0; c.search_active = 0; c.turn_started = 0; c.pending = 0;
            c.arena_state = 1;
        }
        // ---- Game::playTurn (game.cpp:112-133)
        int gs = game_status(root, R);
        if (gs != ST_NOT_ENDED) {  // GameResults::addGame (game.cpp:193-213)
            if (lane_id() == 0) {
                atomicAdd(&E.arena_res[0], 1);
                if (gs == ST_DRAW) atomicAdd(&E.arena_res[1], 1);
                if (gs == 0 || gs == 1) {
                    atomicAdd(&E.arena_res[2 + 2 * gs], 1);
                    if ((int)c.player_start == gs) atomicAdd(&E.arena_res[3 + 2 * gs], 1);
                }
                if (c.slot_games < (uint32_t)ALOG) {
                    E.alog_status[(size_t)g * ALOG + c.slot_games] = (int8_t)gs;
                    E.alog_rounds[(size_t)g * ALOG + c.slot_games] = (uint16_t)root.round;
                }
            }
            if (c.slot_games < (uint32_t)ALOG) ws_store(root, E.alog_final + ((size_t)g * ALOG + c.slot_games) * GREC);
            if (E.arena_collect && c.nsamples) {  // Player::gameFinished -> NNTrainDataStorage::updateValues for both players
                wave_mem_sync();
                flush_samples(E, g, c.nsamples, gs, k.ringdrop);
                k.samples += c.nsamples;
                c.nsamples = 0;
            }
            c.slot_games++;
            k.games++;
            c.player_start ^= 1u;  // Game::incPlayerStart (the concurrent form sets it per game)
            c.pair_phase ^= 1u;
            c.arena_state = 0;
            continue;
        }
        const uint32_t p = root.cur;
        const int kind = p == 0 ? E.kind0 : E.kind1;
        bool fail = false;
        if (kind == 1) {
            if (p == 0) script_take_turn(sp[0], root, R); else script_take_turn(sp[1], root, R);
            fail = root.err != 0 || (root.cur == p && game_status(root, R) == ST_NOT_ENDED);  // "Turn was not incremented"
        } else if (kind == 2) {
            random_take_turn(root, R);
            fail = root.err != 0 || (root.cur == p && game_status(root, R) == ST_NOT_ENDED);
        } else {
            const uint32_t w = kind == 3 ? 1u : 0u;   // which AlphaZeroPlayer: its tree and its network
            const Tree& tt = w ? t2 : t;
            if (w) swap_tree_ctl(c, x2);
            // the player's own trimNodes at the start of a turn, setRootState's at the start of every search: at a turn's first decision
            // both run back to back, which leaves an empty tree (tree_trim_twice)
            if (!c.turn_started && !c.search_active) tree_trim_twice(tt, c);
            else if (!c.turn_started || !c.search_active) tree_trim(tt, c);
            c.turn_started = 1;
            if (!c.search_active) { c.sims_done = 0; c.sims_started = 0; c.search_active = 1; }
            c.search_tree = w;
            c.rng = root.rng;
            uint32_t err = 0;
            int r = search_round(E, g, tt, c, root, scratch, k, err);
            root.rng = c.rng;
            if (r == RD_LEAF) { if (w) swap_tree_ctl(c, x2); break; }
            if (r == RD_FAIL) { fail = true; root.err = err; }
            else {
                uint32_t N; uint64_t valid;
                uint32_t mv = NONE;
                if (root_node(tt, root, N, valid) != NO_NODE) {
                    const float pi = root_policy(N, valid);
                    mv = pick_highest(pi);
                    if (E.arena_collect) {  // AlphaZeroPlayer::takeTurn with trainStorage set (alphazero_player.cpp:15-18)
                        if (c.nsamples < (uint32_t)E.SCAP) {
                            uint8_t* rec = E.stage + ((size_t)g * E.SCAP + c.nsamples) * STAGE_BYTES;
                            encode88(root, rec);
                            if (lane_id() < MOVES) reinterpret_cast<float*>(rec + 88)[lane_id()] = pi;
                            if (lane_id() == 0) rec[260] = (uint8_t)root.cur;
                            c.nsamples++;
                        } else k.ringdrop++;
                    }
                }
                if (mv != NONE) make_move(root, mv, R); else root.err = E_LOGIC;
                c.search_active = 0;
                c.last_move = mv;
                k.dec++;
                fail = root.err != 0;
                if (root.cur != p || game_status(root, R) != ST_NOT_ENDED) c.turn_started = 0;
            }
            if (w) swap_tree_ctl(c, x2);
        }
        if (fail) {  // the reference would have thrown out of GameGroup: drop the game, start a fresh pair
            k.err++;
            c.error = root.err ? root.err : (uint32_t)E_LOGIC;
            root.err = 0;
            c.player_start = 0; c.pair_phase = 0; c.arena_state = 0;
            c.pending = 0; c.search_active = 0; c.turn_started = 0;
            if (E.arena_mirror == AZR_MIRROR_CONCURRENT) c.slot_games++;   // statically assigned: go on with the slot's next pair
        }
    }
    if (two && lane_id() == 0) { uint32_t* d2 = E.tctl2 + (size_t)g * 4; d2[0] = x2.search_id; d2[1] = x2.nfree; d2[2] = x2.hiwater; }
    // hand every waiting leaf to the network of the player that is searching (list 0 when there is one network): a pass
    // evaluates the listed slots only — most slots of an arena idle (scripted players' turns, finished quotas)
    if (c.pending) {
        const uint32_t tree = two ? c.search_tree : 0u;
        int base = 0;
        if (lane_id() == 0) base = atomicAdd(&E.leaf_count[E.lc_base + tree], (int)__builtin_popcount(c.pending));   // one atomic per slot
        base = (int)rfl((uint32_t)base);
        const uint32_t l = lane_id();
        if (l < (uint32_t)E.T && ((c.pending >> l) & 1u))
            E.leaf_list[(size_t)tree * E.G * E.T + base + (int)__builtin_popcount(c.pending & ((1u << l) - 1u))] = g * E.T + (int)l;
    }
    c.rng = root.rng;
    ws_store(root, E.state + (size_t)g * GREC);
    {
        ScriptW* dst = reinterpret_cast<ScriptW*>(E.script) + (size_t)g * 2;
        if (lane_id() == 0) { dst[0] = sp[0]; dst[1] = sp[1]; }
    }
    ctl_store(c, &E.ctl[g]);
    flush_counters(E, g, c, k, false, cnt0);
}

__global__ __launch_bounds__(64) void k_arena_start(Dev E)
{
    const int g = blockIdx.x;
    Ctl c;
    ctl_load(c, &E.ctl[g]);
    Tree t = tree_of(E, g);
    c.hiwater = (uint32_t)E.C;
    tree_clear(t, c);
    if (E.nodes2) {  // the opponent player's tree starts empty with its own stamp counter
        Ctl c2 = c;
        c2.hiwater = (uint32_t)E.C;
        tree_clear(tree2_of(E, g), c2);
        if (lane_id() == 0) { uint32_t* d2 = E.tctl2 + (size_t)g * 4; d2[0] = c2.search_id; d2[1] = c2.nfree; d2[2] = c2.hiwater; d2[3] = 0; }
    }
    c.search_tree = 0; c.nsamples = 0;
    c.mode = 3; c.sims_done = 0; c.sims_started = 0; c.pending = 0; c.search_done = 0; c.error = 0;
    c.arena_state = 0; c.player_start = 0; c.pair_phase = 0; c.turn_started = 0; c.search_active = 0; c.slot_games = 0;
    c.seed = E.base_seed + (uint32_t)g;
    c.rng = rng_seed(c.seed);
    ScriptW* dst = reinterpret_cast<ScriptW*>(E.script) + (size_t)g * 2;
    if (lane_id() == 0) { ScriptW w; script_init(w); dst[0] = w; dst[1] = w; }
    ctl_store(c, &E.ctl[g]);
}

// root statistics / policy / pick for host-stepped use
__global__ __launch_bounds__(64) void k_root_stats(Dev E, uint32_t* n_out, float* q_out, float* p_out, float* pi_out)
{
    const int g = blockIdx.x;
    Ctl c;
    ctl_load(c, &E.ctl[g]);
    Tree t = tree_of(E, g);
    WS root;
    ws_load(root, E.state + (size_t)g * GREC);
    uint32_t kd = ws_record_dword(root);
    uint32_t idx = tree_lookup(t, kd, key_hash(kd));
    const uint32_t l = lane_id();
    uint32_t N = 0; float Q = 0, P = 0, pi = 0;
    if (idx != NO_NODE) {
        const uint8_t* n = node_ptr(t, idx);
        N = reinterpret_cast<const uint32_t*>(n + ND_N)[l < MOVES ? l : 0] & N_MASK;
        Q = reinterpret_cast<const float*>(n + ND_Q)[l < MOVES ? l : 0];
        P = reinterpret_cast<const float*>(n + ND_P)[l < MOVES ? l : 0];
        uint64_t valid = (uint64_t)rfl(*reinterpret_cast<const uint32_t*>(n + ND_VALID_LO)) |
                         ((uint64_t)rfl(*reinterpret_cast<const uint32_t*>(n + ND_VALID_HI)) << 32);
        pi = root_policy(N, valid);
    }
    if (l < MOVES) {
        if (n_out) n_out[(size_t)g * MOVES + l] = N;
        if (q_out) q_out[(size_t)g * MOVES + l] = Q;
        if (p_out) p_out[(size_t)g * MOVES + l] = P;
        if (pi_out) pi_out[(size_t)g * MOVES + l] = pi;
    }
}

__global__ __launch_bounds__(64) void k_pick(Dev E, int sample, uint8_t* moves)
{
    const int g = blockIdx.x;
    Ctl c;
    ctl_load(c, &E.ctl[g]);
    Tree t = tree_of(E, g);
    WS root;
    ws_load(root, E.state + (size_t)g * GREC);
    root.rng = c.rng;
    uint32_t kd = ws_record_dword(root);
    uint32_t idx = tree_lookup(t, kd, key_hash(kd));
    uint32_t mv = NONE;
    if (idx != NO_NODE) {
        const uint8_t* n = node_ptr(t, idx);
        const uint32_t l = lane_id();
        uint32_t N = reinterpret_cast<const uint32_t*>(n + ND_N)[l < MOVES ? l : 0] & N_MASK;
        uint64_t valid = (uint64_t)rfl(*reinterpret_cast<const uint32_t*>(n + ND_VALID_LO)) |
                         ((uint64_t)rfl(*reinterpret_cast<const uint32_t*>(n + ND_VALID_HI)) << 32);
        float pi = root_policy(N, valid);
        mv = sample ? pick_random(root, pi) : pick_highest(pi);
    }
    if (lane_id() == 0) {
        moves[g] = (uint8_t)mv;
        E.ctl[g].rng = root.rng;
    }
}

// keep != 0: the games go on from the states and RNG streams the caller has set (azr_selfplay_start_from_states)
__global__ __launch_bounds__(64) void k_selfplay_start(Dev E, int keep)
{
    const int g = blockIdx.x;
    Ctl c;
    ctl_load(c, &E.ctl[g]);
    Tree t = tree_of(E, g);
    c.hiwater = (uint32_t)E.C;
    tree_clear(t, c);
    tree_trim(t, c);
    c.mode = 2; c.sims_done = 0; c.sims_started = 0; c.pending = 0; c.search_done = 0; c.error = 0;
    c.game_no = 0; c.nsamples = 0; c.decisions = 0; c.status = ST_NOT_ENDED;
    c.seed = E.base_seed + (uint32_t)g;
    if (E.sp_quota && (unsigned long long)g >= E.sp_quota) c.mode = 0;  // fewer games asked for than slots
    if (!keep) {
        WS s;
        ws_blank(s);
        s.rng = rng_seed(c.seed);
        new_game(s);
        c.rng = s.rng;
        ws_store(s, E.state + (size_t)g * GREC);
    }
    ctl_store(c, &E.ctl[g]);
}

// ================================================================================================
// host side
// ================================================================================================
static int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

extern "C" void azr_default_settings(azr_settings* s)
{
    memset(s, 0, sizeof *s);
    s->device = 0;
    s->games = 32;
    s->blocks = 20;
    s->net_dtype = AZR_NET_BF16;
    s->mcts_simulations = 32;
    s->mcts_threads = 2;
    s->allow_yield = 1;
    s->limit_reinforcement = 1;
    s->limit_attack = 0;
    s->max_game_rounds = 30 + 28;
    s->min_unit_move = 3;
    s->temperature_threshold = 15 + 28;
    s->hp_exploration = 1.1f;
    s->dir_noise_value = 0.3f;
    s->dir_noise_epsi = 0.25f;
    s->node_capacity = 0;
    s->sample_capacity = 0;
}

template <typename T>
static hipError_t dmalloc(T** p, size_t n) { return hipMalloc((void**)p, n * sizeof(T)); }

static thread_local std::string g_create_err;   // why the last azr_engine_create of this thread failed
static int engine_init(azr_engine* h, const azr_settings* s);

extern "C" int azr_engine_create(const azr_settings* s, azr_engine** out)
{
    if (!out) { g_create_err = "azr_engine_create: out is NULL"; return AZR_E_INVALID_ARGUMENT; }
    *out = nullptr;   // first: every failure below leaves NULL behind and its reason in g_create_err
    auto bad = [&](const std::string& why) { g_create_err = "azr_engine_create: " + why; return (int)AZR_E_INVALID_ARGUMENT; };
    if (!s) return bad("settings is NULL");
    if (s->games <= 0) return bad("games = " + std::to_string(s->games) + " (need > 0)");
    if (s->blocks <= 0) return bad("blocks = " + std::to_string(s->blocks) + " (need > 0)");
    if (s->net_dtype < AZR_NET_F32 || s->net_dtype > AZR_NET_F16) return bad("net_dtype = " + std::to_string(s->net_dtype) + " (AZR_NET_F32 .. AZR_NET_F16)");
    if (s->mcts_simulations < 0) return bad("mcts_simulations = " + std::to_string(s->mcts_simulations) + " (need >= 0)");
    if (s->mcts_threads < 1 || s->mcts_threads > MAX_THREADS)
        return bad("mcts_threads = " + std::to_string(s->mcts_threads) + " (need 1.." + std::to_string(MAX_THREADS) + ")");
    if (s->mcts_simulations > 0 && s->mcts_simulations < s->mcts_threads)   // count = S - S % T would be 0
        return bad("mcts_simulations = " + std::to_string(s->mcts_simulations) + " < mcts_threads = " + std::to_string(s->mcts_threads) +
                   " (S - S % T simulations would be none)");
    // node indices are 16-bit (azr_tree.hpp): a pool above 65 534 nodes per game cannot be addressed.  Rejected, not
    // clamped: a silently smaller pool would change which expansions are dropped.
    const long long want_nodes = s->node_capacity > 0 ? (long long)s->node_capacity : 16ll * (s->mcts_simulations + 1);
    if (want_nodes > 65534) {
        g_create_err = "azr_engine_create: node pool of " + std::to_string(want_nodes) + " nodes per game (node_capacity, or the default "
                       "16 * (mcts_simulations + 1)) exceeds the 65534 a 16-bit node index addresses; pass node_capacity <= 65534";
        return AZR_E_INVALID_ARGUMENT;
    }
    azr_engine* h = new (std::nothrow) azr_engine();
    if (!h) { g_create_err = "azr_engine_create: out of host memory"; return AZR_E_HIP; }
    const int rc_create = engine_init(h, s);
    if (rc_create) {  // nothing half-built is handed out: free what was allocated, keep the message for azr_last_error(NULL)
        g_create_err = h->err;
        azr_engine_destroy(h);
        return rc_create;
    }
    *out = h;
    return AZR_OK;
}

static int engine_init(azr_engine* h, const azr_settings* s)
{
    h->cfg = *s;
    h->mode = 0;
    h->weights_set = false;
    h->prof_net_ms = h->prof_tree_ms = h->prof_tower_ms = 0;
    h->prof_launches = 0;
    h->stream = nullptr;
    Dev& d = h->d;
    memset(&d, 0, sizeof d);
    memset(&h->net, 0, sizeof h->net);
    HIPCHK(h, hipSetDevice(s->device));
    HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    d.G = s->games;
    d.T = s->mcts_threads;
    int C = s->node_capacity > 0 ? s->node_capacity : 16 * (s->mcts_simulations + 1);
    if (C < 64) C = 64;
    d.C = C;
    d.H = next_pow2(2 * C);
    d.DMAX = std::min(C, 1024);
    d.SCAP = s->sample_capacity > 0 ? s->sample_capacity : 4096;
    d.rules = Rules{s->allow_yield, s->limit_reinforcement, s->limit_attack, s->max_game_rounds, s->min_unit_move};
    // count = MCTS_SIMULATIONS - MCTS_SIMULATIONS % THREADS_PER_MCTS (alphazero_mcts.cpp:265)
    d.search.simulations = s->mcts_simulations - s->mcts_simulations % s->mcts_threads;
    d.search.c1 = 1 - s->dir_noise_epsi;
    d.search.c2 = s->dir_noise_epsi * s->dir_noise_value;
    d.search.hp = s->hp_exploration;
    d.search.temperature_threshold = s->temperature_threshold;
    const size_t G = d.G, GT = G * d.T;
    HIPCHK(h, dmalloc(&d.state, G * GREC));
    HIPCHK(h, dmalloc(&d.ctl, G));
    HIPCHK(h, dmalloc(&d.nodes, G * C * NODE_BYTES));
    HIPCHK(h, dmalloc(&d.touch, G * C));
    HIPCHK(h, dmalloc(&d.nhash, G * C));
    HIPCHK(h, dmalloc(&d.table, G * d.H));
    HIPCHK(h, dmalloc(&d.freel, G * C));
    HIPCHK(h, dmalloc(&d.path, GT * d.DMAX));
    HIPCHK(h, dmalloc(&d.leaf_in, GT * LEAF_STRIDE));
    HIPCHK(h, dmalloc(&d.leaf_key, GT * GREC));
    HIPCHK(h, dmalloc(&d.leaf_valid, GT));
    HIPCHK(h, dmalloc(&d.leaf_hash, GT));
    HIPCHK(h, dmalloc(&d.net_pi, GT * PI_STRIDE));
    HIPCHK(h, dmalloc(&d.net_v, GT));
    HIPCHK(h, dmalloc(&d.stage, G * d.SCAP * STAGE_BYTES));
    d.ring_cap = (unsigned long long)G * d.SCAP;
    HIPCHK(h, dmalloc(&d.ring, (size_t)d.ring_cap * AZR_RECORD_BYTES));
    HIPCHK(h, dmalloc(&d.ring_count, 1));
    HIPCHK(h, dmalloc(&d.counters, G));   // one row per game
    HIPCHK(h, dmalloc(&d.active, 1));
    HIPCHK(h, dmalloc(&d.arena_taken, 1));
    HIPCHK(h, dmalloc(&d.sp_started, 1));
    HIPCHK(h, dmalloc(&d.leaf_list, 2 * GT));      // leaf slots waiting for net A / net B (two-net arena; self-play tail uses [0])
    HIPCHK(h, dmalloc(&d.leaf_count, (size_t)4));
    d.lc_base = 0; d.lc_zero = -1;
    HIPCHK(h, dmalloc(&d.arena_res, 8));
    HIPCHK(h, dmalloc(&d.prev_start, G * GREC));
    HIPCHK(h, dmalloc(&d.script, G * 2 * 32));
    HIPCHK(h, dmalloc(&d.alog_status, G * ALOG));
    HIPCHK(h, dmalloc(&d.alog_rounds, G * ALOG));
    HIPCHK(h, dmalloc(&d.alog_final, G * ALOG * GREC));
    HIPCHK(h, hipMemsetAsync(d.state, 0, G * GREC, h->stream));
    HIPCHK(h, hipMemsetAsync(d.ctl, 0, G * sizeof(Ctl), h->stream));
    HIPCHK(h, hipMemsetAsync(d.touch, 0, G * C * sizeof(uint32_t), h->stream));
    HIPCHK(h, hipMemsetAsync(d.table, 0, G * d.H * sizeof(uint32_t), h->stream));
    HIPCHK(h, hipMemsetAsync(d.leaf_in, 0, GT * LEAF_STRIDE, h->stream));
    HIPCHK(h, hipMemsetAsync(d.net_pi, 0, GT * PI_STRIDE * sizeof(float), h->stream));
    HIPCHK(h, hipMemsetAsync(d.net_v, 0, GT * sizeof(float), h->stream));
    HIPCHK(h, hipMemsetAsync(d.ring_count, 0, sizeof(unsigned long long), h->stream));
    HIPCHK(h, hipMemsetAsync(d.counters, 0, (size_t)d.G * sizeof(Counters), h->stream));
    HIPCHK(h, hipMemsetAsync(d.active, 0, sizeof(uint32_t), h->stream));
    int rc = net_alloc(h);
    if (rc) return rc;
    hipLaunchKernelGGL(k_tree_clear, dim3(d.G), dim3(64), 0, h->stream, d);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return AZR_OK;
}

extern "C" int azr_engine_destroy(azr_engine* h)
{
    if (!h) return AZR_E_BAD_HANDLE;
    hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    Dev& d = h->d;
    void* ptrs[] = {d.sp_started, d.state, d.ctl, d.nodes, d.touch, d.nhash, d.table, d.freel, d.path, d.leaf_in, d.leaf_key,
                    d.leaf_valid, d.leaf_hash, d.net_pi, d.net_v, d.stage, d.ring, d.ring_count, d.counters, d.active,
                    d.arena_taken, d.arena_res, d.prev_start, d.script, d.alog_status, d.alog_rounds, d.alog_final};
    for (void* p : ptrs) if (p) hipFree(p);
    for (void* p : h->tree2) if (p) hipFree(p);
    if (d.leaf_list) hipFree(d.leaf_list);
    if (d.leaf_count) hipFree(d.leaf_count);
    dp_free(h);
    train_free(h);
    net_free(h);
    for (hipEvent_t e : h->ev) hipEventDestroy(e);
#ifdef AZR_TREE_PROF
    {
        unsigned long long t[25] = {0}, u[25] = {0}, hh[16] = {0};
        (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(azr::g_tprof), sizeof t);
        (void)hipMemcpyFromSymbol(u, HIP_SYMBOL(azr::g_tslow), sizeof u);
        (void)hipMemcpyFromSymbol(hh, HIP_SYMBOL(azr::g_thist), sizeof hh);
        static const char* nm[24] = {"prologue", "consume: read leaf", "consume: dup lookup", "consume: normalize", "consume: expand", "backup",
                                     "loop/copy root", "game_status", "valid_moves", "record+hash", "lookup", "leaf write", "select", "make_move", "decision: trim", "epilogue: leaf list",
                                     "decision: policy+pick+stage", "decision: make_move+status", "decision: flush+next game", "epilogue: ws_store", "epilogue: ctl_store", "epilogue: counters", "make_move: fortify", "make_move: attack"};
        if (t[24]) {
            fprintf(stderr, "tree step profile: %llu waves, us per wave:", t[24]); double sum = 0;
            for (int i = 0; i < 24; i++) { fprintf(stderr, " %s %.2f |", nm[i], t[i] * 0.01 / t[24]); sum += t[i] * 0.01 / t[24]; } fprintf(stderr, " total %.2f\n", sum);
            if (u[24]) { fprintf(stderr, "  waves over 60 us: %llu, us per wave:", u[24]); for (int i = 0; i < 24; i++) fprintf(stderr, " %s %.2f |", nm[i], u[i] * 0.01 / u[24]); fprintf(stderr, "\n"); }
            fprintf(stderr, "  waves by total time, 10-us bins:"); for (int i = 0; i < 16; i++) fprintf(stderr, " %llu", hh[i]); fprintf(stderr, "\n");
            unsigned long long z[25] = {0};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(azr::g_tprof), z, sizeof z); (void)hipMemcpyToSymbol(HIP_SYMBOL(azr::g_tslow), z, sizeof z); (void)hipMemcpyToSymbol(HIP_SYMBOL(azr::g_thist), z, sizeof(unsigned long long) * 16);
        }
    }
#endif
    if (h->arena_ev) hipEventDestroy(h->arena_ev);
    if (h->arena_ev2) hipEventDestroy(h->arena_ev2);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return AZR_OK;
}

extern "C" const char* azr_last_error(const azr_engine* h)
{
    if (h) return h->err.c_str();
    return g_create_err.empty() ? "bad handle" : g_create_err.c_str();   // NULL: the calling thread's last failed create
}
extern "C" int azr_engine_games(const azr_engine* h) { return h ? h->d.G : 0; }

// staging helpers: synchronous copies through temporary device buffers (boundary calls are not the hot path)
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
};
#define H2D(h, dst, src, n) HIPCHK(h, hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, (h)->stream))
#define D2H(h, dst, src, n) HIPCHK(h, hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, (h)->stream))
#define SYNC(h) HIPCHK(h, hipStreamSynchronize((h)->stream))
#define LAUNCH(h, kern, ...)                                                          \
    do {                                                                              \
        hipLaunchKernelGGL(kern, dim3((h)->d.G), dim3(64), 0, (h)->stream, __VA_ARGS__); \
        HIPCHK(h, hipGetLastError());                                                 \
    } while (0)
#define ENTER(h)                                 \
    if (!(h)) return AZR_E_BAD_HANDLE;           \
    HIPCHK(h, hipSetDevice((h)->cfg.device))

extern "C" int azr_engine_new_games(azr_engine* h, const uint32_t* seeds)
{
    ENTER(h);
    if (!seeds) return AZR_E_INVALID_ARGUMENT;
    DevBuf b;
    HIPCHK(h, b.alloc(h->d.G * 4));
    H2D(h, b.p, seeds, (size_t)h->d.G * 4);
    LAUNCH(h, k_new_games, h->d, (const uint32_t*)b.p);
    LAUNCH(h, k_tree_clear, h->d);
    SYNC(h);
    return AZR_OK;
}

extern "C" int azr_engine_set_states(azr_engine* h, const void* data160)
{
    ENTER(h);
    if (!data160) return AZR_E_INVALID_ARGUMENT;
    DevBuf b;
    HIPCHK(h, b.alloc((size_t)h->d.G * 160));
    H2D(h, b.p, data160, (size_t)h->d.G * 160);
    LAUNCH(h, k_import160, h->d, (const uint8_t*)b.p);
    SYNC(h);
    return AZR_OK;
}

extern "C" int azr_engine_get_states(azr_engine* h, void* data160)
{
    ENTER(h);
    if (!data160) return AZR_E_INVALID_ARGUMENT;
    DevBuf b;
    HIPCHK(h, b.alloc((size_t)h->d.G * 160));
    LAUNCH(h, k_export160, h->d, (const uint8_t*)h->d.state, (uint8_t*)b.p);
    D2H(h, data160, b.p, (size_t)h->d.G * 160);
    SYNC(h);
    return AZR_OK;
}

extern "C" int azr_engine_set_rng(azr_engine* h, const uint32_t* st)
{
    ENTER(h);
    if (!st) return AZR_E_INVALID_ARGUMENT;
    HIPCHK(h, hipMemcpy2DAsync(&h->d.ctl[0].rng, sizeof(Ctl), st, 4, 4, h->d.G, hipMemcpyHostToDevice, h->stream));
    SYNC(h);
    return AZR_OK;
}

extern "C" int azr_engine_get_rng(azr_engine* h, uint32_t* st)
{
    ENTER(h);
    if (!st) return AZR_E_INVALID_ARGUMENT;
    HIPCHK(h, hipMemcpy2DAsync(st, 4, &h->d.ctl[0].rng, sizeof(Ctl), 4, h->d.G, hipMemcpyDeviceToHost, h->stream));
    SYNC(h);
    return AZR_OK;
}

extern "C" int azr_engine_valid_moves(azr_engine* h, uint64_t* masks)
{
    ENTER(h);
    if (!masks) return AZR_E_INVALID_ARGUMENT;
    DevBuf b;
    HIPCHK(h, b.alloc((size_t)h->d.G * 8));
    LAUNCH(h, k_valid_moves, h->d, (uint64_t*)b.p);
    D2H(h, masks, b.p, (size_t)h->d.G * 8);
    SYNC(h);
    return AZR_OK;
}

extern "C" int azr_engine_make_moves(azr_engine* h, const uint8_t* moves, uint8_t* rc)
{
    ENTER(h);
    if (!moves) return AZR_E_INVALID_ARGUMENT;
    DevBuf b, r;
    HIPCHK(h, b.alloc(h->d.G));
    HIPCHK(h, r.alloc(h->d.G));
    H2D(h, b.p, moves, (size_t)h->d.G);
    LAUNCH(h, k_make_moves, h->d, (const uint8_t*)b.p, (uint8_t*)r.p);
    std::vector<uint8_t> tmp(h->d.G);
    D2H(h, tmp.data(), r.p, (size_t)h->d.G);
    SYNC(h);
    int worst = AZR_OK;
    for (int g = 0; g < h->d.G; g++) {
        if (rc) rc[g] = tmp[g];
        if (tmp[g] && !worst) worst = tmp[g];
    }
    return rc ? AZR_OK : worst;
}

extern "C" int azr_engine_status(azr_engine* h, int8_t* status)
{
    ENTER(h);
    if (!status) return AZR_E_INVALID_ARGUMENT;
    DevBuf b;
    HIPCHK(h, b.alloc(h->d.G));
    LAUNCH(h, k_status, h->d, (int8_t*)b.p);
    D2H(h, status, b.p, (size_t)h->d.G);
    SYNC(h);
    return AZR_OK;
}

extern "C" int azr_engine_encode(azr_engine* h, void* in88)
{
    ENTER(h);
    if (!in88) return AZR_E_INVALID_ARGUMENT;
    DevBuf b;
    HIPCHK(h, b.alloc((size_t)h->d.G * 88));
    HIPCHK(h, hipMemsetAsync(b.p, 0, (size_t)h->d.G * 88, h->stream));
    LAUNCH(h, k_encode, h->d, (uint8_t*)b.p);
    D2H(h, in88, b.p, (size_t)h->d.G * 88);
    SYNC(h);
    return AZR_OK;
}

// ---- search ---------------------------------------------------------------------------------------
extern "C" int azr_mcts_clear(azr_engine* h)
{
    ENTER(h);
    LAUNCH(h, k_tree_clear, h->d);
    SYNC(h);
    return AZR_OK;
}

extern "C" int azr_mcts_trim(azr_engine* h)
{
    ENTER(h);
    LAUNCH(h, k_tree_trim, h->d);
    SYNC(h);
    return AZR_OK;
}

extern "C" int azr_mcts_begin(azr_engine* h)
{
    ENTER(h);
    h->mode = 1;
    LAUNCH(h, k_search_begin, h->d);
    SYNC(h);
    return AZR_OK;
}

static int tree_step_host(azr_engine* h, uint32_t* active)
{
    HIPCHK(h, hipMemsetAsync(h->d.active, 0, 4, h->stream));
    LAUNCH(h, k_tree_step<false>, h->d);
    D2H(h, active, h->d.active, 4);
    SYNC(h);
    return AZR_OK;
}

extern "C" int azr_mcts_leaves(azr_engine* h, void* in88, uint8_t* need, int* active_out)
{
    ENTER(h);
    uint32_t active = 0;
    int rc = tree_step_host(h, &active);
    if (rc) return rc;
    const int G = h->d.G, T = h->d.T;
    if (in88) HIPCHK(h, hipMemcpy2DAsync(in88, 88, h->d.leaf_in, LEAF_STRIDE, 88, (size_t)G * T, hipMemcpyDeviceToHost, h->stream));
    if (need) {
        std::vector<uint32_t> p(G);
        HIPCHK(h, hipMemcpy2DAsync(p.data(), 4, &h->d.ctl[0].pending, sizeof(Ctl), 4, G, hipMemcpyDeviceToHost, h->stream));
        SYNC(h);
        for (int g = 0; g < G; g++)
            for (int k = 0; k < T; k++) need[g * T + k] = (uint8_t)((p[g] >> k) & 1u);
    }
    SYNC(h);
    if (active_out) *active_out = (int)active;
    return AZR_OK;
}

extern "C" int azr_mcts_apply(azr_engine* h, const float* pi, const float* v)
{
    ENTER(h);
    if (!pi || !v) return AZR_E_INVALID_ARGUMENT;
    const size_t GT = (size_t)h->d.G * h->d.T;
    HIPCHK(h, hipMemcpy2DAsync(h->d.net_pi, PI_STRIDE * 4, pi, MOVES * 4, MOVES * 4, GT, hipMemcpyHostToDevice, h->stream));
    H2D(h, h->d.net_v, v, GT * 4);
    SYNC(h);
    return AZR_OK;
}

extern "C" int azr_mcts_simulate(azr_engine* h)
{
    ENTER(h);
    if (!h->weights_set) { h->err = "azr_mcts_simulate: no weights (azr_nn_init_random / azr_nn_set_weights / azr_nn_load)"; return AZR_E_STATE; }
    int rc = azr_mcts_begin(h);
    if (rc) return rc;
    for (;;) {
        uint32_t active = 0;
        rc = tree_step_host(h, &active);
        if (rc) return rc;
        if (active == 0) break;
        rc = net_forward(h, h->d.leaf_in, LEAF_STRIDE, h->d.G * h->d.T, h->d.net_pi, h->d.net_v);
        if (rc) return rc;
    }
    // surface per-game errors
    std::vector<uint32_t> e(h->d.G);
    HIPCHK(h, hipMemcpy2DAsync(e.data(), 4, &h->d.ctl[0].error, sizeof(Ctl), 4, h->d.G, hipMemcpyDeviceToHost, h->stream));
    SYNC(h);
    for (int g = 0; g < h->d.G; g++)
        if (e[g]) { h->err = "search error in game " + std::to_string(g) + " code " + std::to_string(e[g]); return (int)e[g]; }
    return AZR_OK;
}

extern "C" int azr_mcts_root_stats(azr_engine* h, uint32_t* n, float* q, float* p)
{
    ENTER(h);
    const size_t sz = (size_t)h->d.G * MOVES * 4;
    DevBuf bn, bq, bp;
    HIPCHK(h, bn.alloc(sz)); HIPCHK(h, bq.alloc(sz)); HIPCHK(h, bp.alloc(sz));
    LAUNCH(h, k_root_stats, h->d, (uint32_t*)bn.p, (float*)bq.p, (float*)bp.p, (float*)nullptr);
    if (n) D2H(h, n, bn.p, sz);
    if (q) D2H(h, q, bq.p, sz);
    if (p) D2H(h, p, bp.p, sz);
    SYNC(h);
    return AZR_OK;
}

extern "C" int azr_mcts_policy(azr_engine* h, float* pi)
{
    ENTER(h);
    if (!pi) return AZR_E_INVALID_ARGUMENT;
    const size_t sz = (size_t)h->d.G * MOVES * 4;
    DevBuf b;
    HIPCHK(h, b.alloc(sz));
    LAUNCH(h, k_root_stats, h->d, (uint32_t*)nullptr, (float*)nullptr, (float*)nullptr, (float*)b.p);
    D2H(h, pi, b.p, sz);
    SYNC(h);
    return AZR_OK;
}

extern "C" int azr_mcts_pick(azr_engine* h, int sample, uint8_t* moves)
{
    ENTER(h);
    if (!moves) return AZR_E_INVALID_ARGUMENT;
    DevBuf b;
    HIPCHK(h, b.alloc(h->d.G));
    LAUNCH(h, k_pick, h->d, sample, (uint8_t*)b.p);
    D2H(h, moves, b.p, (size_t)h->d.G);
    SYNC(h);
    return AZR_OK;
}

// ---- device-resident self-play ------------------------------------------------------------------------
static int selfplay_start(azr_engine* h, uint32_t base_seed, unsigned long long quota, int keep = 0)
{
    h->d.base_seed = base_seed;
    h->d.sp_quota = quota;
    h->d.sp_compact = 0;
    h->sp_tail = false;
    h->mode = 2;
    const unsigned long long started = quota ? std::min<unsigned long long>(quota, (unsigned long long)h->d.G) : 0ull;
    HIPCHK(h, hipMemcpyAsync(h->d.sp_started, &started, sizeof started, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d.counters, 0, (size_t)h->d.G * sizeof(Counters), h->stream));
    HIPCHK(h, hipMemsetAsync(h->d.ring_count, 0, sizeof(unsigned long long), h->stream));
    LAUNCH(h, k_selfplay_start, h->d, keep);
    SYNC(h);
    return AZR_OK;
}

extern "C" int azr_selfplay_start_from_states(azr_engine* h, uint32_t base_seed)
{
    ENTER(h);
    return selfplay_start(h, base_seed, 0, 1);
}

extern "C" int azr_selfplay_start(azr_engine* h, uint32_t base_seed)
{
    ENTER(h);
    return selfplay_start(h, base_seed, 0);
}

extern "C" int azr_selfplay_start_games(azr_engine* h, uint32_t base_seed, uint64_t games)
{
    ENTER(h);
    if (games == 0) return AZR_E_INVALID_ARGUMENT;
    return selfplay_start(h, base_seed, games);
}

extern "C" int azr_selfplay_run(azr_engine* h, int passes)
{
    ENTER(h);
    if (h->mode != 2) { h->err = "azr_selfplay_run: call azr_selfplay_start first"; return AZR_E_STATE; }
    if (!h->weights_set) { h->err = "azr_selfplay_run: no weights"; return AZR_E_STATE; }
    // launches timed with HIP events, spread evenly over the run.  A sample, not every pass: an event is a marker packet the
    // queue has to retire, and five of them per pass cost ~18 us of a 1.1 ms pass.
    const int PROF_MAX = 24;
    const int nprof = std::min(passes, PROF_MAX);
    while ((int)h->ev.size() < 5 * PROF_MAX) {
        hipEvent_t e;
        HIPCHK(h, hipEventCreate(&e));
        h->ev.push_back(e);
    }
    const int stride = passes > nprof ? passes / nprof : 1;
    int k = 0;
    const int GT = h->d.G * h->d.T;
    for (int p = 0; p < passes; p++) {
        // Quota mode (azr_selfplay_start_games): once every game has been started the slots go idle one by one.  From then
        // on a pass evaluates the waiting leaf slots only (listed by the tree step, one count read-back per pass) and the
        // tile plan follows the shrinking batch — a launch over 1024 mostly idle slots costs as much as a full one.
        if (h->d.sp_quota && !h->sp_tail && p % 32 == 0) {
            unsigned long long started = 0;
            D2H(h, &started, h->d.sp_started, sizeof started);
            SYNC(h);
            h->sp_tail = started >= h->d.sp_quota;
            h->d.sp_compact = h->sp_tail ? 1 : 0;
        }
        int n_eval = GT;
        const int* map = nullptr;
        const bool prof = (p % stride == 0) && k < nprof;
        if (h->sp_tail) HIPCHK(h, hipMemsetAsync(h->d.leaf_count, 0, sizeof(int), h->stream));
        if (prof) HIPCHK(h, hipEventRecord(h->ev[3 * k + 0], h->stream));
        LAUNCH(h, k_tree_step<true>, h->d);
        if (prof) HIPCHK(h, hipEventRecord(h->ev[3 * k + 1], h->stream));
        if (h->sp_tail) {
            D2H(h, &n_eval, h->d.leaf_count, sizeof(int));
            SYNC(h);
            map = h->d.leaf_list;
            if (n_eval == 0) {   // nothing waits for the net: every game of the quota is over
                if (prof) { HIPCHK(h, hipEventRecord(h->ev[3 * k + 2], h->stream)); h->pe_tower0 = h->pe_tower1 = nullptr; }
                break;
            }
        }
        h->pe_tower0 = prof ? h->ev[3 * PROF_MAX + 2 * k] : nullptr;
        h->pe_tower1 = prof ? h->ev[3 * PROF_MAX + 2 * k + 1] : nullptr;
        int rc = net_forward_ex(h, h->d.leaf_in, LEAF_STRIDE, n_eval, h->d.net_pi, h->d.net_v, map, h->stream);
        h->pe_tower0 = h->pe_tower1 = nullptr;
        if (rc) return rc;
        if (prof) { HIPCHK(h, hipEventRecord(h->ev[3 * k + 2], h->stream)); k++; }
    }
    SYNC(h);
    double tn = 0, tt = 0, tw = 0;
    const bool tower_timed = h->cfg.net_dtype == AZR_NET_BF16 || h->cfg.net_dtype == AZR_NET_F32X || h->cfg.net_dtype == AZR_NET_F16;   // one kernel = one net forward, bracketed by events
    for (int i = 0; i < k; i++) {
        float a = 0, b = 0, c = 0;
        HIPCHK(h, hipEventElapsedTime(&a, h->ev[3 * i + 0], h->ev[3 * i + 1]));
        HIPCHK(h, hipEventElapsedTime(&b, h->ev[3 * i + 1], h->ev[3 * i + 2]));
        if (tower_timed) HIPCHK(h, hipEventElapsedTime(&c, h->ev[3 * PROF_MAX + 2 * i], h->ev[3 * PROF_MAX + 2 * i + 1]));
        tt += a; tn += b; tw += c;
    }
    h->prof_tower_ms = k ? (float)(tw / k) : 0;
    h->prof_launches = k;
    h->prof_tree_ms = k ? (float)(tt / k) : 0;
    h->prof_net_ms = k ? (float)(tn / k) : 0;
    return AZR_OK;
}

extern "C" int azr_profile_last_run(azr_engine* h, float* net_ms, float* tree_ms, int* launches)
{
    if (!h) return AZR_E_BAD_HANDLE;
    // net_ms: the dominant kernel alone (k_tower_bf16) when the bf16 path is active, else the whole fp32 forward
    if (net_ms) *net_ms = h->prof_tower_ms > 0 ? h->prof_tower_ms : h->prof_net_ms;
    if (tree_ms) *tree_ms = h->prof_tree_ms;
    if (launches) *launches = h->prof_launches;
    return AZR_OK;
}

extern "C" int azr_selfplay_counters(azr_engine* h, azr_counters* out)
{
    ENTER(h);
    if (!out) return AZR_E_INVALID_ARGUMENT;
    std::vector<Counters> rows(h->d.G);
    D2H(h, rows.data(), h->d.counters, rows.size() * sizeof(Counters));
    SYNC(h);
    Counters c;
    memset(&c, 0, sizeof c);
    for (const Counters& r : rows) {
        c.simulations += r.simulations; c.evaluations += r.evaluations; c.levels += r.levels; c.decisions += r.decisions;
        c.games_finished += r.games_finished; c.samples += r.samples; c.nodes_dropped += r.nodes_dropped; c.errors += r.errors;
        c.ring_dropped += r.ring_dropped;
    }
    out->simulations = c.simulations; out->evaluations = c.evaluations; out->levels = c.levels;
    out->decisions = c.decisions; out->games_finished = c.games_finished; out->samples = c.samples;
    out->nodes_dropped = c.nodes_dropped; out->errors = c.errors;
    out->records_dropped = c.ring_dropped;
    out->tower_fallbacks = 0;
    {
        unsigned long long f = 0;
        int rc = net_fallbacks(h, &f);
        if (rc) return rc;
        out->tower_fallbacks = f;
        if (h->opponent) {   // two-net arena: the opponent's launches ran on its own handle
            rc = net_fallbacks(h->opponent, &f);
            if (rc) { h->err = h->opponent->err; return rc; }
            out->tower_fallbacks += f;
        }
    }
    return AZR_OK;
}

extern "C" int azr_samples_drain(azr_engine* h, void* rec265, size_t cap, size_t* n_out)
{
    ENTER(h);
    unsigned long long n = 0;
    D2H(h, &n, h->d.ring_count, 8);
    SYNC(h);
    if (n > h->d.ring_cap) n = h->d.ring_cap;
    if (!rec265) cap = (size_t)n;   // no buffer: discard everything (reset of the ring)
    size_t take = std::min((size_t)n, cap);
    if (take && rec265) D2H(h, rec265, h->d.ring, take * AZR_RECORD_BYTES);
    const unsigned long long left = n - take;
    if (left) {  // partial drain: the records that did not fit move to the front of the ring and stay
        DevBuf tmp;
        HIPCHK(h, tmp.alloc((size_t)left * AZR_RECORD_BYTES));
        HIPCHK(h, hipMemcpyAsync(tmp.p, h->d.ring + take * AZR_RECORD_BYTES, (size_t)left * AZR_RECORD_BYTES, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d.ring, tmp.p, (size_t)left * AZR_RECORD_BYTES, hipMemcpyDeviceToDevice, h->stream));
        SYNC(h);
    }
    HIPCHK(h, hipMemcpyAsync(h->d.ring_count, &left, 8, hipMemcpyHostToDevice, h->stream));
    SYNC(h);
    if (n_out) *n_out = take;
    return AZR_OK;
}

extern "C" int azr_samples_device_view(azr_engine* h, void** dev_ptr, size_t* n_out)
{
    ENTER(h);
    unsigned long long n = 0;
    D2H(h, &n, h->d.ring_count, 8);
    SYNC(h);
    if (n > h->d.ring_cap) n = h->d.ring_cap;
    if (dev_ptr) *dev_ptr = h->d.ring;
    if (n_out) *n_out = (size_t)n;
    return AZR_OK;
}

extern "C" int azr_samples_copy_device(azr_engine* h, void* dst_device, size_t cap, size_t* n_out)
{
    ENTER(h);
    unsigned long long n = 0;
    D2H(h, &n, h->d.ring_count, 8);
    SYNC(h);
    if (n > h->d.ring_cap) n = h->d.ring_cap;
    const size_t take = std::min((size_t)n, cap);
    if (take && !dst_device) return AZR_E_INVALID_ARGUMENT;
    if (take) HIPCHK(h, hipMemcpyAsync(dst_device, h->d.ring, take * AZR_RECORD_BYTES, hipMemcpyDeviceToDevice, h->stream));
    SYNC(h);
    if (n_out) *n_out = take;
    return AZR_OK;
}

extern "C" int azr_device_synchronize(azr_engine* h)
{
    ENTER(h);
    SYNC(h);
    return AZR_OK;
}

// ---- arena: GameGroup::playGames (game/game.cpp:256-312) ----------------------------------------------------------
extern "C" int azr_arena_start(azr_engine* h, int player1, int player2, int games, int games_per_slot_cap, int mirror_games,
                               uint32_t base_seed)
{
    ENTER(h);
    if (player1 < 0 || player1 > 3 || player2 < 0 || player2 > 3 || games < 0) return AZR_E_INVALID_ARGUMENT;
    if (mirror_games < AZR_MIRROR_OFF || mirror_games > AZR_MIRROR_CONCURRENT) { h->err = "azr_arena_start: mirror_games must be AZR_MIRROR_OFF / _SEQUENTIAL / _CONCURRENT"; return AZR_E_INVALID_ARGUMENT; }
    if (mirror_games == AZR_MIRROR_CONCURRENT && h->d.G < 2) { h->err = "azr_arena_start: AZR_MIRROR_CONCURRENT needs at least 2 slots (a pair's halves play on slots 2j and 2j + 1)"; return AZR_E_INVALID_ARGUMENT; }
    if (player1 == player2 && (player1 == AZR_PLAYER_ALPHAZERO || player1 == AZR_PLAYER_ALPHAZERO_B)) {
        h->err = "azr_arena_start: two AlphaZero players need one tree each: use AZR_PLAYER_ALPHAZERO vs AZR_PLAYER_ALPHAZERO_B "
                 "(azr_arena_set_opponent_net(h, h) for the same network on both sides)";
        return AZR_E_STATE;
    }
    const bool usesA = player1 == AZR_PLAYER_ALPHAZERO || player2 == AZR_PLAYER_ALPHAZERO;
    const bool usesB = player1 == AZR_PLAYER_ALPHAZERO_B || player2 == AZR_PLAYER_ALPHAZERO_B;
    if (usesA && !h->weights_set) { h->err = "azr_arena_start: no weights"; return AZR_E_STATE; }
    if (usesB && (!h->opponent || !h->opponent->weights_set)) {
        h->err = "azr_arena_start: AZR_PLAYER_ALPHAZERO_B needs azr_arena_set_opponent_net with a handle that has weights";
        return AZR_E_STATE;
    }
    Dev& d = h->d;
    d.nodes2 = usesB ? h->tree2[0] ? (uint8_t*)h->tree2[0] : nullptr : nullptr;
    if (d.arena_collect) HIPCHK(h, hipMemsetAsync(d.ring_count, 0, sizeof(unsigned long long), h->stream));
    d.kind0 = player1; d.kind1 = player2; d.arena_total = games; d.arena_slot_cap = games_per_slot_cap;
    d.arena_mirror = mirror_games; d.base_seed = base_seed;
    h->mode = 3;
    HIPCHK(h, hipMemsetAsync(d.arena_taken, 0, sizeof(int), h->stream));
    HIPCHK(h, hipMemsetAsync(d.leaf_count, 0, 4 * sizeof(int), h->stream));
    h->arena_pass = 0;
    HIPCHK(h, hipMemsetAsync(d.arena_res, 0, 8 * sizeof(int), h->stream));
    HIPCHK(h, hipMemsetAsync(d.counters, 0, (size_t)d.G * sizeof(Counters), h->stream));
    HIPCHK(h, hipMemsetAsync(d.alog_status, 0, (size_t)d.G * ALOG, h->stream));
    LAUNCH(h, k_arena_start, d);
    SYNC(h);
    return AZR_OK;
}

extern "C" int azr_arena_run(azr_engine* h, int passes, int* finished_out)
{
    ENTER(h);
    if (h->mode != 3) { h->err = "azr_arena_run: call azr_arena_start first"; return AZR_E_STATE; }
    const bool needs_net = h->d.kind0 == AZR_PLAYER_ALPHAZERO || h->d.kind1 == AZR_PLAYER_ALPHAZERO;
    const int GT = h->d.G * h->d.T;
    // Up to 256 waiting leaves on the 16-bit towers: the net launches read the tree step's leaf counts from device memory themselves
    // (net_forward_counted), so a pass is queued without a read-back and the host looks at the slots' states once per CHUNK passes —
    // the passes of a slot that went idle meanwhile end at their first instruction.  (AZR_ARENA_COUNTED=0, test build: the read-back form.)
    // (At most min(slots, games) slots ever play: an engine of 512 slots that plays 100 compare games has at most 200 leaves waiting.)
    const int NB_MAX = std::min(h->d.G, std::max(1, h->d.arena_total)) * h->d.T;
    const bool counted = hook_env_int("AZR_ARENA_COUNTED", 1) != 0 && net_forward_counted_ok(h, NB_MAX) &&
                         (!h->d.nodes2 || net_forward_counted_ok(h->opponent, NB_MAX));
    if (counted && (needs_net || h->d.nodes2)) {
        constexpr int CHUNK = 16;
        if (!h->arena_ev) HIPCHK(h, hipEventCreateWithFlags(&h->arena_ev, hipEventDisableTiming));
        if (!h->arena_ev2) HIPCHK(h, hipEventCreateWithFlags(&h->arena_ev2, hipEventDisableTiming));
        std::vector<uint32_t> st0(h->d.G);
        for (int p = 0; p < passes; p++) {
            // the counts of this pass go to row (pass & 1) of leaf_count — zero since the pass before the last (or azr_arena_start) — and the
            // step zeroes the other row for the next pass: no memset between a pass's launches either
            const int row = 2 * (int)(h->arena_pass++ & 1u);
            Dev e = h->d;
            e.lc_base = row; e.lc_zero = 2 - row;
            LAUNCH(h, k_arena_step, e);
            const int* cnt_dev = h->d.leaf_count + row;
            const bool beside = h->d.nodes2 && h->opponent != h;   // two launches on two streams (one handle on both sides: one stream, one after the other)
            if (h->d.nodes2) {   // the opponent's net on the opponent's stream, side by side with this one's: after the tree step, before the next
                HIPCHK(h, hipEventRecord(h->arena_ev2, h->stream));
                HIPCHK(h, hipStreamWaitEvent(h->opponent->stream, h->arena_ev2, 0));
                int rc = net_forward_counted(h->opponent, h->d.leaf_in, LEAF_STRIDE, NB_MAX, cnt_dev + 1, beside ? cnt_dev : nullptr, h->d.net_pi, h->d.net_v, h->d.leaf_list + GT, h->opponent->stream);
                if (rc) { h->err = h->opponent->err; return rc; }
                HIPCHK(h, hipEventRecord(h->arena_ev, h->opponent->stream));
            }
            int rc = net_forward_counted(h, h->d.leaf_in, LEAF_STRIDE, NB_MAX, cnt_dev, beside ? cnt_dev + 1 : nullptr, h->d.net_pi, h->d.net_v, h->d.leaf_list, h->stream);
            if (rc) return rc;
            if (h->d.nodes2) HIPCHK(h, hipStreamWaitEvent(h->stream, h->arena_ev, 0));
            if (p % CHUNK == CHUNK - 1 && p + 1 < passes) {
                HIPCHK(h, hipMemcpy2DAsync(st0.data(), 4, &h->d.ctl[0].arena_state, sizeof(Ctl), 4, h->d.G, hipMemcpyDeviceToHost, h->stream));
                SYNC(h);
                bool all_idle = true;
                for (uint32_t v : st0) all_idle = all_idle && v == 2;
                if (all_idle) break;
            }
        }
    } else
    if (h->d.nodes2) {  // two networks: every pass evaluates each net on the leaves of its own player only
        if (!h->arena_ev) HIPCHK(h, hipEventCreateWithFlags(&h->arena_ev, hipEventDisableTiming));
        for (int p = 0; p < passes; p++) {
            HIPCHK(h, hipMemsetAsync(h->d.leaf_count, 0, 2 * sizeof(int), h->stream));
            LAUNCH(h, k_arena_step, h->d);
            int cnt[2] = {0, 0};
            D2H(h, cnt, h->d.leaf_count, sizeof cnt);
            SYNC(h);
            if (cnt[0] == 0 && cnt[1] == 0) break;  // every slot is idle: the quota is exhausted
            // The two launches are independent (own weights, disjoint leaf slots) and small — an arena of 100 games is
            // 1 board per workgroup on fewer than half of the CUs each — so they run side by side: this net on this
            // handle's stream, the opponent's on the opponent's; the next tree step waits for both.  (The tree step that
            // wrote the leaves has completed: the count read-back above synchronised the stream.)
            int rc = net_forward_ex(h, h->d.leaf_in, LEAF_STRIDE, cnt[0], h->d.net_pi, h->d.net_v, h->d.leaf_list, h->stream);
            if (rc) return rc;
            if (cnt[1] > 0) {
                rc = net_forward_ex(h->opponent, h->d.leaf_in, LEAF_STRIDE, cnt[1], h->d.net_pi, h->d.net_v, h->d.leaf_list + GT, h->opponent->stream);
                if (rc) { h->err = h->opponent->err; return rc; }
                HIPCHK(h, hipEventRecord(h->arena_ev, h->opponent->stream));
                HIPCHK(h, hipStreamWaitEvent(h->stream, h->arena_ev, 0));
            }
        }
    } else
    for (int p = 0; p < passes; p++) {
        if (needs_net) HIPCHK(h, hipMemsetAsync(h->d.leaf_count, 0, sizeof(int), h->stream));
        LAUNCH(h, k_arena_step, h->d);
        if (needs_net) {   // one network: evaluate the waiting leaf slots only (one count read-back per pass)
            int cnt = 0;
            D2H(h, &cnt, h->d.leaf_count, sizeof cnt);
            SYNC(h);
            if (cnt == 0) {
                // nobody waits for the net: either every slot is idle (quota exhausted) or only scripted players moved
                std::vector<uint32_t> st0(h->d.G);
                HIPCHK(h, hipMemcpy2DAsync(st0.data(), 4, &h->d.ctl[0].arena_state, sizeof(Ctl), 4, h->d.G, hipMemcpyDeviceToHost, h->stream));
                SYNC(h);
                bool all_idle = true;
                for (uint32_t v : st0) all_idle = all_idle && v == 2;
                if (all_idle) break;
                continue;
            }
            int rc = net_forward_ex(h, h->d.leaf_in, LEAF_STRIDE, cnt, h->d.net_pi, h->d.net_v, h->d.leaf_list, h->stream);
            if (rc) return rc;
        }
    }
    std::vector<uint32_t> st(h->d.G);
    HIPCHK(h, hipMemcpy2DAsync(st.data(), 4, &h->d.ctl[0].arena_state, sizeof(Ctl), 4, h->d.G, hipMemcpyDeviceToHost, h->stream));
    SYNC(h);
    int idle = 0;
    for (uint32_t v : st) idle += v == 2;
    if (finished_out) *finished_out = idle == h->d.G;
    return AZR_OK;
}

extern "C" int azr_arena_set_opponent_net(azr_engine* h, azr_engine* other)
{
    ENTER(h);
    if (!other) { h->opponent = nullptr; return AZR_OK; }
    if (other->cfg.device != h->cfg.device || other->net.blocks != h->net.blocks || other->cfg.net_dtype != h->cfg.net_dtype ||
        other->d.G * other->d.T < h->d.G * h->d.T) {
        h->err = "azr_arena_set_opponent_net: the opponent handle must be on the same device with the same net shape and >= leaf slots";
        return AZR_E_INVALID_ARGUMENT;
    }
    Dev& d = h->d;
    if (!h->tree2[0]) {  // the second AlphaZeroPlayer's tree per slot + the per-net leaf lists
        const size_t G = d.G, C = d.C;
        uint8_t* n2 = nullptr; uint32_t *t2 = nullptr, *h2 = nullptr, *tb2 = nullptr, *tc2 = nullptr; uint16_t* f2 = nullptr;
        HIPCHK(h, dmalloc(&n2, G * C * NODE_BYTES)); h->tree2[0] = n2;
        HIPCHK(h, dmalloc(&t2, G * C)); h->tree2[1] = t2;
        HIPCHK(h, dmalloc(&h2, G * C)); h->tree2[2] = h2;
        HIPCHK(h, dmalloc(&tb2, G * d.H)); h->tree2[3] = tb2;
        HIPCHK(h, dmalloc(&f2, G * C)); h->tree2[4] = f2;
        HIPCHK(h, dmalloc(&tc2, G * 4)); h->tree2[5] = tc2;
        HIPCHK(h, hipMemsetAsync(t2, 0, G * C * sizeof(uint32_t), h->stream));
        HIPCHK(h, hipMemsetAsync(tb2, 0, G * d.H * sizeof(uint32_t), h->stream));
        HIPCHK(h, hipMemsetAsync(tc2, 0, G * 4 * sizeof(uint32_t), h->stream));
        d.touch2 = t2; d.nhash2 = h2; d.table2 = tb2; d.freel2 = f2; d.tctl2 = tc2;
        SYNC(h);
    }
    h->opponent = other;
    return AZR_OK;
}

extern "C" int azr_arena_collect_samples(azr_engine* h, int on)
{
    if (!h) return AZR_E_BAD_HANDLE;
    h->d.arena_collect = on ? 1 : 0;
    return AZR_OK;
}

extern "C" int azr_arena_results(azr_engine* h, azr_game_results* out)
{
    ENTER(h);
    if (!out) return AZR_E_INVALID_ARGUMENT;
    int r[8];
    D2H(h, r, h->d.arena_res, sizeof r);
    SYNC(h);
    out->count = r[0]; out->draw = r[1]; out->win[0] = r[2]; out->win_and_started[0] = r[3];
    out->win[1] = r[4]; out->win_and_started[1] = r[5];
    return AZR_OK;
}

extern "C" int azr_arena_log(azr_engine* h, int32_t* games_per_slot, int8_t* status, uint16_t* rounds, void* finals160)
{
    ENTER(h);
    const int G = h->d.G;
    if (games_per_slot)
        HIPCHK(h, hipMemcpy2DAsync(games_per_slot, 4, &h->d.ctl[0].slot_games, sizeof(Ctl), 4, G, hipMemcpyDeviceToHost, h->stream));
    if (status) D2H(h, status, h->d.alog_status, (size_t)G * ALOG);
    if (rounds) D2H(h, rounds, h->d.alog_rounds, (size_t)G * ALOG * 2);
    if (finals160) {
        DevBuf b;
        HIPCHK(h, b.alloc((size_t)G * ALOG * 160));
        hipLaunchKernelGGL(k_export160, dim3(G * ALOG), dim3(64), 0, h->stream, h->d, (const uint8_t*)h->d.alog_final, (uint8_t*)b.p);
        HIPCHK(h, hipGetLastError());
        D2H(h, finals160, b.p, (size_t)G * ALOG * 160);
        SYNC(h);
    }
    SYNC(h);
    return AZR_OK;
}
