// azr_tree.hpp — flattened-array MCTS for one game per wavefront (gfx950, wave64).
//
// Replaces the reference's `unordered_map<State, shared_ptr<StateSimulations>>` transposition table with its
// per-node `unordered_map<LandIndex, SimulationValue>` (player/alpha_zero/alphazero_mcts.h:16-78) by, per game:
//   * a node pool in HBM, one 640-byte node = 64-B state key | header | P[44] | Q[44] | N[44]  (lane i <-> move i),
//   * an open-addressing table (u32 = tag16 | node+1) keyed by the full 64-byte game record (state equality, as
//     StateSimulationsStorage::exist / State::equalFields, alphazero_mcts.cpp:189-201, state/state.cpp:111-135),
//   * per-node "touch" stamps instead of the visited flag: trimNodes (alphazero_mcts.cpp:229-245) keeps exactly
//     the nodes created or selected-through since the previous trim,
//   * a path stack per search thread so backup (alphazero_mcts.cpp:367-375) is one lane-parallel pass after the leaf's
//     value arrives,
//   * THREADS_PER_MCTS (src/settings.h:44, alphazero_mcts.cpp:255-320) as T lock-stepped descents per game: the
//     reference's T threads block together in predictFuture until the batch is evaluated; here thread k = leaf slot
//     g * T + k of the net batch, threads run in index order, and SimulationValue::active_N (the virtual-loss count of
//     getNextBestMoveAndSetVisited) lives in the top byte of the N word.
// Everything is private to the game's wave: no atomics, no inter-wave sharing.
#pragma once
#include "azr_wave.hpp"

namespace azr {

constexpr int NODE_BYTES = 640;
constexpr int ND_KEY = 0, ND_SUMN = 64, ND_VALID_LO = 68, ND_VALID_HI = 72, ND_P = 80, ND_Q = 256, ND_N = 432;
constexpr uint32_t NO_NODE = 0xffffffffu;
constexpr int MAX_THREADS = 8;             // THREADS_PER_MCTS upper bound of this build
constexpr uint32_t N_MASK = 0x00ffffffu;   // N word: visit count in the low 24 bits, active_N in the top 8
constexpr uint32_t ACT_ONE = 0x01000000u;

struct Search {  // reference Settings the search reads (src/settings.h:45,61-64)
    int simulations;
    float c1;        // (1 - DIR_NOISE_EPSI)
    float c2;        // DIR_NOISE_EPSI * DIR_NOISE_VALUE
    float hp;        // HP_EXPLORATION
    int temperature_threshold;
};

// per-game control block (one 128-byte line)
struct alignas(128) Ctl {
    uint32_t mode;         // 0 idle | 1 host-stepped search | 2 device self-play
    uint32_t search_id;    // stamp of the current search (trim count)
    uint32_t sims_done;
    uint32_t pending;      // bit k: search thread k has a leaf written, waiting for the net's (pi, v)
    uint32_t sims_started; // Counter::i of AlphaZeroMCTS::simulate: descents claimed (sims_done = descents backed up)
    uint32_t nfree;
    uint32_t hiwater;
    uint32_t search_done;
    uint32_t rng;          // the game's minstd_rand0 state
    uint32_t game_no;      // games started in this slot
    uint32_t nsamples;     // records staged for the running game
    int32_t status;        // gameStatus of the running game (-1 running)
    uint32_t error;        // sticky rules / capacity error
    uint32_t last_move;
    uint32_t decisions;    // decisions taken in the running game
    uint32_t seed;         // seed of the running game
    // arena mode (game/game.cpp): the slot is one "thread" of GameGroup::playGames
    uint32_t arena_state;  // 0 = needs Game::newGame | 1 = playing | 2 = idle (quota exhausted)
    uint32_t player_start; // Game::playerStart
    uint32_t pair_phase;   // 0 = first game of a pair (Counter::hasNext(2) taken here) | 1 = second
    uint32_t turn_started; // AlphaZeroPlayer::takeTurn in progress (its own trimNodes done)
    uint32_t search_active;// AlphaZeroMCTS::simulate in progress
    uint32_t slot_games;   // games finished in this slot
    uint32_t dup_dropped;  // StateSimulationsStorage::duplicatedStatesDropped
    uint32_t plen[MAX_THREADS];  // path length of thread k's pending descent (0 = setRootState's root expansion)
    uint32_t search_tree;  // two-net arena: tree (= net) of the search in flight, 0 = this handle's, 1 = the opponent's
};
static_assert(sizeof(Ctl) == 128, "Ctl must be one line");

struct Tree {  // this game's slices of the engine's HBM arrays
    uint8_t* nodes;      // [C][NODE_BYTES]
    uint32_t* touch;     // [C]   0 = free slot, else search_id of the last touch
    uint32_t* nhash;     // [C]   32-bit key hash
    uint32_t* table;     // [H]   0 = empty, else tag16 << 16 | (node + 1)
    uint16_t* freel;     // [C]   free-slot stack
    uint32_t* path;      // [DMAX] node | move << 16 | flip << 24   (the current search thread's stack)
    int C, H, DMAX;
};

// global counters (one per engine), bumped with one atomic per wave per event
struct Counters {
    unsigned long long simulations, evaluations, levels, decisions, games_finished, samples, nodes_dropped, errors,
        ring_dropped;
};

// Lanes of one wave exchange data through HBM inside one launch (lane 0 writes a table slot, lane 7 updates Q,
// all lanes read them back later): make earlier stores of the wave visible to its later loads.
__device__ __forceinline__ void wave_mem_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }

__device__ __forceinline__ uint8_t* node_ptr(const Tree& t, uint32_t idx) { return t.nodes + (size_t)idx * NODE_BYTES; }

// 32-bit hash of the 64-byte record held as one dword per lane (lanes 0..15 significant)
__device__ __forceinline__ uint32_t key_hash(uint32_t kd)
{
    uint32_t l = lane_id() & 15u;
    uint32_t x = kd * 0x9E3779B1u + (l + 1u) * 0x85EBCA77u;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) x += (uint32_t)__shfl_xor((int)x, m) * 0x01000193u ^ (x >> 7);
    // the butterfly above is order-sensitive per lane; fold to one uniform value
    uint32_t h = rfl(x);
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h == 0 ? 1u : h;
}

// One round of the table's probe sequence, 16 slots at a time: lanes 0..15 (and their copies 16..63) read the 16 consecutive slots
// from `slot` on in ONE coalesced request.  Returns the candidates — slots whose tag matches, in probe order, up to the first empty
// slot — as a bit mask, `e` holding each lane's entry; `ended` = the sequence ends inside this round (an empty slot was seen).
// A lookup walked the sequence one dependent load per slot; a clustered table (after a trim re-inserts the survivors) made that
// the longest segment of the slow waves a launch waits for (20 us, profiles/r03_tree_step_profile.txt).  Same slots, same order,
// same first hit: results do not change.
__device__ __forceinline__ uint32_t probe16(const Tree& t, uint32_t slot, uint32_t tag, uint32_t& e, bool& ended)
{
    e = t.table[(slot + (lane_id() & 15u)) & (uint32_t)(t.H - 1)];
    const uint32_t empty = (uint32_t)ballot64(e == 0) & 0xffffu;
    uint32_t match = (uint32_t)ballot64(e != 0 && (e >> 16) == tag) & 0xffffu;
    ended = empty != 0;
    if (ended) match &= (1u << __builtin_ctz(empty)) - 1u;
    return match;
}

// StateSimulationsStorage::exist + getStateSimulation: node index of the record `kd`, or NO_NODE
__device__ __forceinline__ uint32_t tree_lookup(const Tree& t, uint32_t kd, uint32_t h)
{
    const uint32_t tag = h >> 16;
    uint32_t slot = h & (uint32_t)(t.H - 1);
    for (int probes = 0; probes < t.H; probes += 16) {
        uint32_t e;
        bool ended;
        uint32_t match = probe16(t, slot, tag, e, ended);
        while (match) {
            const uint32_t j = (uint32_t)__builtin_ctz(match);
            match &= match - 1u;
            const uint32_t idx = (rdl(e, j) & 0xffffu) - 1u;
            uint32_t nk = reinterpret_cast<const uint32_t*>(node_ptr(t, idx))[lane_id() & 15u];
            if (ballot64(nk != kd) == 0) return idx;
        }
        if (ended) return NO_NODE;
        slot = (slot + 16) & (uint32_t)(t.H - 1);
    }
    return NO_NODE;
}

__device__ __forceinline__ void tree_insert(const Tree& t, uint32_t h, uint32_t idx)
{
    const uint32_t mask = (uint32_t)(t.H - 1);
    uint32_t slot = h & mask;
    for (int probes = 0; probes < t.H; probes += 16) {   // the first empty slot of the probe sequence, 16 slots per round trip
        const uint32_t e = t.table[(slot + (lane_id() & 15u)) & mask];
        const uint32_t empty = (uint32_t)ballot64(e == 0) & 0xffffu;
        if (empty) {
            if (lane_id() == 0) t.table[(slot + (uint32_t)__builtin_ctz(empty)) & mask] = (h & 0xffff0000u) | (idx + 1u);
            wave_mem_sync();
            return;
        }
        slot = (slot + 16) & mask;
    }
}

__device__ __forceinline__ uint32_t tree_alloc(const Tree& t, Ctl& c)
{
    if (c.nfree > 0) {
        c.nfree--;
        return rfl((uint32_t)t.freel[c.nfree]);
    }
    if ((int)c.hiwater < t.C) return c.hiwater++;
    return NO_NODE;
}

// StateSimulationsStorage::clearNodes (alphazero_mcts.cpp:223-227)
__device__ __forceinline__ void tree_clear(const Tree& t, Ctl& c)
{
    for (int i = (int)lane_id(); i < t.H; i += 64) t.table[i] = 0;
    for (int i = (int)lane_id(); i < (int)c.hiwater; i += 64) t.touch[i] = 0;
    c.nfree = 0;
    c.hiwater = 0;
    c.search_id = 1;
    wave_mem_sync();
}

// StateSimulationsStorage::trimNodes (alphazero_mcts.cpp:229-245): survivors = touched under the previous stamp.
// Rebuilds the table and the free stack; node bodies do not move.  The survivors of a 64-node chunk are re-inserted by their
// lanes side by side (compare-and-swap on the empty slot, linear probing): which slot a key lands in depends on the order the
// L2 retires the lanes' atomics, what a lookup returns does not (it compares tag and full key along the probe sequence, and
// nothing is ever deleted from a table between two rebuilds).  One at a time the re-insertion was ~1 us per survivor — the
// slowest wave of a tree step, the one the launch waits for, is a game that has just moved.
__device__ __forceinline__ void tree_trim(const Tree& t, Ctl& c)
{
    const uint32_t prev = c.search_id;
    c.search_id = prev + 1;
    for (int i = (int)lane_id(); i < t.H; i += 64) t.table[i] = 0;
    wave_mem_sync();
    uint32_t nfree = 0;
    const uint32_t hw = c.hiwater;
    const uint32_t mask = (uint32_t)(t.H - 1);
    for (uint32_t base = 0; base < hw; base += 64) {
        uint32_t i = base + lane_id();
        uint32_t tc = i < hw ? t.touch[i] : 0u;
        uint32_t hv = i < hw ? t.nhash[i] : 0u;
        bool alive = i < hw && tc == prev;
        bool dead = i < hw && !alive;
        uint64_t dm = ballot64(dead);
        if (dead) {
            uint32_t pos = nfree + (uint32_t)popc64(dm & ((1ULL << lane_id()) - 1ULL));
            t.freel[pos] = (uint16_t)i;
            t.touch[i] = 0;
        }
        nfree += (uint32_t)popc64(dm);
        if (alive) {
            const uint32_t entry = (hv & 0xffff0000u) | (i + 1u);
            uint32_t slot = hv & mask;
            for (int probes = 0; probes < t.H; probes++) {
                if (atomicCAS(&t.table[slot], 0u, entry) == 0u) break;
                slot = (slot + 1) & mask;
            }
        }
    }
    c.nfree = nfree;
    wave_mem_sync();
    // the atomics ran in L2: drop this CU's L1 lines of the table (cleared by plain stores above) before the wave's next lookups
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// Two trimNodes in a row with no search between them — AlphaZeroPlayer::takeTurn's own trim followed by setRootState's at the first
// decision of a turn (alphazero_player.cpp:5, alphazero_mcts.cpp:255-270) — keep NOTHING: the first leaves the survivors' stamps as
// they were, the second finds no node stamped since.  The state two tree_trim calls leave behind, written directly: empty table, every
// slot below the high-water mark free in ascending order, the stamp counter two further — without the first trim's scan, its
// survivors' re-insertion and the second clearing of the table.
__device__ __forceinline__ void tree_trim_twice(const Tree& t, Ctl& c)
{
    c.search_id += 2;
    for (int i = (int)lane_id(); i < t.H; i += 64) t.table[i] = 0;
    const uint32_t hw = c.hiwater;
    for (uint32_t i = lane_id(); i < hw; i += 64) {
        t.freel[i] = (uint16_t)i;
        t.touch[i] = 0;
    }
    c.nfree = hw;
    wave_mem_sync();
}

// libstdc++ unordered_map<LandIndex,...> iteration order (alphazero_mcts.cpp:78): returns the key
// of `ties` that comes first when the set bits of `valid` were inserted in ascending order.  Emulates
// _Hashtable::_M_insert_bucket_begin / _M_rehash_aux with the prime policy 13 -> 29 -> 59.  Executed by lane 0
// only (ties between exactly equal PUCT scores are rare); `scratch` = 128 bytes of LDS private to the wave.
__device__ __noinline__ uint32_t umap_first(uint64_t valid, uint64_t ties, int8_t* scratch)
{
    uint32_t result = NONE;
    if (lane_id() == 0) {
        int8_t* next = scratch;        // [64], index 63 = before-begin
        int8_t* bucket = scratch + 64; // [64]
        for (int i = 0; i < 64; i++) { next[i] = -1; bucket[i] = -1; }
        const int BB = 63;
        int nb = 1, size = 0, next_resize = 0;
        for (int k = 0; k < MOVES; k++) {
            if (!(valid & (1ULL << k))) continue;
            if (size + 1 > next_resize) {
                int want = size + 1;
                if (next_resize == 0 && want < 11) want = 11;
                if (want >= nb) {
                    int a = want + 1, b = nb * 2;
                    int n = a > b ? a : b;
                    int newnb = n <= 13 ? 13 : n <= 29 ? 29 : 59;  // _M_next_bkt for the sizes that can occur
                    next_resize = newnb;
                    // _M_rehash_aux (unique keys)
                    int p = next[BB];
                    for (int i = 0; i < 64; i++) bucket[i] = -1;
                    next[BB] = -1;
                    int bbegin = 0;
                    while (p >= 0) {
                        int nx = next[p];
                        int bk = p % newnb;
                        if (bucket[bk] < 0) {
                            next[p] = next[BB];
                            next[BB] = (int8_t)p;
                            bucket[bk] = BB;
                            if (next[p] >= 0) bucket[bbegin] = (int8_t)p;
                            bbegin = bk;
                        } else {
                            next[p] = next[bucket[bk]];
                            next[bucket[bk]] = (int8_t)p;
                        }
                        p = nx;
                    }
                    nb = newnb;
                } else next_resize = nb;
            }
            int bk = k % nb;
            if (bucket[bk] >= 0) {
                next[k] = next[bucket[bk]];
                next[bucket[bk]] = (int8_t)k;
            } else {
                next[k] = next[BB];
                next[BB] = (int8_t)k;
                if (next[k] >= 0) bucket[next[k] % nb] = (int8_t)k;
                bucket[bk] = BB;
            }
            size++;
        }
        for (int p = next[BB]; p >= 0; p = next[p])
            if (ties & (1ULL << p)) { result = (uint32_t)p; break; }
    }
    return rfl(result);
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// the fields of a node the selection reads, one lane per move (lanes >= 43 hold move 0's)
struct NodeRegs { uint32_t sumN, valid_lo, valid_hi, W; float P, Q; };

__device__ __forceinline__ NodeRegs node_load(const uint8_t* n)
{
    const uint32_t l = lane_id(), ll = l < MOVES ? l : 0;
    NodeRegs r;
    r.sumN = *reinterpret_cast<const uint32_t*>(n + ND_SUMN);
    r.valid_lo = *reinterpret_cast<const uint32_t*>(n + ND_VALID_LO);
    r.valid_hi = *reinterpret_cast<const uint32_t*>(n + ND_VALID_HI);
    r.P = reinterpret_cast<const float*>(n + ND_P)[ll];
    r.Q = reinterpret_cast<const float*>(n + ND_Q)[ll];
    r.W = reinterpret_cast<const uint32_t*>(n + ND_N)[ll];
    return r;
}

// tree_lookup for the descent: the node's selection fields are requested together with its key (both hang off the table entry
// alone), so a hit hands them to tree_select without a second dependent round trip to L2 / HBM — a level of the descent is a chain of
// such round trips and little else.  A tag collision wastes the six loads; a miss issues none.
__device__ __forceinline__ uint32_t tree_lookup_node(const Tree& t, uint32_t kd, uint32_t h, NodeRegs& nr)
{
    const uint32_t tag = h >> 16;
    uint32_t slot = h & (uint32_t)(t.H - 1);
    for (int probes = 0; probes < t.H; probes += 16) {
        uint32_t e;
        bool ended;
        uint32_t match = probe16(t, slot, tag, e, ended);
        while (match) {
            const uint32_t j = (uint32_t)__builtin_ctz(match);
            match &= match - 1u;
            const uint32_t idx = (rdl(e, j) & 0xffffu) - 1u;
            const uint8_t* n = node_ptr(t, idx);
            uint32_t nk = reinterpret_cast<const uint32_t*>(n)[lane_id() & 15u];
            nr = node_load(n);
            if (ballot64(nk != kd) == 0) return idx;
        }
        if (ended) return NO_NODE;
        slot = (slot + 16) & (uint32_t)(t.H - 1);
    }
    return NO_NODE;
}

// StateSimulations::getNextBestMoveAndSetVisited (alphazero_mcts.cpp:67-119).  Float ops in the reference's order, no
// FMA.  The reference's loop keeps the first strict maximum in unordered_map iteration order over the moves that are
// not "skipped" (N == 0 && active_N == 1: another thread is already exploring that unobserved move); only when every
// candidate is skipped does it fall back to the best skipped one.  active_N++ on the chosen move.
__device__ __forceinline__ uint32_t tree_select(const Tree& t, uint32_t idx, const NodeRegs& nr, const Search& S, uint32_t stamp, int8_t* scratch)
{
    const uint8_t* n = node_ptr(t, idx);
    const uint32_t l = lane_id();
    const uint32_t sumN = rfl(nr.sumN);
    const uint64_t valid = (uint64_t)rfl(nr.valid_lo) | ((uint64_t)rfl(nr.valid_hi) << 32);
    const float P = nr.P;
    const float Q = nr.Q;
    const uint32_t W = nr.W;
    const uint32_t N = W & N_MASK, act = W >> 24;
    if (l == 0) t.touch[idx] = stamp;  // visited = true
    const float noiseP = __fadd_rn(__fmul_rn(S.c1, P), S.c2);
    const float v = __fmul_rn(__fmul_rn(noiseP, S.hp), __fsqrt_rn(__fadd_rn(1.0f, (float)sumN)));
    const float nn = __fadd_rn(1.0f, (float)N);
    float u = __fadd_rn(Q, __fdiv_rn(v, nn));
    const bool ok = l < MOVES && ((valid >> l) & 1ULL);
    const bool skip = ok && N == 0 && act == 1;
    float best = wave_max((ok && !skip) ? u : -INFINITY);
    uint64_t ties = ballot64(ok && !skip && u == best && u > -INFINITY);
    if (ties == 0) {  // bestMove == None: duplicate the best skipped request (alphazero_mcts.cpp:111-114)
        best = wave_max(skip ? u : -INFINITY);
        ties = ballot64(skip && u == best && u > -INFINITY);
        if (ties == 0) return NONE;  // all NaN / -inf: the reference's moveValues.at(None) throws
    }
    const uint32_t mv = (ties & (ties - 1)) == 0 ? (uint32_t)ctz64(ties) : umap_first(valid, ties, scratch);
    if (l == mv) reinterpret_cast<uint32_t*>(const_cast<uint8_t*>(n) + ND_N)[l] = W + ACT_ONE;  // sv.active_N++
#ifndef AZR_EXP_LATE_FENCE
    wave_mem_sync();
#endif
    return mv;
}
// NNOutputData::normalize (alphazero_nn_data.cpp:3-27): sequential fp32 sum over the legal entries, index order
__device__ __forceinline__ float normalize_prior(float pi, uint64_t valid)
{
    // (an illegal entry adds +0.0f instead of being skipped: x + 0.0f == x bit for bit for the non-negative softmax outputs summed
    //  here, and the 43 dependent adds run as straight-line code — with a branch per entry this was 1.5 us per leaf)
    float sum = 0.0f;
#pragma unroll
    for (uint32_t i = 0; i < MOVES; i++) {
        const float x = rdlf(pi, i);
        sum = __fadd_rn(sum, ((valid >> i) & 1ULL) ? x : 0.0f);
    }
    const uint32_t l = lane_id();
    float p = (l < MOVES && ((valid >> l) & 1ULL)) ? pi : 0.0f;
    if (p > 0.0f) p = __fdiv_rn(p, sum);
    return p;
}

// StateSimulations ctor + StateSimulationsStorage::add (alphazero_mcts.cpp:26-42,203-215)
__device__ __forceinline__ uint32_t tree_expand(const Tree& t, Ctl& c, uint32_t kd, uint32_t h, uint64_t valid, float prior)
{
    uint32_t idx = tree_alloc(t, c);
    if (idx == NO_NODE) return NO_NODE;
    uint8_t* n = node_ptr(t, idx);
    const uint32_t l = lane_id();
    if (l < 16) reinterpret_cast<uint32_t*>(n + ND_KEY)[l] = kd;
    if (l == 0) {
        *reinterpret_cast<uint32_t*>(n + ND_SUMN) = 0;
        *reinterpret_cast<uint32_t*>(n + ND_VALID_LO) = (uint32_t)valid;
        *reinterpret_cast<uint32_t*>(n + ND_VALID_HI) = (uint32_t)(valid >> 32);
        t.touch[idx] = c.search_id;
        t.nhash[idx] = h;
    }
    if (l < 44) {
        reinterpret_cast<float*>(n + ND_P)[l] = l < MOVES ? prior : 0.0f;
        reinterpret_cast<float*>(n + ND_Q)[l] = 0.0f;
        reinterpret_cast<uint32_t*>(n + ND_N)[l] = 0u;
    }
    tree_insert(t, h, idx);   // (ends with the fence that makes the node's body AND its table entry visible to this wave's later loads)
    return idx;
}

// SimulationValue::addValue / StateSimulations::addValue along the whole path (alphazero_mcts.cpp:8-21,55-60,
// 367-375).  Lane i handles path entry i; the value's sign at entry i is flipped once for every "player changed"
// at entries >= i.
// `fresh_path`: the path entries were written by lane 0 in THIS launch (a descent that ended in a finished game) and have to be made
// visible to the other lanes first; a leaf's path was written by the launch that found the leaf.
__device__ __forceinline__ void tree_backup(const Tree& t, uint32_t path_len, float leaf_value, bool fresh_path = true)
{
    if (fresh_path) wave_mem_sync();
    uint32_t carry = 0;  // parity of flips below the current chunk
    for (int base = ((int)path_len - 1) & ~63; base >= 0; base -= 64) {
        uint32_t i = (uint32_t)base + lane_id();
        bool on = i < path_len;
        uint32_t e = on ? t.path[i] : 0u;
        uint64_t fm = ballot64(on && ((e >> 24) & 1u));
        uint32_t flips = (uint32_t)popc64(fm >> lane_id()) + carry;
        if (on) {
            float v = (flips & 1u) ? -leaf_value : leaf_value;
            uint8_t* n = node_ptr(t, e & 0xffffu);
            uint32_t mv = (e >> 16) & 0xffu;
            float* q = reinterpret_cast<float*>(n + ND_Q) + mv;
            uint32_t* nn = reinterpret_cast<uint32_t*>(n + ND_N) + mv;
            const uint32_t W = *nn, N = W & N_MASK;
            float Q = *q;
            Q = N == 0 ? v : __fdiv_rn(__fadd_rn(__fmul_rn((float)N, Q), v), (float)(N + 1u));
            *q = Q;
            *nn = W + 1u - ((W >> 24) ? ACT_ONE : 0u);  // N++, active_N--
            *reinterpret_cast<uint32_t*>(n + ND_SUMN) += 1u;
        }
        carry = (carry + (uint32_t)popc64(fm)) & 1u;
    }
    wave_mem_sync();
}

// StateSimulations::calculateMoveProbability(1.0f) (alphazero_mcts.cpp:121-149): pow(N, 1.0) == N exactly
__device__ __forceinline__ float root_policy(uint32_t N, uint64_t valid)
{
    const uint32_t l = lane_id();
    float prob = (l < MOVES && ((valid >> l) & 1ULL)) ? (float)N : 0.0f;
    float sum = 0.0f;
    for (uint32_t i = 0; i < MOVES; i++)
        if ((valid >> i) & 1ULL) sum = __fadd_rn(sum, rdlf(prob, i));
    return __fdiv_rn(prob, sum);
}

// AlphaZeroMCTS::pickHigestWeightedMove (alphazero_mcts.cpp:397-412): strict >, lowest index wins ties, None if all <= 0
__device__ __forceinline__ uint32_t pick_highest(float pi)
{
    const uint32_t l = lane_id();
    float v = l < MOVES ? pi : -1.0f;
    if (!(v > 0.0f)) v = -1.0f;  // NaN and non-positive never win
    float best = wave_max(v);
    if (!(best > 0.0f)) return NONE;
    return (uint32_t)ctz64(ballot64(v == best));
}

// AlphaZeroMCTS::pickRandomWeightedMove (alphazero_mcts.cpp:379-395): one rFloat from the game's stream
__device__ __forceinline__ uint32_t pick_random(WS& s, float pi)
{
    float sum = 0.0f;
    for (uint32_t i = 0; i < MOVES; i++) sum = __fadd_rn(sum, rdlf(pi, i));
    float ra = __fmul_rn(sum, rng_float(s));
    float it = 0.0f;
    for (uint32_t i = 0; i < MOVES; i++) {
        it = __fadd_rn(it, rdlf(pi, i));
        if (it >= ra) return i;
    }
    return NONE;
}

}  // namespace azr
