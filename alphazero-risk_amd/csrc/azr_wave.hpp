// azr_wave.hpp — wavefront-resident Risk game state and rules for gfx950 (wave64).
//
// Design: ONE 64-lane wavefront owns ONE game.  Lane l (< 42) holds territory l's byte
// (army:6 | owner:2, the reference's `LandArmy`, state/state.h:24-34); every other field of the
// reference's `Data` (state/state.h:86-105) is wave-uniform and lives in SGPRs.  The reference's five
// incrementally maintained 48-bit masks per player (`PlayerStatus`, state/state.h:59-84, updated in
// State::setLandArmy, state/state.cpp:279-385) are NOT stored: each is one `v_cmp` + ballot over the
// lanes (owned = ballot(owner == p); attackable = ballot(owner != p && nbmask & owned) ...), which is
// what the reference's own consistencyCheck (state/state.cpp:1209-1429) defines them to be.
// Rules control flow is wave-uniform (scalar branches); the only per-lane work is compares for ballots
// and the one or two territory writes of a move.
//
// A game is persisted in HBM as a 64-byte record (one cache line, `GREC` below), which is also the key
// of a search-tree node.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace azr {

constexpr int LANDS = 42, MOVES = 43, SKIP = 42, NONE = 43, NEUTRAL = 2, ARMY_MAX = 32;
enum : uint32_t { PH_SETUP = 0, PH_SETUP_NEUTRAL, PH_REINFORCEMENT, PH_ATTACK, PH_ATTACK_MOBILIZATION, PH_FORTIFY };
enum : int { ST_NOT_ENDED = -1, ST_DRAW = -2 };
enum : uint32_t { E_OK = 0, E_INVALID_ARGUMENT = 1, E_LOGIC = 2 };
constexpr uint64_t ALL_LANDS = 0x3ffffffffffULL;
constexpr uint64_t SKIP_MASK = 1ULL << SKIP;

// 64-byte game record layout (bytes)
constexpr int GREC = 64;
constexpr int GR_CUR = 42, GR_CARD_SETS = 43, GR_REINF = 44, GR_PHASE = 45, GR_MOB_FROM = 46, GR_MOB_TO = 47,
              GR_ALLOW_DRAW = 48, GR_ATTACKS = 49, GR_ROUND_LO = 50, GR_ROUND_HI = 51, GR_CARDS0 = 52, GR_CARDS1 = 53;

struct Rules {  // src/settings.h:51-56
    int allow_yield, limit_reinforcement, limit_attack, max_game_rounds, min_unit_move;
};

// Map tables: adjacency in the reference's declaration order (land/land.cpp:246-297) — the ORDER decides
// attack-source and fortify-source ties — and continents in the order State::calculateReinforcementValue
// tests them (state/state.cpp:461-483: NA, SA, AF, EU, AS, AU; bonuses land/land_index.h:5-10).
static __constant__ uint8_t c_deg[LANDS] = {3, 4, 4, 4, 6, 3, 4, 4, 3, 3, 3, 4, 2, 3, 4, 4, 6, 5, 6, 4, 6,
                                            4, 3, 6, 3, 2, 4, 5, 3, 5, 4, 2, 5, 5, 6, 6, 4, 3, 3, 3, 3, 2};
static __constant__ __attribute__((aligned(8))) uint8_t c_nb[LANDS][8] = {
    {1, 3, 29},         {0, 3, 4, 2},        {1, 4, 5, 13},       {0, 1, 4, 6},        {1, 3, 6, 7, 5, 2},
    {4, 7, 2},          {3, 4, 7, 8},        {8, 6, 4, 5},        {6, 7, 9},           {8, 10, 11},
    {9, 11, 12},        {9, 10, 12, 20},     {10, 11},            {2, 14, 15},         {13, 19, 15, 17},
    {13, 14, 16, 17},   {15, 17, 18, 35, 33, 26}, {15, 14, 18, 19, 16}, {19, 17, 16, 20, 21, 35}, {20, 14, 18, 17},
    {11, 19, 18, 21, 23, 22}, {18, 20, 23, 35}, {20, 23, 24},      {21, 20, 22, 24, 25, 35}, {22, 23, 25},
    {24, 23},           {16, 33, 34, 27},    {26, 34, 32, 30, 28}, {27, 30, 29},       {28, 30, 32, 31, 0},
    {28, 29, 32, 27},   {29, 32},            {27, 30, 29, 31, 34}, {16, 26, 34, 36, 35}, {32, 27, 26, 33, 36, 37},
    {21, 23, 18, 16, 33, 36}, {35, 33, 34, 37}, {36, 34, 38},      {37, 39, 40},        {38, 41, 40},
    {41, 39, 38},       {40, 39}};
static __constant__ uint64_t c_cont_mask[6] = {0x1ffULL, 0x1e00ULL, 0x3f00000ULL, 0xfe000ULL, 0x3ffc000000ULL,
                                               0x3c000000000ULL};
static __constant__ int c_cont_bonus[6] = {5, 2, 3, 5, 7, 2};

// ---- wave primitives -------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Timing-experiment builds only (-DAZR_TREE_PROF, tools/tree_prof.sh; never the product library): lane 0 of the tree-step wave
// adds the 100-MHz ticks since the previous mark to segment i; the kernel's last mark flushes the segments to g_tprof.
#ifdef AZR_TREE_PROF
__shared__ unsigned long long tp_acc[24];
__shared__ unsigned long long tp_last;
__device__ unsigned long long g_tprof[25];
__device__ unsigned long long g_tslow[25];   // the same for waves that took more than 60 us
__device__ unsigned long long g_thist[16];   // waves by total time, 10-us bins
__device__ __forceinline__ void TP_BEGIN() { if (threadIdx.x == 0) { for (int i = 0; i < 24; i++) tp_acc[i] = 0; tp_last = wall_clock64(); } }
__device__ __forceinline__ void TP(int i) { if (threadIdx.x == 0) { const unsigned long long n = wall_clock64(); tp_acc[i] += n - tp_last; tp_last = n; } }
__device__ __forceinline__ void TP_END()
{
    if (threadIdx.x == 0) {
        unsigned long long tot = 0;
        for (int i = 0; i < 24; i++) { tot += tp_acc[i]; if (tp_acc[i]) atomicAdd(&g_tprof[i], tp_acc[i]); }
        atomicAdd(&g_tprof[24], 1ull);
        if (tot > 6000) { for (int i = 0; i < 24; i++) if (tp_acc[i]) atomicAdd(&g_tslow[i], tp_acc[i]); atomicAdd(&g_tslow[24], 1ull); }
        atomicAdd(&g_thist[tot / 1000 > 15 ? 15 : tot / 1000], 1ull);
    }
}
#else
__device__ __forceinline__ void TP_BEGIN() {}
__device__ __forceinline__ void TP(int) {}
__device__ __forceinline__ void TP_END() {}
#endif
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t rdl(uint32_t v, uint32_t l)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)rfl(l));
}
__device__ __forceinline__ float rdlf(float v, uint32_t l) { return __uint_as_float(rdl(__float_as_uint(v), l)); }
__device__ __forceinline__ uint32_t wrl(uint32_t old, uint32_t l, uint32_t val)
{
    return lane_id() == l ? val : old;  // l, val wave-uniform
}
__device__ __forceinline__ uint64_t ballot64(bool p) { return (uint64_t)__ballot(p); }
__device__ __forceinline__ uint64_t rfl64(uint64_t v)
{
    return (uint64_t)rfl((uint32_t)v) | ((uint64_t)rfl((uint32_t)(v >> 32)) << 32);
}
__device__ __forceinline__ uint64_t rdl64(uint64_t v, uint32_t l)
{
    return (uint64_t)rdl((uint32_t)v, l) | ((uint64_t)rdl((uint32_t)(v >> 32), l) << 32);
}
__device__ __forceinline__ int popc64(uint64_t x) { return __builtin_popcountll(x); }
__device__ __forceinline__ int ctz64(uint64_t x) { return __builtin_ctzll(x); }

// PER LANE: the adjacency list of land `lane`, one byte per neighbour in declaration order, the degree in byte 7 (0 for lanes >= 42).
// The depth-first walks below step through these lists one neighbour at a time: from a VGPR (v_readlane) a step is a few cycles,
// from the constant tables it was two dependent scalar loads — a fortify move cost 30 - 40 us, the slowest waves of a mid-game tree step.
__device__ __forceinline__ uint64_t lane_nbpack()
{
    const uint32_t l = lane_id();
    if (l >= LANDS) return 0;
    return *reinterpret_cast<const uint64_t*>(&c_nb[l][0]) | ((uint64_t)c_deg[l] << 56);
}

// ---- the wave-resident game ------------------------------------------------------------------------------
struct WS {
    uint32_t la;   // PER LANE: land byte for lanes < 42; 0xC0 (owner 3 = nobody) for lanes >= 42
    uint64_t nbm;  // PER LANE: neighbour mask of land `lane` (0 for lanes >= 42)
    uint64_t pk;   // PER LANE: neighbour LIST of land `lane` (lane_nbpack: one byte per neighbour in declaration order, degree in byte 7)
    // wave-uniform
    uint32_t cur, card_sets, reinf, phase, mob_from, mob_to, allow_draw, attacks, round, cards0, cards1;
    uint32_t rng;  // minstd_rand0 engine state of this game's stream
    uint32_t err;  // first rules error (E_*)
};

__device__ __forceinline__ uint64_t nbmask_of(uint64_t pk)
{
    uint64_t m = 0;
    const uint32_t d = (uint32_t)(pk >> 56);
#pragma unroll
    for (uint32_t i = 0; i < 6; i++) m |= i < d ? 1ULL << ((pk >> (8u * i)) & 0xffu) : 0ULL;
    return m;
}

__device__ __forceinline__ uint32_t w_army(const WS& s) { return s.la & 63u; }
__device__ __forceinline__ uint32_t w_owner(const WS& s) { return s.la >> 6; }
__device__ __forceinline__ uint32_t land_army(const WS& s, uint32_t land) { return rdl(s.la, land) & 63u; }
__device__ __forceinline__ uint32_t land_owner(const WS& s, uint32_t land) { return rdl(s.la, land) >> 6; }

__device__ __forceinline__ uint64_t m_owned(const WS& s, uint32_t p) { return ballot64(w_owner(s) == p); }
__device__ __forceinline__ uint64_t m_owned_army(const WS& s, uint32_t p)
{
    return ballot64(w_owner(s) == p && w_army(s) > 1);
}
__device__ __forceinline__ uint64_t m_owned_full(const WS& s, uint32_t p)
{
    return ballot64(w_owner(s) == p && w_army(s) == ARMY_MAX);
}
// lands not owned by p with an owned (resp. owned-with-army) neighbour
__device__ __forceinline__ uint64_t m_attack(const WS& s, uint32_t p)
{
    uint64_t o = m_owned(s, p);
    return ballot64(lane_id() < LANDS && w_owner(s) != p && (s.nbm & o) != 0);
}
__device__ __forceinline__ uint64_t m_attack_army(const WS& s, uint32_t p)
{
    uint64_t oa = m_owned_army(s, p);
    return ballot64(lane_id() < LANDS && w_owner(s) != p && (s.nbm & oa) != 0);
}
__device__ __forceinline__ int total_army(const WS& s, uint32_t p)
{
    bool mine = w_owner(s) == p;
    int t = 0;
#pragma unroll
    for (int b = 0; b < 6; b++) t += popc64(ballot64(mine && ((s.la >> b) & 1u))) << b;
    return t;
}
__device__ __forceinline__ uint32_t cards_of(const WS& s, uint32_t p) { return p == 0 ? s.cards0 : s.cards1; }

__device__ __forceinline__ void ws_blank(WS& s)  // `State()` (state/state.h:86-105 default member initialisers)
{
    s.la = lane_id() < LANDS ? 0x80u : 0xC0u;
    s.pk = lane_nbpack();
    s.nbm = nbmask_of(s.pk);
    s.cur = 0; s.card_sets = 0; s.reinf = 0; s.phase = PH_SETUP; s.mob_from = NONE; s.mob_to = NONE;
    s.allow_draw = 0; s.attacks = 0; s.round = 1; s.cards0 = 0; s.cards1 = 0;
    s.err = 0;
}

// ---- 64-byte record <-> wave ------------------------------------------------------------------------------
__device__ __forceinline__ void ws_load(WS& s, const uint8_t* rec)
{
    uint32_t l = lane_id();
    uint32_t b = rec[l];
    s.la = l < LANDS ? b : 0xC0u;
    s.pk = lane_nbpack();
    s.nbm = nbmask_of(s.pk);
    s.cur = rdl(b, GR_CUR); s.card_sets = rdl(b, GR_CARD_SETS); s.reinf = rdl(b, GR_REINF);
    s.phase = rdl(b, GR_PHASE); s.mob_from = rdl(b, GR_MOB_FROM); s.mob_to = rdl(b, GR_MOB_TO);
    s.allow_draw = rdl(b, GR_ALLOW_DRAW); s.attacks = rdl(b, GR_ATTACKS);
    s.round = rdl(b, GR_ROUND_LO) | (rdl(b, GR_ROUND_HI) << 8);
    s.cards0 = rdl(b, GR_CARDS0); s.cards1 = rdl(b, GR_CARDS1);
    s.err = 0;
}

// the byte lane `l` contributes to the 64-byte record
__device__ __forceinline__ uint32_t ws_record_byte(const WS& s)
{
    uint32_t l = lane_id();
    uint32_t b = l < LANDS ? s.la : 0u;
    b = l == GR_CUR ? s.cur : b;
    b = l == GR_CARD_SETS ? s.card_sets : b;
    b = l == GR_REINF ? s.reinf : b;
    b = l == GR_PHASE ? s.phase : b;
    b = l == GR_MOB_FROM ? s.mob_from : b;
    b = l == GR_MOB_TO ? s.mob_to : b;
    b = l == GR_ALLOW_DRAW ? s.allow_draw : b;
    b = l == GR_ATTACKS ? s.attacks : b;
    b = l == GR_ROUND_LO ? (s.round & 0xffu) : b;
    b = l == GR_ROUND_HI ? (s.round >> 8) : b;
    b = l == GR_CARDS0 ? s.cards0 : b;
    b = l == GR_CARDS1 ? s.cards1 : b;
    return b & 0xffu;
}
__device__ __forceinline__ void ws_store(const WS& s, uint8_t* rec) { rec[lane_id()] = (uint8_t)ws_record_byte(s); }

// record as 16 dwords: lane l gets dword (l & 15) of the 64-byte record (for hashing / key compares)
__device__ __forceinline__ uint32_t ws_record_dword(const WS& s)
{
    const uint32_t b = ws_record_byte(s);
    const uint32_t base = (lane_id() & 15u) * 4u;
    // gather the 4 bytes held by lanes base..base+3 (ds_bpermute takes a byte address = lane * 4)
    uint32_t b0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((base + 0) << 2), (int)b);
    uint32_t b1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((base + 1) << 2), (int)b);
    uint32_t b2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((base + 2) << 2), (int)b);
    uint32_t b3 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((base + 3) << 2), (int)b);
    return b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
}

// ---- RNG: minstd_rand0 + libstdc++ 11 distributions (src/rng.h:5-50) -----------------------------
constexpr uint32_t RNG_M = 2147483647u, RNG_RANGE = 2147483645u;
__device__ __forceinline__ uint32_t rng_seed(uint32_t seed)
{
    uint32_t s = seed % RNG_M;
    return s == 0 ? 1u : s;
}
__device__ __forceinline__ uint32_t rng_next(WS& s)
{
    // x * 16807 mod (2^31 - 1) by the Mersenne fold (exact)
    uint64_t p = (uint64_t)s.rng * 16807ull;
    uint32_t r = (uint32_t)(p & RNG_M) + (uint32_t)(p >> 31);
    r = r >= RNG_M ? r - RNG_M : r;
    s.rng = r;
    return r;
}
// uniform_int_distribution's two-division fallback path for [0, urange], urange < engine range
__device__ __forceinline__ uint32_t rng_downscale(WS& s, uint32_t urange)
{
    const uint32_t uerange = urange + 1;
    const uint32_t scaling = RNG_RANGE / uerange;
    const uint32_t past = uerange * scaling;
    uint32_t ret;
    do {
        ret = rng_next(s) - 1u;
    } while (ret >= past);
    return ret / scaling;
}
__device__ __forceinline__ uint32_t rng_dice(WS& s) { return rng_downscale(s, 5) + 1; }  // rDice
__device__ __forceinline__ uint32_t rng_int(WS& s)  // rInt: uniform_int_distribution<int>(0, RAND_MAX), up-scaling
{
    const uint64_t urange = 2147483647ull, uerng = (uint64_t)RNG_RANGE + 1;
    uint64_t tmp, ret;
    do {
        tmp = uerng * rng_downscale(s, (uint32_t)(urange / uerng));
        ret = tmp + (uint64_t)(rng_next(s) - 1u);
    } while (ret > urange || ret < tmp);
    return (uint32_t)ret;
}
__device__ __forceinline__ float rng_float(WS& s)  // rFloat: generate_canonical<float,24>, k = 1
{
    float r = (float)(rng_next(s) - 1u) / 2147483648.0f;  // float(2147483646.0L) == 2^31
    return r >= 1.0f ? 0.99999994f : r;                   // nextafterf(1, 0)
}
// Utility::randomMask (land/land.cpp:100-112)
__device__ __forceinline__ uint64_t rng_random_mask(WS& s, uint64_t masks)
{
    int count = popc64(masks);
    int rindex = (int)(rng_int(s) % (uint32_t)count);
    uint64_t mask = 1ULL << ctz64(masks);
    for (int i = 0; i < rindex; i++) {
        masks &= ~mask;
        mask = 1ULL << ctz64(masks);
    }
    return mask;
}

// ---- rules --------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void set_land(WS& s, uint32_t land, uint32_t value, uint32_t owner)
{
    // State::setLandArmy (state/state.cpp:279-385); the mask maintenance there is implicit here
    if (lane_id() == land) s.la = (value & 63u) | (owner << 6);
}

// State::calculateReinforcementValue (state/state.cpp:457-491)
__device__ __forceinline__ int reinforcement_value(uint64_t owned)
{
    int count = popc64(owned) / 3;
#pragma unroll
    for (int c = 0; c < 6; c++) {
        uint64_t cm = c_cont_mask[c];
        if ((owned & cm) == cm) count += c_cont_bonus[c];
    }
    return count < 3 ? 3 : count;
}

__device__ __forceinline__ void goto_fortify(WS& s)  // state/state.cpp:42-49
{
    if (s.phase != PH_ATTACK) { s.err = E_INVALID_ARGUMENT; return; }
    s.phase = PH_FORTIFY;
}
__device__ __forceinline__ void goto_attack(WS& s)  // state/state.cpp:20-40
{
    if (s.phase != PH_REINFORCEMENT && s.phase != PH_ATTACK_MOBILIZATION) { s.err = E_INVALID_ARGUMENT; return; }
    s.phase = PH_ATTACK;
    s.mob_from = NONE;
    s.mob_to = NONE;
    s.reinf = 0;
    if (m_attack_army(s, s.cur) == 0) goto_fortify(s);
}
__device__ __forceinline__ void next_player_setup_turn(WS& s)  // state/state.cpp:725-746
{
    s.phase = PH_SETUP;
    s.round = (s.round + 1) & 0xffffu;
    s.cur ^= 1u;
    if (s.reinf == 0) {
        s.phase = PH_REINFORCEMENT;
        s.reinf = (uint32_t)reinforcement_value(m_owned(s, s.cur)) & 0xffu;
    }
}
__device__ __forceinline__ void next_player_game_turn(WS& s)  // state/state.cpp:748-766 (+ drawCard :618-626)
{
    if (s.allow_draw) {
        if (s.cur == 0) s.cards0 = (s.cards0 + 1) & 0xffu; else s.cards1 = (s.cards1 + 1) & 0xffu;
        s.allow_draw = 0;
    }
    s.round = (s.round + 1) & 0xffffu;
    s.cur ^= 1u;
    s.attacks = 0;
    s.phase = PH_REINFORCEMENT;
    s.reinf = (uint32_t)reinforcement_value(m_owned(s, s.cur)) & 0xffu;
}
// GameHelper::playCards -> State::playCards, STATE_SIMPLE_CARDS (game_helper.cpp:3-17, state/state.cpp:1091-1117)
__device__ __forceinline__ void play_cards(WS& s)
{
    uint32_t c = cards_of(s, s.cur);
    if (c >= 3) {
        c -= 3;
        if (s.cur == 0) s.cards0 = c; else s.cards1 = c;
        s.card_sets = (s.card_sets + 1) & 0xffu;
        uint32_t k = s.card_sets, gained;
        if (k >= 1 && k <= 5) gained = 2 + 2 * k;       // 4,6,8,10,12
        else if (k == 6) gained = 15;
        else gained = (uint32_t)(15 + ((int)k - 6) * 5) & 0xffffu;  // `default:` (k == 0 after u8 wrap included)
        s.reinf = (s.reinf + gained) & 0xffu;
    }
}

// State::getDiceRolls (state/state.cpp:645-684): descending sort network as written
__device__ __forceinline__ void dice_rolls(WS& s, int n, uint32_t& r1, uint32_t& r2, uint32_t& r3)
{
    r1 = r2 = r3 = 0;
    if (n > 0) r1 = rng_dice(s);
    if (n > 1) {
        r2 = rng_dice(s);
        if (r1 < r2) { uint32_t t = r2; r2 = r1; r1 = t; }
    }
    if (n > 2) {
        r3 = rng_dice(s);
        if (r1 < r3) { uint32_t t = r3; r3 = r2; r2 = r1; r1 = t; }
        else if (r2 < r3) { uint32_t t = r3; r3 = r2; r2 = t; }
    }
}

// State::attackMove (state/state.cpp:769-918)
__device__ __forceinline__ void attack_move(WS& s, uint32_t from, uint32_t to)
{
    s.attacks = (s.attacks + 1) & 0xffu;
    if (s.phase != PH_ATTACK || from == NONE || to == NONE) { s.err = E_INVALID_ARGUMENT; return; }
    uint32_t fa = rdl(s.la, from), ta = rdl(s.la, to);
    uint32_t a_army = fa & 63u, d_army = ta & 63u, attacker = fa >> 6, defender = ta >> 6;
    if (attacker != s.cur || attacker == defender || a_army <= 1) { s.err = E_INVALID_ARGUMENT; return; }
    int units = 1;
    uint32_t attack_amount = a_army, defend_amount = d_army;
    if (d_army > 0) {
        int an = attack_amount >= 4 ? 3 : attack_amount == 3 ? 2 : 1;
        units = an;
        int dn = defend_amount >= 2 ? 2 : 1;
        uint32_t a1, a2, a3, d1, d2, d3;
        dice_rolls(s, an, a1, a2, a3);  // attacker's dice first
        dice_rolls(s, dn, d1, d2, d3);
        if (a1 > d1) defend_amount--; else { attack_amount--; units--; }
        if (an >= 2 && dn == 2) {
            if (a2 > d2) defend_amount--; else { attack_amount--; units--; }
        }
    }
    if (defend_amount == 0) {
        attack_amount = (attack_amount - (uint32_t)units) & 0xffu;
        if (attack_amount > 1) {
            s.phase = PH_ATTACK_MOBILIZATION;
            s.mob_from = from;
            s.mob_to = to;
        }
        s.allow_draw = 1;
        set_land(s, from, attack_amount, attacker);
        set_land(s, to, (uint32_t)units, attacker);
    } else {
        set_land(s, from, attack_amount, attacker);
        set_land(s, to, defend_amount, defender);
    }
    if (s.phase == PH_ATTACK && m_attack_army(s, s.cur) == 0) goto_fortify(s);
}

// State::getNeutralPlayerAttackLands (state/state.cpp:1067-1083); adjacency is symmetric, so "in the union of
// the neutral lands' neighbour masks" == "has a neutral neighbour"
__device__ __forceinline__ uint64_t neutral_attack_lands(const WS& s)
{
    uint64_t neutral = ALL_LANDS & ~m_owned(s, 0) & ~m_owned(s, 1);
    return ballot64(lane_id() < LANDS && (s.nbm & neutral) != 0) & ~neutral;
}

// UtilityNN::getValidMoves (player/alpha_zero/alphazero_moves.cpp:3-70)
__device__ __forceinline__ uint64_t valid_moves(const WS& s, const Rules& R)
{
    const uint32_t p = s.cur, e = s.cur ^ 1u;
    switch (s.phase) {
    case PH_SETUP:
    case PH_REINFORCEMENT: {
        uint64_t owned = m_owned(s, p) & ~m_owned_full(s, p);
        if (owned == 0) return SKIP_MASK;
        if (R.limit_reinforcement) {
            uint64_t nb = owned & (m_attack(s, e) | neutral_attack_lands(s));
            return nb != 0 ? nb : owned;
        }
        return owned;
    }
    case PH_SETUP_NEUTRAL:
        return ALL_LANDS & ~m_owned(s, p) & ~m_owned(s, e);
    case PH_ATTACK: {
        uint64_t aa = m_attack_army(s, p);
        if (R.limit_attack) return aa != 0 ? aa : SKIP_MASK;
        return aa | SKIP_MASK;
    }
    case PH_ATTACK_MOBILIZATION:
        return (1ULL << s.mob_from) | (1ULL << s.mob_to);
    case PH_FORTIFY:
        if (R.limit_reinforcement) return (m_owned(s, p) & m_attack(s, e)) | SKIP_MASK;  // `a & b | SKIP`
        return m_owned(s, p) | SKIP_MASK;
    default:
        return 0;
    }
}

// State::gameStatus (state/state.cpp:518-565)
__device__ __forceinline__ int game_status(const WS& s, const Rules& R)
{
    int p0 = popc64(m_owned(s, 0));
    if (p0 == 0) return 1;
    int p1 = popc64(m_owned(s, 1));
    if (p1 == 0) return 0;
    if (R.allow_yield) {
        if (p0 >= 30) return 0;
        else if (p1 >= 30) return 1;
    }
    if ((int)s.round > R.max_game_rounds) {
        if (p0 > p1) return 0;
        else if (p0 < p1) return 1;
        else return ST_DRAW;
    }
    return ST_NOT_ENDED;
}

// FORTIFY source choice (alphazero_moves.cpp:176-226 with GameHelper::PlayerMovement / LandSetMovement::add,
// game_helper.cpp:51-109): pre-order flood of the target's owned component in neighbour-list order from the
// component's lowest-index land; among lands != target take max (army-1) > 0, lands with ALL neighbours owned
// first, first-in-pre-order wins ties.  Wave-uniform scalar DFS; the explicit stack lives in the lanes of one
// VGPR (v_writelane / v_readlane).
__device__ __forceinline__ void fortify_pick(const WS& s, uint32_t target, uint32_t& from_out, uint32_t& amount_out)
{
    const uint64_t owned = m_owned(s, s.cur);
    // component of target by mask flooding (ballots), to find its lowest-index land
    uint64_t comp = 1ULL << target;
    for (;;) {
        uint64_t grow = comp | (owned & ballot64((s.nbm & comp) != 0));
        if (grow == comp) break;
        comp = grow;
    }
    // The walk below only decides TIES: a source is the candidate of its class (all neighbours owned / some not) with the largest
    // movable army, the first such in pre-order.  A unique maximum needs no order: found lane-parallel, one ballot per value bit.
    {
        const uint32_t l = lane_id();
        const uint32_t value = ((s.la & 63u) - 1u) & 0xffu;
        const bool cand = ((comp >> l) & 1ULL) && l != target && value > 0;
        const bool inner = (s.nbm & owned) == s.nbm;
        uint64_t ci = ballot64(cand && inner), cb = ballot64(cand && !inner);
#pragma unroll
        for (int b = 7; b >= 0; b--) {
            const uint64_t hi = ballot64((value >> b) & 1u);
            ci = (ci & hi) ? (ci & hi) : ci;
            cb = (cb & hi) ? (cb & hi) : cb;
        }
        const uint64_t pick = ci ? ci : cb;
        if ((pick & (pick - 1)) == 0) {   // none, or exactly one
            from_out = pick ? (uint32_t)ctz64(pick) : NONE;
            amount_out = pick ? ((rdl(s.la, from_out) & 63u) - 1u) & 0xffu : 0u;
            return;
        }
    }
    const uint32_t start = (uint32_t)ctz64(comp);
    uint32_t best_nn = 0, best = 0, from_nn = NONE, from = NONE;
    uint64_t visited = 1ULL << start;
    const uint64_t pk = s.pk;
    uint32_t stk = 0;      // lane i = stack entry i: land | next_neighbour_index << 8
    int sp = 0;
    stk = wrl(stk, 0, start);
    // visit(start)
    uint32_t v_land = start;
    for (;;) {
        // "emit" v_land: evaluate as a source candidate in pre-order
        if (v_land != target) {
            uint32_t value = ((rdl(s.la, v_land) & 63u) - 1u) & 0xffu;
            uint64_t nm = rdl64(s.nbm, v_land);
            if ((nm & owned) == nm) {
                if (value > best_nn) { best_nn = value; from_nn = v_land; }
            } else {
                if (value > best) { best = value; from = v_land; }
            }
        }
        // advance DFS to the next unvisited owned neighbour in list order
        bool found = false;
        while (sp >= 0) {
            uint32_t e = rdl(stk, (uint32_t)sp);
            uint32_t l = e & 0xffu, i = e >> 8;
            const uint64_t row = rdl64(pk, l);
            if (i >= (uint32_t)(row >> 56)) { sp--; continue; }
            stk = wrl(stk, (uint32_t)sp, l | ((i + 1) << 8));
            uint32_t n = (uint32_t)(row >> (8u * i)) & 0xffu;
            uint64_t nbit = 1ULL << n;
            if ((owned & nbit) && !(visited & nbit)) {
                visited |= nbit;
                sp++;
                stk = wrl(stk, (uint32_t)sp, n);
                v_land = n;
                found = true;
                break;
            }
        }
        if (!found) break;
    }
    if (from_nn != NONE) { from = from_nn; best = best_nn; }
    from_out = from;
    amount_out = best;
}

// UtilityNN::makeMove (player/alpha_zero/alphazero_moves.cpp:72-233).  On a rules error s.err is set and the
// caller discards the wave copy (the reference throws).
__device__ __forceinline__ void make_move(WS& s, uint32_t li, const Rules& R)
{
    if (li == NONE) { s.err = E_INVALID_ARGUMENT; return; }
    if (li == SKIP) {
        switch (s.phase) {
        case PH_REINFORCEMENT: goto_attack(s); return;
        case PH_ATTACK: goto_fortify(s); return;
        case PH_FORTIFY: next_player_game_turn(s); return;
        default: s.err = E_LOGIC; return;
        }
    }
    if (li > 41) { s.err = E_LOGIC; return; }
    const uint32_t p = s.cur;
    const uint32_t tb = rdl(s.la, li);
    const uint32_t t_army = tb & 63u, t_owner = tb >> 6;
    if (s.phase == PH_SETUP) {  // State::setupReinforcementMove (state/state.cpp:1009-1030)
        if (s.reinf == 0) { s.err = E_INVALID_ARGUMENT; return; }
        s.reinf = (s.reinf - 2) & 0xffu;
        if (t_owner != p) { s.err = E_INVALID_ARGUMENT; return; }
        if (t_army + 2 > ARMY_MAX) { s.err = E_LOGIC; return; }  // addLandArmy (state/state.cpp:241-256)
        set_land(s, li, t_army + 2, p);
        s.phase = PH_SETUP_NEUTRAL;
    } else if (s.phase == PH_SETUP_NEUTRAL) {  // State::setupReinforcementNeutralMove (state/state.cpp:1032-1053)
        if (t_owner != NEUTRAL) { s.err = E_INVALID_ARGUMENT; return; }
        set_land(s, li, t_army + 1, NEUTRAL);
        next_player_setup_turn(s);
    } else if (s.phase == PH_REINFORCEMENT) {  // alphazero_moves.cpp:104-121 + State::reinforcementMove (:976-998)
        play_cards(s);
        uint32_t amount = s.reinf / 2;
        if ((int)amount < R.min_unit_move) amount = (uint32_t)(R.min_unit_move < (int)s.reinf ? R.min_unit_move : (int)s.reinf) & 0xffu;
        uint32_t space = (uint32_t)(ARMY_MAX - (int)t_army) & 0xffu;
        amount = space < amount ? space : amount;
        if (s.reinf < amount) { s.err = E_INVALID_ARGUMENT; return; }
        s.reinf -= amount;
        if (t_army > 0 && t_owner != p) { s.err = E_LOGIC; return; }
        if (t_army + amount > ARMY_MAX) { s.err = E_LOGIC; return; }
        set_land(s, li, t_army + amount, p);
        if (s.reinf == 0) goto_attack(s);
    } else if (s.phase == PH_ATTACK) {  // alphazero_moves.cpp:122-144
        // the owned neighbour with the largest army to spare, the first such in the target's neighbour list: lane-parallel maximum
        // (one ballot per value bit); the list (s.pk, a register) is walked only when the maximum is tied
        uint32_t best_from = NONE;
        {
            const uint32_t l = lane_id();
            const uint64_t nbt = rdl64(s.nbm, li);
            const uint32_t value = ((s.la & 63u) - 1u) & 0xffu;
            uint64_t cs = ballot64(((nbt >> l) & 1ULL) && w_owner(s) == p && w_army(s) > 1);   // (army > 1 => value > 0)
#pragma unroll
            for (int b = 7; b >= 0; b--) {
                const uint64_t hi = ballot64((value >> b) & 1u);
                cs = (cs & hi) ? (cs & hi) : cs;
            }
            if (cs & (cs - 1)) {
                const uint64_t row = rdl64(s.pk, li);
                const uint32_t d = (uint32_t)(row >> 56);
                for (uint32_t i = 0; i < d; i++) {
                    const uint32_t nl = (uint32_t)(row >> (8u * i)) & 0xffu;
                    if ((cs >> nl) & 1ULL) { best_from = nl; break; }
                }
            } else if (cs) best_from = (uint32_t)ctz64(cs);
        }
        attack_move(s, best_from, li);
    } else if (s.phase == PH_ATTACK_MOBILIZATION) {  // alphazero_moves.cpp:145-171 + attackReinforcementMove (:920-947)
        if (li == s.mob_from) goto_attack(s);
        else if (li == s.mob_to) {
            uint32_t from_army = land_army(s, s.mob_from);
            uint32_t value = (from_army - 1u) & 0xffu;
            uint32_t amount = value / 2;
            if ((int)amount < R.min_unit_move) amount = (uint32_t)(R.min_unit_move < (int)value ? R.min_unit_move : (int)value) & 0xffu;
            uint32_t after = (from_army - amount) & 0xffu;
            if (after < 1) { s.err = E_INVALID_ARGUMENT; return; }
            uint32_t from = s.mob_from, to = s.mob_to;
            set_land(s, from, after, p);
            set_land(s, to, (t_army + amount) & 0xffu, p);
            if ((after & 63u) == 1) goto_attack(s);
        } else { s.err = E_INVALID_ARGUMENT; return; }
    } else if (s.phase == PH_FORTIFY) {  // alphazero_moves.cpp:172-231 + State::fortifyMove (:949-974)
        if (t_army != ARMY_MAX && t_owner == p) {
            uint32_t from, best;
            fortify_pick(s, li, from, best);
            if (from != NONE) {
                uint32_t space = (uint32_t)(ARMY_MAX - (int)t_army) & 0xffu;
                uint32_t amount = space < best ? space : best;
                uint32_t from_army = land_army(s, from);
                uint32_t after_from = (from_army - amount) & 0xffu;
                if (after_from < 1) { s.err = E_INVALID_ARGUMENT; return; }
                if (t_army + amount > ARMY_MAX) { s.err = E_INVALID_ARGUMENT; return; }
                set_land(s, from, after_from, p);
                set_land(s, li, t_army + amount, p);
            }
        }
        next_player_game_turn(s);
    } else {
        s.err = E_LOGIC;
    }
}

// State::newGame (state/state.cpp:137-167)
__device__ __forceinline__ void new_game(WS& s)
{
    uint32_t keep_rng = s.rng;
    ws_blank(s);
    s.rng = keep_rng;
    uint64_t avail = ALL_LANDS;
    while (avail != 0) {
        uint64_t m = rng_random_mask(s, avail);
        avail &= ~m;
        set_land(s, (uint32_t)ctz64(m), 1, s.cur);
        if (s.cur == 1) {
            m = rng_random_mask(s, avail);
            avail &= ~m;
            set_land(s, (uint32_t)ctz64(m), 1, NEUTRAL);
        }
        s.cur ^= 1u;
    }
    s.reinf = (40 - 14) * 2;
}

// NNInputData(const State&) (neural_network/alphazero_nn_data.cpp:165-196) -> the reference's 88-byte image
__device__ __forceinline__ void encode88(const WS& s, uint8_t* in88)
{
    const uint32_t l = lane_id();
    const uint32_t p = s.cur, e = s.cur ^ 1u;
    float ref = (float)reinforcement_value(m_owned(s, p));
    float eref = (float)reinforcement_value(m_owned(s, e));
    float ta = (float)total_army(s, p), eta = (float)total_army(s, e);
    float af = (float)s.attacks / 8.0f;
    // bytes 0..47
    uint32_t b = l < LANDS ? s.la : 0u;
    b = l == 42 ? s.cur : b;
    b = l == 44 ? (s.round & 0xffu) : b;
    b = l == 45 ? (s.round >> 8) : b;
    if (l < 48) in88[l] = (uint8_t)b;
    // floats 48..87
    float f = 0.0f;
    f = l == 0 ? ref / (ref + eref) : f;
    f = l == 1 ? (af < 1.0f ? af : 1.0f) : f;
    f = l == 2 ? (s.allow_draw ? 1.0f : 0.0f) : f;
    f = (l >= 3 && l < 9) ? (s.phase == l - 3 ? 1.0f : 0.0f) : f;
    f = l == 9 ? ta / (ta + eta) : f;
    if (l < 10) reinterpret_cast<float*>(in88 + 48)[l] = f;
}

}  // namespace azr
