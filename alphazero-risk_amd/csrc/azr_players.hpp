// azr_players.hpp — the reference's scripted opponents as wave-resident device code:
// ScriptPlayer (player/script/script_player.cpp) and RandomPlayer (player/random/random_player.cpp), driven directly
// on the wave-resident game state of azr_wave.hpp so that an arena (AlphaZero vs Script / Random, game/game.cpp) runs
// entirely on the device, one wavefront per game.  Control flow is wave-uniform; the member variables the reference's
// ScriptPlayer object carries across turns and games live in `ScriptW`.
#pragma once
#include "azr_wave.hpp"

namespace azr {

// LandSet constructor orders (land/land_set.cpp:12-33) in ScriptPlayer()'s priority-vector order
// ASIA, NORTH_AMERICA, SOUTH_AMERICA, EUROPE, AFRICA, AUSTRALIA (script_player.cpp:12-14)
static __constant__ uint8_t c_set_lands[6][12] = {
    {26, 33, 35, 36, 27, 28, 29, 30, 31, 32, 34, 37}, {0, 1, 2, 3, 4, 5, 6, 7, 8}, {9, 10, 11, 12},
    {13, 14, 15, 16, 17, 19, 18},                      {20, 21, 22, 24, 25, 23},    {38, 39, 40, 41}};
static __constant__ uint8_t c_set_count[6] = {12, 9, 4, 7, 6, 4};
static __constant__ uint64_t c_set_mask[6] = {0x3ffc000000ULL, 0x1ffULL, 0x1e00ULL, 0xfe000ULL, 0x3f00000ULL,
                                              0x3c000000000ULL};

struct ScriptW {  // 32 bytes in HBM per (game slot, player)
    uint32_t order;           // attackLandSetPriority: 6 x 4-bit set ids, position i at bits 4i
    uint32_t attacking_set;   // 7 = none yet
    uint32_t land_to, land_from, attack_from_army;
    uint32_t pad;
    uint64_t owned_attack_mask_unused;  // kept for layout; masks are re-derived at the start of every turn
};
static_assert(sizeof(ScriptW) == 32, "ScriptW layout");

__device__ __forceinline__ void script_init(ScriptW& p)
{
    p.order = 0x543210u;
    p.attacking_set = 7;
    p.land_to = NONE; p.land_from = NONE; p.attack_from_army = 0; p.pad = 0; p.owned_attack_mask_unused = 0;
}

// ---- State:: move primitives the players call directly (state/state.cpp) ---------------------------------------
__device__ __forceinline__ void reinforcement_move(WS& s, uint32_t amount, uint32_t to)  // :976-998 (+ addLandArmy :241-256)
{
    if (s.phase != PH_REINFORCEMENT || s.reinf < amount) { s.err = E_INVALID_ARGUMENT; return; }
    s.reinf -= amount;
    const uint32_t tb = rdl(s.la, to), army = tb & 63u, owner = tb >> 6;
    if (army > 0 && owner != s.cur) { s.err = E_LOGIC; return; }
    if (army + amount > ARMY_MAX) { s.err = E_LOGIC; return; }
    set_land(s, to, army + amount, s.cur);
    if (s.reinf == 0) goto_attack(s);
}
__device__ __forceinline__ void attack_reinforcement_move(WS& s, uint32_t amount)  // :920-947
{
    if (s.phase != PH_ATTACK_MOBILIZATION) { s.err = E_INVALID_ARGUMENT; return; }
    const uint32_t from = s.mob_from, to = s.mob_to;
    const uint32_t from_army = land_army(s, from), to_army = land_army(s, to);
    const uint32_t after = (from_army - amount) & 0xffu;
    if (after < 1) { s.err = E_INVALID_ARGUMENT; return; }
    set_land(s, from, after, s.cur);
    set_land(s, to, (to_army + amount) & 0xffu, s.cur);
    if ((after & 63u) == 1) goto_attack(s);
}
__device__ __forceinline__ void fortify_move(WS& s, uint32_t amount, uint32_t from, uint32_t to)  // :949-974
{
    if (s.phase != PH_FORTIFY) { s.err = E_INVALID_ARGUMENT; return; }
    const uint32_t from_army = land_army(s, from), to_army = land_army(s, to);
    const uint32_t after_from = (from_army - amount) & 0xffu;
    if (after_from < 1) { s.err = E_INVALID_ARGUMENT; return; }
    if (to_army + amount > ARMY_MAX) { s.err = E_INVALID_ARGUMENT; return; }
    set_land(s, from, after_from, s.cur);
    set_land(s, to, to_army + amount, s.cur);
}
__device__ __forceinline__ void setup_reinforcement_move(WS& s, uint32_t to)  // :1009-1030
{
    if (s.phase != PH_SETUP || s.reinf == 0) { s.err = E_INVALID_ARGUMENT; return; }
    s.reinf = (s.reinf - 2) & 0xffu;
    const uint32_t tb = rdl(s.la, to), army = tb & 63u, owner = tb >> 6;
    if (owner != s.cur) { s.err = E_INVALID_ARGUMENT; return; }
    if (army + 2 > ARMY_MAX) { s.err = E_LOGIC; return; }
    set_land(s, to, army + 2, s.cur);
    s.phase = PH_SETUP_NEUTRAL;
}
__device__ __forceinline__ void setup_reinforcement_neutral_move(WS& s, uint32_t to)  // :1032-1053
{
    if (s.phase != PH_SETUP_NEUTRAL) { s.err = E_INVALID_ARGUMENT; return; }
    const uint32_t tb = rdl(s.la, to);
    if ((tb >> 6) != NEUTRAL) { s.err = E_INVALID_ARGUMENT; return; }
    set_land(s, to, (tb & 63u) + 1, NEUTRAL);
    next_player_setup_turn(s);
}

// State::invertPlayers (state/state.cpp:493-516): swap the players' lands and cards
__device__ __forceinline__ void invert_players(WS& s)
{
    const uint32_t o = w_owner(s);
    if (o < 2) s.la = (s.la & 63u) | ((o ^ 1u) << 6);
    const uint32_t c0 = s.cards0;
    s.cards0 = s.cards1;
    s.cards1 = c0;
}

// ---- GameHelper::PlayerMovement (game_helper.cpp:51-109) --------------------------------------------------------
// One owned component flooded in the reference's recursive pre-order from `start`: returns its mask and, evaluated in
// pre-order with strict ">", the fortify-from land (no non-owned neighbour, largest army) and the fortify-to land
// (most non-owned neighbours).
struct CompW { uint64_t mask; uint32_t from, to, from_amount, to_nb; };

__device__ __forceinline__ CompW flood_component(const WS& s, uint64_t owned, uint32_t start)
{
    CompW c{1ULL << start, NONE, NONE, 0, 0};
    const uint64_t pk = s.pk;   // adjacency lists in a VGPR (azr_wave.hpp)
    uint32_t stk = 0;  // lane i = stack entry i: land | next_neighbour_index << 8
    int sp = 0;
    stk = wrl(stk, 0, start);
    uint32_t v = start;
    for (;;) {
        const uint64_t nm = rdl64(s.nbm, v);
        const uint64_t att = ~owned & nm;
        const uint32_t army = rdl(s.la, v) & 63u;
        if (att == 0) {
            if (army > c.from_amount) { c.from = v; c.from_amount = army; }
        } else {
            const uint32_t cnt = (uint32_t)popc64(att);
            if (cnt > c.to_nb) { c.to_nb = cnt; c.to = v; }
        }
        bool found = false;
        while (sp >= 0) {
            const uint32_t e = rdl(stk, (uint32_t)sp);
            const uint32_t l = e & 0xffu, i = e >> 8;
            const uint64_t row = rdl64(pk, l);
            if (i >= (uint32_t)(row >> 56)) { sp--; continue; }
            stk = wrl(stk, (uint32_t)sp, l | ((i + 1) << 8));
            const uint32_t n = (uint32_t)(row >> (8u * i)) & 0xffu;
            const uint64_t nbit = 1ULL << n;
            if ((owned & nbit) && !(c.mask & nbit)) {
                c.mask |= nbit;
                sp++;
                stk = wrl(stk, (uint32_t)sp, n);
                v = n;
                found = true;
                break;
            }
        }
        if (!found) break;
    }
    return c;
}

// landSetMovements[0] after the (stable, < 17 elements) sort by fortify-from amount, descending: the first component
// in discovery order with the largest amount
__device__ __forceinline__ CompW best_component(const WS& s)
{
    const uint64_t owned = m_owned(s, s.cur);
    uint64_t covered = 0;
    CompW best{0, NONE, NONE, 0, 0};
    bool have = false;
    uint64_t todo = owned;
    while (todo) {
        const uint32_t start = (uint32_t)ctz64(todo);
        CompW c = flood_component(s, owned, start);
        covered |= c.mask;
        todo &= ~covered;
        if (!have || c.from_amount > best.from_amount) { best = c; have = true; }
    }
    return best;
}

// ---- ScriptPlayer (script_player.cpp) ------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ord_get(uint32_t order, int i) { return (order >> (4 * i)) & 15u; }

// GameHelper::sortLandSet (game_helper.cpp:19-39) over the packed per-set counters
__device__ __forceinline__ bool set_before(uint64_t no, uint64_t noa, uint32_t a, uint32_t b)
{
    const uint32_t na = (uint32_t)(no >> (8 * a)) & 255u, nb = (uint32_t)(no >> (8 * b)) & 255u;
    if (na == nb) {
        const uint32_t aa = (uint32_t)(noa >> (8 * a)) & 255u, ab = (uint32_t)(noa >> (8 * b)) & 255u;
        if (aa == ab) return c_set_mask[a] > c_set_mask[b];
        return aa > ab;
    }
    return na < nb;
}

// updateAttackLandSetPriority + updateAttackLandSet + updateAttackLandTo + updateAttackLandFrom (:17-80)
__device__ __forceinline__ void script_update(ScriptW& p, const WS& s, uint64_t owned_attack_mask, uint64_t attack_mask)
{
    const uint64_t owned = m_owned(s, s.cur);
    uint64_t no = 0, noa = 0;
    for (int k = 0; k < 6; k++) {
        const uint64_t m = c_set_mask[k] & ~owned;
        no |= (uint64_t)popc64(m) << (8 * k);
        noa |= (uint64_t)popc64(m & attack_mask) << (8 * k);
    }
    uint32_t order = p.order;
    for (int i = 1; i < 6; i++) {  // insertion sort == std::sort for 6 elements; the order is total
        const uint32_t v = ord_get(order, i);
        int j = i;
        while (j > 0 && set_before(no, noa, v, ord_get(order, j - 1))) {
            order = (order & ~(15u << (4 * j))) | (ord_get(order, j - 1) << (4 * j));
            j--;
        }
        order = (order & ~(15u << (4 * j))) | (v << (4 * j));
    }
    p.order = order;
    for (int i = 0; i < 6; i++) {
        const uint32_t k = ord_get(order, i);
        if (((noa >> (8 * k)) & 255u) > 0) { p.attacking_set = k; break; }
    }
    if (p.attacking_set < 6) {
        const uint32_t k = p.attacking_set;
        const int n = c_set_count[k];
        for (int i = 0; i < n; i++) {
            const uint32_t l = c_set_lands[k][i];
            if ((attack_mask >> l) & 1ULL) { p.land_to = l; break; }
        }
    }
    p.attack_from_army = 0;
    if (p.land_to != NONE) {
        const int d = c_deg[p.land_to];
        for (int i = 0; i < d; i++) {
            const uint32_t nl = c_nb[p.land_to][i];
            if ((owned_attack_mask >> nl) & 1ULL) {
                const uint32_t army = rdl(s.la, nl) & 63u;
                if (army > p.attack_from_army) { p.attack_from_army = army; p.land_from = nl; }
            }
        }
    }
}

// ScriptPlayer::attackLand (:82-135)
__device__ __forceinline__ void script_attack_land(ScriptW& p, WS& s, const Rules& R)
{
    if (p.land_from == NONE || p.land_to == NONE) { s.err = E_LOGIC; return; }
    while (s.reinf > 0) {
        const uint64_t not_full = m_owned(s, s.cur) & ~m_owned_full(s, s.cur);
        uint32_t to = p.land_from;
        if (((not_full >> p.land_from) & 1ULL) == 0) {
            uint64_t nb = rdl64(s.nbm, p.land_to) & not_full;
            if (nb == 0) nb = not_full & (m_attack(s, s.cur ^ 1u) | neutral_attack_lands(s));
            if (nb == 0) nb = not_full;
            if (nb == 0) { s.err = E_LOGIC; return; }
            to = (uint32_t)ctz64(nb);
        }
        const uint32_t space = (uint32_t)(ARMY_MAX - (int)land_army(s, to)) & 0xffu;
        uint32_t reinforcement = space < s.reinf ? space : s.reinf;
        if (reinforcement == 0) { s.err = E_LOGIC; return; }
        while (reinforcement > 0) {
            const uint32_t step = (int)reinforcement < R.min_unit_move ? reinforcement : (uint32_t)R.min_unit_move;
            reinforcement_move(s, step, to);
            if (s.err) return;
            reinforcement -= step;
        }
    }
    p.attack_from_army = land_army(s, p.land_from);
    while (p.attack_from_army > 1) {
        const uint32_t owner_before = land_owner(s, p.land_to);
        attack_move(s, p.land_from, p.land_to);
        if (s.err) return;
        const bool captured = land_owner(s, p.land_to) != owner_before;
        p.attack_from_army = land_army(s, p.land_from);
        if (captured && p.attack_from_army > 1) {
            uint32_t max_move = p.attack_from_army - 1;
            while (max_move > 0) {
                const uint32_t step = (int)max_move < R.min_unit_move ? max_move : (uint32_t)R.min_unit_move;
                max_move -= step;
                attack_reinforcement_move(s, step);
                if (s.err) return;
            }
            break;
        }
    }
}

// ScriptPlayer::takeTurn (:162-227)
__device__ __forceinline__ void script_take_turn(ScriptW& p, WS& s, const Rules& R)
{
    const uint32_t me = s.cur, en = s.cur ^ 1u;
    if (s.phase == PH_SETUP) {
        script_update(p, s, m_owned(s, me), m_attack(s, me));
        if (p.land_from == NONE) { s.err = E_LOGIC; return; }
        setup_reinforcement_move(s, p.land_from);
        if (s.err) return;
        const uint64_t neutral = ALL_LANDS & ~m_owned(s, me) & ~m_owned(s, en);
        uint64_t nte = neutral & m_attack(s, en) & ~m_attack(s, me);
        if (nte == 0) nte = neutral & m_attack(s, en);
        const uint64_t pick = nte != 0 ? rng_random_mask(s, nte) : rng_random_mask(s, neutral);
        setup_reinforcement_neutral_move(s, (uint32_t)ctz64(pick));
        return;
    }
    uint64_t owned_attack_mask = m_owned(s, me), attack_mask = m_attack(s, me);
    play_cards(s);
    while (attack_mask != 0 || s.reinf > 0) {
        script_update(p, s, owned_attack_mask, attack_mask);
        script_attack_land(p, s, R);
        if (s.err) return;
        owned_attack_mask = m_owned_army(s, me);
        attack_mask = m_attack_army(s, me);
    }
    if (m_owned_army(s, me) != 0) {  // fortify (:138-160)
        const CompW c = best_component(s);
        if (c.from_amount > 0 && c.to != NONE) {
            uint32_t amount = (land_army(s, c.from) - 1u) & 0xffu;
            const uint32_t space = (uint32_t)(ARMY_MAX - (int)land_army(s, c.to)) & 0xffu;
            amount = amount < space ? amount : space;
            fortify_move(s, amount, c.from, c.to);
            if (s.err) return;
        }
    }
    next_player_game_turn(s);
}

// ---- RandomPlayer (random_player.cpp:22-111) ----------------------------------------------------------------------
__device__ __forceinline__ void random_take_turn(WS& s, const Rules& R)
{
    const uint32_t me = s.cur, en = s.cur ^ 1u;
    while (s.err == 0 && game_status(s, R) == ST_NOT_ENDED && s.cur == me) {
        if (s.phase == PH_SETUP) {
            const uint64_t m = m_owned(s, me);
            if (m == 0) { s.err = E_INVALID_ARGUMENT; return; }
            setup_reinforcement_move(s, (uint32_t)ctz64(rng_random_mask(s, m)));
        } else if (s.phase == PH_SETUP_NEUTRAL) {
            const uint64_t m = ALL_LANDS & ~m_owned(s, me) & ~m_owned(s, en);
            if (m == 0) { s.err = E_INVALID_ARGUMENT; return; }
            setup_reinforcement_neutral_move(s, (uint32_t)ctz64(rng_random_mask(s, m)));
        } else if (s.phase == PH_REINFORCEMENT) {
            play_cards(s);
            const uint64_t m = m_owned(s, me) & ~m_owned_full(s, me);
            if (m == 0) { s.err = E_INVALID_ARGUMENT; return; }
            reinforcement_move(s, 1, (uint32_t)ctz64(rng_random_mask(s, m)));
        } else if (s.phase == PH_ATTACK) {
            const uint64_t mv = rng_random_mask(s, m_attack_army(s, me) | SKIP_MASK);
            if (mv & SKIP_MASK) goto_fortify(s);
            else {
                const uint32_t to = (uint32_t)ctz64(mv);
                const uint64_t fm = rdl64(s.nbm, to) & m_owned_army(s, me);
                if (fm == 0) { s.err = E_INVALID_ARGUMENT; return; }
                attack_move(s, (uint32_t)ctz64(rng_random_mask(s, fm)), to);
            }
        } else if (s.phase == PH_ATTACK_MOBILIZATION) {
            if (rng_float(s) > 0.5f) {
                const int v = (int)land_army(s, s.mob_from) - 1;
                attack_reinforcement_move(s, (uint32_t)(v < R.min_unit_move ? v : R.min_unit_move) & 0xffu);
            } else goto_attack(s);
        } else if (s.phase == PH_FORTIFY) {
            const uint64_t mv = rng_random_mask(s, (m_owned(s, me) & ~m_owned_full(s, me)) | SKIP_MASK);
            if (mv != SKIP_MASK) {
                const uint32_t to = (uint32_t)ctz64(mv);
                // component of `to` (order-free: mask flooding by ballots)
                const uint64_t owned = m_owned(s, me);
                uint64_t comp = mv;
                for (;;) {
                    const uint64_t grow = comp | (owned & ballot64((s.nbm & comp) != 0));
                    if (grow == comp) break;
                    comp = grow;
                }
                const uint64_t with_army = comp & ~mv & m_owned_army(s, me);
                if (with_army != 0) {
                    const uint32_t from = (uint32_t)ctz64(rng_random_mask(s, with_army));
                    uint32_t amount = (land_army(s, from) - 1u) & 0xffu;
                    const uint32_t space = (uint32_t)(ARMY_MAX - (int)land_army(s, to)) & 0xffu;
                    amount = space < amount ? space : amount;
                    if (amount == 0) { s.err = E_LOGIC; return; }
                    const uint32_t ra = rng_int(s) % amount;
                    fortify_move(s, ra, from, to);
                    if (s.err) return;
                }
            }
            next_player_game_turn(s);
        }
    }
}

}  // namespace azr
