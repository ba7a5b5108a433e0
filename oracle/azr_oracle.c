/* TEST INFRASTRUCTURE — CPU oracle for the AlphaZero-Risk hot path.  NOT product code.
 * See azr_oracle.h for scope and pinning status.  Citations are relative to /root/reference/.
 * Plain C11, scalar, AoS, written to follow the reference's control flow line by line so that a
 * reader can diff behaviour, not to be fast.  Compile with -ffp-contract=off (oracle/Makefile). */
#define _GNU_SOURCE
#include "azr_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * settings (src/settings.h:41-64)
 * ---------------------------------------------------------------------------------------------- */
void orc_default_settings(orc_settings* s)
{
    s->allow_yield = 1;
    s->limit_reinforcement = 1;
    s->limit_attack = 0;
    s->max_game_rounds = 30 + 28;
    s->min_unit_move = 3;
    s->mcts_simulations = 32;
    s->hp_exploration = 1.1f;
    s->dir_noise_value = 0.3f; /* `float DIR_NOISE_VALUE = 0.3;` double literal narrowed to float */
    s->dir_noise_epsi = 0.25f;
    s->temperature_threshold = 15 + 28;
    s->mcts_threads = 1;
}

/* ------------------------------------------------------------------------------------------------
 * static map tables (land/land.cpp:246-297 adjacency in declaration order; land/land_set.cpp:12-33;
 * land/land_index.h:5-10 bonuses).  Neighbour ORDER is behaviour (attack source and fortify DFS
 * tie-breaks), so it is kept exactly.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint8_t n; uint8_t nb[6]; } orc_land;
static const orc_land LAND[ORC_LANDS] = {
    {3, { 1,  3, 29, 255, 255, 255}}, /*  0 ALASKA */
    {4, { 0,  3,  4,  2, 255, 255}},  /*  1 NORTHWEST_TERRIOTRY */
    {4, { 1,  4,  5, 13, 255, 255}},  /*  2 GREENLAND */
    {4, { 0,  1,  4,  6, 255, 255}},  /*  3 ALBERTA */
    {6, { 1,  3,  6,  7,  5,  2}},    /*  4 ONTARIO */
    {3, { 4,  7,  2, 255, 255, 255}}, /*  5 QUEBEC */
    {4, { 3,  4,  7,  8, 255, 255}},  /*  6 WESTERN_UNITED_STATES */
    {4, { 8,  6,  4,  5, 255, 255}},  /*  7 EASTERN_UNITED_STATES */
    {3, { 6,  7,  9, 255, 255, 255}}, /*  8 CENTRAL_AMERICA */
    {3, { 8, 10, 11, 255, 255, 255}}, /*  9 VENEZUELA */
    {3, { 9, 11, 12, 255, 255, 255}}, /* 10 PERU */
    {4, { 9, 10, 12, 20, 255, 255}},  /* 11 BRAZIL */
    {2, {10, 11, 255, 255, 255, 255}},/* 12 ARGENTINA */
    {3, { 2, 14, 15, 255, 255, 255}}, /* 13 ICELAND */
    {4, {13, 19, 15, 17, 255, 255}},  /* 14 GREAT_BRITAIN */
    {4, {13, 14, 16, 17, 255, 255}},  /* 15 SCANDINAVIA */
    {6, {15, 17, 18, 35, 33, 26}},    /* 16 UKRAINE */
    {5, {15, 14, 18, 19, 16, 255}},   /* 17 NORTHERN_EUROPE */
    {6, {19, 17, 16, 20, 21, 35}},    /* 18 SOUTHERN_EUROPE */
    {4, {20, 14, 18, 17, 255, 255}},  /* 19 WESTERN_EUROPE */
    {6, {11, 19, 18, 21, 23, 22}},    /* 20 NORTH_AFRICA */
    {4, {18, 20, 23, 35, 255, 255}},  /* 21 EGYPT */
    {3, {20, 23, 24, 255, 255, 255}}, /* 22 CONGO */
    {6, {21, 20, 22, 24, 25, 35}},    /* 23 EAST_AFRICA */
    {3, {22, 23, 25, 255, 255, 255}}, /* 24 SOUTH_AFRICA */
    {2, {24, 23, 255, 255, 255, 255}},/* 25 MADAGASKAR */
    {4, {16, 33, 34, 27, 255, 255}},  /* 26 URAL */
    {5, {26, 34, 32, 30, 28, 255}},   /* 27 SIBERIA */
    {3, {27, 30, 29, 255, 255, 255}}, /* 28 YAKUTSK */
    {5, {28, 30, 32, 31,  0, 255}},   /* 29 KAMCHATKA */
    {4, {28, 29, 32, 27, 255, 255}},  /* 30 IRKUTSK */
    {2, {29, 32, 255, 255, 255, 255}},/* 31 JAPAN */
    {5, {27, 30, 29, 31, 34, 255}},   /* 32 MONGOLIA */
    {5, {16, 26, 34, 36, 35, 255}},   /* 33 AFGHANISTAN */
    {6, {32, 27, 26, 33, 36, 37}},    /* 34 CHINA */
    {6, {21, 23, 18, 16, 33, 36}},    /* 35 MIDDLE_EAST */
    {4, {35, 33, 34, 37, 255, 255}},  /* 36 INDIA */
    {3, {36, 34, 38, 255, 255, 255}}, /* 37 SIAM */
    {3, {37, 39, 40, 255, 255, 255}}, /* 38 INDONESIA */
    {3, {38, 41, 40, 255, 255, 255}}, /* 39 NEW_GUINEA */
    {3, {41, 39, 38, 255, 255, 255}}, /* 40 WESTERN_AUSTRALIA */
    {2, {40, 39, 255, 255, 255, 255}},/* 41 EASTERN_AUSTRALIA */
};

/* order of the tests in State::calculateReinforcementValue (state.cpp:461-483): NA, SA, AF, EU, AS, AU */
static const uint64_t CONT_MASK[6] = {
    0x1ffULL,            /* NORTH_AMERICA  0..8   */
    0x1e00ULL,           /* SOUTH_AMERICA  9..12  */
    0x3f00000ULL,        /* AFRICA        20..25  */
    0xfe000ULL,          /* EUROPE        13..19  */
    0x3ffc000000ULL,     /* ASIA          26..37  */
    0x3c000000000ULL,    /* AUSTRALIA     38..41  */
};
static const int CONT_BONUS[6] = {5, 2, 3, 5, 7, 2};
#define ALL_LANDS 0x3ffffffffffULL
#define SKIP_MASK (1ULL << ORC_SKIP)

static uint64_t nb_mask(int land)
{
    uint64_t m = 0;
    for (int i = 0; i < LAND[land].n; i++) m |= 1ULL << LAND[land].nb[i];
    return m;
}

int orc_neighbours(int land, uint8_t* out6)
{
    for (int i = 0; i < LAND[land].n; i++) out6[i] = LAND[land].nb[i];
    return LAND[land].n;
}
uint64_t orc_neighbour_mask(int land) { return nb_mask(land); }
uint64_t orc_continent_mask(int c) { return c >= 0 && c < 6 ? CONT_MASK[c] : ALL_LANDS; }

static int popc(uint64_t x) { return __builtin_popcountll(x); }
static int ctz(uint64_t x) { return __builtin_ctzll(x); }

/* ------------------------------------------------------------------------------------------------
 * RNG (src/rng.h:5-50; libstdc++ 11 <random>: minstd_rand0, uniform_int_distribution fallback
 * path of bits/uniform_int_dist.h, generate_canonical<float,24> of bits/random.tcc). SURVEY App-D.
 * ---------------------------------------------------------------------------------------------- */
#define MINSTD_M 2147483647u
#define MINSTD_A 16807u
#define URNG_MIN 1u
#define URNG_RANGE 2147483645u /* max - min */

void orc_rng_seed(orc_rng* r, uint32_t seed)
{
    uint32_t s = seed % MINSTD_M; /* linear_congruential_engine::seed, c == 0 */
    r->x = s == 0 ? 1u : s;
}

uint32_t orc_rng_next(orc_rng* r)
{
    r->x = (uint32_t)(((uint64_t)r->x * MINSTD_A) % MINSTD_M);
    return r->x;
}

/* uniform_int_distribution down-scaling branch for a range [0, urange], urange < URNG_RANGE */
static uint64_t rng_downscale(orc_rng* r, uint64_t urange)
{
    const uint64_t uerange = urange + 1;
    const uint64_t scaling = URNG_RANGE / uerange;
    const uint64_t past = uerange * scaling;
    uint64_t ret;
    do {
        ret = (uint64_t)orc_rng_next(r) - URNG_MIN;
    } while (ret >= past);
    return ret / scaling;
}

int orc_rng_dice(orc_rng* r) { return (int)rng_downscale(r, 5) + 1; } /* uniform_int_distribution<int>(1,6) */

int orc_rng_int(orc_rng* r) /* uniform_int_distribution<int>(0, RAND_MAX): up-scaling branch */
{
    const uint64_t urange = 2147483647u;
    const uint64_t uerngrange = (uint64_t)URNG_RANGE + 1;
    uint64_t tmp, ret;
    do {
        tmp = uerngrange * rng_downscale(r, urange / uerngrange);
        ret = tmp + ((uint64_t)orc_rng_next(r) - URNG_MIN);
    } while (ret > urange || ret < tmp);
    return (int)ret;
}

float orc_rng_float(orc_rng* r) /* uniform_real_distribution<float>(0,1) */
{
    /* generate_canonical<float,24>: k = 1; r = long double(max) - long double(min) + 1 = 2147483646,
     * converted to float when multiplied into __tmp (1.0f) -> 2147483648.0f */
    const float range = (float)2147483646.0L;
    float sum = (float)(orc_rng_next(r) - URNG_MIN) * 1.0f;
    float ret = sum / (1.0f * range);
    if (ret >= 1.0f) ret = nextafterf(1.0f, 0.0f);
    return ret * (1.0f - 0.0f) + 0.0f;
}

/* Utility::randomMask (land/land.cpp:100-112) */
uint64_t orc_random_mask(orc_rng* r, uint64_t masks)
{
    int count = popc(masks);
    int rindex = orc_rng_int(r) % count;
    uint64_t mask = 1ULL << ctz(masks);
    for (int i = 0; i < rindex; i++) {
        masks &= ~mask;
        mask = 1ULL << ctz(masks);
    }
    return mask;
}

/* ------------------------------------------------------------------------------------------------
 * State image (state/state.h:24-105; byte offsets probed from the compiled reference)
 * ---------------------------------------------------------------------------------------------- */
void orc_state_blank(orc_state* s)
{
    memset(s, 0, sizeof *s);
    for (int i = 0; i < ORC_LANDS; i++) s->owner[i] = ORC_NEUTRAL; /* LandArmy{army 0, playerIndex 2} */
    s->round = 1;
    s->phase = ORC_SETUP;
    s->mob_from = ORC_NONE;
    s->mob_to = ORC_NONE;
}

static void put48(uint8_t* p, uint64_t v) { for (int i = 0; i < 6; i++) p[i] = (uint8_t)(v >> (8 * i)); }
static uint64_t get48(const uint8_t* p) { uint64_t v = 0; for (int i = 0; i < 6; i++) v |= (uint64_t)p[i] << (8 * i); return v; }

void orc_state_pack(const orc_state* s, uint8_t* d)
{
    memset(d, 0, 160);
    for (int i = 0; i < ORC_LANDS; i++) d[i] = (uint8_t)((s->army[i] & 63) | (s->owner[i] << 6));
    for (int p = 0; p < 2; p++) {
        uint8_t* q = d + 48 + 48 * p;
        put48(q + 0, s->ps[p].owned);
        put48(q + 8, s->ps[p].owned_army);
        put48(q + 16, s->ps[p].owned_full);
        put48(q + 24, s->ps[p].attack);
        put48(q + 32, s->ps[p].attack_army);
        q[38] = (uint8_t)(s->ps[p].total_army & 0xff);
        q[39] = (uint8_t)((uint16_t)s->ps[p].total_army >> 8);
        q[40] = s->ps[p].cards;
    }
    d[144] = (uint8_t)(s->round & 0xff);
    d[145] = (uint8_t)(s->round >> 8);
    d[146] = (uint8_t)s->cur;
    d[147] = s->card_sets;
    d[148] = s->reinf;
    d[149] = s->phase;
    d[150] = s->mob_from;
    d[151] = s->mob_to;
    d[152] = s->allow_draw;
    d[153] = s->attacks;
    d[154] = (uint8_t)(s->drawn & 0xff);
    d[155] = (uint8_t)(s->drawn >> 8);
}

void orc_state_unpack(orc_state* s, const uint8_t* d)
{
    memset(s, 0, sizeof *s);
    for (int i = 0; i < ORC_LANDS; i++) { s->army[i] = d[i] & 63; s->owner[i] = d[i] >> 6; }
    for (int p = 0; p < 2; p++) {
        const uint8_t* q = d + 48 + 48 * p;
        s->ps[p].owned = get48(q + 0);
        s->ps[p].owned_army = get48(q + 8);
        s->ps[p].owned_full = get48(q + 16);
        s->ps[p].attack = get48(q + 24);
        s->ps[p].attack_army = get48(q + 32);
        s->ps[p].total_army = (int16_t)(q[38] | (q[39] << 8));
        s->ps[p].cards = q[40];
    }
    s->round = (uint16_t)(d[144] | (d[145] << 8));
    s->cur = (int8_t)d[146];
    s->card_sets = d[147];
    s->reinf = d[148];
    s->phase = d[149];
    s->mob_from = d[150];
    s->mob_to = d[151];
    s->allow_draw = d[152];
    s->attacks = d[153];
    s->drawn = (uint16_t)(d[154] | (d[155] << 8));
}

/* State::equalFields (state.cpp:111-135) */
int orc_state_equal(const orc_state* a, const orc_state* b)
{
    if (a->mob_from != b->mob_from) return 0;
    if (a->mob_to != b->mob_to) return 0;
    if (a->card_sets != b->card_sets) return 0;
    if (a->cur != b->cur) return 0;
    if (a->drawn != b->drawn) return 0;
    if (a->allow_draw != b->allow_draw) return 0;
    if (a->reinf != b->reinf) return 0;
    if (a->round != b->round) return 0;
    if (a->phase != b->phase) return 0;
    if (a->attacks != b->attacks) return 0;
    for (int p = 0; p < 2; p++) {
        const orc_player *x = &a->ps[p], *y = &b->ps[p];
        if (x->owned != y->owned || x->owned_army != y->owned_army || x->owned_full != y->owned_full ||
            x->attack != y->attack || x->attack_army != y->attack_army || x->total_army != y->total_army ||
            x->cards != y->cards)
            return 0;
    }
    for (int i = 0; i < ORC_LANDS; i++)
        if (a->army[i] != b->army[i] || a->owner[i] != b->owner[i]) return 0;
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * Rules (state/state.cpp)
 * ---------------------------------------------------------------------------------------------- */
#define TRY(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

static orc_player* cur_ps(orc_state* s) { return &s->ps[s->cur]; }
static const orc_player* ccur_ps(const orc_state* s) { return &s->ps[s->cur]; }
static const orc_player* cenemy_ps(const orc_state* s) { return &s->ps[s->cur == 0 ? 1 : 0]; }

/* updateAttackBitMask (state.cpp:258-272) */
static void update_attack_bitmask(orc_player* p, int land)
{
    uint64_t lm = 1ULL << land, nm = nb_mask(land);
    if ((p->owned & lm) == 0) {
        if ((nm & p->owned) > 0) p->attack |= lm;
        if ((nm & p->owned_army) > 0) p->attack_army |= lm;
    }
}

/* State::setLandArmy(idx, value, playerIndex) (state.cpp:279-385) */
static int set_land_army(orc_state* s, int land, uint8_t value, uint8_t player)
{
    if (land > 41) return ORC_LOGIC_ERROR;
    if (player > 2) return ORC_LOGIC_ERROR;
    const uint8_t old_owner_idx = s->owner[land];
    const uint8_t new_value = value, old_value = s->army[land];
    if (new_value != old_value || old_owner_idx != player) {
        orc_player* old_owner = old_owner_idx == ORC_NEUTRAL ? NULL : &s->ps[old_owner_idx];
        orc_player* new_owner = player == ORC_NEUTRAL ? NULL : &s->ps[player];
        const uint64_t lm = 1ULL << land, nm = nb_mask(land);

        if (new_owner != NULL && new_value == ORC_ARMY_MAX) new_owner->owned_full |= lm;
        else if (old_owner != NULL && old_value == ORC_ARMY_MAX) old_owner->owned_full &= ~lm;

        if (old_owner_idx == player) { /* land did not change owner */
            if (new_owner != NULL) {
                new_owner->total_army = (int16_t)(new_owner->total_army + (new_value - old_value));
                if (old_value == 1 && new_value > 1) {
                    new_owner->owned_army |= lm;
                    new_owner->attack_army |= nm;
                    new_owner->attack_army &= ~new_owner->owned;
                } else if (old_value > 1 && new_value == 1) {
                    new_owner->owned_army &= ~lm;
                    new_owner->attack_army &= ~nm;
                    for (int i = 0; i < LAND[land].n; i++) {
                        int nl = LAND[land].nb[i];
                        if ((nb_mask(nl) & new_owner->owned_army) > 0) new_owner->attack_army |= 1ULL << nl;
                    }
                    new_owner->attack_army &= ~new_owner->owned;
                }
            }
        } else { /* land changed owner */
            if (new_owner != NULL) {
                new_owner->total_army = (int16_t)(new_owner->total_army + new_value);
                new_owner->owned |= lm;
                new_owner->attack |= nm;
                new_owner->attack &= ~new_owner->owned;
                if (new_value > 1) {
                    new_owner->owned_army |= lm;
                    new_owner->attack_army |= nm;
                }
                new_owner->attack_army &= ~new_owner->owned;
            }
            if (old_owner != NULL) {
                old_owner->total_army = (int16_t)(old_owner->total_army - old_value);
                old_owner->owned &= ~lm;
                old_owner->owned_army &= ~lm;
                old_owner->attack &= ~nm;
                old_owner->attack_army &= ~nm;
                update_attack_bitmask(old_owner, land);
                for (int i = 0; i < LAND[land].n; i++) update_attack_bitmask(old_owner, LAND[land].nb[i]);
            }
        }
        s->army[land] = value & 63; /* 6-bit field (state.h:26) */
        s->owner[land] = player;
    }
    return ORC_OK;
}

/* State::addLandArmy(idx, value) for the current player (state.cpp:241-256) */
static int add_land_army(orc_state* s, int land, uint8_t value)
{
    if (s->army[land] > 0 && s->owner[land] != (uint8_t)s->cur) return ORC_LOGIC_ERROR;
    int combined = (int)s->army[land] + value;
    if (combined > ORC_ARMY_MAX) return ORC_LOGIC_ERROR;
    return set_land_army(s, land, (uint8_t)combined, (uint8_t)s->cur);
}

/* State::calculateReinforcementValue(ownedLand) (state.cpp:457-491) */
int orc_reinforcement_value(uint64_t owned)
{
    int8_t count = (int8_t)(popc(owned) / 3);
    for (int c = 0; c < 6; c++)
        if ((owned & CONT_MASK[c]) == CONT_MASK[c]) count = (int8_t)(count + CONT_BONUS[c]);
    if (count < 3) count = 3;
    return count;
}

static int goto_fortify(orc_state* s) /* state.cpp:42-49 */
{
    if (s->phase != ORC_ATTACK) return ORC_INVALID_ARGUMENT;
    s->phase = ORC_FORTIFY;
    return ORC_OK;
}

static int goto_attack(orc_state* s) /* state.cpp:20-40 */
{
    if (s->phase != ORC_REINFORCEMENT && s->phase != ORC_ATTACK_MOBILIZATION) return ORC_INVALID_ARGUMENT;
    s->phase = ORC_ATTACK;
    s->mob_from = ORC_NONE;
    s->mob_to = ORC_NONE;
    if (s->reinf > 0) s->reinf = 0;
    if (ccur_ps(s)->attack_army == 0) return goto_fortify(s);
    return ORC_OK;
}

static void next_player_turn(orc_state* s) /* state.cpp:702-712 */
{
    uint8_t p = (uint8_t)s->cur;
    p++;
    if (p >= 2) p = 0;
    s->cur = (int8_t)p;
}

static void next_player_setup_turn(orc_state* s) /* state.cpp:725-746 */
{
    s->phase = ORC_SETUP;
    s->round++;
    next_player_turn(s);
    if (s->reinf == 0) {
        s->phase = ORC_REINFORCEMENT;
        s->reinf = (uint8_t)(int8_t)orc_reinforcement_value(ccur_ps(s)->owned);
    }
}

static void next_player_game_turn(orc_state* s) /* state.cpp:748-766, drawCard :618-626 */
{
    if (s->allow_draw) {
        s->ps[s->cur].cards = (uint8_t)(s->ps[s->cur].cards + 1);
        s->allow_draw = 0;
    }
    s->round++;
    next_player_turn(s);
    s->attacks = 0;
    s->phase = ORC_REINFORCEMENT;
    s->reinf = (uint8_t)(int8_t)orc_reinforcement_value(ccur_ps(s)->owned);
}

/* State::playCards, STATE_SIMPLE_CARDS (state.cpp:1091-1117), via GameHelper::playCards (game_helper.cpp:3-17) */
static void play_cards(orc_state* s)
{
    if (ccur_ps(s)->cards >= 3) {
        cur_ps(s)->cards = (uint8_t)(cur_ps(s)->cards - 3);
        s->card_sets = (uint8_t)(s->card_sets + 1);
        uint16_t gained;
        switch (s->card_sets) {
        case 1: gained = 4; break;
        case 2: gained = 6; break;
        case 3: gained = 8; break;
        case 4: gained = 10; break;
        case 5: gained = 12; break;
        case 6: gained = 15; break;
        default: gained = (uint16_t)(15 + (s->card_sets - 6) * 5); break;
        }
        s->reinf = (uint8_t)(s->reinf + gained);
    }
}

/* State::getDiceRolls (state.cpp:645-684) */
static void dice_rolls(orc_rng* r, int n, uint8_t roll[3])
{
    roll[0] = roll[1] = roll[2] = 0;
    if (n > 0) roll[0] = (uint8_t)orc_rng_dice(r);
    if (n > 1) {
        roll[1] = (uint8_t)orc_rng_dice(r);
        if (roll[0] < roll[1]) { uint8_t t = roll[1]; roll[1] = roll[0]; roll[0] = t; }
    }
    if (n > 2) {
        roll[2] = (uint8_t)orc_rng_dice(r);
        if (roll[0] < roll[2]) { uint8_t t = roll[2]; roll[2] = roll[1]; roll[1] = roll[0]; roll[0] = t; }
        else if (roll[1] < roll[2]) { uint8_t t = roll[2]; roll[2] = roll[1]; roll[1] = t; }
    }
}

/* State::attackMove (state.cpp:769-918) */
static int attack_move(orc_state* s, int from, int to, orc_rng* r)
{
    s->attacks = (uint8_t)(s->attacks + 1);
    if (s->phase != ORC_ATTACK) return ORC_INVALID_ARGUMENT;
    if (from == ORC_NONE) return ORC_INVALID_ARGUMENT;
    if (to == ORC_NONE) return ORC_INVALID_ARGUMENT;
    const uint8_t a_army = s->army[from], d_army = s->army[to];
    const int8_t attacker = (int8_t)s->owner[from], defender = (int8_t)s->owner[to];
    if (attacker != s->cur) return ORC_INVALID_ARGUMENT;
    if (attacker == defender) return ORC_INVALID_ARGUMENT;
    if (a_army <= 1) return ORC_INVALID_ARGUMENT;

    int attacking_units = 1;
    uint8_t attack_amount = a_army, defend_amount = d_army;
    uint8_t ar[3] = {0, 0, 0}, dr[3] = {0, 0, 0};
    if (d_army > 0) {
        int attacking_n = attack_amount >= 4 ? 3 : attack_amount == 3 ? 2 : 1;
        attacking_units = attacking_n;
        int defending_n = defend_amount >= 2 ? 2 : 1;
        dice_rolls(r, attacking_n, ar); /* attacker first */
        dice_rolls(r, defending_n, dr);
        if (ar[0] > dr[0]) defend_amount--;
        else { attack_amount--; attacking_units--; }
        if (attacking_n >= 2 && defending_n == 2) {
            if (ar[1] > dr[1]) defend_amount--;
            else { attack_amount--; attacking_units--; }
        }
    }

    if (defend_amount == 0) {
        attack_amount = (uint8_t)(attack_amount - attacking_units);
        if (attack_amount > 1) {
            s->phase = ORC_ATTACK_MOBILIZATION;
            s->mob_from = (uint8_t)from;
            s->mob_to = (uint8_t)to;
        }
        s->allow_draw = 1;
        TRY(set_land_army(s, from, attack_amount, (uint8_t)attacker));
        TRY(set_land_army(s, to, (uint8_t)attacking_units, (uint8_t)attacker));
    } else {
        TRY(set_land_army(s, from, attack_amount, (uint8_t)attacker));
        TRY(set_land_army(s, to, defend_amount, (uint8_t)defender));
    }
    if (s->phase == ORC_ATTACK && ccur_ps(s)->attack_army == 0) return goto_fortify(s);
    return ORC_OK;
}

/* State::attackReinforcementMove (state.cpp:920-947) */
static int attack_reinforcement_move(orc_state* s, uint8_t amount)
{
    if (s->phase != ORC_ATTACK_MOBILIZATION) return ORC_INVALID_ARGUMENT;
    int from = s->mob_from, to = s->mob_to;
    uint8_t from_army = s->army[from];
    uint8_t after = (uint8_t)(from_army - amount);
    if (after < 1) return ORC_INVALID_ARGUMENT;
    uint8_t to_army = s->army[to];
    TRY(set_land_army(s, from, (uint8_t)(from_army - amount), (uint8_t)s->cur));
    TRY(set_land_army(s, to, (uint8_t)(to_army + amount), (uint8_t)s->cur));
    if (s->army[from] == 1) return goto_attack(s);
    return ORC_OK;
}

/* State::fortifyMove (state.cpp:949-974) */
static int fortify_move(orc_state* s, uint8_t amount, int from, int to)
{
    if (s->phase != ORC_FORTIFY) return ORC_INVALID_ARGUMENT;
    uint8_t after_from = (uint8_t)(s->army[from] - amount);
    if (after_from < 1) return ORC_INVALID_ARGUMENT;
    int after_to = (int)s->army[to] + amount;
    if (after_to > ORC_ARMY_MAX) return ORC_INVALID_ARGUMENT;
    TRY(set_land_army(s, from, after_from, (uint8_t)s->cur));
    TRY(set_land_army(s, to, (uint8_t)after_to, (uint8_t)s->cur));
    return ORC_OK;
}

/* State::reinforcementMove (state.cpp:976-998) */
static int reinforcement_move(orc_state* s, uint8_t amount, int to)
{
    if (s->phase != ORC_REINFORCEMENT) return ORC_INVALID_ARGUMENT;
    if (s->reinf < amount) return ORC_INVALID_ARGUMENT;
    s->reinf = (uint8_t)(s->reinf - amount);
    TRY(add_land_army(s, to, amount));
    if (s->reinf == 0) return goto_attack(s);
    return ORC_OK;
}

/* State::setupReinforcementMove (state.cpp:1009-1030) */
static int setup_reinforcement_move(orc_state* s, int to)
{
    if (s->phase != ORC_SETUP) return ORC_INVALID_ARGUMENT;
    if (s->reinf <= 0) return ORC_INVALID_ARGUMENT;
    s->reinf = (uint8_t)(s->reinf - 2);
    if ((ccur_ps(s)->owned & (1ULL << to)) == 0) return ORC_INVALID_ARGUMENT;
    TRY(add_land_army(s, to, 2));
    s->phase = ORC_SETUP_NEUTRAL; /* gotoSetupNeutral (state.cpp:11-18) */
    return ORC_OK;
}

/* State::setupReinforcementNeutralMove (state.cpp:1032-1053) */
static int setup_reinforcement_neutral_move(orc_state* s, int to)
{
    if (s->phase != ORC_SETUP_NEUTRAL) return ORC_INVALID_ARGUMENT;
    uint64_t neutral = ~ccur_ps(s)->owned & ~cenemy_ps(s)->owned;
    if ((neutral & (1ULL << to)) == 0) return ORC_INVALID_ARGUMENT;
    if (s->owner[to] != ORC_NEUTRAL) return ORC_INVALID_ARGUMENT;
    TRY(set_land_army(s, to, (uint8_t)(s->army[to] + 1), ORC_NEUTRAL));
    next_player_setup_turn(s);
    return ORC_OK;
}

/* State::newGame (state.cpp:137-167) */
void orc_new_game(orc_state* s, orc_rng* r)
{
    orc_state_blank(s);
    uint64_t avail = ALL_LANDS;
    while (avail != 0) {
        uint64_t m = orc_random_mask(r, avail);
        avail &= ~m;
        set_land_army(s, ctz(m), 1, (uint8_t)s->cur);
        if (s->cur == 1) {
            m = orc_random_mask(r, avail);
            avail &= ~m;
            set_land_army(s, ctz(m), 1, ORC_NEUTRAL);
        }
        next_player_turn(s);
    }
    s->reinf = (40 - 14) * 2;
}

/* State::gameStatus (state.cpp:518-565) */
int orc_game_status(const orc_state* s, const orc_settings* cfg)
{
    int p0 = popc(s->ps[0].owned);
    if (p0 == 0) return 1;
    int p1 = popc(s->ps[1].owned);
    if (p1 == 0) return 0;
    if (cfg->allow_yield) {
        if (p0 >= 30) return 0;
        else if (p1 >= 30) return 1;
    }
    if (s->round > cfg->max_game_rounds) {
        if (p0 > p1) return 0;
        else if (p0 < p1) return 1;
        else return ORC_DRAW;
    }
    return ORC_NOT_ENDED;
}

/* State::invertPlayers (state.cpp:493-516) */
void orc_invert_players(orc_state* s)
{
    orc_player t = s->ps[0];
    s->ps[0] = s->ps[1];
    s->ps[1] = t;
    for (int i = 0; i < ORC_LANDS; i++) {
        if (s->owner[i] == 0) s->owner[i] = 1;
        else if (s->owner[i] == 1) s->owner[i] = 0;
    }
}

/* State::getNeutralPlayerAttackLands (state.cpp:1067-1083) */
static uint64_t neutral_attack_lands(const orc_state* s)
{
    uint64_t neutral = ALL_LANDS & ~ccur_ps(s)->owned & ~cenemy_ps(s)->owned;
    uint64_t out = 0, it = neutral;
    while (it > 0) {
        int l = ctz(it);
        it &= ~(1ULL << l);
        out |= nb_mask(l);
    }
    return out & ~neutral;
}

/* UtilityNN::getValidMoves (alphazero_moves.cpp:3-70) */
uint64_t orc_valid_moves(const orc_state* s, const orc_settings* cfg)
{
    const orc_player* pls = ccur_ps(s);
    const orc_player* epls = cenemy_ps(s);
    switch (s->phase) {
    case ORC_SETUP:
    case ORC_REINFORCEMENT: {
        uint64_t owned = pls->owned & ~pls->owned_full;
        if (owned == 0) return SKIP_MASK;
        else if (cfg->limit_reinforcement) {
            uint64_t nb = owned & (epls->attack | neutral_attack_lands(s));
            if (nb != 0) return nb;
            return owned;
        } else return owned;
    }
    case ORC_SETUP_NEUTRAL:
        return ALL_LANDS & ~pls->owned & ~epls->owned;
    case ORC_ATTACK:
        if (cfg->limit_attack) {
            if (popc(pls->attack_army) > 0) return pls->attack_army;
            else return SKIP_MASK;
        } else return pls->attack_army | SKIP_MASK;
    case ORC_ATTACK_MOBILIZATION:
        return (1ULL << s->mob_from) | (1ULL << s->mob_to);
    case ORC_FORTIFY:
        if (cfg->limit_reinforcement) return (pls->owned & epls->attack) | SKIP_MASK; /* `a & b | SKIP` */
        else return pls->owned | SKIP_MASK;
    default:
        return 0;
    }
}

/* GameHelper::LandSetMovement::add (game_helper.cpp:51-82): recursive pre-order flood */
static void lsm_add(int land, uint64_t owned, uint64_t* set_mask, uint8_t* list, int* n)
{
    if (((1ULL << land) & owned & ~*set_mask) > 0) {
        *set_mask |= 1ULL << land;
        list[(*n)++] = (uint8_t)land;
        for (int i = 0; i < LAND[land].n; i++) lsm_add(LAND[land].nb[i], owned, set_mask, list, n);
    }
}

/* UtilityNN::makeMove (alphazero_moves.cpp:72-233) */
int orc_make_move(orc_state* s, int li, orc_rng* r, const orc_settings* cfg)
{
    if (li == ORC_NONE) return ORC_INVALID_ARGUMENT;
    if (li == ORC_SKIP) {
        switch (s->phase) {
        case ORC_REINFORCEMENT: return goto_attack(s);
        case ORC_ATTACK: return goto_fortify(s);
        case ORC_FORTIFY: next_player_game_turn(s); return ORC_OK;
        default: return ORC_LOGIC_ERROR;
        }
    }
    if (li < 0 || li > 41) return ORC_LOGIC_ERROR;
    const orc_player* pls = ccur_ps(s);
    if (s->phase == ORC_SETUP) {
        return setup_reinforcement_move(s, li);
    } else if (s->phase == ORC_SETUP_NEUTRAL) {
        return setup_reinforcement_neutral_move(s, li);
    } else if (s->phase == ORC_REINFORCEMENT) {
        play_cards(s);
        uint8_t reinforcement = (uint8_t)(s->reinf / 2);
        if (reinforcement < cfg->min_unit_move)
            reinforcement = (uint8_t)(cfg->min_unit_move < (int)s->reinf ? cfg->min_unit_move : (int)s->reinf);
        uint8_t max_value = (uint8_t)(ORC_ARMY_MAX - s->army[li]);
        reinforcement = max_value < reinforcement ? max_value : reinforcement;
        return reinforcement_move(s, reinforcement, li);
    } else if (s->phase == ORC_ATTACK) {
        uint8_t best_army = 0;
        int best_from = ORC_NONE;
        for (int i = 0; i < LAND[li].n; i++) {
            int nl = LAND[li].nb[i];
            if (((1ULL << nl) & pls->owned_army) > 0) {
                uint8_t attack_army = (uint8_t)(s->army[nl] - 1);
                if (attack_army > best_army) { best_army = attack_army; best_from = nl; }
            }
        }
        return attack_move(s, best_from, li, r);
    } else if (s->phase == ORC_ATTACK_MOBILIZATION) {
        if (li == s->mob_from) return goto_attack(s);
        else if (li == s->mob_to) {
            uint8_t value = (uint8_t)(s->army[s->mob_from] - 1);
            uint8_t reinforcement = (uint8_t)(value / 2);
            if (reinforcement < cfg->min_unit_move)
                reinforcement = (uint8_t)(cfg->min_unit_move < (int)value ? cfg->min_unit_move : (int)value);
            return attack_reinforcement_move(s, reinforcement);
        } else return ORC_INVALID_ARGUMENT;
    } else if (s->phase == ORC_FORTIFY) {
        uint8_t value_to = s->army[li];
        if (value_to != ORC_ARMY_MAX) {
            /* GameHelper::PlayerMovement (game_helper.cpp:90-109): components in index order of their
             * lowest land; the std::sort by landFortifyFromAmount cannot change which component holds li */
            uint64_t owned = pls->owned, covered = 0;
            for (int i = 0; i < ORC_LANDS; i++) {
                if (((1ULL << i) & owned & ~covered) > 0) {
                    uint64_t set_mask = 0;
                    uint8_t list[ORC_LANDS];
                    int n = 0;
                    lsm_add(i, owned, &set_mask, list, &n);
                    covered |= set_mask;
                    if ((set_mask & (1ULL << li)) > 0) {
                        uint8_t best_nn = 0, best = 0;
                        int from_nn = ORC_NONE, from = ORC_NONE;
                        for (int j = 0; j < n; j++) {
                            int lf = list[j];
                            if (lf != li) {
                                uint8_t value = (uint8_t)(s->army[lf] - 1);
                                uint64_t owned_nb = nb_mask(lf) & pls->owned;
                                if (owned_nb == nb_mask(lf)) {
                                    if (value > best_nn) { best_nn = value; from_nn = lf; }
                                } else {
                                    if (value > best) { best = value; from = lf; }
                                }
                            }
                        }
                        if (from_nn != ORC_NONE) { from = from_nn; best = best_nn; }
                        if (from != ORC_NONE) {
                            uint8_t max_value = (uint8_t)(ORC_ARMY_MAX - s->army[li]);
                            TRY(fortify_move(s, max_value < best ? max_value : best, from, li));
                        }
                        break;
                    }
                }
            }
        }
        next_player_game_turn(s);
        return ORC_OK;
    }
    return ORC_LOGIC_ERROR;
}

/* State::consistencyCheck + consistencyCheckArmyValue (state.cpp:1181-1429), restated as a
 * recomputation of every derived field from landArmy[] */
int orc_consistency_check(const orc_state* s)
{
    int bad = 0;
    for (int p = 0; p < 2; p++) {
        uint64_t owned = 0, owned_army = 0, full = 0;
        int total = 0;
        for (int i = 0; i < ORC_LANDS; i++)
            if (s->owner[i] == p) {
                owned |= 1ULL << i;
                if (s->army[i] > 1) owned_army |= 1ULL << i;
                if (s->army[i] == ORC_ARMY_MAX) full |= 1ULL << i;
                total += s->army[i];
            }
        uint64_t attack = 0, attack_army = 0;
        for (int i = 0; i < ORC_LANDS; i++) {
            if (owned & (1ULL << i)) continue;
            if (nb_mask(i) & owned) attack |= 1ULL << i;
            if (nb_mask(i) & owned_army) attack_army |= 1ULL << i;
        }
        const orc_player* q = &s->ps[p];
        if (q->owned != owned) bad |= 1;
        if (q->owned_army != owned_army) bad |= 2;
        if (q->owned_full != full) bad |= 4;
        if (q->attack != attack) bad |= 8;
        if (q->attack_army != attack_army) bad |= 16;
        if (q->total_army != total) bad |= 32;
    }
    return bad;
}

/* ------------------------------------------------------------------------------------------------
 * NN data seams (neural_network/alphazero_nn_data.{h,cpp}, alphazero_nn.cpp:31-67)
 * ---------------------------------------------------------------------------------------------- */
static void putf(uint8_t* p, float f) { memcpy(p, &f, 4); }
static float getf(const uint8_t* p) { float f; memcpy(&f, p, 4); return f; }

/* NNInputData(const State&) (alphazero_nn_data.cpp:165-196); byte layout probed from the reference */
void orc_encode(const orc_state* s, uint8_t* in)
{
    memset(in, 0, 88);
    for (int i = 0; i < ORC_LANDS; i++) in[i] = (uint8_t)((s->army[i] & 63) | (s->owner[i] << 6));
    in[42] = (uint8_t)s->cur;
    in[44] = (uint8_t)(s->round & 0xff);
    in[45] = (uint8_t)(s->round >> 8);
    const orc_player* ps = ccur_ps(s);
    const orc_player* eps = cenemy_ps(s);
    float ref = (float)(int8_t)orc_reinforcement_value(ps->owned);
    float eref = (float)(int8_t)orc_reinforcement_value(eps->owned);
    putf(in + 48, ref / (ref + eref));
    float af = s->attacks / 8.0f;
    putf(in + 52, af < 1.0f ? af : 1.0f);
    putf(in + 56, s->allow_draw ? 1.0f : 0.0f);
    for (int ph = 0; ph < 6; ph++) putf(in + 60 + 4 * ph, s->phase == ph ? 1.0f : 0.0f);
    float ta = ps->total_army, eta = eps->total_army;
    putf(in + 84, ta / (ta + eta));
}

/* setInStateTensor (alphazero_nn.cpp:31-67) with plane indices of alphazero_nn_data.h:13-39 (V2):
 * 0 current, 1 enemy, 2 neutral, 3 armyShare, 4 reinforcementShare, 5 attacks, 6 canDraw, 7..12 phases */
void orc_planes(const uint8_t* in, float* t)
{
    int cur = in[42], enemy = cur == 0 ? 1 : 0;
    for (int pos = 0; pos < ORC_LANDS; pos++) {
        float* c = t + pos * 13;
        int army = in[pos] & 63, owner = in[pos] >> 6;
        float fa = (float)army / ORC_ARMY_MAX;
        c[0] = cur == owner ? fa : 0.0f;
        c[1] = enemy == owner ? fa : 0.0f;
        c[2] = ORC_NEUTRAL == owner ? fa : 0.0f;
        c[3] = getf(in + 84);
        c[4] = getf(in + 48);
        c[5] = getf(in + 52);
        c[6] = getf(in + 56);
        for (int ph = 0; ph < 6; ph++) c[7 + ph] = getf(in + 60 + 4 * ph);
    }
}

/* NNOutputData::normalize (alphazero_nn_data.cpp:3-27) */
void orc_normalize(float* pi, uint64_t valid)
{
    float sum = 0.0f;
    uint64_t bit = 1;
    for (int i = 0; i < ORC_MOVES; i++) {
        if ((valid & bit) > 0) sum += pi[i];
        else pi[i] = 0.0f;
        bit <<= 1;
    }
    for (int i = 0; i < ORC_MOVES; i++)
        if (pi[i] > 0.0f) pi[i] /= sum;
}

/* NNTrainDataStorage::updateValues (alphazero_nn_data.cpp:51-65), ROUND_WEIGHTED_VALUE off */
void orc_update_values(const int8_t* player_index, int n, int game_status, float* z)
{
    for (int i = 0; i < n; i++)
        z[i] = game_status == ORC_DRAW ? 0.0f : player_index[i] == game_status ? 1.0f : -1.0f;
}

int orc_play_random_game(uint32_t seed, int cap, uint8_t* states160, uint64_t* masks, uint8_t* moves,
                         int* status, uint8_t* final160, const orc_settings* cfg)
{
    orc_rng r;
    orc_rng_seed(&r, seed);
    orc_state s;
    orc_new_game(&s, &r);
    int n = 0, st = orc_game_status(&s, cfg);
    while (st == ORC_NOT_ENDED && n < cap) {
        uint64_t vm = orc_valid_moves(&s, cfg);
        uint64_t m = orc_random_mask(&r, vm);
        int mv = ctz(m);
        if (states160) orc_state_pack(&s, states160 + (size_t)n * 160);
        if (masks) masks[n] = vm;
        if (moves) moves[n] = (uint8_t)mv;
        n++;
        int rc = orc_make_move(&s, mv, &r, cfg);
        if (rc) { *status = -100 - rc; orc_state_pack(&s, final160); return n; }
        st = orc_game_status(&s, cfg);
    }
    *status = st;
    orc_state_pack(&s, final160);
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * libstdc++ unordered_map<LandIndex, SimulationValue> iteration order.
 * std::hash of an enum is the identity; _Hashtable inserts a node at the beginning of its bucket, or
 * at the front of the global singly-linked list when the bucket is empty; _M_rehash_aux walks the
 * old list and re-inserts the same way; _Prime_rehash_policy (max load 1.0, growth 2): first
 * allocation 13 buckets, then 29, then 59.  Validated against the host library in
 * tests/test_umap_order.py.
 * ---------------------------------------------------------------------------------------------- */
#define UM_BB 63 /* before-begin sentinel */
typedef struct {
    int next[64];
    int bucket[64]; /* node preceding the bucket's first node, -1 = empty */
    int nb, size, next_resize;
} um_t;

static const int UM_PRIMES[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 103, 109, 113, 127, 0};

static int um_next_bkt(um_t* u, int n)
{
    static const unsigned char fast[] = {2, 2, 2, 3, 5, 5, 7, 7, 11, 11, 11, 11, 13, 13};
    if (n < (int)sizeof fast) {
        if (n == 0) return 1;
        u->next_resize = fast[n];
        return fast[n];
    }
    for (int i = 6; UM_PRIMES[i]; i++)
        if (UM_PRIMES[i] >= n) { u->next_resize = UM_PRIMES[i]; return UM_PRIMES[i]; }
    return 127;
}

static void um_insert_bucket_begin(um_t* u, int bkt, int node)
{
    if (u->bucket[bkt] >= 0) {
        u->next[node] = u->next[u->bucket[bkt]];
        u->next[u->bucket[bkt]] = node;
    } else {
        u->next[node] = u->next[UM_BB];
        u->next[UM_BB] = node;
        if (u->next[node] >= 0) u->bucket[u->next[node] % u->nb] = node;
        u->bucket[bkt] = UM_BB;
    }
}

static void um_rehash(um_t* u, int n)
{
    int p = u->next[UM_BB];
    for (int i = 0; i < 64; i++) u->bucket[i] = -1;
    u->next[UM_BB] = -1;
    int bbegin_bkt = 0;
    while (p >= 0) {
        int nx = u->next[p];
        int bkt = p % n;
        if (u->bucket[bkt] < 0) {
            u->next[p] = u->next[UM_BB];
            u->next[UM_BB] = p;
            u->bucket[bkt] = UM_BB;
            if (u->next[p] >= 0) u->bucket[bbegin_bkt] = p;
            bbegin_bkt = bkt;
        } else {
            u->next[p] = u->next[u->bucket[bkt]];
            u->next[u->bucket[bkt]] = p;
        }
        p = nx;
    }
    u->nb = n;
}

int orc_umap_order(uint64_t mask, uint8_t* out)
{
    um_t u;
    for (int i = 0; i < 64; i++) { u.next[i] = -1; u.bucket[i] = -1; }
    u.nb = 1; u.size = 0; u.next_resize = 0;
    for (int k = 0; k < ORC_MOVES; k++) {
        if (!(mask & (1ULL << k))) continue;
        /* _M_need_rehash(n_bkt, n_elt, 1) */
        if (u.size + 1 > u.next_resize) {
            int want = u.size + 1;
            if (u.next_resize == 0 && want < 11) want = 11;
            double min_bkts = (double)want / 1.0;
            if (min_bkts >= u.nb) {
                int a = (int)floor(min_bkts) + 1, b = u.nb * 2;
                um_rehash(&u, um_next_bkt(&u, a > b ? a : b));
            } else u.next_resize = u.nb;
        }
        um_insert_bucket_begin(&u, k % u.nb, k);
        u.size++;
    }
    int n = 0;
    for (int p = u.next[UM_BB]; p >= 0; p = u.next[p]) out[n++] = (uint8_t)p;
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * Policy/value net, fp32 (python/src/build_graph.py:37-90; shapes from python/model/model_txt_V2_5.pb)
 *
 * AZRW flat parameter layout (floats), B = blocks, F = 256:
 *   stem   : conv W[3][3][13][F] (HWIO) ; conv_bn gamma[7] beta[7] mean[7] var[7]   (BN over board ROW y)
 *   block i: 2a W[3][3][F][F] ; bn2a gamma[F] beta[F] mean[F] var[F] ; 2b W[3][3][F][F] ; bn2b (same)
 *   policy : pi W[F][2] ; bn_pi g,b,m,v [2] each ; dense W[84][43] ; bias[43]
 *   value  : v W[F][1] ; bn_v g,b,m,v [1] each ; dense_1 W[42][256] ; bias[256] ; dense_2 W[256][1] ; bias[1]
 * ---------------------------------------------------------------------------------------------- */
#define NF 256
#define NPOS 42
#define NIN 13
#define BN_EPS 1e-3f

size_t orc_net_param_count(int blocks)
{
    size_t n = 9 * NIN * NF + 4 * 7;
    n += (size_t)blocks * 2 * (9 * NF * NF + 4 * NF);
    n += NF * 2 + 4 * 2 + 84 * 43 + 43;
    n += NF * 1 + 4 * 1 + 42 * 256 + 256 + 256 + 1;
    return n;
}

static uint64_t splitmix64(uint64_t* s)
{
    uint64_t z = (*s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static float uni(uint64_t* s) { return (float)((splitmix64(s) >> 40) * (1.0 / 16777216.0)); } /* [0,1) */

static float* fill_glorot(float* p, size_t n, int fan_in, int fan_out, uint64_t* s)
{
    float lim = sqrtf(6.0f / (float)(fan_in + fan_out));
    for (size_t i = 0; i < n; i++) p[i] = (2.0f * uni(s) - 1.0f) * lim;
    return p + n;
}
static float* fill_bn(float* p, int c)
{
    for (int i = 0; i < c; i++) p[i] = 1.0f;           /* gamma */
    for (int i = 0; i < c; i++) p[c + i] = 0.0f;       /* beta */
    for (int i = 0; i < c; i++) p[2 * c + i] = 0.0f;   /* moving mean */
    for (int i = 0; i < c; i++) p[3 * c + i] = 1.0f;   /* moving variance */
    return p + 4 * c;
}
static float* fill_zero(float* p, size_t n) { memset(p, 0, n * sizeof(float)); return p + n; }

void orc_net_init_random(float* flat, int blocks, uint64_t seed)
{
    uint64_t s = seed;
    float* p = flat;
    p = fill_glorot(p, 9 * NIN * NF, 9 * NIN, 9 * NF, &s);
    p = fill_bn(p, 7);
    for (int b = 0; b < blocks * 2; b++) {
        p = fill_glorot(p, 9 * NF * NF, 9 * NF, 9 * NF, &s);
        p = fill_bn(p, NF);
    }
    p = fill_glorot(p, NF * 2, NF, 2, &s);
    p = fill_bn(p, 2);
    p = fill_glorot(p, 84 * 43, 84, 43, &s);
    p = fill_zero(p, 43);
    p = fill_glorot(p, NF, NF, 1, &s);
    p = fill_bn(p, 1);
    p = fill_glorot(p, 42 * 256, 42, 256, &s);
    p = fill_zero(p, 256);
    p = fill_glorot(p, 256, 256, 1, &s);
    p = fill_zero(p, 1);
}

/* SAME 3x3 conv on the 7x6 board, NHWC, HWIO weights, no bias */
static void conv3x3(const float* in, int cin, const float* w, float* out)
{
    memset(out, 0, sizeof(float) * NPOS * NF);
    for (int y = 0; y < 7; y++)
        for (int x = 0; x < 6; x++) {
            float* o = out + (y * 6 + x) * NF;
            for (int ky = 0; ky < 3; ky++) {
                int yy = y + ky - 1;
                if (yy < 0 || yy >= 7) continue;
                for (int kx = 0; kx < 3; kx++) {
                    int xx = x + kx - 1;
                    if (xx < 0 || xx >= 6) continue;
                    const float* ip = in + (yy * 6 + xx) * cin;
                    const float* wp = w + (size_t)(ky * 3 + kx) * cin * NF;
                    for (int ci = 0; ci < cin; ci++) {
                        float xv = ip[ci];
                        const float* wr = wp + (size_t)ci * NF;
                        for (int co = 0; co < NF; co++) o[co] += xv * wr[co];
                    }
                }
            }
        }
}

static float bn_apply(float x, const float* bn, int c, int i)
{
    float g = bn[i], b = bn[c + i], m = bn[2 * c + i], v = bn[3 * c + i];
    return (x - m) * (g / sqrtf(v + BN_EPS)) + b;
}

static void net_forward_one(const orc_net* net, const uint8_t* in88, float* pi, float* v)
{
    float x0[NPOS * NIN];
    float* a = (float*)malloc(sizeof(float) * NPOS * NF * 3);
    float *X = a, *T = a + NPOS * NF, *U = a + 2 * NPOS * NF;
    const float* p = net->flat;
    orc_planes(in88, x0);
    conv3x3(x0, NIN, p, X); p += 9 * NIN * NF;
    for (int pos = 0; pos < NPOS; pos++) /* conv_bn: axis=1 => per board row y (build_graph.py:68) */
        for (int c = 0; c < NF; c++) {
            float t = bn_apply(X[pos * NF + c], p, 7, pos / 6);
            X[pos * NF + c] = t > 0 ? t : 0;
        }
    p += 28;
    for (int b = 0; b < net->blocks; b++) {
        conv3x3(X, NF, p, T); p += 9 * NF * NF;
        for (int i = 0; i < NPOS * NF; i++) { float t = bn_apply(T[i], p, NF, i % NF); T[i] = t > 0 ? t : 0; }
        p += 4 * NF;
        conv3x3(T, NF, p, U); p += 9 * NF * NF;
        for (int i = 0; i < NPOS * NF; i++) { float t = bn_apply(U[i], p, NF, i % NF) + X[i]; X[i] = t > 0 ? t : 0; }
        p += 4 * NF;
    }
    /* policy head (build_graph.py:76-81) */
    float ph[84];
    const float* wpi = p; p += NF * 2;
    const float* bnpi = p; p += 8;
    for (int pos = 0; pos < NPOS; pos++)
        for (int c = 0; c < 2; c++) {
            float s = 0;
            for (int ci = 0; ci < NF; ci++) s += X[pos * NF + ci] * wpi[ci * 2 + c];
            float t = bn_apply(s, bnpi, 2, c);
            ph[pos * 2 + c] = t > 0 ? t : 0;
        }
    const float* wd = p; p += 84 * 43;
    const float* bd = p; p += 43;
    float logit[43], mx = -INFINITY;
    for (int j = 0; j < 43; j++) {
        float s = 0;
        for (int i = 0; i < 84; i++) s += ph[i] * wd[i * 43 + j];
        logit[j] = s + bd[j];
        if (logit[j] > mx) mx = logit[j];
    }
    float se = 0;
    for (int j = 0; j < 43; j++) { logit[j] = expf(logit[j] - mx); se += logit[j]; }
    for (int j = 0; j < 43; j++) pi[j] = logit[j] / se;
    /* value head (build_graph.py:83-90) */
    float vh[42];
    const float* wv = p; p += NF;
    const float* bnv = p; p += 4;
    for (int pos = 0; pos < NPOS; pos++) {
        float s = 0;
        for (int ci = 0; ci < NF; ci++) s += X[pos * NF + ci] * wv[ci];
        float t = bn_apply(s, bnv, 1, 0);
        vh[pos] = t > 0 ? t : 0;
    }
    const float* w1 = p; p += 42 * 256;
    const float* b1 = p; p += 256;
    const float* w2 = p; p += 256;
    const float* b2 = p; p += 1;
    float h[256];
    for (int j = 0; j < 256; j++) {
        float s = 0;
        for (int i = 0; i < 42; i++) s += vh[i] * w1[i * 256 + j];
        s += b1[j];
        h[j] = s > 0 ? s : 0;
    }
    float s = 0;
    for (int j = 0; j < 256; j++) s += h[j] * w2[j];
    *v = tanhf(s + b2[0]);
    free(a);
}

void orc_net_forward(const orc_net* net, const uint8_t* in88, int n, float* pi, float* v)
{
    for (int i = 0; i < n; i++) net_forward_one(net, in88 + (size_t)i * 88, pi + (size_t)i * 43, v + i);
}

typedef struct { const orc_net* net; const uint8_t* in; float* pi; float* v; int lo, hi; } fw_job;
static void* fw_thread(void* a)
{
    fw_job* j = (fw_job*)a;
    for (int i = j->lo; i < j->hi; i++) net_forward_one(j->net, j->in + (size_t)i * 88, j->pi + (size_t)i * 43, j->v + i);
    return NULL;
}
void orc_net_forward_mt(const orc_net* net, const uint8_t* in88, int n, float* pi, float* v, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    if (threads > n) threads = n > 0 ? n : 1;
    pthread_t th[64];
    fw_job jobs[64];
    for (int t = 0; t < threads; t++) {
        jobs[t] = (fw_job){net, in88, pi, v, (int)((long)n * t / threads), (int)((long)n * (t + 1) / threads)};
        pthread_create(&th[t], NULL, fw_thread, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
}

void orc_net_eval(void* ctx, const uint8_t* in88, float* pi43, float* v) { net_forward_one((const orc_net*)ctx, in88, pi43, v); }

/* stub nets for tree-only parity tests */
void orc_hash_eval(void* ctx, const uint8_t* in88, float* pi, float* v)
{
    (void)ctx;
    uint64_t h = 0xcbf29ce484222325ULL;
    for (int i = 0; i < 88; i++) { h ^= in88[i]; h *= 0x100000001b3ULL; }
    uint64_t s = h;
    float sum = 0.0f;
    for (int i = 0; i < 43; i++) { pi[i] = 0.5f + (float)((splitmix64(&s) >> 40) * (1.0 / 16777216.0)); sum += pi[i]; }
    for (int i = 0; i < 43; i++) pi[i] /= sum;
    *v = (float)((splitmix64(&s) >> 40) * (1.0 / 16777216.0)) * 2.0f - 1.0f;
}
void orc_uniform_eval(void* ctx, const uint8_t* in88, float* pi, float* v)
{
    (void)ctx; (void)in88;
    for (int i = 0; i < 43; i++) pi[i] = 1.0f / 43.0f;
    *v = 0.0f;
}

/* ------------------------------------------------------------------------------------------------
 * MCTS (player/alpha_zero/alphazero_mcts.{h,cpp}).  THREADS_PER_MCTS = T search threads are run in
 * lock-step: the reference's threads all block in predictFuture until the batch is evaluated
 * (alphazero_mcts.cpp:262-265), so one round = [every thread with an evaluated leaf: store.add +
 * backup, in thread order] then [every thread: claim simulations and descend until it needs the net, in
 * thread order].  That is one of the reference's possible schedules and the only one at T = 1.
 * ---------------------------------------------------------------------------------------------- */
#define ORC_MAX_THREADS 8
#define ORC_MAX_DEPTH 1024
typedef struct {
    int pending, plen;
    int node[ORC_MAX_DEPTH];
    uint8_t mv[ORC_MAX_DEPTH], flip[ORC_MAX_DEPTH];
    orc_state leaf;
    uint64_t valid;
} orc_thread;

typedef struct {
    orc_state key;
    float Q[ORC_MOVES], P[ORC_MOVES];
    uint32_t N[ORC_MOVES];
    uint8_t active[ORC_MOVES];
    uint64_t valid;
    uint8_t order[ORC_MOVES];
    int norder;
    float value;
    int visited;
    uint32_t sumN;
} orc_node;

struct orc_mcts {
    orc_settings cfg;
    orc_node* nodes;
    int count, cap;
    uint64_t sims, evals, levels, dup_dropped;
    orc_thread* th;
};

orc_mcts* orc_mcts_create(const orc_settings* cfg)
{
    orc_mcts* m = (orc_mcts*)calloc(1, sizeof *m);
    m->cfg = *cfg;
    m->cap = 256;
    m->nodes = (orc_node*)malloc(sizeof(orc_node) * (size_t)m->cap);
    if (m->cfg.mcts_threads < 1) m->cfg.mcts_threads = 1;
    if (m->cfg.mcts_threads > ORC_MAX_THREADS) m->cfg.mcts_threads = ORC_MAX_THREADS;
    m->th = (orc_thread*)calloc((size_t)m->cfg.mcts_threads, sizeof(orc_thread));
    return m;
}
void orc_mcts_destroy(orc_mcts* m) { if (m) { free(m->nodes); free(m->th); free(m); } }
uint64_t orc_mcts_dup_count(const orc_mcts* m) { return m->dup_dropped; }
void orc_mcts_clear(orc_mcts* m) { m->count = 0; }                     /* clearNodes (:223-227) */
int orc_mcts_node_count(const orc_mcts* m) { return m->count; }
uint64_t orc_mcts_sim_count(const orc_mcts* m) { return m->sims; }
uint64_t orc_mcts_eval_count(const orc_mcts* m) { return m->evals; }
uint64_t orc_mcts_level_count(const orc_mcts* m) { return m->levels; }

/* StateSimulationsStorage::trimNodes (:229-245) */
void orc_mcts_trim(orc_mcts* m)
{
    int w = 0;
    for (int i = 0; i < m->count; i++) {
        if (m->nodes[i].visited) {
            m->nodes[i].visited = 0;
            if (w != i) m->nodes[w] = m->nodes[i];
            w++;
        }
    }
    m->count = w;
}

/* exist / getStateSimulation (:189-221): keyed by field equality */
static int find_node(const orc_mcts* m, const orc_state* s)
{
    for (int i = 0; i < m->count; i++)
        if (orc_state_equal(&m->nodes[i].key, s)) return i;
    return -1;
}

/* StateSimulations ctor (:26-42) + store.add (:203-215) */
static int add_node(orc_mcts* m, const orc_state* s, const float* pi, float value, uint64_t valid)
{
    if (m->count == m->cap) {
        m->cap *= 2;
        m->nodes = (orc_node*)realloc(m->nodes, sizeof(orc_node) * (size_t)m->cap);
    }
    orc_node* n = &m->nodes[m->count];
    memset(n, 0, sizeof *n);
    n->key = *s;
    n->value = value;
    n->visited = 1;
    n->sumN = 0;
    n->valid = valid;
    for (int i = 0; i < ORC_MOVES; i++)
        if (valid & (1ULL << i)) n->P[i] = pi[i];
    n->norder = orc_umap_order(valid, n->order);
    return m->count++;
}

/* StateSimulations::getNextBestMoveAndSetVisited (:67-119) */
static int next_best_move(const orc_settings* cfg, orc_node* n)
{
    n->visited = 1;
    int best = ORC_NONE, dup_best = ORC_NONE;
    float best_u = -INFINITY, dup_u = -INFINITY;
    for (int k = 0; k < n->norder; k++) {
        int mv = n->order[k];
        float P = n->P[mv];
        float noiseP = (1 - cfg->dir_noise_epsi) * P + cfg->dir_noise_epsi * cfg->dir_noise_value;
        float v = noiseP * cfg->hp_exploration * sqrtf(1.0f + n->sumN);
        float nn = 1.0f + n->N[mv];
        float u = n->Q[mv] + (v / nn);
        if (u > best_u) {
            if (n->N[mv] == 0 && n->active[mv] == 1) {
                if (u > dup_u) { dup_u = u; dup_best = mv; }
            } else {
                best_u = u;
                best = mv;
            }
        }
    }
    if (best == ORC_NONE) best = dup_best;
    if (best == ORC_NONE) return ORC_NONE;
    n->active[best]++;
    return best;
}

/* SimulationValue::addValue + StateSimulations::addValue (:8-21,55-60) */
static void add_value(orc_node* n, int mv, float v)
{
    if (n->N[mv] == 0) n->Q[mv] = v;
    else n->Q[mv] = (n->N[mv] * n->Q[mv] + v) / (n->N[mv] + 1);
    n->N[mv]++;
    n->active[mv]--;
    n->sumN++;
}

/* the unwinding of AlphaZeroMCTS::search's recursion (:367-375): sign flips where the mover changed */
static void backup(orc_mcts* m, const orc_thread* t, float v)
{
    for (int i = t->plen - 1; i >= 0; i--) {
        if (t->flip[i]) v = -v;
        add_value(&m->nodes[t->node[i]], t->mv[i], v); /* nodes[] may have been realloc'ed: index, not pointer */
    }
}

/* AlphaZeroMCTS::search (:322-377) from the root down to a terminal state (*leaf = 0, backed up) or to a state that
 * is not in the store (*leaf = 1: the thread blocks in predictFuture) */
static int descend(orc_mcts* m, orc_thread* t, const orc_state* root, orc_rng* r, int* leaf)
{
    orc_state s = *root;
    t->plen = 0;
    for (;;) {
        int gs = orc_game_status(&s, &m->cfg);
        if (gs != ORC_NOT_ENDED) {
            backup(m, t, gs == ORC_DRAW ? 0.0f : (gs == s.cur ? 1.0f : -1.0f));
            *leaf = 0;
            return ORC_OK;
        }
        uint64_t valid = orc_valid_moves(&s, &m->cfg);
        if (valid == 0) return ORC_INVALID_ARGUMENT;
        int idx = find_node(m, &s);
        if (idx < 0) {
            t->leaf = s;
            t->valid = valid;
            t->pending = 1;
            *leaf = 1;
            return ORC_OK;
        }
        m->levels++;
        int best = next_best_move(&m->cfg, &m->nodes[idx]);
        if (best == ORC_NONE) return ORC_LOGIC_ERROR; /* moveValues.at(None) throws out_of_range */
        int cur = s.cur;
        TRY(orc_make_move(&s, best, r, &m->cfg));
        if (t->plen >= ORC_MAX_DEPTH) return ORC_LOGIC_ERROR;
        t->node[t->plen] = idx;
        t->mv[t->plen] = (uint8_t)best;
        t->flip[t->plen] = (uint8_t)(cur != s.cur);
        t->plen++;
    }
}

/* the leaf branch of search after the future resolved (:350-356) */
static void expand_leaf(orc_mcts* m, const orc_state* s, uint64_t valid, orc_eval_fn eval, void* ctx, float* v)
{
    uint8_t in88[88];
    float pi[ORC_MOVES];
    orc_encode(s, in88);
    eval(ctx, in88, pi, v);
    m->evals++;
    orc_normalize(pi, valid);
    if (find_node(m, s) < 0) add_node(m, s, pi, *v, valid);
    else m->dup_dropped++; /* StateSimulationsStorage::add (:203-215) */
}

/* AlphaZeroMCTS::simulate + setRootState + threadSimulateJob (:255-320) */
int orc_mcts_simulate(orc_mcts* m, const orc_state* root, orc_rng* r, orc_eval_fn eval, void* ctx)
{
    const int T = m->cfg.mcts_threads;
    orc_mcts_trim(m);
    if (find_node(m, root) < 0) {
        float v;
        expand_leaf(m, root, orc_valid_moves(root, &m->cfg), eval, ctx, &v);
    }
    const int count = m->cfg.mcts_simulations - m->cfg.mcts_simulations % T;
    int started = 0;
    for (int k = 0; k < T; k++) m->th[k].pending = 0;
    for (;;) {
        for (int k = 0; k < T; k++) {
            orc_thread* t = &m->th[k];
            if (!t->pending) continue;
            float v;
            expand_leaf(m, &t->leaf, t->valid, eval, ctx, &v);
            backup(m, t, v);
            m->sims++;
            t->pending = 0;
        }
        int waiting = 0;
        for (int k = 0; k < T; k++) {
            orc_thread* t = &m->th[k];
            while (started < count) { /* Counter::hasNext */
                int leaf = 0;
                started++;
                TRY(descend(m, t, root, r, &leaf));
                if (leaf) { waiting++; break; }
                m->sims++;
            }
        }
        if (!waiting) break;
    }
    return ORC_OK;
}

int orc_mcts_root_stats(orc_mcts* m, const orc_state* root, uint32_t* n43, float* q43, float* p43, uint32_t* sumN)
{
    int idx = find_node(m, root);
    if (idx < 0) return ORC_LOGIC_ERROR;
    const orc_node* n = &m->nodes[idx];
    for (int i = 0; i < ORC_MOVES; i++) {
        if (n43) n43[i] = n->N[i];
        if (q43) q43[i] = n->Q[i];
        if (p43) p43[i] = n->P[i];
    }
    if (sumN) *sumN = n->sumN;
    return ORC_OK;
}

/* StateSimulations::calculateMoveProbability(1.0f) (:121-149) */
int orc_mcts_policy(orc_mcts* m, const orc_state* root, float* pi)
{
    int idx = find_node(m, root);
    if (idx < 0) return ORC_LOGIC_ERROR;
    const orc_node* n = &m->nodes[idx];
    float sum = 0.0f;
    for (int i = 0; i < ORC_MOVES; i++) {
        if (n->valid & (1ULL << i)) {
            float prob = (float)pow((double)n->N[i], 1.0 / 1.0f);
            pi[i] = prob;
            sum += prob;
        } else pi[i] = 0.0f;
    }
    for (int i = 0; i < ORC_MOVES; i++) pi[i] /= sum;
    return ORC_OK;
}

/* AlphaZeroMCTS::pickHigestWeightedMove (:397-412) */
int orc_pick_highest(const float* pi)
{
    float best = 0.0f;
    int li = ORC_NONE;
    for (int i = 0; i < ORC_MOVES; i++)
        if (pi[i] > best) { best = pi[i]; li = i; }
    return li;
}

/* AlphaZeroMCTS::pickRandomWeightedMove (:379-395) */
int orc_pick_random(const float* pi, orc_rng* r)
{
    float sum = 0.0f;
    for (int i = 0; i < ORC_MOVES; i++) sum += pi[i];
    float ra = sum * orc_rng_float(r);
    float it = 0.0f;
    for (int i = 0; i < ORC_MOVES; i++) {
        it += pi[i];
        if (it >= ra) return i;
    }
    return ORC_NONE; /* throws invalid_argument in the reference */
}

/* AlphaZeroTrainer::threadExecuteTrainingGame, one game (alphazero_trainer.cpp:80-119).  The
 * reference draws dice, newGame picks and sampled moves from ONE global engine; with one game at a
 * time that is this per-game stream seeded at game start. */
int orc_selfplay_game(const orc_settings* cfg, uint32_t seed, orc_eval_fn eval, void* ctx,
                      uint8_t* rec265, int cap, int* status, int* rounds, uint8_t* moves_out, int max_decisions,
                      uint64_t* sims_out, uint64_t* evals_out)
{
    orc_rng r;
    orc_rng_seed(&r, seed);
    orc_mcts* m = orc_mcts_create(cfg);
    orc_state s;
    orc_new_game(&s, &r);
    int gs = ORC_NOT_ENDED, n = 0;
    int8_t* players = (int8_t*)malloc((size_t)(cap > 0 ? cap : 1));
    for (int i = 0; gs == ORC_NOT_ENDED; i++) {
        if (max_decisions > 0 && i >= max_decisions) break;
        if (orc_mcts_simulate(m, &s, &r, eval, ctx)) { n = -1; break; }
        float pi[ORC_MOVES];
        if (orc_mcts_policy(m, &s, pi)) { n = -1; break; }
        int li = s.round > cfg->temperature_threshold ? orc_pick_highest(pi) : orc_pick_random(pi, &r);
        if (n < cap) {
            uint8_t* rec = rec265 + (size_t)n * 265;
            rec[0] = (uint8_t)s.cur;
            orc_encode(&s, rec + 1);
            memset(rec + 89, 0, 4);
            memcpy(rec + 93, pi, 43 * 4);
            players[n] = s.cur;
            if (moves_out) moves_out[n] = (uint8_t)li;
            n++;
        }
        if (orc_make_move(&s, li, &r, cfg)) { n = -1; break; }
        gs = orc_game_status(&s, cfg);
    }
    if (n >= 0 && gs != ORC_NOT_ENDED) {
        float* z = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
        orc_update_values(players, n, gs, z);
        for (int i = 0; i < n; i++) memcpy(rec265 + (size_t)i * 265 + 89, &z[i], 4);
        free(z);
    }
    if (status) *status = gs;
    if (rounds) *rounds = s.round;
    if (sims_out) *sims_out = m->sims;
    if (evals_out) *evals_out = m->evals;
    free(players);
    orc_mcts_destroy(m);
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * cpu_baseline leg of bench.py: `threads` independent self-play games (one per thread, t = 1, fp32 CPU net),
 * each playing `decisions` decisions of the trainer's move loop from newGame.  Returns total completed
 * simulations and wall seconds.
 * ---------------------------------------------------------------------------------------------- */
#include <time.h>
typedef struct { const orc_settings* cfg; const orc_net* net; uint32_t seed; int decisions; uint64_t sims, evals; } bench_job;
static void* bench_thread(void* a)
{
    bench_job* j = (bench_job*)a;
    uint8_t* rec = (uint8_t*)malloc((size_t)265 * (size_t)(j->decisions + 1));
    int st, rounds;
    orc_selfplay_game(j->cfg, j->seed, orc_net_eval, (void*)j->net, rec, j->decisions + 1, &st, &rounds, NULL,
                      j->decisions, &j->sims, &j->evals);
    free(rec);
    return NULL;
}
int orc_bench_selfplay(const orc_settings* cfg, const orc_net* net, uint32_t base_seed, int threads, int decisions,
                       uint64_t* sims, uint64_t* evals, double* seconds)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256];
    bench_job jobs[256];
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; t++) {
        jobs[t] = (bench_job){cfg, net, base_seed + (uint32_t)t, decisions, 0, 0};
        pthread_create(&th[t], NULL, bench_thread, &jobs[t]);
    }
    uint64_t s = 0, e = 0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); s += jobs[t].sims; e += jobs[t].evals; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *sims = s; *evals = e;
    *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    return 0;
}

/* ================================================================================================
 * Opponent players and the host game driver: ScriptPlayer (player/script/script_player.cpp),
 * RandomPlayer (player/random/random_player.cpp), Game / GameResults (game/game.cpp:101-235).  PINNED against the
 * real reference (these units are TensorFlow-free and are part of oracle/_ref).
 * ============================================================================================== */

/* LandSet constructor orders (land/land_set.cpp:12-33); priority vector initial order of ScriptPlayer() is
 * ASIA, NORTH_AMERICA, SOUTH_AMERICA, EUROPE, AFRICA, AUSTRALIA (script_player.cpp:12-14) */
static const uint8_t SET_LANDS[6][12] = {
    {26, 33, 35, 36, 27, 28, 29, 30, 31, 32, 34, 37},      /* ASIA */
    {0, 1, 2, 3, 4, 5, 6, 7, 8},                           /* NORTH_AMERICA */
    {9, 10, 11, 12},                                       /* SOUTH_AMERICA */
    {13, 14, 15, 16, 17, 19, 18},                          /* EUROPE */
    {20, 21, 22, 24, 25, 23},                              /* AFRICA */
    {38, 39, 40, 41},                                      /* AUSTRALIA */
};
static const uint8_t SET_COUNT[6] = {12, 9, 4, 7, 6, 4};
static const uint64_t SET_MASK[6] = {0x3ffc000000ULL, 0x1ffULL, 0x1e00ULL, 0xfe000ULL, 0x3f00000ULL, 0x3c000000000ULL};

int orc_landset_lands(int set, uint8_t* out12) { memcpy(out12, SET_LANDS[set], SET_COUNT[set]); return SET_COUNT[set]; }

void orc_script_init(orc_script* p)
{
    memset(p, 0, sizeof *p);
    for (int i = 0; i < 6; i++) p->order[i] = (uint8_t)i;
    p->attacking_set = -1;
    p->land_to = ORC_NONE;
    p->land_from = ORC_NONE;
}

/* GameHelper::sortLandSet (game_helper.cpp:19-39): strict weak order that is total (mask tie-break) */
static int landset_before(const orc_script* p, int a, int b)
{
    if (p->not_owned[a] == p->not_owned[b]) {
        if (p->not_owned_attack[a] == p->not_owned_attack[b]) return SET_MASK[a] > SET_MASK[b];
        return p->not_owned_attack[a] > p->not_owned_attack[b];
    }
    return p->not_owned[a] < p->not_owned[b];
}

/* ScriptPlayer::updateAttackLandSetPriority / updateAttackLandSet / updateAttackLandTo / updateAttackLandFrom
 * (script_player.cpp:17-80) */
static void script_update(orc_script* p, const orc_state* s)
{
    const orc_player* ps = &s->ps[s->cur];
    for (int k = 0; k < 6; k++) {
        uint64_t no = SET_MASK[k] & ~ps->owned;
        p->not_owned[k] = (uint8_t)popc(no);
        p->not_owned_attack[k] = (uint8_t)popc(no & p->attack_mask);
    }
    for (int i = 1; i < 6; i++) { /* std::sort on 6 elements == insertion sort; the order is total */
        uint8_t v = p->order[i];
        int j = i;
        while (j > 0 && landset_before(p, v, p->order[j - 1])) { p->order[j] = p->order[j - 1]; j--; }
        p->order[j] = v;
    }
    for (int i = 0; i < 6; i++)
        if (p->not_owned_attack[p->order[i]] > 0) { p->attacking_set = p->order[i]; break; }
    if (p->attacking_set >= 0) {
        int k = p->attacking_set;
        for (int i = 0; i < SET_COUNT[k]; i++)
            if (((1ULL << SET_LANDS[k][i]) & p->attack_mask) > 0) { p->land_to = SET_LANDS[k][i]; break; }
    }
    p->attack_from_army = 0;
    if (p->land_to != ORC_NONE)
        for (int i = 0; i < LAND[p->land_to].n; i++) {
            int nl = LAND[p->land_to].nb[i];
            if ((1ULL << nl) & p->owned_attack_mask) {
                if (s->army[nl] > p->attack_from_army) { p->attack_from_army = s->army[nl]; p->land_from = nl; }
            }
        }
}

/* ScriptPlayer::attackLand (script_player.cpp:82-135) */
static int script_attack_land(orc_script* p, orc_state* s, orc_rng* r, const orc_settings* cfg)
{
    if (p->land_from == ORC_NONE || p->land_to == ORC_NONE) return ORC_LOGIC_ERROR; /* null deref in the reference */
    while (s->reinf > 0) {
        const orc_player* ps = &s->ps[s->cur];
        uint64_t not_full = ps->owned & ~ps->owned_full;
        int to = p->land_from;
        if (((1ULL << p->land_from) & not_full) == 0) {
            uint64_t nb = nb_mask(p->land_to) & not_full;
            if (nb > 0) to = ctz(nb);
            else {
                nb = not_full & (cenemy_ps(s)->attack | neutral_attack_lands(s));
                if (nb > 0) to = ctz(nb);
                else to = ctz(not_full); /* not_full == 0: lm2li(garbage) in the reference */
            }
        }
        uint8_t max_reinf = (uint8_t)(ORC_ARMY_MAX - s->army[to]);
        uint8_t reinforcement = max_reinf < s->reinf ? max_reinf : s->reinf;
        if (reinforcement == 0) return ORC_LOGIC_ERROR; /* the reference would spin forever */
        while (reinforcement > 0) {
            uint8_t step = (int)reinforcement < cfg->min_unit_move ? reinforcement : (uint8_t)cfg->min_unit_move;
            TRY(reinforcement_move(s, step, to));
            reinforcement = (uint8_t)(reinforcement - step);
        }
    }
    p->attack_from_army = s->army[p->land_from];
    while (p->attack_from_army > 1) {
        const uint8_t owner_before = s->owner[p->land_to];
        TRY(attack_move(s, p->land_from, p->land_to, r));
        int captured = s->owner[p->land_to] != owner_before; /* attackMove's return value */
        p->attack_from_army = s->army[p->land_from];
        if (captured && p->attack_from_army > 1) {
            uint8_t max_move = (uint8_t)(p->attack_from_army - 1);
            while (max_move > 0) {
                uint8_t step = (int)max_move < cfg->min_unit_move ? max_move : (uint8_t)cfg->min_unit_move;
                max_move = (uint8_t)(max_move - step);
                TRY(attack_reinforcement_move(s, step));
            }
            break;
        }
    }
    return ORC_OK;
}

/* GameHelper::PlayerMovement (game_helper.cpp:51-109): per owned component in discovery order — fortify-from = the
 * land with no non-owned neighbour and the largest army (first wins), fortify-to = the land with most non-owned
 * neighbours (first wins); components then std::sort'ed by fortify-from amount, descending.  libstdc++ sorts fewer
 * than 17 elements by insertion sort (stable); more than 16 owned components cannot occur in practice. */
typedef struct { uint64_t mask; int from, to; uint8_t from_amount, to_nb; } orc_lsm;

static void lsm_flood(const orc_state* s, int land, uint64_t owned, orc_lsm* c)
{
    if (((1ULL << land) & owned & ~c->mask) > 0) {
        c->mask |= 1ULL << land;
        uint64_t att = ~owned & nb_mask(land);
        if (att == 0) {
            if (s->army[land] > c->from_amount) { c->from = land; c->from_amount = s->army[land]; }
        } else {
            uint8_t cnt = (uint8_t)popc(att);
            if (cnt > c->to_nb) { c->to_nb = cnt; c->to = land; }
        }
        for (int i = 0; i < LAND[land].n; i++) lsm_flood(s, LAND[land].nb[i], owned, c);
    }
}

static int player_movement(const orc_state* s, orc_lsm* out)
{
    uint64_t owned = s->ps[s->cur].owned, covered = 0;
    int n = 0;
    for (int i = 0; i < ORC_LANDS; i++)
        if (((1ULL << i) & owned & ~covered) > 0) {
            orc_lsm c = {0, ORC_NONE, ORC_NONE, 0, 0};
            lsm_flood(s, i, owned, &c);
            covered |= c.mask;
            out[n++] = c;
        }
    for (int i = 1; i < n; i++) { /* stable insertion sort, descending by from_amount */
        orc_lsm v = out[i];
        int j = i;
        while (j > 0 && v.from_amount > out[j - 1].from_amount) { out[j] = out[j - 1]; j--; }
        out[j] = v;
    }
    return n;
}

/* ScriptPlayer::takeTurn (script_player.cpp:162-227) */
int orc_script_take_turn(orc_script* p, orc_state* s, orc_rng* r, const orc_settings* cfg)
{
    const orc_player* ps = &s->ps[s->cur];
    if (s->phase == ORC_SETUP) {
        p->owned_attack_mask = ps->owned;
        p->attack_mask = ps->attack;
        script_update(p, s);
        if (p->land_from == ORC_NONE) return ORC_LOGIC_ERROR;
        TRY(setup_reinforcement_move(s, p->land_from));
        const orc_player* eps = cenemy_ps(s);
        uint64_t neutral = ALL_LANDS & ~ps->owned & ~eps->owned;
        uint64_t nte = neutral & eps->attack & ~ps->attack;
        if (nte == 0) nte = neutral & eps->attack;
        uint64_t pick = popc(nte) > 0 ? orc_random_mask(r, nte) : orc_random_mask(r, neutral);
        return setup_reinforcement_neutral_move(s, ctz(pick));
    }
    p->owned_attack_mask = ps->owned;
    p->attack_mask = ps->attack;
    play_cards(s);
    while (p->attack_mask > 0 || s->reinf > 0) {
        script_update(p, s);
        TRY(script_attack_land(p, s, r, cfg));
        p->owned_attack_mask = ps->owned_army;
        p->attack_mask = ps->attack_army;
    }
    /* fortify (script_player.cpp:138-160) */
    if (popc(ps->owned_army) > 0) {
        orc_lsm comps[ORC_LANDS];
        int n = player_movement(s, comps);
        if (n > 0 && comps[0].from_amount > 0 && comps[0].to != ORC_NONE) {
            uint8_t amount = (uint8_t)(s->army[comps[0].from] - 1);
            uint8_t max_amount = (uint8_t)(ORC_ARMY_MAX - s->army[comps[0].to]);
            amount = amount < max_amount ? amount : max_amount;
            TRY(fortify_move(s, amount, comps[0].from, comps[0].to));
        }
    }
    next_player_game_turn(s);
    return ORC_OK;
}

/* RandomPlayer::pickRandomMove == Utility::randomMask with a throw on the empty set (random_player.cpp:3-20) */
static int pick_random_move(orc_rng* r, uint64_t moves, uint64_t* out)
{
    if (popc(moves) == 0) return ORC_INVALID_ARGUMENT;
    *out = orc_random_mask(r, moves);
    return ORC_OK;
}

/* RandomPlayer::takeTurn (random_player.cpp:22-111) */
int orc_random_take_turn(orc_state* s, int me, orc_rng* r, const orc_settings* cfg)
{
    while (orc_game_status(s, cfg) == ORC_NOT_ENDED && s->cur == me) {
        const orc_player* ps = &s->ps[s->cur];
        const orc_player* eps = cenemy_ps(s);
        uint64_t mv;
        if (s->phase == ORC_SETUP) {
            TRY(pick_random_move(r, ps->owned, &mv));
            TRY(setup_reinforcement_move(s, ctz(mv)));
        } else if (s->phase == ORC_SETUP_NEUTRAL) {
            TRY(pick_random_move(r, ALL_LANDS & ~ps->owned & ~eps->owned, &mv));
            TRY(setup_reinforcement_neutral_move(s, ctz(mv)));
        } else if (s->phase == ORC_REINFORCEMENT) {
            play_cards(s);
            TRY(pick_random_move(r, ps->owned & ~ps->owned_full, &mv));
            TRY(reinforcement_move(s, 1, ctz(mv)));
        } else if (s->phase == ORC_ATTACK) {
            TRY(pick_random_move(r, ps->attack_army | SKIP_MASK, &mv));
            if ((SKIP_MASK & mv) > 0) { TRY(goto_fortify(s)); }
            else {
                int to = ctz(mv);
                uint64_t from;
                TRY(pick_random_move(r, nb_mask(to) & ps->owned_army, &from));
                TRY(attack_move(s, ctz(from), to, r));
            }
        } else if (s->phase == ORC_ATTACK_MOBILIZATION) {
            if (orc_rng_float(r) > 0.5f) {
                int v = (int)s->army[s->mob_from] - 1;
                uint8_t amount = (uint8_t)(v < cfg->min_unit_move ? v : cfg->min_unit_move);
                TRY(attack_reinforcement_move(s, amount));
            } else { TRY(goto_attack(s)); }
        } else if (s->phase == ORC_FORTIFY) {
            TRY(pick_random_move(r, (ps->owned & ~ps->owned_full) | SKIP_MASK, &mv));
            if (mv != SKIP_MASK) {
                int to = ctz(mv);
                orc_lsm comps[ORC_LANDS];
                int n = player_movement(s, comps);
                for (int i = 0; i < n; i++)
                    if ((comps[i].mask & mv) > 0) {
                        uint64_t with_army = comps[i].mask & ~mv & ps->owned_army;
                        if (with_army > 0) {
                            uint64_t fm;
                            TRY(pick_random_move(r, with_army, &fm));
                            int from = ctz(fm);
                            uint8_t amount = (uint8_t)(s->army[from] - 1);
                            uint8_t max_amount = (uint8_t)(ORC_ARMY_MAX - s->army[to]);
                            amount = max_amount < amount ? max_amount : amount;
                            if (amount == 0) return ORC_LOGIC_ERROR; /* rInt() % 0 in the reference */
                            uint64_t ra = (uint64_t)(orc_rng_int(r) % amount);
                            TRY(fortify_move(s, (uint8_t)ra, from, to));
                        }
                        break;
                    }
            }
            next_player_game_turn(s);
        }
    }
    return ORC_OK;
}

/* AlphaZeroPlayer::takeTurn (alphazero_player.cpp:3-21) at one search thread */
/* optional (s, pi, player) log of the AlphaZero decisions of the running game (alphazero_player.cpp:15-18) */
typedef struct { uint8_t* rec; int cap, n, game_start; } orc_reclog;

static int az_take_turn(orc_mcts* m, orc_state* s, int me, orc_rng* r, const orc_settings* cfg, orc_eval_fn eval, void* ctx,
                        orc_reclog* log)
{
    orc_mcts_trim(m);
    while (orc_game_status(s, cfg) == ORC_NOT_ENDED && s->cur == me) {
        TRY(orc_mcts_simulate(m, s, r, eval, ctx));
        float pi[ORC_MOVES];
        TRY(orc_mcts_policy(m, s, pi));
        int li = orc_pick_highest(pi);
        if (log && log->rec && log->n < log->cap) {
            uint8_t* d = log->rec + (size_t)log->n * 265;
            d[0] = (uint8_t)s->cur;
            orc_encode(s, d + 1);
            memset(d + 89, 0, 4);
            memcpy(d + 93, pi, 43 * 4);
            log->n++;
        }
        TRY(orc_make_move(s, li, r, cfg));
    }
    return ORC_OK;
}

/* One thread of GameGroup::playGames (game.cpp:238-254) = Game::playGames(1) repeated `games` times with
 * alternating starts and mirrored pairs (Game::newGame, game.cpp:170-191).  kinds: 0 = AlphaZero (eval), 1 = Script,
 * 2 = Random.  One RNG stream for everything (the reference's global engine).  Optional per-game outputs. */
int orc_play_games(const orc_settings* cfg, int kind0, int kind1, int games, int mirror, uint32_t seed,
                   orc_eval_fn eval, void* ctx, orc_results* res, int8_t* status_out, uint8_t* finals160,
                   uint16_t* rounds_out)
{
    return orc_play_games2(cfg, kind0, kind1, games, mirror, seed, eval, ctx, eval, ctx, res, status_out, finals160, rounds_out,
                           NULL, 0, NULL, NULL);
}

/* Game::gameLoop + the bookkeeping after it (game.cpp:101-168) for ONE game that starts in state *s: both players'
 * newGame, turns until the game ends, updateValues on the records logged during the game, GameResults::addGame. */
typedef struct {
    const orc_settings* cfg;
    int kind[2];
    orc_script sp[2];
    orc_mcts* mc[2];
    orc_eval_fn eval, eval_b;
    void *ctx, *ctx_b;
    orc_reclog log;
} orc_table;

static int play_out(orc_table* t, orc_state* s, orc_rng* r, int player_start, int gi, orc_results* res, int8_t* status_out,
                    uint8_t* finals160, uint16_t* rounds_out, int* rec_game_end)
{
    const orc_settings* cfg = t->cfg;
    int rc = ORC_OK;
    for (int p = 0; p < 2; p++)
        if (t->mc[p]) orc_mcts_clear(t->mc[p]); /* AlphaZeroPlayer::newGame */
    int gs = ORC_NOT_ENDED;
    while (gs == ORC_NOT_ENDED && rc == ORC_OK) { /* Game::gameLoop / playTurn (game.cpp:101-133) */
        int cur = s->cur;
        int setup = s->phase == ORC_SETUP;
        if (t->kind[cur] == 1) rc = orc_script_take_turn(&t->sp[cur], s, r, cfg);
        else if (t->kind[cur] == 2) rc = orc_random_take_turn(s, cur, r, cfg);
        else if (t->kind[cur] == 3) rc = az_take_turn(t->mc[cur], s, cur, r, cfg, t->eval_b, t->ctx_b, &t->log);
        else rc = az_take_turn(t->mc[cur], s, cur, r, cfg, t->eval, t->ctx, &t->log);
        if (rc) break;
        gs = setup ? ORC_NOT_ENDED : orc_game_status(s, cfg);
        if (cur == s->cur && gs == ORC_NOT_ENDED) { rc = ORC_LOGIC_ERROR; break; } /* "Turn was not incremented" */
    }
    if (rc) return rc;
    for (int i = t->log.game_start; i < t->log.n; i++) { /* updateValues (alphazero_nn_data.cpp:51-65) */
        float z = gs == ORC_DRAW ? 0.0f : ((int)t->log.rec[(size_t)i * 265] == gs ? 1.0f : -1.0f);
        memcpy(t->log.rec + (size_t)i * 265 + 89, &z, 4);
    }
    t->log.game_start = t->log.n;
    if (rec_game_end) rec_game_end[gi] = t->log.n;
    res->count++;
    if (gs == ORC_DRAW) res->draw++;
    for (int p = 0; p < 2; p++)
        if (gs == p) { res->win[p]++; if (player_start == p) res->win_started[p]++; }
    if (status_out) status_out[gi] = (int8_t)gs;
    if (finals160) orc_state_pack(s, finals160 + (size_t)gi * 160);
    if (rounds_out) rounds_out[gi] = s->round;
    return ORC_OK;
}

static void table_open(orc_table* t, const orc_settings* cfg, int kind0, int kind1, orc_eval_fn eval, void* ctx, orc_eval_fn eval_b,
                       void* ctx_b, uint8_t* rec265, int rec_cap)
{
    t->cfg = cfg;
    t->kind[0] = kind0; t->kind[1] = kind1;
    t->eval = eval; t->ctx = ctx; t->eval_b = eval_b; t->ctx_b = ctx_b;
    t->log.rec = rec265; t->log.cap = rec_cap; t->log.n = 0; t->log.game_start = 0;
    for (int p = 0; p < 2; p++) {
        orc_script_init(&t->sp[p]);
        t->mc[p] = (t->kind[p] == 0 || t->kind[p] == 3) ? orc_mcts_create(cfg) : NULL;
    }
}

static void table_close(orc_table* t)
{
    for (int p = 0; p < 2; p++)
        if (t->mc[p]) orc_mcts_destroy(t->mc[p]);
}

/* the same with kind 3 = an AlphaZero player on a second network (eval_b) — GameGroup::playGames(trainAZPG,
 * generateAZPG, ...) of the trainer — and optionally the 265-byte (s, pi, z) records of the AlphaZero decisions, game
 * by game in decision order, z filled in when the game ends (NNTrainDataStorage::updateValues) */
int orc_play_games2(const orc_settings* cfg, int kind0, int kind1, int games, int mirror, uint32_t seed,
                    orc_eval_fn eval, void* ctx, orc_eval_fn eval_b, void* ctx_b, orc_results* res, int8_t* status_out,
                    uint8_t* finals160, uint16_t* rounds_out, uint8_t* rec265, int rec_cap, int* rec_n, int* rec_game_end)
{
    orc_table t;
    table_open(&t, cfg, kind0, kind1, eval, ctx, eval_b, ctx_b, rec265, rec_cap);
    orc_rng r;
    orc_rng_seed(&r, seed);
    memset(res, 0, sizeof *res);
    orc_state s, prev_start;
    orc_state_blank(&prev_start);
    int player_start = 0, rc = ORC_OK;
    for (int gi = 0; gi < games && rc == ORC_OK; gi++) {
        if (mirror && player_start != 0) {
            s = prev_start;
            orc_invert_players(&s);
            s.cur = (int8_t)player_start;
        } else {
            orc_new_game(&s, &r);
            s.cur = (int8_t)player_start;
            prev_start = s;
        }
        rc = play_out(&t, &s, &r, player_start, gi, res, status_out, finals160, rounds_out, rec_game_end);
        player_start = (player_start + 1) % 2;
    }
    table_close(&t);
    res->rng_state = r.x;
    if (rec_n) *rec_n = t.log.n;
    return rc;
}

/* One SLOT of the concurrent-halves form of a mirrored arena (include/azr.h, AZR_MIRROR_CONCURRENT): the two games of a pair are
 * played at the same time by two tables — nothing in Game orders them (game.cpp:238-254), they only share the deal
 * (game.cpp:170-191).  This slot plays half `half` of the pairs pair_seed0, pair_seed0 + pair_stride, ...: the deal of pair seed q
 * comes from minstd_rand0(q); half 0 (player 0 starts) goes on with that stream, half 1 plays invertPlayers of the deal with
 * player 1 starting and draws from minstd_rand0(q + 2^30).  Players (and a ScriptPlayer's memory) persist over the slot's games
 * as they do over a reference thread's. */
int orc_play_half_games(const orc_settings* cfg, int kind0, int kind1, int games, int half, uint32_t pair_seed0, uint32_t pair_stride,
                        orc_eval_fn eval, void* ctx, orc_eval_fn eval_b, void* ctx_b, orc_results* res, int8_t* status_out,
                        uint8_t* finals160, uint16_t* rounds_out, uint8_t* rec265, int rec_cap, int* rec_n, int* rec_game_end)
{
    orc_table t;
    table_open(&t, cfg, kind0, kind1, eval, ctx, eval_b, ctx_b, rec265, rec_cap);
    memset(res, 0, sizeof *res);
    orc_rng r;
    r.x = 0;
    int rc = ORC_OK;
    for (int gi = 0; gi < games && rc == ORC_OK; gi++) {
        const uint32_t q = pair_seed0 + (uint32_t)gi * pair_stride;
        orc_state s;
        orc_rng_seed(&r, q);
        orc_new_game(&s, &r);
        if (half) {
            orc_invert_players(&s);
            orc_rng_seed(&r, q + (1u << 30));
        }
        s.cur = (int8_t)half;
        rc = play_out(&t, &s, &r, half, gi, res, status_out, finals160, rounds_out, rec_game_end);
    }
    table_close(&t);
    res->rng_state = r.x;
    if (rec_n) *rec_n = t.log.n;
    return rc;
}
