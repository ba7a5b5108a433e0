// TEST INFRASTRUCTURE — not product code.
//
// C-ABI harness around the *real* reference implementation, compiled from the
// sources where they lie under /root/reference (never copied into this repo) by
// oracle/Makefile into oracle/_ref/libazr_ref.so.  It is used to
//   (1) generate the golden vectors under tests/golden/ (tests/golden/make_golden.py), and
//   (2) pin the plain-C restatement in oracle/azr_oracle.c against the reference in this
//       container (tests/test_oracle_vs_ref.py; skipped where oracle/_ref is absent).
//
// Only the TensorFlow-free translation units of the reference are linked (land, land_set,
// state, game_helper, alphazero_moves, alphazero_nn_data, player, game + xxhash).  The MCTS /
// NN-service units (alphazero_mcts.cpp, alphazero_nn.cpp, alphazero_gpu_cluster.cpp, ...)
// include TensorFlow headers which this image lacks, so they are unbuildable here and are
// covered by the restatement only (DESIGN.md "parity unpinned" rows).
//
// Everything in this file is harness glue written for this repo; it calls the reference's
// public API:  State (state/state.h:109-218), UtilityNN::getValidMoves/makeMove
// (player/alpha_zero/alphazero_moves.cpp:3-233), NNInputData / NNOutputData::normalize /
// NNTrainDataStorage::updateValues (neural_network/alphazero_nn_data.cpp), Rng (src/rng.h).

#include <cstring>
#include <cstdint>
#include <stdexcept>
#include <sstream>
#include <cstdio>
#include <algorithm>

#include "risk_game/player/alpha_zero/alphazero_moves.h"
#include "risk_game/player/script/script_player.h"
#include "risk_game/player/random/random_player.h"

static_assert(sizeof(Data) == 160, "reference Data layout changed");
static_assert(sizeof(NNInputData) == 88, "reference NNInputData layout changed");

static char g_err[512];

static void load_state(State& s, const void* data160)
{
    // Data is the first member of State; getData() hands back a reference to it.
    std::memcpy(const_cast<Data*>(&s.getData()), data160, sizeof(Data));
}

static void store_state(const State& s, void* data160)
{
    std::memcpy(data160, &s.getData(), sizeof(Data));
}

extern "C" {

const char* ref_last_error() { return g_err; }

int ref_sizeof_data() { return (int)sizeof(Data); }
int ref_sizeof_input() { return (int)sizeof(NNInputData); }

// ---- settings that the rules read (src/settings.h:41-56) -------------------------------
void ref_set_rules(int allow_yield, int limit_reinforcement, int limit_attack, int max_rounds, int min_unit_move)
{
    SETTINGS.ALLOW_YIELD = allow_yield != 0;
    SETTINGS.LIMIT_REINFORCEMENT_MOVES = limit_reinforcement != 0;
    SETTINGS.LIMIT_ATTACK_MOVES = limit_attack != 0;
    SETTINGS.MAX_GAME_ROUNDS = max_rounds;
    SETTINGS.MIN_UNIT_MOVE = min_unit_move;
}

// ---- RNG (src/rng.h:5-50) -----------------------------------------------------------------
void ref_seed(uint32_t seed) { RNG.getEngine().seed(seed); }
int ref_rng_dice() { return RNG.rDice(); }
int ref_rng_int() { return RNG.rInt(); }
float ref_rng_float() { return RNG.rFloat(); }
// raw engine state, so a restatement with per-game streams can be lined up with the global one
uint32_t ref_rng_state()
{
    std::ostringstream os; os << RNG.getEngine();
    return (uint32_t)std::stoul(os.str());
}
uint64_t ref_random_mask(uint64_t masks) { return Utility::randomMask(masks); }

// ---- static tables (land/land.cpp:246-297, land/land_set.cpp:12-33) -----------------------------
int ref_neighbours(int land, uint8_t* out)
{
    const Land* l = Land::getLand((uint8_t)land);
    int n = (int)l->neihboursLandIndex.size();
    for (int i = 0; i < n; i++) out[i] = Utility::li2i(l->neihboursLandIndex[i]);
    return n;
}
uint64_t ref_neighbour_mask(int land) { return Land::getLand((uint8_t)land)->neighboursLandIndexBitMask; }
uint64_t ref_continent_mask(int c)
{
    switch (c) {
    case 0: return LandSet::NORTH_AMERICA.landSetIndexBitMask;
    case 1: return LandSet::SOUTH_AMERICA.landSetIndexBitMask;
    case 2: return LandSet::AFRICA.landSetIndexBitMask;
    case 3: return LandSet::EUROPE.landSetIndexBitMask;
    case 4: return LandSet::ASIA.landSetIndexBitMask;
    case 5: return LandSet::AUSTRALIA.landSetIndexBitMask;
    default: return LandSet::ALL_LANDS_MASK;
    }
}

// ---- rules ------------------------------------------------------------------------------------
void ref_new_game(void* data160)
{
    State s;
    s.newGame();
    store_state(s, data160);
}

// a default-constructed State (what `State rootState = State()` holds before newGame)
void ref_blank_state(void* data160)
{
    State s;
    store_state(s, data160);
}

uint64_t ref_valid_moves(const void* data160)
{
    State s; load_state(s, data160);
    return UtilityNN::getValidMoves(s);
}

int ref_game_status(const void* data160)
{
    State s; load_state(s, data160);
    return s.gameStatus();
}

int ref_reinforcement_value(uint64_t owned)
{
    State s;
    return s.calculateReinforcementValue(owned);
}

// returns 0 on success; 1 = std::invalid_argument, 2 = std::logic_error, 3 = other
int ref_make_move(void* data160, int move)
{
    State s; load_state(s, data160);
    try {
        UtilityNN::makeMove(s, Utility::i2li((uint8_t)move));
    } catch (const std::invalid_argument& e) {
        std::snprintf(g_err, sizeof g_err, "invalid_argument: %s", e.what()); return 1;
    } catch (const std::logic_error& e) {
        std::snprintf(g_err, sizeof g_err, "logic_error: %s", e.what()); return 2;
    } catch (...) {
        std::snprintf(g_err, sizeof g_err, "unknown exception"); return 3;
    }
    store_state(s, data160);
    return 0;
}

void ref_invert_players(void* data160)
{
    State s; load_state(s, data160);
    s.invertPlayers();
    store_state(s, data160);
}

// ---- NN data seams (alphazero_nn_data.cpp:3-27,51-65,165-196) -------------------------------
void ref_encode(const void* data160, void* in88)
{
    State s; load_state(s, data160);
    NNInputData in(s);
    // zero the destination first so struct padding is deterministic in the fixture
    std::memset(in88, 0, sizeof(NNInputData));
    NNInputData* o = (NNInputData*)in88;
    std::memcpy(o->land, in.land, sizeof(in.land));
    o->playerIndex = in.playerIndex;
    o->round = in.round;
    o->featureReinforcementShare = in.featureReinforcementShare;
    o->featureAttackFrequency = in.featureAttackFrequency;
    o->featureCanDrawCard = in.featureCanDrawCard;
    o->featureIsPhaseSetup = in.featureIsPhaseSetup;
    o->featureIsPhaseSetupNeutral = in.featureIsPhaseSetupNeutral;
    o->featureIsPhaseReinforcement = in.featureIsPhaseReinforcement;
    o->featureIsPhaseAttack = in.featureIsPhaseAttack;
    o->featureIsPhaseAttackMobilization = in.featureIsPhaseAttackMobilization;
    o->featureIsPhaseFortify = in.featureIsPhaseFortify;
    o->featureArmyShare = in.featureArmyShare;
}

void ref_normalize(float* pi43, uint64_t valid)
{
    NNOutputData out;
    out.policy.assign(pi43, pi43 + TF_OUTPUT_POLICY_TENSOR_SIZE);
    out.normalize(valid);
    std::memcpy(pi43, out.policy.data(), sizeof(float) * TF_OUTPUT_POLICY_TENSOR_SIZE);
}

// z back-fill: n records with playerIndex[i]; returns z[i]
void ref_update_values(const int8_t* player_index, int n, int game_status, int round_count, float* z)
{
    NNTrainDataStorage st;
    for (int i = 0; i < n; i++) {
        NNTrainData d; d.playerIndex = player_index[i];
        st.data.push_back(d);
    }
    st.updateValues(game_status, round_count);
    for (int i = 0; i < n; i++) z[i] = st.data[i].out.value;
}

// ---- one whole seeded random-policy game (used for bulk pinning; moves via Utility::randomMask) ----
// Plays from newGame with moves drawn by randomMask(valid) from the SAME global RNG stream that
// the dice use.  Writes up to `cap` steps: states160[cap*160] (state BEFORE the move),
// masks[cap], moves[cap].  Returns number of steps, *status = final gameStatus, final160 = last state.
int ref_play_random_game(uint32_t seed, int cap, uint8_t* states160, uint64_t* masks, uint8_t* moves,
                         int* status, void* final160)
{
    ref_seed(seed);
    State s;
    s.newGame();
    int n = 0;
    int st = s.gameStatus();
    while (st == State::NOT_ENDED && n < cap) {
        uint64_t vm = UtilityNN::getValidMoves(s);
        uint64_t m = Utility::randomMask(vm);
        uint8_t mv = Utility::lm2i(m);
        if (states160) store_state(s, states160 + (size_t)n * sizeof(Data));
        if (masks) masks[n] = vm;
        if (moves) moves[n] = mv;
        n++;
        try {
            UtilityNN::makeMove(s, Utility::i2li(mv));
        } catch (const std::exception& e) {
            std::snprintf(g_err, sizeof g_err, "exception at step %d: %s", n, e.what());
            *status = -100;
            store_state(s, final160);
            return n;
        }
        st = s.gameStatus();
    }
    *status = st;
    store_state(s, final160);
    return n;
}


// ---- opponents + host game driver (player/script, player/random, game/game.cpp): one thread of GameGroup::playGames ----
// kinds: 1 = ScriptPlayer, 2 = RandomPlayer.  results6 = {count, draw, win0, winStarted0, win1, winStarted1}.
int ref_play_games(int kind0, int kind1, int games, int mirror, uint32_t seed, int* results6, int8_t* status,
                   uint8_t* finals160, uint16_t* rounds, uint32_t* rng_state)
{
    SETTINGS.MIRROR_GAMES = mirror != 0;
    ref_seed(seed);
    auto mk = [](int kind) -> std::shared_ptr<Player> {
        if (kind == 1) return std::shared_ptr<Player>(new ScriptPlayer());
        return std::shared_ptr<Player>(new RandomPlayer());
    };
    Game game;
    game.addPlayer(mk(kind0));
    game.addPlayer(mk(kind1));
    GameResults total;
    try {
        for (int i = 0; i < games; i++) {
            GameResults gr = game.playGames(1);
            total.add(gr);
            if (status) status[i] = (int8_t)game.state.gameStatus();
            if (finals160) store_state(game.state, finals160 + (size_t)i * sizeof(Data));
            if (rounds) rounds[i] = game.state.getRound();
        }
    } catch (const std::exception& e) {
        std::snprintf(g_err, sizeof g_err, "exception: %s", e.what());
        return 1;
    }
    results6[0] = total.count; results6[1] = total.draw;
    results6[2] = total.players[0].win; results6[3] = total.players[0].winAndStartedGame;
    results6[4] = total.players[1].win; results6[5] = total.players[1].winAndStartedGame;
    if (rng_state) *rng_state = ref_rng_state();
    return 0;
}

int ref_landset_lands(int set, uint8_t* out12)
{
    const LandSet* sets[6] = {&LandSet::ASIA, &LandSet::NORTH_AMERICA, &LandSet::SOUTH_AMERICA, &LandSet::EUROPE,
                              &LandSet::AFRICA, &LandSet::AUSTRALIA};
    int n = (int)sets[set]->lands.size();
    for (int i = 0; i < n; i++) out12[i] = Utility::li2i(sets[set]->lands[i]->landIndex);
    return n;
}

// ---- sample storage: saveTrainingSamples / loadTrainingSamples / trimOldExamples / extend / updateOldGamesIndex
//      (neural_network/alphazero_nn_data.cpp:67-138,158-167).  Records cross this boundary in the 265-byte on-disk layout
//      (i8 player | 88 B NNInputData | f32 z | f32 pi[43]).
static void fill_storage(NNTrainDataStorage& st, const uint8_t* rec265, int n)
{
    for (int i = 0; i < n; i++, rec265 += 265) {
        NNTrainData d;
        d.playerIndex = (int8_t)rec265[0];
        std::memcpy((void*)&d.in, rec265 + 1, sizeof(NNInputData));
        std::memcpy(&d.out.value, rec265 + 89, 4);
        d.out.policy.resize(TF_OUTPUT_POLICY_TENSOR_SIZE);
        std::memcpy(d.out.policy.data(), rec265 + 93, 4 * TF_OUTPUT_POLICY_TENSOR_SIZE);
        st.data.push_back(d);
    }
}
static void dump_storage(const NNTrainDataStorage& st, uint8_t* rec265, int cap)
{
    for (int i = 0; i < (int)st.data.size() && i < cap; i++, rec265 += 265) {
        const NNTrainData& d = st.data[i];
        rec265[0] = (uint8_t)d.playerIndex;
        std::memcpy(rec265 + 1, (const void*)&d.in, sizeof(NNInputData));
        std::memcpy(rec265 + 89, &d.out.value, 4);
        std::memset(rec265 + 93, 0, 4 * TF_OUTPUT_POLICY_TENSOR_SIZE);
        std::memcpy(rec265 + 93, d.out.policy.data(), 4 * std::min<size_t>(d.out.policy.size(), TF_OUTPUT_POLICY_TENSOR_SIZE));
    }
}

// the reference's WRITER on n records
int ref_save_samples(const uint8_t* rec265, int n, const char* path)
{
    NNTrainDataStorage st;
    fill_storage(st, rec265, n);
    st.saveTrainingSamples(path);
    return 0;
}

// the reference's READER: returns the number of records it believes the file holds (its 4-byte header read), and the
// records as it parsed them
int ref_load_samples(const char* path, uint8_t* rec265_out, int cap)
{
    NNTrainDataStorage st;
    st.loadTrainingSamples(path);
    dump_storage(st, rec265_out, cap);
    return (int)st.data.size();
}

// trimOldExamples on n records (record i carries z = i as a marker) with the given oldGameIndex and storage limits;
// returns the new count, *first_kept = marker of the first surviving record, *old_out = oldGameIndex afterwards
int ref_trim_old_examples(int n, long old_game_index, int smin, int smax, int* first_kept, long* old_out)
{
    const int keep_min = SETTINGS.SAMPLES_STORAGE_MIN, keep_max = SETTINGS.SAMPLES_STORAGE_MAX;
    SETTINGS.SAMPLES_STORAGE_MIN = smin;
    SETTINGS.SAMPLES_STORAGE_MAX = smax;
    NNTrainDataStorage st;
    st.oldGameIndex = (size_t)old_game_index;
    for (int i = 0; i < n; i++) {
        NNTrainData d;
        d.out.value = (float)i;
        st.data.push_back(d);
    }
    st.trimOldExamples();
    SETTINGS.SAMPLES_STORAGE_MIN = keep_min;
    SETTINGS.SAMPLES_STORAGE_MAX = keep_max;
    if (first_kept) *first_kept = st.data.empty() ? -1 : (int)st.data.front().out.value;
    if (old_out) *old_out = (long)st.oldGameIndex;
    return (int)st.data.size();
}

// extend (per-GPU storages appended in GPU order, alphazero_trainer.cpp:59-62) + updateOldGamesIndex
int ref_extend(const uint8_t* a265, int na, const uint8_t* b265, int nb, uint8_t* out265, int cap, long* old_index_after_update)
{
    NNTrainDataStorage a, b;
    fill_storage(a, a265, na);
    fill_storage(b, b265, nb);
    a.extend(b);
    a.updateOldGamesIndex();
    dump_storage(a, out265, cap);
    if (old_index_after_update) *old_index_after_update = (long)a.oldGameIndex;
    return (int)a.data.size();
}

} // extern "C"
