/* TEST INFRASTRUCTURE — CPU oracle for the AlphaZero-Risk hot path.  NOT product code.
 *
 * Plain-C restatement of the reference's algorithm for SURVEY.md §8(a) rows a1-a24.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product path (alphazero-risk_amd/, include/azr.h) never links, imports or calls it.
 *
 * Pinning status:
 *   rules / legal masks / dice / RNG / feature struct / normalize / z back-fill
 *       -> PINNED against the real reference compiled in oracle/_ref (tests/test_oracle_vs_ref.py)
 *          and against the committed fixtures in tests/golden/ generated from it.
 *   MCTS (a16-a23), tensor plane packing (a13), net forward (a14)
 *       -> "parity unpinned": the reference units need TensorFlow (absent, un-vendored, unpinned
 *          version) and the reference ships no tests/golden vectors; restated from the cited lines.
 */
#ifndef AZR_ORACLE_H
#define AZR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_LANDS = 42, ORC_MOVES = 43, ORC_SKIP = 42, ORC_NONE = 43, ORC_NEUTRAL = 2, ORC_ARMY_MAX = 32 };
enum { ORC_SETUP = 0, ORC_SETUP_NEUTRAL, ORC_REINFORCEMENT, ORC_ATTACK, ORC_ATTACK_MOBILIZATION, ORC_FORTIFY };
enum { ORC_NOT_ENDED = -1, ORC_DRAW = -2 };
enum { ORC_OK = 0, ORC_INVALID_ARGUMENT = 1, ORC_LOGIC_ERROR = 2 };

/* src/settings.h:41-64 — the knobs the hot path reads */
typedef struct {
    int allow_yield;            /* ALLOW_YIELD                1 */
    int limit_reinforcement;    /* LIMIT_REINFORCEMENT_MOVES  1 */
    int limit_attack;           /* LIMIT_ATTACK_MOVES         0 */
    int max_game_rounds;        /* MAX_GAME_ROUNDS           58 */
    int min_unit_move;          /* MIN_UNIT_MOVE              3 */
    int mcts_simulations;       /* MCTS_SIMULATIONS          32 */
    float hp_exploration;       /* HP_EXPLORATION           1.1 */
    float dir_noise_value;      /* DIR_NOISE_VALUE          0.3 */
    float dir_noise_epsi;       /* DIR_NOISE_EPSI          0.25 */
    int temperature_threshold;  /* TEMPERATURE_TRESHOLD      43 */
    int mcts_threads;           /* THREADS_PER_MCTS (reference default 2; here 1 = the deterministic schedule).  T > 1
                                   = T search threads in lock-step, thread order 0..T-1 (see orc_mcts_simulate) */
} orc_settings;

void orc_default_settings(orc_settings* s);

/* state/state.h:59-105 */
typedef struct {
    uint64_t owned, owned_army, owned_full, attack, attack_army;
    int16_t total_army;
    uint8_t cards;
} orc_player;

typedef struct {
    uint8_t army[ORC_LANDS];
    uint8_t owner[ORC_LANDS];
    orc_player ps[2];
    uint16_t round;
    int8_t cur;
    uint8_t card_sets;
    uint8_t reinf;
    uint8_t phase;
    uint8_t mob_from, mob_to;
    uint8_t allow_draw;
    uint8_t attacks;
    uint16_t drawn;
} orc_state;

/* per-stream RNG (src/rng.h) */
typedef struct { uint32_t x; } orc_rng;
void orc_rng_seed(orc_rng* r, uint32_t seed);
uint32_t orc_rng_next(orc_rng* r);
int orc_rng_dice(orc_rng* r);
int orc_rng_int(orc_rng* r);
float orc_rng_float(orc_rng* r);
uint64_t orc_random_mask(orc_rng* r, uint64_t masks);

/* static tables */
int orc_neighbours(int land, uint8_t* out6);
uint64_t orc_neighbour_mask(int land);
uint64_t orc_continent_mask(int c); /* 0..5, 6 = all lands */

/* 160-byte reference `Data` image <-> oracle state */
void orc_state_blank(orc_state* s);
void orc_state_pack(const orc_state* s, uint8_t* data160);
void orc_state_unpack(orc_state* s, const uint8_t* data160);
int orc_state_equal(const orc_state* a, const orc_state* b);

/* rules */
void orc_new_game(orc_state* s, orc_rng* r);
uint64_t orc_valid_moves(const orc_state* s, const orc_settings* cfg);
int orc_make_move(orc_state* s, int move, orc_rng* r, const orc_settings* cfg);
int orc_game_status(const orc_state* s, const orc_settings* cfg);
int orc_reinforcement_value(uint64_t owned);
int orc_consistency_check(const orc_state* s); /* 0 = all derived masks/totals match a recomputation */
void orc_invert_players(orc_state* s);

/* NN data seams */
void orc_encode(const orc_state* s, uint8_t* in88);
void orc_planes(const uint8_t* in88, float* t546); /* [7][6][13] */
void orc_normalize(float* pi43, uint64_t valid);
void orc_update_values(const int8_t* player_index, int n, int game_status, float* z);

/* flat bulk helpers over byte images (used by the ctypes tests) */
int orc_play_random_game(uint32_t seed, int cap, uint8_t* states160, uint64_t* masks, uint8_t* moves,
                         int* status, uint8_t* final160, const orc_settings* cfg);

/* emulation of libstdc++'s unordered_map<LandIndex,...> iteration order after inserting the set
 * bits of `mask` in ascending order (alphazero_mcts.cpp:32-41,78) */
int orc_umap_order(uint64_t mask, uint8_t* out43);

/* ---- policy/value net (python/src/build_graph.py:54-90), fp32 ---- */
typedef struct {
    int blocks;
    const float* flat;     /* AZRW flat parameter vector, see orc_net_param_count */
} orc_net;
size_t orc_net_param_count(int blocks);
void orc_net_init_random(float* flat, int blocks, uint64_t seed);  /* Glorot-uniform kernels, BN identity, zero bias */
void orc_net_forward(const orc_net* net, const uint8_t* in88, int n, float* pi /*[n][43]*/, float* v /*[n]*/);
void orc_net_forward_mt(const orc_net* net, const uint8_t* in88, int n, float* pi, float* v, int threads);

/* ---- MCTS ---- */
typedef void (*orc_eval_fn)(void* ctx, const uint8_t* in88, float* pi43, float* v);

typedef struct orc_mcts orc_mcts;
orc_mcts* orc_mcts_create(const orc_settings* cfg);
void orc_mcts_destroy(orc_mcts* m);
void orc_mcts_clear(orc_mcts* m);
void orc_mcts_trim(orc_mcts* m);
int orc_mcts_node_count(const orc_mcts* m);
/* AlphaZeroMCTS::simulate with cfg.mcts_threads lock-stepped search threads; returns ORC_OK or an error */
int orc_mcts_simulate(orc_mcts* m, const orc_state* root, orc_rng* r, orc_eval_fn eval, void* ctx);
/* root statistics after simulate */
int orc_mcts_root_stats(orc_mcts* m, const orc_state* root, uint32_t* n43, float* q43, float* p43, uint32_t* sumN);
int orc_mcts_policy(orc_mcts* m, const orc_state* root, float* pi43);
int orc_pick_highest(const float* pi43);
int orc_pick_random(const float* pi43, orc_rng* r);
uint64_t orc_mcts_sim_count(const orc_mcts* m);   /* completed search() descents */
uint64_t orc_mcts_eval_count(const orc_mcts* m);  /* net evaluations (leaf + root) */
uint64_t orc_mcts_level_count(const orc_mcts* m); /* inner-node levels visited (for mean depth) */
uint64_t orc_mcts_dup_count(const orc_mcts* m);   /* duplicatedStatesDropped (T > 1 only) */

/* one self-play game as alphazero_trainer.cpp:80-119; records are the 265-byte on-disk layout.
 * Returns number of records written (<= cap) or -1 on error. */
int orc_selfplay_game(const orc_settings* cfg, uint32_t seed, orc_eval_fn eval, void* ctx,
                      uint8_t* rec265, int cap, int* status, int* rounds, uint8_t* moves_out, int max_decisions,
                      uint64_t* sims_out, uint64_t* evals_out);

/* ---- opponents and the host game driver, pinned against oracle/_ref ---- */
typedef struct {
    uint8_t order[6];               /* attackLandSetPriority as a permutation of {ASIA,NA,SA,EU,AF,AU} */
    uint8_t not_owned[6], not_owned_attack[6];
    int attacking_set;              /* members of ScriptPlayer persist across turns AND games */
    int land_to, land_from;
    uint8_t attack_from_army;
    uint64_t owned_attack_mask, attack_mask;
} orc_script;
typedef struct { int count, draw, win[2], win_started[2]; uint32_t rng_state; } orc_results;
void orc_script_init(orc_script* p);
int orc_script_take_turn(orc_script* p, orc_state* s, orc_rng* r, const orc_settings* cfg);
int orc_random_take_turn(orc_state* s, int me, orc_rng* r, const orc_settings* cfg);
int orc_landset_lands(int set, uint8_t* out12);
/* kinds: 0 AlphaZero (needs eval), 1 ScriptPlayer, 2 RandomPlayer */
int orc_play_games(const orc_settings* cfg, int kind0, int kind1, int games, int mirror, uint32_t seed,
                   orc_eval_fn eval, void* ctx, orc_results* res, int8_t* status_out, uint8_t* finals160,
                   uint16_t* rounds_out);

int orc_play_games2(const orc_settings* cfg, int kind0, int kind1, int games, int mirror, uint32_t seed,
                    orc_eval_fn eval, void* ctx, orc_eval_fn eval_b, void* ctx_b, orc_results* res, int8_t* status_out,
                    uint8_t* finals160, uint16_t* rounds_out, uint8_t* rec265, int rec_cap, int* rec_n,
                    int* rec_game_end /* [games] cumulative record count after each game */);

/* one slot of the concurrent-halves form of a mirrored arena (include/azr.h AZR_MIRROR_CONCURRENT): half `half` of the pairs
 * pair_seed0 + k * pair_stride, k = 0 .. games - 1 */
int orc_play_half_games(const orc_settings* cfg, int kind0, int kind1, int games, int half, uint32_t pair_seed0, uint32_t pair_stride,
                        orc_eval_fn eval, void* ctx, orc_eval_fn eval_b, void* ctx_b, orc_results* res, int8_t* status_out,
                        uint8_t* finals160, uint16_t* rounds_out, uint8_t* rec265, int rec_cap, int* rec_n, int* rec_game_end);

/* bench.py cpu_baseline: `threads` games in parallel, `decisions` decisions each, fp32 CPU net */
int orc_bench_selfplay(const orc_settings* cfg, const orc_net* net, uint32_t base_seed, int threads, int decisions,
                       uint64_t* sims, uint64_t* evals, double* seconds);

/* deterministic integer-hash stub "net" shared by oracle-side tests: pi_i in (0.5,1.5)/sum, v in (-1,1) */
void orc_hash_eval(void* ctx, const uint8_t* in88, float* pi43, float* v);
/* uniform stub: every prior identical (forces the unordered_map tie-break path), v = 0 */
void orc_uniform_eval(void* ctx, const uint8_t* in88, float* pi43, float* v);
/* orc_net adaptor: ctx = orc_net* */
void orc_net_eval(void* ctx, const uint8_t* in88, float* pi43, float* v);

#ifdef __cplusplus
}
#endif
#endif
