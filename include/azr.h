/* azr.h — C-ABI of the MI355X-native AlphaZero-Risk self-play hot path (libazr_hip.so).
 *
 * Drop-in boundary for the reference's three in-process seams (citations relative to the
 * reference tree).  The reference has no FFI; these entry points are what a binding for each seam
 * would call, batched over G concurrent games (one handle = one GPU = one HIP stream set; a handle is
 * NOT re-entrant, different handles are independent — the reference's "one self-play thread per GPU",
 * player/alpha_zero/alphazero_trainer.cpp:48-57).
 *
 * Conventions: every function returns 0 on success or an AZR_E_* code (no exceptions cross the ABI);
 * the caller owns every buffer; `*_host` pointers are host memory, copied through pinned staging;
 * byte images use the reference's own layouts:
 *     state  = `struct Data`            160 B  (state/state.h:86-105)
 *     in88   = `class NNInputData`       88 B  (neural_network/alphazero_nn_data.h:66-96)
 *     rec265 = on-disk training record  265 B  (alphazero_nn_data.cpp:123-130: i8 player | in88 | f32 z | f32 pi[43])
 * Policy index 0..41 = land, 42 = SKIP (land/land.cpp:312), 43 = None.
 */
#ifndef AZR_H
#define AZR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZR_LANDS 42
#define AZR_MOVES 43
#define AZR_STATE_BYTES 160
#define AZR_INPUT_BYTES 88
#define AZR_RECORD_BYTES 265

enum {
    AZR_OK = 0,
    AZR_E_INVALID_ARGUMENT = 1, /* std::invalid_argument in the reference (illegal phase/move) */
    AZR_E_LOGIC = 2,            /* std::logic_error (army overflow, skip in a non-skippable phase) */
    AZR_E_BAD_HANDLE = 3,
    AZR_E_HIP = 4,              /* a HIP runtime call failed (TF_CHECK_OK abort in the reference) */
    AZR_E_CAPACITY = 5,         /* node pool / path stack / sample buffer exhausted */
    AZR_E_IO = 6,
    AZR_E_STATE = 7             /* call not valid in the engine's current mode */
};

/* arithmetic of the policy/value net contractions.
 *   AZR_NET_BF16  bf16 operands on the MFMA, fp32 accumulate: the fast path (|d pi|, |d v| <= 2e-2 of an fp32 evaluation)
 *   AZR_NET_F32   fp32 on the vector ALU: the precise, slow path (tolerance anchor of the tests)
 *   AZR_NET_F32X  fp32-equivalent on the MFMA: every conv operand as an fp16 pair (22 significand bits), three MFMA passes
 *                 per layer, fp32 accumulate / epilogue / residual / heads — the reference evaluates in fp32
 *                 (alphazero_nn.cpp:247-248); <= 2e-5 of the fp32 evaluation.  Conv weights must lie in the fp16 range.
 *   AZR_NET_F16   fp16 operands on the MFMA (the kernels and the rate of AZR_NET_BF16, 11 significand bits instead of 8): ~7x
 *                 closer to the fp32 evaluation than bf16 (<= 3e-3).  Conv weights must lie in the fp16 range (they are packed
 *                 as 2^k w per layer); activations saturate at 65504. */
enum { AZR_NET_F32 = 0, AZR_NET_BF16 = 1, AZR_NET_F32X = 2, AZR_NET_F16 = 3 };

/* Mirrors the fields of `class Settings` the hot path reads (src/settings.h:41-64) + engine sizing. */
typedef struct azr_settings {
    int32_t device;                /* HIP device ordinal */
    int32_t games;                 /* G: concurrent games on this handle (gpu-games, settings.h:163-171) */
    int32_t blocks;                /* residual blocks B (CMakeLists.txt:15 BLOCKS, 20) */
    int32_t net_dtype;             /* AZR_NET_F32 | AZR_NET_BF16 | AZR_NET_F32X | AZR_NET_F16 */
    int32_t mcts_simulations;      /* MCTS_SIMULATIONS (--mcts) */
    int32_t mcts_threads;          /* THREADS_PER_MCTS (-t, settings.h:44; default 2): T lock-stepped search threads per
                                      game with the reference's active_N virtual loss; 1..8.  Leaf slots = games * T. */
    int32_t allow_yield;           /* ALLOW_YIELD (--allow-yield) */
    int32_t limit_reinforcement;   /* LIMIT_REINFORCEMENT_MOVES (--limit-reinforcement) */
    int32_t limit_attack;          /* LIMIT_ATTACK_MOVES (--limit-attack) */
    int32_t max_game_rounds;       /* MAX_GAME_ROUNDS 58 */
    int32_t min_unit_move;         /* MIN_UNIT_MOVE 3 */
    int32_t temperature_threshold; /* TEMPERATURE_TRESHOLD (--temp) */
    float hp_exploration;          /* HP_EXPLORATION (--hp) */
    float dir_noise_value;         /* DIR_NOISE_VALUE (--dnv) */
    float dir_noise_epsi;          /* DIR_NOISE_EPSI (--dne) */
    int32_t node_capacity;         /* tree nodes per game; 0 = 16 * (mcts_simulations + 1) */
    int32_t sample_capacity;       /* (s,pi,z) records buffered per game before a drain; 0 = 4096 */
} azr_settings;

typedef struct azr_engine azr_engine;

/* Settings() defaults (src/settings.h:22-81) */
void azr_default_settings(azr_settings* s);

/* On failure *out is NULL, nothing stays allocated and azr_last_error(NULL) holds the calling thread's reason.
 * AZR_E_INVALID_ARGUMENT also for a node pool (node_capacity, or the default 16 * (mcts_simulations + 1)) above the
 * 65534 nodes per game a 16-bit node index addresses. */
int azr_engine_create(const azr_settings* s, azr_engine** out);
int azr_engine_destroy(azr_engine* h);
const char* azr_last_error(const azr_engine* h);   /* h == NULL: the last failed azr_engine_create of this thread */
int azr_engine_games(const azr_engine* h);

/* ---- game rules: `class State` + UtilityNN (state/state.cpp, alphazero_moves.cpp) -------------------- */
/* State::newGame (state.cpp:137-167) for game g with its own minstd_rand0 stream seeded seeds[g]
 * (replaces the reference's process-global RNG, src/rng.h:50). */
int azr_engine_new_games(azr_engine* h, const uint32_t* seeds_host);
int azr_engine_set_states(azr_engine* h, const void* data160_host); /* [G][160] */
int azr_engine_get_states(azr_engine* h, void* data160_host);       /* [G][160], padding bytes zero */
int azr_engine_set_rng(azr_engine* h, const uint32_t* engine_state_host); /* raw minstd_rand0 state per game */
int azr_engine_get_rng(azr_engine* h, uint32_t* engine_state_host);
/* UtilityNN::getValidMoves (alphazero_moves.cpp:3-70): bit i = land i, bit 42 = SKIP */
int azr_engine_valid_moves(azr_engine* h, uint64_t* masks_host);
/* UtilityNN::makeMove (alphazero_moves.cpp:72-233); rc_host[g] (optional) = AZR_OK / AZR_E_INVALID_ARGUMENT /
 * AZR_E_LOGIC per game, as the reference's throw sites; moves_host[g] = 255 leaves game g untouched. */
int azr_engine_make_moves(azr_engine* h, const uint8_t* moves_host, uint8_t* rc_host);
/* State::gameStatus (state.cpp:518-565): -1 running, 0/1 winner, -2 draw */
int azr_engine_status(azr_engine* h, int8_t* status_host);
/* NNInputData(const State&) (alphazero_nn_data.cpp:165-196) */
int azr_engine_encode(azr_engine* h, void* in88_host); /* [G][88] */

/* ---- NN service: AlphaZeroNNId (alphazero_gpu_cluster.h:14-47) ------------------------------------------ */
size_t azr_nn_param_count(int blocks);                     /* floats in the AZRW flat vector */
int azr_nn_init_random(azr_engine* h, uint64_t seed);      /* `init` op: Glorot-uniform kernels, BN identity */
int azr_nn_set_weights(azr_engine* h, const float* flat_host, size_t count);
int azr_nn_get_weights(azr_engine* h, float* flat_host, size_t count);
int azr_nn_load(azr_engine* h, const char* path);          /* loadCheckpoint (alphazero_nn.cpp:189-204) */
int azr_nn_save(azr_engine* h, const char* path);          /* saveCheckpoint (alphazero_nn.cpp:206-214) */
/* predict / processBatchPrediction (alphazero_nn.cpp:236-267,322-349): n inputs -> softmax pi[n][43], tanh v[n] */
int azr_nn_predict(azr_engine* h, const void* in88_host, int n, float* pi_host, float* v_host);
/* AlphaZeroNN::train (alphazero_nn.cpp:351-410) on n 265-byte records: per epoch shuffle (std::shuffle with a
 * minstd_rand0 — *shuffle_rng_state is the raw engine state in and out, standing in for the process-global RNG,
 * src/rng.h:50; NULL = default-seeded), floor(n / batch_size) minibatch steps of the `optimize` op (fp32 forward in
 * training mode, loss, backward, Adam; python/src/build_graph.py:54-103), remainder dropped.  loss_*_host[e] = the
 * epoch averages the reference prints and logs (NaN when n < batch_size).  Adam moments and step count persist on the
 * handle across calls like the TF session's slots; inference weights are refolded / repacked before returning.
 * Range: the forward conv multiplies fp16 pairs of 2^10 w, so a conv weight with |w| >= 64 (or a weight that is not a number) cannot be
 * represented; that, and a loss that stops being a number, is detected behind every epoch: the call then returns
 * AZR_E_INVALID_ARGUMENT with the reason in azr_last_error, the handle's weights are those from before the call and its optimiser
 * state is dropped (the reference's TensorFlow step would carry the NaNs on silently). */
int azr_nn_train(azr_engine* h, const void* rec265_host, size_t n, int epochs, int batch_size,
                 uint32_t* shuffle_rng_state, float* loss_pi_host, float* loss_v_host);
/* Data-parallel AlphaZeroNN::train: the same epochs / shuffles / minibatches, every minibatch split over `world` ranks
 * (one process per GPU; the reference trains on GPU 0 only and hands the weights over through checkpoints/temp.bin,
 * alphazero_gpu_cluster.cpp:221-231).  Every rank passes ALL n records and the same *shuffle_rng_state and takes slice
 * `rank` of each minibatch (batch_size % world == 0).  Whatever spans the minibatch — batch-norm statistics in the
 * forward pass, their two sums in the backward pass, the losses, and at the end of the step the whole gradient vector
 * (azr_nn_param_count floats) — is summed over the ranks through `allreduce`: in place on DEVICE memory of this GPU,
 * dtype 0 = float32, 1 = float64, return 0 on success; the engine's stream is idle while it runs.  All ranks then take the
 * same Adam step, so their weights stay equal without a broadcast, and equal the single-GPU step's up to summation
 * order.  world = 1 with allreduce == NULL and no communicator is azr_nn_train; world = 1 WITH a callback (or a one-rank
 * communicator, azr_dp_init) runs the data-parallel code path on one rank (every all-reduce is the identity): the single-GPU
 * rehearsal of the RCCL path.  allreduce == NULL with world > 1 needs azr_dp_init(h, rank, world, ...). */
typedef int (*azr_allreduce_fn)(void* ctx, void* device_ptr, size_t count, int dtype);
int azr_nn_train_dp(azr_engine* h, const void* rec265_host, size_t n, int epochs, int batch_size, uint32_t* shuffle_rng_state,
                    int rank, int world, azr_allreduce_fn allreduce, void* ctx, float* loss_pi_host, float* loss_v_host);
/* The handle's own RCCL communicator for azr_nn_train_dp (one process per GPU, RCCL over xGMI).  With it — and allreduce == NULL —
 * every sum of the data-parallel step is an ncclAllReduce on the engine's own stream: stream-ordered, no host hand-over (2B + 6
 * small sums and one of azr_nn_param_count floats per step).  Rank 0 draws the 128-byte id (azr_dp_unique_id) and passes it to the
 * other processes by whatever the launcher offers (torch.distributed broadcast, MPI, a file); then EVERY rank calls azr_dp_init
 * (collective).  RCCL is bound at run time ("librccl.so.1": the copy already loaded into the process, else /opt/rocm's);
 * AZR_E_STATE if there is none.  azr_engine_destroy shuts the communicator down. */
#define AZR_DP_ID_BYTES 128
int azr_dp_unique_id(void* id128);
int azr_dp_init(azr_engine* h, int rank, int world, const void* id128);
int azr_dp_shutdown(azr_engine* h);
/* one `session->Run(..., {optimize})` (alphazero_nn.cpp:389-391) on exactly n records in the given order */
int azr_nn_train_batch(azr_engine* h, const void* rec265_host, int n, float* loss_pi, float* loss_v);
/* diagnostics: gradient vector of the last step in AZRW layout (moving-statistics slots unused) */
int azr_nn_train_grads(azr_engine* h, float* flat_host, size_t count);
/* drop the optimiser state and the training buffers */
int azr_nn_train_reset(azr_engine* h);

/* ---- search: AlphaZeroMCTS / StateSimulationsStorage (alphazero_mcts.h:55-95) ---------------------------- */
int azr_mcts_clear(azr_engine* h);   /* clearNodes (alphazero_mcts.cpp:223-227), all games */
int azr_mcts_trim(azr_engine* h);    /* trimNodes  (alphazero_mcts.cpp:229-245), all games */
/* AlphaZeroMCTS::simulate (alphazero_mcts.cpp:255-307) for all G roots in lock-step: trim, expand the root if unknown,
 * then mcts_simulations - mcts_simulations % mcts_threads searches per game by mcts_threads search threads that block
 * together at the NN seam (thread k of game g = leaf slot g * T + k; threads run in index order — one of the
 * reference's possible schedules, and the only one at T = 1).  Finished games idle. */
int azr_mcts_simulate(azr_engine* h);
/* The same search split at the NN seam (predictFuture, alphazero_mcts.cpp:350-351), so a caller can supply
 * priors/values itself: begin -> { leaves -> [evaluate] -> apply }* until *active_out == 0. */
int azr_mcts_begin(azr_engine* h);
/* in88_host [G*T][88], need_eval_host [G*T], pi_host [G*T][43], v_host [G*T]; slot = g * T + k */
int azr_mcts_leaves(azr_engine* h, void* in88_host, uint8_t* need_eval_host, int* active_out);
int azr_mcts_apply(azr_engine* h, const float* pi_host, const float* v_host);
/* root statistics of the last search: N[G][43]; Q,P optional */
int azr_mcts_root_stats(azr_engine* h, uint32_t* n_host, float* q_host, float* p_host);
/* StateSimulations::calculateMoveProbability(1.0f) (alphazero_mcts.cpp:121-149) */
int azr_mcts_policy(azr_engine* h, float* pi_host);
/* pickHigestWeightedMove / pickRandomWeightedMove (alphazero_mcts.cpp:379-412) on the last search's policy;
 * sample != 0 draws with the game's own RNG stream (one rFloat). */
int azr_mcts_pick(azr_engine* h, int sample, uint8_t* moves_host);

/* ---- device-resident self-play (trainer move loop, alphazero_trainer.cpp:80-119) --------------------------- */
/* (Re)start all G games: game g plays seeds base_seed + g, then base_seed + G + g, ... */
int azr_selfplay_start(azr_engine* h, uint32_t base_seed);
/* The trainer's own loop bound (Counter::hasNext over TRAIN_ITERATION_GAMES, alphazero_trainer.cpp:83): start exactly
 * `games` games — seeds base_seed .. base_seed + games - 1, handed to whichever slot is free next — and play every one
 * of them to its end; slots idle once no game is left to start.  Done when games_finished + errors == games. */
int azr_selfplay_start_games(azr_engine* h, uint32_t base_seed, uint64_t games);
/* The move loop entered in the MIDDLE of games: game g goes on from the state and RNG stream the caller has set
 * (azr_engine_set_states / azr_engine_set_rng; states must be running games), its records start there; a finished game's
 * slot restarts as under azr_selfplay_start (seeds base_seed + G + g, base_seed + 2G + g, ...). */
int azr_selfplay_start_from_states(azr_engine* h, uint32_t base_seed);
/* Run `passes` passes of the hot path: every pass = one tree step (backup/expand + select to the next leaf,
 * decisions, moves, game restarts — all on device) + one batched net evaluation of the G leaves. */
int azr_selfplay_run(azr_engine* h, int passes);
typedef struct azr_counters {
    uint64_t simulations;   /* completed search() descents (root expansions not counted) */
    uint64_t evaluations;   /* net evaluations consumed (leaf + root) */
    uint64_t levels;        /* inner-node levels visited (mean depth = levels / simulations) */
    uint64_t decisions;     /* moves played */
    uint64_t games_finished;
    uint64_t samples;       /* records produced */
    uint64_t nodes_dropped; /* expansions skipped because the node pool was full (should be 0) */
    uint64_t errors;        /* games stopped on a rules error (should be 0) */
    uint64_t records_dropped; /* records lost because a game outgrew sample_capacity or the ring was full (should be 0;
                                 `samples` counts them too) */
    uint64_t tower_fallbacks; /* net launches of <= 128 boards (split-channel tower, 4 co-resident workgroups per board pair) in which a
                                 workgroup waited for its partners longer than the spin limit — the GPU was shared with something that kept
                                 them from running — and which the one-board-per-workgroup kernel queued behind them recomputed, in
                                 stream order: results are unaffected, the count says how often it happened since the handle exists
                                 (normally 0) */
} azr_counters;
int azr_selfplay_counters(azr_engine* h, azr_counters* out);
/* finished games' records, z filled (NNTrainDataStorage::updateValues, alphazero_nn_data.cpp:51-65).  Copies the first
 * min(available, cap_records) records; the others STAY buffered for the next call (drain in a loop until *n_out <
 * cap_records).  rec265_host == NULL discards everything buffered. */
int azr_samples_drain(azr_engine* h, void* rec265_host, size_t cap_records, size_t* n_out);
/* the same copy to DEVICE memory of this GPU (e.g. a collective's send buffer), on the engine's own stream, without
 * removing anything: min(available, cap_records) records */
int azr_samples_copy_device(azr_engine* h, void* rec265_device, size_t cap_records, size_t* n_out);
/* device-side view for RCCL gathers: pointer to the packed record ring and its count; valid until the next
 * azr_selfplay_run / azr_samples_drain */
int azr_samples_device_view(azr_engine* h, void** dev_ptr_out, size_t* n_out);

/* ---- arena: GameGroup::playGames (game/game.cpp:256-312) on the device ------------------------------------------------ */
/* The G slots of the engine are the reference's G player pairs (threads): each plays Game::playGames(1) repeatedly —
 * mirrored pairs with alternating starts (Game::newGame, game.cpp:170-191) — until Counter::hasNext(2) fails.
 * Players: AlphaZeroPlayer (alphazero_player.cpp:3-21, argmax, tree trimmed at every turn), ScriptPlayer
 * (player/script/script_player.cpp), RandomPlayer (player/random/random_player.cpp). */
enum { AZR_PLAYER_ALPHAZERO = 0, AZR_PLAYER_SCRIPT = 1, AZR_PLAYER_RANDOM = 2,
       AZR_PLAYER_ALPHAZERO_B = 3 /* AlphaZeroPlayer on a second network: azr_arena_set_opponent_net */ };
typedef struct azr_game_results {   /* GameResults (game/game.h:17-29) */
    int32_t count, draw;
    int32_t win[2], win_and_started[2];
} azr_game_results;
/* mirror_games (SETTINGS.MIRROR_GAMES, game.cpp:170-191):
 *   AZR_MIRROR_OFF         every game a fresh deal
 *   AZR_MIRROR_SEQUENTIAL  the reference's thread-per-pair form: a slot deals, plays the game, then plays the same deal with the
 *                          players inverted (State::invertPlayers, state.cpp:493-516); slot g draws everything — deals and dice of
 *                          all its games — from ONE minstd_rand0 stream seeded base_seed + g (the reference's global engine)
 *   AZR_MIRROR_CONCURRENT  the two games of a pair at the same time on the slots 2j and 2j + 1 (nothing in Game orders them, only
 *                          the shared initial deal does).  Pair p = j + k (G/2) is the k-th pair of slot pair j; both halves deal from
 *                          minstd_rand0(base_seed + p); half 0 (player 0 starts) goes on with that stream for its dice, half 1
 *                          (invertPlayers of the same deal, player 1 starts) draws its dice from minstd_rand0(base_seed + p + 2^30).
 *                          Pairs are assigned statically (p < games / 2), so a run is a function of its arguments alone; G even.
 * games = Counter::count; games_per_slot_cap > 0 additionally limits every slot (deterministic splits for tests) */
enum { AZR_MIRROR_OFF = 0, AZR_MIRROR_SEQUENTIAL = 1, AZR_MIRROR_CONCURRENT = 2 };
int azr_arena_start(azr_engine* h, int player1, int player2, int games, int games_per_slot_cap, int mirror_games,
                    uint32_t base_seed);
int azr_arena_run(azr_engine* h, int passes, int* finished_out);
/* New-vs-old arena (GameGroup::playGames(trainAZPG, generateAZPG, ...), alphazero_trainer.cpp:147-152): player
 * AZR_PLAYER_ALPHAZERO_B searches a tree of its own in every slot (every AlphaZeroPlayer owns an AlphaZeroMCTS) and is
 * evaluated by `other`'s network (same device, same net shape; `other` may be h itself; its weights are used in place,
 * so keep `other` alive and pass NULL here before destroying it).  Each pass runs the two networks on the leaves of
 * their own players only. */
int azr_arena_set_opponent_net(azr_engine* h, azr_engine* other);
/* INCLUDE_COMPARE_GAMES_TRAIN_SAMPLES (alphazero_trainer.cpp:143-146): AlphaZero players push (s, pi) at every decision
 * (alphazero_player.cpp:15-18); a finished game's records get their z and go to the record ring (azr_samples_drain),
 * game by game in decision order (the reference appends player by player).  Set before azr_arena_start. */
int azr_arena_collect_samples(azr_engine* h, int on);
int azr_arena_results(azr_engine* h, azr_game_results* out);
/* per slot: games finished, and for its first 16 games status / round count / final state image */
int azr_arena_log(azr_engine* h, int32_t* games_per_slot_host, int8_t* status_host /*[G][16]*/,
                  uint16_t* rounds_host /*[G][16]*/, void* finals160_host /*[G][16][160]*/);

/* ---- measurement hooks (bench.py) --------------------------------------------------------------------------- */
/* average duration in ms of the net-forward launches and of the tree-step launches over the last
 * azr_selfplay_run, measured with HIP events on the engine's stream */
int azr_profile_last_run(azr_engine* h, float* net_ms_avg, float* tree_ms_avg, int* launches);
int azr_device_synchronize(azr_engine* h);
/* diagnostics of the tower kernels (tools/tower_clock.py, tools/tower_trace.py): sustained in-kernel shader clock and
 * workgroup-0 time after `warm` back-to-back launches on n leaf slots; per-workgroup time stamps of one launch */
int azr_debug_tower_clock(azr_engine* h, int n, int warm, double* ghz_out, double* tower_ms_out);
int azr_debug_tower_trace(azr_engine* h, int n, int warm, unsigned long long* out8, int cap_wgs, int* wgs_out);
/* the tile plan of a bf16 net launch of n boards: boards per workgroup (the largest, in a mixed launch) and workgroups;
 * 2..4 boards = k_tower_sb<NB> (one LDS image), 1 = k_tower_bf16<1> (AZR_TOWER_SB=0: k_tower_bf16<1..3>) */
int azr_debug_tower_plan(azr_engine* h, int n, int* boards_per_wg, int* wgs);

#ifdef __cplusplus
}
#endif
#endif /* AZR_H */
