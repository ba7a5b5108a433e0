// azr_internal.hpp — engine object shared by the C-ABI translation units (host side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "../../include/azr.h"
#include "azr_tree.hpp"

// Test hooks.  Environment switches that select an older or alternative formulation of the same arithmetic (the parity tests compare it
// with the default one as an independently written implementation), force a hand-off to give up, or stand in for a communicator exist
// ONLY in libazr_hip_test.so — the same sources compiled with -DAZR_TEST_HOOKS by `make test` (csrc/Makefile), loaded by the tests that
// need them.  The product library libazr_hip.so reads no environment variable and carries one path.
#ifdef AZR_TEST_HOOKS
#include <stdlib.h>
namespace azr {
inline const char* hook_env(const char* name) { return getenv(name); }
}
#else
namespace azr {
constexpr const char* hook_env(const char*) { return nullptr; }
}
#endif
namespace azr {
inline int hook_env_int(const char* name, int dflt) { const char* v = hook_env(name); return v ? atoi(v) : dflt; }

constexpr int LEAF_STRIDE = 96;    // in88 padded to 96 B per game
constexpr int PI_STRIDE = 44;      // pi[43] padded
constexpr int STAGE_BYTES = 264;   // staged record: in88 @0 | pi[43] @88 | player @260
constexpr int NF = 256;            // FILTERS (python/src/build_graph.py:32)
constexpr int NPOS = 42;

// device view handed to kernels by value
struct Dev {
    int G, C, H, DMAX, SCAP;
    int T;                 // THREADS_PER_MCTS: search threads (= leaf slots) per game; slot = g * T + k
    Rules rules;
    Search search;
    uint8_t* state;        // [G][64] game records
    Ctl* ctl;              // [G]
    uint8_t* nodes;        // [G][C][NODE_BYTES]
    uint32_t* touch;       // [G][C]
    uint32_t* nhash;       // [G][C]
    uint32_t* table;       // [G][H]
    uint16_t* freel;       // [G][C]
    uint32_t* path;        // [G][T][DMAX]
    uint8_t* leaf_in;      // [G*T][LEAF_STRIDE]
    uint8_t* leaf_key;     // [G*T][64]
    uint64_t* leaf_valid;  // [G*T]
    uint32_t* leaf_hash;   // [G*T]
    float* net_pi;         // [G*T][PI_STRIDE]
    float* net_v;          // [G*T]
    uint8_t* stage;        // [G][SCAP][STAGE_BYTES]
    uint8_t* ring;         // [RCAP][265]
    unsigned long long* ring_count;
    unsigned long long ring_cap;
    Counters* counters;
    uint32_t* active;      // number of games with a pending leaf after the last tree step
    uint32_t base_seed;
    unsigned long long sp_quota;      // self-play quota mode: games to start in all (0 = unlimited)
    unsigned long long* sp_started;   // ... and the ticket counter
    int sp_compact;                   // self-play tail: the tree step lists the waiting leaf slots (leaf_list[0], leaf_count[0]) and
                                      // the net runs on that list only
    // arena (GameGroup::playGames, game/game.cpp:277-312)
    int kind0, kind1;          // AZR_PLAYER_* of player index 0 / 1
    int arena_total;           // Counter::count
    int arena_slot_cap;        // optional cap of games per slot (0 = none)
    int arena_mirror;          // SETTINGS.MIRROR_GAMES
    int* arena_taken;          // Counter::i
    int* arena_res;            // GameResults: count, draw, win0, winStarted0, win1, winStarted1
    uint8_t* prev_start;       // [G][64]  Game::previousStartState
    uint8_t* script;           // [G][2][32]  ScriptW
    int8_t* alog_status;       // [G][ALOG]
    uint16_t* alog_rounds;     // [G][ALOG]
    uint8_t* alog_final;       // [G][ALOG][64]
    // two-net arena (trainAZPG vs generateAZPG, alphazero_trainer.cpp:147-152): AZR_PLAYER_ALPHAZERO_B searches its own
    // tree (every AlphaZeroPlayer owns an AlphaZeroMCTS) and is evaluated by the opponent handle's network
    uint8_t* nodes2;           // [G][C][NODE_BYTES]   (null until azr_arena_set_opponent_net)
    uint32_t* touch2;
    uint32_t* nhash2;
    uint32_t* table2;
    uint16_t* freel2;
    uint32_t* tctl2;           // [G][4]  tree 2's {search_id, nfree, hiwater, -}
    int* leaf_list;            // [2][G*T] leaf slots waiting for net A / net B after an arena step
    int* leaf_count;           // [2][2]: [0][net] by default; the arena's passes without a read-back alternate between the two rows
    int lc_base;               // ... row (x 2) the arena step counts into
    int lc_zero;               // ... row (x 2) the arena step zeroes for the next pass (nobody reads it any more), or -1
    int arena_collect;         // stage (s, pi, player) per AlphaZero decision and flush finished games to the record ring
};
constexpr int ALOG = 16;

// folded network parameters on device
struct NetDev {
    int blocks;
    // fp32 path
    float* stem_w;     // [9][13][256]
    float* stem_scale; // [7] per board row (conv_bn, axis=1)
    float* stem_shift; // [7]
    float* tower_w;    // [2B][9][256][256]
    float* tower_scale;// [2B][256]
    float* tower_shift;// [2B][256]
    float* d_flat;     // whole AZRW vector on device (fp32 conv kernels and the heads read it in place)
    float* d_fold;     // folded BN: stem scale[7] shift[7]; per conv layer scale[256] shift[256]
    const float* head; // head section of d_flat
    // bf16 MFMA path
    void* bf16ctx;      // azr_net_bf16.hip: packed weight fragments + tower output buffer
    void* fxctx;        // azr_tower_fx.hip (AZR_NET_F32X): packed fp16-pair weight fragments
    // azr_nn_predict staging, created by the first call: device [G] x (96 B in | 44 floats pi | 1 float v), the same in pinned host memory
    uint8_t* pred_dev;
    uint8_t* pred_host;
    // activations (fp32 path)
    float* actX;       // [G][42][256]
    float* actT;       // [G][42][256]
};

}  // namespace azr

struct azr_engine {
    azr_settings cfg;
    azr::Dev d;
    azr::NetDev net;
    hipStream_t stream;
    std::string err;
    std::vector<float> flat;      // AZRW host copy
    int mode;                     // 0 rules only / stepwise, 2 self-play
    // profiling of the last azr_selfplay_run
    std::vector<hipEvent_t> ev;
    float prof_net_ms, prof_tree_ms, prof_tower_ms;
    hipEvent_t pe_tower0 = nullptr, pe_tower1 = nullptr;  // when set, the net records these around its dominant kernel
    int prof_launches;
    bool weights_set;
    void* tree2[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // nodes2, touch2, nhash2, table2, freel2, tctl2
    azr_engine* opponent = nullptr;  // handle whose network plays AZR_PLAYER_ALPHAZERO_B (azr_arena_set_opponent_net)
    unsigned arena_pass = 0;         // passes of the running arena that were queued without a read-back (parity = leaf_count row)
    hipEvent_t arena_ev2 = nullptr;  // ... and this stream's tree step done -> the opponent's net launch may start
    hipEvent_t arena_ev = nullptr;   // two-net arena: the opponent's net launch (on ITS stream) done -> this stream may go on
    bool sp_tail = false;         // quota self-play: no game is left to start, slots go idle -> compacted net batches
    void* train = nullptr;        // azr_train.hip: optimiser state + activation slabs, created by the first azr_nn_train*
    void* dp_comm = nullptr;      // azr_dp_init: this handle's RCCL communicator (ncclComm_t), rank and world
    int dp_rank = 0, dp_world = 0;
};

namespace azr {
// net (azr_net.hip)
int net_alloc(azr_engine* h);
void net_free(azr_engine* h);
int net_upload(azr_engine* h);  // fold BN, pack, copy h->flat to the device
int net_forward(azr_engine* h, const uint8_t* d_in88, int in_stride, int n, float* d_pi, float* d_v);
int net_forward_ex(azr_engine* h, const uint8_t* d_in88, int in_stride, int n, float* d_pi, float* d_v, const int* d_map, hipStream_t st);
bool net_forward_counted_ok(azr_engine* h, int n_max);
int net_forward_counted(azr_engine* h, const uint8_t* d_in88, int in_stride, int n_max, const int* n_dev, const int* n_other, float* d_pi, float* d_v, const int* d_map, hipStream_t st);
size_t net_param_count(int blocks);
void net_init_random(float* flat, int blocks, uint64_t seed);
int net_fallbacks(azr_engine* h, unsigned long long* out);   // split-channel tower launches recomputed after a hand-off gave up
// NET_F32X (azr_tower_fx.hip)
int net_fx_alloc(azr_engine* h);
void net_fx_free(azr_engine* h);
int net_fx_upload(azr_engine* h, const float* fold_host);
int net_fx_forward(azr_engine* h, const uint8_t* d_in88, int in_stride, int n, float* d_pi, float* d_v, const int* d_map, hipStream_t st);
// train (azr_train.hip)
void train_free(azr_engine* h);
void dp_free(azr_engine* h);
}  // namespace azr
