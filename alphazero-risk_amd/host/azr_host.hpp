// azr_host.hpp — C++ host side above the C-ABI (include/azr.h), shaped like the reference's own seams so that
// code written against JGasp/alphazero-risk reads the same:  Settings (src/settings.h), State (state/state.h),
// NNInputData / NNOutputData / NNTrainData / NNTrainDataStorage (neural_network/alphazero_nn_data.h),
// AlphaZeroNNId / AlphaZeroNNGroup / AlphaZeroCluster (neural_network/alphazero_gpu_cluster.h),
// AlphaZeroMCTS (alphazero_mcts.h), Player / PlayerGroup / AlphaZeroPlayerGroup (player/base/player.h,
// alphazero_player.h), AlphaZeroTrainer (alphazero_trainer.h).  MI355X-first difference: every object is BATCHED over
// the G games of one engine (one engine per GPU) — the reference's thread-per-game fan-out becomes one wavefront per
// game on the device.  No rules, search or net arithmetic happens on the host.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/azr.h"

namespace azrhost {

// ------------------------------------------------------------------------------------------------------------------
// Settings — same field names, flags, defaults and side effects (log/settings.txt) as src/settings.h:19-211.
// `-m learn` is accepted as an alias of the reference's `-m train`.  Extra flags of this build: --blocks, --dtype.
// ------------------------------------------------------------------------------------------------------------------
class Settings {
public:
    std::string MODE = "play";
    std::string DEFAULT_GRAPH_DEF_PB = "model_bin.pb";
    std::string DEFAULT_CHECKPOINT_DIR = "checkpoints";
    std::string DEFAULT_BEST_CHECKPOINT = DEFAULT_CHECKPOINT_DIR + "/best-checkpoint.bin";
    std::string DEFAULT_LATEST_CHECKPOINT = DEFAULT_CHECKPOINT_DIR + "/latest-checkpoint.bin";
    std::string DEFAULT_CHECKPOINT_TEMP = DEFAULT_CHECKPOINT_DIR + "/temp.bin";
    std::string PLAYER_1 = "az", GRAPH_DEF_PB_1 = DEFAULT_GRAPH_DEF_PB, CHECKPOINT_1 = DEFAULT_LATEST_CHECKPOINT;
    std::string PLAYER_2 = "sp", GRAPH_DEF_PB_2 = DEFAULT_GRAPH_DEF_PB, CHECKPOINT_2 = DEFAULT_LATEST_CHECKPOINT;
    std::string DEFAULT_DATA = "data";
    std::string DEFAULT_SAMPLES = DEFAULT_DATA + "/training_samples.bin";
    int NUMBER_OF_GPUS = 1;
    int NUMBER_OF_CONCURENT_GAMES_PER_GPU = 4;
    int AVG_PRED_BATCH_SIZE = 32;
    int THREADS_PER_MCTS = 2;
    int MCTS_SIMULATIONS = 32;
    bool LOG_STATE = false, LOG_NN_TRAINING = true, PERSIST_SAMPLES_DATA = false;
    int MIN_UNIT_MOVE = 3;
    int MAX_GAME_ROUNDS = 30 + 28;
    bool LIMIT_REINFORCEMENT_MOVES = true, LIMIT_ATTACK_MOVES = false, MIRROR_GAMES = true, ALLOW_YIELD = true;
    long TRAIN_ITERATIONS = 10000;
    int TRAIN_ITERATION_GAMES = 1000;
    float HP_EXPLORATION = 1.1f, DIR_NOISE_VALUE = 0.3f, DIR_NOISE_EPSI = 0.25f;
    int TEMPERATURE_TRESHOLD = 15 + 28;
    int COMPARE_GAMES = 1000;
    float COMPARE_TRESHOLD = 0.55f;
    bool INCLUDE_COMPARE_GAMES_TRAIN_SAMPLES = true;
    int BENCHMARK_GAMES_RANDOM = 10, BENCHMARK_GAMES_SCRIPT = 100;
    bool TRAINING_REVERT_MODEL = true;
    int EPOCHS = 10;
    int BATCH_SIZE = 512;
    int SAMPLES_STORAGE_MIN = 1024 * BATCH_SIZE;
    int SAMPLES_STORAGE_MAX = 16384 * BATCH_SIZE;
    float DYNAMIC_EPOCH_THRESHOLD = 0.01f;
    int DATA_GAMES_SS = 5000, DATA_GAMES_SR = 5000, DATA_TRAIN_LOOPS = 1000;
    // this build
    int BLOCKS = 20;              // CMakeLists.txt:15 `set(BLOCKS 20)` is compile-time in the reference
    std::string NET_DTYPE = "bf16";
    uint32_t BASE_SEED = 20260001;
    std::vector<int> DEVICE_MAP;         // --devices: HIP device of logical gpu i (empty = i); "0,0" rehearses --gpus 2 on one card
    bool CONCURRENT_PAIR_HALVES = true;  // --pair-halves: the two games of a mirrored pair on two slots at the same time (AZR_MIRROR_CONCURRENT)

    int deviceOf(int gpu) const { return DEVICE_MAP.empty() ? gpu : DEVICE_MAP.at(gpu); }
    int arenaMirrorMode() const { return !MIRROR_GAMES ? AZR_MIRROR_OFF : CONCURRENT_PAIR_HALVES ? AZR_MIRROR_CONCURRENT : AZR_MIRROR_SEQUENTIAL; }

    int getNumberOfPlayers() const { return NUMBER_OF_GPUS * NUMBER_OF_CONCURENT_GAMES_PER_GPU; }
    void init(int argc, char* argv[]);   // parses every flag of SURVEY App-G; exits on -h/--help
    void toEngine(azr_settings& s, int device) const;
    std::string describe() const;
};
extern Settings SETTINGS;

// one host thread per GPU; a failure inside a thread is raised in the caller after every thread has been joined
void forEachGpu(int P, const char* what, const std::function<void(int)>& body);

// ------------------------------------------------------------------------------------------------------------------
// byte images with the reference's layouts
// ------------------------------------------------------------------------------------------------------------------
enum class RoundPhase : uint8_t { SETUP, SETUP_NEUTRAL, REINFORCEMENT, ATTACK, ATTACK_MOBILIZATION, FORTIFY };

class State {  // image of `struct Data` (state/state.h:86-105); all mutation happens on the device
public:
    uint8_t data[AZR_STATE_BYTES] = {0};
    static const int DRAW = -2, NOT_ENDED = -1;
    uint16_t getRound() const { return (uint16_t)(data[144] | (data[145] << 8)); }
    int8_t getCurrentPlayerTurn() const { return (int8_t)data[146]; }
    RoundPhase getRoundPhase() const { return (RoundPhase)data[149]; }
    uint8_t getReinforcement() const { return data[148]; }
    uint8_t landArmy(int i) const { return data[i] & 63; }
    uint8_t landOwner(int i) const { return data[i] >> 6; }
    uint64_t ownedLands(int player) const { uint64_t v = 0; memcpy(&v, data + 48 + 48 * player, 6); return v; }
};

class NNInputData { public: uint8_t bytes[AZR_INPUT_BYTES] = {0}; };

class NNOutputData {
public:
    std::vector<float> policy;
    float value = 0.0f;
};

class NNTrainData {  // one 265-byte record (alphazero_nn_data.cpp:123-130)
public:
    int8_t playerIndex = 0;
    NNInputData in;
    NNOutputData out;
};

class NNTrainDataStorage {
public:
    std::vector<NNTrainData> data;
    size_t lastGameIndex = 0, oldGameIndex = 0;
    void appendPacked(const uint8_t* rec265, size_t n);
    void extend(NNTrainDataStorage& s) { data.insert(data.end(), s.data.begin(), s.data.end()); }
    void trimOldExamples();  // alphazero_nn_data.cpp:67-84
    void updateValues(int gameStatus, int roundCount);  // alphazero_nn_data.cpp:51-65 (roundCount unused by default)
    void updateOldGamesIndex() { oldGameIndex = data.size() - 1; }  // alphazero_nn_data.cpp:160-163
    std::vector<uint8_t> packed() const;                // 265-byte records, file order
    // writer = the reference's (8-byte size_t count); reader accepts that and the 4-byte count the reference's own
    // reader expects
    void saveTrainingSamples(const std::string& path) const;
    void loadTrainingSamples(const std::string& path);
};

// ------------------------------------------------------------------------------------------------------------------
// NN service
// ------------------------------------------------------------------------------------------------------------------
class Engine {  // RAII over azr_engine
public:
    azr_engine* h = nullptr;
    int games = 0;
    Engine(const Settings& s, int device, int games);
    ~Engine();
    void check(int rc, const char* what) const;
};

class AlphaZeroNNId {  // alphazero_gpu_cluster.h:14-47
public:
    std::shared_ptr<Engine> engine;
    int gpu;
    AlphaZeroNNId(std::shared_ptr<Engine> e, int gpu) : engine(e), gpu(gpu) {}
    void loadCheckpoint(const std::string& path);  // missing file => random init + save (alphazero_nn.cpp:197-202)
    void saveCheckpoint(const std::string& path);
    NNOutputData predict(const NNInputData& in);
    std::vector<NNOutputData> predict(const std::vector<NNInputData>& in);
    // AlphaZeroNN::train (alphazero_nn.cpp:351-410) through azr_nn_train: EPOCH loop, shuffle, floor(n/BATCH_SIZE)
    // optimiser steps, loss prints and log/azr-nn-training-log.txt columns
    void train(const std::vector<NNTrainData>& trainData, int epochs);
};

class AlphaZeroNNGroup {  // alphazero_gpu_cluster.h:76-95: the same net on every GPU
public:
    std::string name;
    std::vector<std::shared_ptr<AlphaZeroNNId>> neuralNetworkIds;
    size_t size() const { return neuralNetworkIds.size(); }
    std::shared_ptr<AlphaZeroNNId> getNN(int i) { return neuralNetworkIds.at(i); }
    void loadCheckpoint(const std::string& path) { for (auto& n : neuralNetworkIds) n->loadCheckpoint(path); }
    void saveCheckpoint(const std::string& path) { neuralNetworkIds.at(0)->saveCheckpoint(path); }
    // alphazero_gpu_cluster.cpp:221-231: train on GPU 0's net, hand the weights to the others through temp.bin
    void train(const std::vector<NNTrainData>& trainData, int epochs);
};

class AlphaZeroCluster {  // alphazero_gpu_cluster.h:97-111
public:
    int gpus = 0;
    std::vector<std::shared_ptr<AlphaZeroNNGroup>> groups;
    void initGpus(int n) { gpus = n; }
    std::shared_ptr<AlphaZeroNNGroup> initPlayerGroup(const std::string& name, const std::string& graphPath);
};

// ------------------------------------------------------------------------------------------------------------------
// search + players, batched
// ------------------------------------------------------------------------------------------------------------------
class AlphaZeroMCTS {  // alphazero_mcts.h:80-95 over the G games of an engine
public:
    std::shared_ptr<AlphaZeroNNId> nn;
    explicit AlphaZeroMCTS(std::shared_ptr<AlphaZeroNNId> nn) : nn(nn) {}
    void clearNodes();
    void trimNodes();
    void simulate(const std::vector<State>& roots);                // all G roots in lock-step
    std::vector<std::vector<float>> calculateMoveProbability();    // temperature 1.0
    std::vector<uint8_t> pickHigestWeightedMove();
    std::vector<uint8_t> pickRandomWeightedMove();
};

class Player {  // player/base/player.h:10-26
public:
    int8_t playerIndexTurn = 0;
    NNTrainDataStorage* trainStorage = nullptr;
    virtual ~Player() {}
    virtual void newGame() {}
    virtual void takeTurn(State&) {}
    virtual void gameFinished(int, int) {}
};

class PlayerGroup {  // player/base/player.h:28-36
public:
    virtual ~PlayerGroup() {}
    virtual size_t size() { return 0; }
    virtual std::shared_ptr<Player> getPlayer(int) { return std::make_shared<Player>(); }
};

// AlphaZeroPlayerGroup (alphazero_player.cpp:36-55): `gpu-games` players per GPU.  takeTurns() is the batched
// AlphaZeroPlayer::takeTurn (alphazero_player.cpp:3-21): for every game whose mover is this group's player index,
// loop {simulate -> pi -> argmax -> makeMove} until the turn passes or the game ends.
class AlphaZeroPlayerGroup : public PlayerGroup {
public:
    std::shared_ptr<AlphaZeroNNGroup> nnGroup;
    std::vector<std::shared_ptr<Player>> players;
    explicit AlphaZeroPlayerGroup(std::shared_ptr<AlphaZeroNNGroup> g);
    size_t size() override { return players.size(); }
    std::shared_ptr<Player> getPlayer(int i) override { return players.at(i); }
    // storages (optional, one per game slot): the records AlphaZeroPlayer::takeTurn pushes when trainStorage is set
    void takeTurns(int gpu, std::vector<State>& states, int8_t playerIndexTurn, std::vector<NNTrainDataStorage>* storages = nullptr);
};

// game/game.h GameResults + operator<< (game.cpp:193-235)
struct GameResults {
    int count = 0, draw = 0;
    struct { int win = 0, winAndStartedGame = 0; } players[2];
    void add(const GameResults& o);
    void addGame(int gameStatus, int startingPlayer);
};
std::ostream& operator<<(std::ostream& os, const GameResults& gr);

// GameGroup::playGames (game.cpp:256-312) on the device arena (azr_arena_*), one host thread per GPU.  AlphaZero vs
// AlphaZero (two nets = two engines per GPU): pg2's network plays AZR_PLAYER_ALPHAZERO_B inside pg1's engine, every slot
// a pair of AlphaZeroPlayers with their own trees playing mirrored pairs with alternating starts (game.cpp:153-191).
// `tds` (optional) collects the (s, pi, z) records both AlphaZero players push during the games
// (alphazero_player.cpp:15-18,24-29), game by game.
class GameGroup {
public:
    static GameResults playGames(AlphaZeroPlayerGroup& pg1, AlphaZeroPlayerGroup& pg2, int games, NNTrainDataStorage* tds = nullptr);
    static GameResults playGames(AlphaZeroPlayerGroup& pg1, int otherKind /* AZR_PLAYER_SCRIPT | RANDOM */, int games);
};

struct SelfPlayReport {
    uint64_t games = 0, decisions = 0, simulations = 0, samples = 0, errors = 0;
    double seconds = 0;
};

class AlphaZeroTrainer {  // alphazero_trainer.h
public:
    NNTrainDataStorage trainStorage;
    long trainIteration = 0;
    std::vector<uint64_t> selfPlayStarted;   // per GPU: self-play games started so far (seed stream position)
    // generateTrainData (alphazero_trainer.cpp:36-78): one host thread per GPU, TRAIN_ITERATION_GAMES games in total,
    // device-resident self-play; the per-GPU storages are concatenated in GPU order
    SelfPlayReport generateTrainData(std::shared_ptr<AlphaZeroNNGroup> generate);
    void train(std::shared_ptr<AlphaZeroNNGroup> trainGroup, std::shared_ptr<AlphaZeroNNGroup> generateGroup);
    bool updateIfImprovement(std::shared_ptr<AlphaZeroNNGroup> trainGroup, std::shared_ptr<AlphaZeroNNGroup> generateGroup, bool doBenchmark);
    void benchmark(AlphaZeroPlayerGroup& azpg);
    static bool isModelImproved(const GameResults& gr);
};

}  // namespace azrhost
