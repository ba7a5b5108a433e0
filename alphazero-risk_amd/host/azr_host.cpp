// azr_host.cpp — implementation of the reference-shaped host classes over the C-ABI (see azr_host.hpp).
#include "azr_host.hpp"

#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <map>
#include <sstream>

namespace azrhost {

Settings SETTINGS;

// ---- Settings ---------------------------------------------------------------------------------------------------
namespace {
struct Opt { const char* name; const char* desc; std::string def; bool is_bool; };

bool parse_bool(const std::string& v) { return !(v == "0" || v == "false" || v == "False" || v == "no"); }

void mkdirs(const std::string& path)
{
    std::string cur;
    for (size_t i = 0; i < path.size(); i++) {
        cur += path[i];
        if (path[i] == '/' || i + 1 == path.size()) mkdir(cur.c_str(), 0777);
    }
}
}  // namespace

// One host thread per GPU (the reference's structure: alphazero_trainer.cpp:48-57, game.cpp:277-312).  An exception thrown in a
// thread body would end the process in std::terminate: it is caught there, every thread is joined, and the first failure is
// raised in the caller's thread with the GPU it came from.
void forEachGpu(int P, const char* what, const std::function<void(int)>& body)
{
    std::vector<std::string> failed(P);
    std::vector<std::thread> threads;
    for (int i = 0; i < P; i++)
        threads.emplace_back([&, i]() {
            try { body(i); }
            catch (const std::exception& ex) { failed[i] = ex.what()[0] ? ex.what() : "unknown error"; }
            catch (...) { failed[i] = "unknown error"; }
        });
    for (auto& t : threads) t.join();
    for (int i = 0; i < P; i++)
        if (!failed[i].empty()) throw std::runtime_error(std::string(what) + " on gpu " + std::to_string(i) + ": " + failed[i]);
}

void Settings::init(int argc, char* argv[])
{
    // name, description (as the reference's help text), default — settings.h:91-137
    std::vector<Opt> opts = {
        {"m", "Mode [train/play]", MODE, false},
        {"g", "Default graph file path", DEFAULT_GRAPH_DEF_PB, false},
        {"c", "Checkpoint file path", DEFAULT_LATEST_CHECKPOINT, false},
        {"p1", "Player 1 [az/sp]", PLAYER_1, false},
        {"g1", "Graph file path for player 1", GRAPH_DEF_PB_1, false},
        {"c1", "Checkpoint file path for player 1", CHECKPOINT_1, false},
        {"p2", "Player 2 [az/sp]", PLAYER_2, false},
        {"g2", "Graph file path for player 2", GRAPH_DEF_PB_2, false},
        {"c2", "Checkpoint file path for player 2", CHECKPOINT_2, false},
        {"gpus", "Number of gpu units", std::to_string(NUMBER_OF_GPUS), false},
        {"gpu-games", "Number of concurent games per gpu", std::to_string(NUMBER_OF_CONCURENT_GAMES_PER_GPU), false},
        {"t", "Number of threads", std::to_string(THREADS_PER_MCTS), false},
        {"apbs", "Set number of games per gpu to get avg. prediction batch size", std::to_string(AVG_PRED_BATCH_SIZE), false},
        {"lnt", "Log nn training", std::to_string(LOG_NN_TRAINING), true},
        {"ls", "Log state", std::to_string(LOG_STATE), true},
        {"dgss", "Number of data games for trin-data script vs script", std::to_string(DATA_GAMES_SS), false},
        {"dgsr", "Number of data games for trin-data script vs random", std::to_string(DATA_GAMES_SR), false},
        {"dtl", "Number of train loops for data games", std::to_string(DATA_TRAIN_LOOPS), false},
        {"allow-yield", "Allow yield when enemy ownes 3/4 of lands", std::to_string(ALLOW_YIELD), true},
        {"limit-reinforcement", "Limit reinforcement moves", std::to_string(LIMIT_REINFORCEMENT_MOVES), true},
        {"limit-attack", "Limit attack moves", std::to_string(LIMIT_ATTACK_MOVES), true},
        {"mirror-games", "Play games in pair with mirrored initial position", std::to_string(MIRROR_GAMES), true},
        {"ti", "Number of train iterations", std::to_string(TRAIN_ITERATIONS), false},
        {"tg", "Games played per train iteration", std::to_string(TRAIN_ITERATION_GAMES), false},
        {"mcts", "Number of MCTS simulations", std::to_string(MCTS_SIMULATIONS), false},
        {"hp", "Exploration factor", std::to_string(HP_EXPLORATION), false},
        {"dnv", "Dirchlet noise value", std::to_string(DIR_NOISE_VALUE), false},
        {"dne", "Dirchlet noise epsi", std::to_string(DIR_NOISE_EPSI), false},
        {"temp", "Temperature trehsold", std::to_string(TEMPERATURE_TRESHOLD), false},
        {"e", "Number of epochs per train iteration", std::to_string(EPOCHS), false},
        {"bs", "Batch size", std::to_string(BATCH_SIZE), false},
        {"cg", "Number of games for comparison", std::to_string(COMPARE_GAMES), false},
        {"ct", "Treshold for accepted improvement", std::to_string(COMPARE_TRESHOLD), false},
        {"s", "Number of stored samples", std::to_string(SAMPLES_STORAGE_MIN), false},
        {"blocks", "[this build] residual blocks of the net (reference: compile-time BLOCKS)", std::to_string(BLOCKS), false},
        {"dtype", "[this build] net arithmetic bf16|f16|f32x|f32 (f32x = fp32-equivalent on the MFMA)", NET_DTYPE, false},
        {"seed", "[this build] base seed of the per-game RNG streams", std::to_string(BASE_SEED), false},
        {"devices", "[this build] HIP device of every logical gpu, comma separated (default 0,1,..; \"0,0\" rehearses --gpus 2 on one card)", "", false},
        {"pair-halves", "[this build] mirrored pairs: 1 = both games of a pair at the same time on two slots, 0 = one after the other on one slot", std::to_string(CONCURRENT_PAIR_HALVES), true},
        {"help", "Display help", "0", true},
    };
    std::map<std::string, std::string> val;
    std::map<std::string, bool> given;
    auto find = [&](const std::string& n) -> const Opt* {
        for (auto& o : opts) if (n == o.name) return &o;
        return nullptr;
    };
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "-h") a = "--help";
        std::string name, value;
        bool has_value = false;
        if (a.rfind("--", 0) == 0) {
            size_t eq = a.find('=');
            name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
            if (eq != std::string::npos) { value = a.substr(eq + 1); has_value = true; }
        } else if (a.size() >= 2 && a[0] == '-') {
            name = a.substr(1, 1);  // cxxopts: one-letter names are short options (-m train, -mtrain, -t 1)
            if (a.size() > 2) { value = a.substr(a[2] == '=' ? 3 : 2); has_value = true; }
        } else {
            fprintf(stderr, "unexpected argument '%s'\n", a.c_str());
            exit(2);
        }
        const Opt* o = find(name);
        if (!o) { fprintf(stderr, "Option '%s' does not exist\n", name.c_str()); exit(2); }
        if (!has_value) {
            if (o->is_bool && (i + 1 >= argc || argv[i + 1][0] == '-')) value = "1";
            else if (i + 1 < argc) value = argv[++i];
            else { fprintf(stderr, "Option '%s' is missing an argument\n", name.c_str()); exit(2); }
        }
        val[name] = value;
        given[name] = true;
    }
    auto get = [&](const char* n) { return given.count(n) ? val[n] : find(n)->def; };
    if (given.count("help")) {
        printf("AlphaZero implementation for game Risk (MI355X-native hot path)\nUsage:\n  AlphaZero-Risk [OPTION...]\n\n");
        for (auto& o : opts) printf("  %s%-22s %s (default: %s)\n", strlen(o.name) == 1 ? " -" : "--", o.name, o.desc, o.def.c_str());
        exit(0);
    }
    MODE = get("m");
    if (MODE == "learn") MODE = "train";  // BASELINE.json calls the reference's `train` mode `learn`
    DEFAULT_GRAPH_DEF_PB = get("g");
    DEFAULT_LATEST_CHECKPOINT = get("c");
    PLAYER_1 = get("p1"); GRAPH_DEF_PB_1 = get("g1"); CHECKPOINT_1 = get("c1");
    PLAYER_2 = get("p2"); GRAPH_DEF_PB_2 = get("g2"); CHECKPOINT_2 = get("c2");
    THREADS_PER_MCTS = atoi(get("t").c_str());
    NUMBER_OF_GPUS = atoi(get("gpus").c_str());
    if (given.count("gpu-games")) NUMBER_OF_CONCURENT_GAMES_PER_GPU = atoi(get("gpu-games").c_str());
    else NUMBER_OF_CONCURENT_GAMES_PER_GPU = AVG_PRED_BATCH_SIZE / (THREADS_PER_MCTS > 0 ? THREADS_PER_MCTS : 1) * 2;
    DATA_GAMES_SS = atoi(get("dgss").c_str());
    DATA_GAMES_SR = atoi(get("dgsr").c_str());
    DATA_TRAIN_LOOPS = atoi(get("dtl").c_str());
    LOG_STATE = parse_bool(get("ls"));
    ALLOW_YIELD = parse_bool(get("allow-yield"));
    LIMIT_REINFORCEMENT_MOVES = parse_bool(get("limit-reinforcement"));
    LIMIT_ATTACK_MOVES = parse_bool(get("limit-attack"));
    MIRROR_GAMES = parse_bool(get("mirror-games"));
    TRAIN_ITERATIONS = atol(get("ti").c_str());
    TRAIN_ITERATION_GAMES = atoi(get("tg").c_str());
    MCTS_SIMULATIONS = atoi(get("mcts").c_str());
    HP_EXPLORATION = (float)atof(get("hp").c_str());
    DIR_NOISE_VALUE = (float)atof(get("dnv").c_str());
    DIR_NOISE_EPSI = (float)atof(get("dne").c_str());
    TEMPERATURE_TRESHOLD = atoi(get("temp").c_str());
    EPOCHS = atoi(get("e").c_str());
    BATCH_SIZE = atoi(get("bs").c_str());
    COMPARE_GAMES = atoi(get("cg").c_str());
    COMPARE_TRESHOLD = (float)atof(get("ct").c_str());
    SAMPLES_STORAGE_MIN = atoi(get("s").c_str());
    BLOCKS = atoi(get("blocks").c_str());
    NET_DTYPE = get("dtype");
    BASE_SEED = (uint32_t)strtoul(get("seed").c_str(), nullptr, 10);
    CONCURRENT_PAIR_HALVES = parse_bool(get("pair-halves"));
    DEVICE_MAP.clear();
    {
        std::stringstream ss(get("devices"));
        std::string tok;
        while (std::getline(ss, tok, ',')) if (!tok.empty()) DEVICE_MAP.push_back(atoi(tok.c_str()));
        if (!DEVICE_MAP.empty() && (int)DEVICE_MAP.size() != NUMBER_OF_GPUS) {
            fprintf(stderr, "--devices names %d devices for --gpus %d\n", (int)DEVICE_MAP.size(), NUMBER_OF_GPUS);
            exit(2);
        }
    }
    // `--lnt` and `--apbs` are parsed and never applied in the reference either
    mkdirs("log");
    std::ofstream out("log/settings.txt", std::ofstream::out);
    for (auto& o : opts) out << o.name << "(" << o.desc << ")=" << (given.count(o.name) ? val[o.name] : o.def) << std::endl;
}

void Settings::toEngine(azr_settings& s, int device) const
{
    azr_default_settings(&s);
    s.device = device;
    s.games = NUMBER_OF_CONCURENT_GAMES_PER_GPU;
    s.blocks = BLOCKS;
    s.net_dtype = NET_DTYPE == "f32" ? AZR_NET_F32 : NET_DTYPE == "f32x" ? AZR_NET_F32X : NET_DTYPE == "f16" ? AZR_NET_F16 : AZR_NET_BF16;
    s.mcts_simulations = MCTS_SIMULATIONS;
    s.mcts_threads = std::max(1, std::min(8, THREADS_PER_MCTS));
    s.allow_yield = ALLOW_YIELD;
    s.limit_reinforcement = LIMIT_REINFORCEMENT_MOVES;
    s.limit_attack = LIMIT_ATTACK_MOVES;
    s.max_game_rounds = MAX_GAME_ROUNDS;
    s.min_unit_move = MIN_UNIT_MOVE;
    s.temperature_threshold = TEMPERATURE_TRESHOLD;
    s.hp_exploration = HP_EXPLORATION;
    s.dir_noise_value = DIR_NOISE_VALUE;
    s.dir_noise_epsi = DIR_NOISE_EPSI;
}

std::string Settings::describe() const
{
    std::ostringstream o;
    o << "GPUs: " << NUMBER_OF_GPUS << ", Games per GPU " << NUMBER_OF_CONCURENT_GAMES_PER_GPU << ", MCTS threads: "
      << THREADS_PER_MCTS << ", MCTS simulations " << MCTS_SIMULATIONS;
    return o.str();
}

// ---- samples ---------------------------------------------------------------------------------------------------------
void NNTrainDataStorage::appendPacked(const uint8_t* rec, size_t n)
{
    data.reserve(data.size() + n);
    for (size_t i = 0; i < n; i++, rec += AZR_RECORD_BYTES) {
        NNTrainData d;
        d.playerIndex = (int8_t)rec[0];
        memcpy(d.in.bytes, rec + 1, AZR_INPUT_BYTES);
        memcpy(&d.out.value, rec + 89, 4);
        d.out.policy.resize(AZR_MOVES);
        memcpy(d.out.policy.data(), rec + 93, AZR_MOVES * 4);
        data.push_back(std::move(d));
    }
}

void NNTrainDataStorage::trimOldExamples()
{
    if (data.size() > (size_t)SETTINGS.SAMPLES_STORAGE_MAX) {
        size_t excess = data.size() - SETTINGS.SAMPLES_STORAGE_MAX;
        data.erase(data.begin(), data.begin() + excess);
        printf("[MAX] Erased %d oldeset examples\n", int(excess));
    } else if (data.size() > (size_t)SETTINGS.SAMPLES_STORAGE_MIN && oldGameIndex > 0) {
        size_t excess = std::min(oldGameIndex, data.size() - SETTINGS.SAMPLES_STORAGE_MIN);
        oldGameIndex -= excess;
        data.erase(data.begin(), data.begin() + excess);
        printf("[MIN] Erased %d oldeset examples\n", int(excess));
    }
}

void NNTrainDataStorage::updateValues(int gameStatus, int roundCount)
{
    (void)roundCount;  // ROUND_WEIGHTED_VALUE is not a default macro
    for (size_t i = lastGameIndex; i < data.size(); i++)
        data[i].out.value = gameStatus == State::DRAW ? 0.0f : data[i].playerIndex == gameStatus ? 1.0f : -1.0f;
    lastGameIndex = data.size();
}

std::vector<uint8_t> NNTrainDataStorage::packed() const
{
    std::vector<uint8_t> buf(data.size() * AZR_RECORD_BYTES);
    for (size_t i = 0; i < data.size(); i++) {
        uint8_t* r = buf.data() + i * AZR_RECORD_BYTES;
        r[0] = (uint8_t)data[i].playerIndex;
        memcpy(r + 1, data[i].in.bytes, AZR_INPUT_BYTES);
        memcpy(r + 89, &data[i].out.value, 4);
        memcpy(r + 93, data[i].out.policy.data(), 4 * AZR_MOVES);
    }
    return buf;
}

void NNTrainDataStorage::saveTrainingSamples(const std::string& path) const
{
    if (data.empty()) { printf("No training samples\n"); return; }
    size_t slash = path.find_last_of('/');
    if (slash != std::string::npos) mkdirs(path.substr(0, slash));
    std::ofstream out(path, std::ios::out | std::ios::binary);
    uint64_t size = data.size();
    out.write((const char*)&size, 8);
    for (auto& d : data) {
        out.write((const char*)&d.playerIndex, 1);
        out.write((const char*)d.in.bytes, AZR_INPUT_BYTES);
        out.write((const char*)&d.out.value, 4);
        out.write((const char*)d.out.policy.data(), 4 * AZR_MOVES);
    }
    printf("Training samples saved %d\n", int(data.size()));
}

void NNTrainDataStorage::loadTrainingSamples(const std::string& path)
{
    std::ifstream in(path, std::ios::in | std::ios::binary | std::ios::ate);
    if (!in) { printf("File does note exist: %s\n", path.c_str()); return; }
    const uint64_t bytes = (uint64_t)in.tellg();
    in.seekg(0);
    // the reference's writer emits an 8-byte count, its reader consumes 4 bytes: accept whichever fits the file size
    uint64_t n8 = 0;
    in.read((char*)&n8, 8);
    size_t header = 8;
    uint64_t n = n8;
    if (bytes != 8 + n8 * AZR_RECORD_BYTES) {
        uint32_t n4 = (uint32_t)n8;
        if (bytes == 4 + (uint64_t)n4 * AZR_RECORD_BYTES) { n = n4; header = 4; }
        else { printf("Unrecognised sample file: %s\n", path.c_str()); return; }
    }
    in.seekg(header);
    std::vector<uint8_t> buf(n * AZR_RECORD_BYTES);
    in.read((char*)buf.data(), buf.size());
    appendPacked(buf.data(), n);
}

// ---- engine / NN service ------------------------------------------------------------------------------------------
Engine::Engine(const Settings& s, int device, int g) : games(g)
{
    azr_settings es;
    s.toEngine(es, device);
    es.games = g;
    int rc = azr_engine_create(&es, &h);
    if (rc) {
        std::string msg = azr_last_error(nullptr);   // *out is NULL on failure; the reason is kept per thread
        h = nullptr;
        throw std::runtime_error("engine: " + msg);
    }
}
Engine::~Engine() { if (h) azr_engine_destroy(h); }
void Engine::check(int rc, const char* what) const
{
    if (rc == AZR_E_INVALID_ARGUMENT) throw std::invalid_argument(std::string(what) + ": " + azr_last_error(h));
    if (rc == AZR_E_LOGIC) throw std::logic_error(std::string(what) + ": " + azr_last_error(h));
    if (rc) throw std::runtime_error(std::string(what) + ": " + azr_last_error(h));
}

void AlphaZeroNNId::loadCheckpoint(const std::string& path)
{
    struct stat st;
    if (stat(path.c_str(), &st) == 0) {
        engine->check(azr_nn_load(engine->h, path.c_str()), "loadCheckpoint");
        printf("Loaded checkpoint %s\n", path.c_str());
    } else {  // alphazero_nn.cpp:197-202: no checkpoint => initialise and save one
        printf("Checkpoint %s not found, initializing random weights\n", path.c_str());
        engine->check(azr_nn_init_random(engine->h, 20260002ull), "init");
        saveCheckpoint(path);
    }
}
void AlphaZeroNNId::saveCheckpoint(const std::string& path)
{
    size_t slash = path.find_last_of('/');
    if (slash != std::string::npos) mkdirs(path.substr(0, slash));
    engine->check(azr_nn_save(engine->h, path.c_str()), "saveCheckpoint");
}
NNOutputData AlphaZeroNNId::predict(const NNInputData& in) { return predict(std::vector<NNInputData>{in})[0]; }
std::vector<NNOutputData> AlphaZeroNNId::predict(const std::vector<NNInputData>& in)
{
    const int n = (int)in.size();
    std::vector<uint8_t> x((size_t)n * AZR_INPUT_BYTES);
    for (int i = 0; i < n; i++) memcpy(x.data() + (size_t)i * AZR_INPUT_BYTES, in[i].bytes, AZR_INPUT_BYTES);
    std::vector<float> pi((size_t)n * AZR_MOVES), v(n);
    engine->check(azr_nn_predict(engine->h, x.data(), n, pi.data(), v.data()), "predict");
    std::vector<NNOutputData> out(n);
    for (int i = 0; i < n; i++) {
        out[i].policy.assign(pi.begin() + (size_t)i * AZR_MOVES, pi.begin() + (size_t)(i + 1) * AZR_MOVES);
        out[i].value = v[i];
    }
    return out;
}

// stands in for the process-global RNG engine the reference shuffles with (src/rng.h:50); raw minstd_rand0 state
static uint32_t g_shuffle_state = 1;
static std::ofstream& logFile(const char* path)
{
    static std::map<std::string, std::ofstream> files;
    auto& f = files[path];
    if (!f.is_open()) { mkdirs("log"); f.open(path, std::ofstream::out | std::ofstream::app); }
    return f;
}

void AlphaZeroNNId::train(const std::vector<NNTrainData>& trainData, int epochs)
{
    std::vector<uint8_t> buf(trainData.size() * AZR_RECORD_BYTES);
    for (size_t i = 0; i < trainData.size(); i++) {
        uint8_t* r = buf.data() + i * AZR_RECORD_BYTES;
        r[0] = (uint8_t)trainData[i].playerIndex;
        memcpy(r + 1, trainData[i].in.bytes, AZR_INPUT_BYTES);
        memcpy(r + 89, &trainData[i].out.value, 4);
        memcpy(r + 93, trainData[i].out.policy.data(), 4 * AZR_MOVES);
    }
    printf("Started training\n");
    std::vector<float> lp(std::max(epochs, 1)), lv(std::max(epochs, 1));
    auto t0 = std::chrono::steady_clock::now();
    engine->check(azr_nn_train(engine->h, buf.data(), trainData.size(), epochs, SETTINGS.BATCH_SIZE, &g_shuffle_state, lp.data(), lv.data()),
                  "train");
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int e = 0; e < epochs; e++) {
        printf("EPOCH %d\nLoss Policy / Value: %f / %f\n", e, lp[e], lv[e]);
        if (SETTINGS.LOG_NN_TRAINING) logFile("log/azr-nn-training-log.txt") << lp[e] << ", " << lv[e] << ", ";
    }
    if (SETTINGS.LOG_NN_TRAINING) logFile("log/azr-nn-training-log.txt") << std::endl;
    const size_t steps = (size_t)epochs * (trainData.size() / SETTINGS.BATCH_SIZE);
    printf("Trained %zu minibatch steps of %d in %.2f s (%.1f ms/step)\n", steps, SETTINGS.BATCH_SIZE, dt, steps ? 1e3 * dt / steps : 0.0);
}

void AlphaZeroNNGroup::train(const std::vector<NNTrainData>& trainData, int epochs)
{
    auto& nn = neuralNetworkIds.at(0);
    nn->train(trainData, epochs);
    nn->saveCheckpoint(SETTINGS.DEFAULT_CHECKPOINT_TEMP);
    for (size_t i = 1; i < neuralNetworkIds.size(); i++) neuralNetworkIds[i]->loadCheckpoint(SETTINGS.DEFAULT_CHECKPOINT_TEMP);
}

std::shared_ptr<AlphaZeroNNGroup> AlphaZeroCluster::initPlayerGroup(const std::string& name, const std::string& graphPath)
{
    for (auto& g : groups)
        if (g->name == name) throw std::invalid_argument("Duplicated player group");  // alphazero_gpu_cluster.cpp:160-163
    (void)graphPath;  // the TF graph-def is not used: the net is built into the HIP library (blocks = --blocks)
    auto grp = std::make_shared<AlphaZeroNNGroup>();
    grp->name = name;
    for (int gpu = 0; gpu < gpus; gpu++) {
        auto eng = std::make_shared<Engine>(SETTINGS, SETTINGS.deviceOf(gpu), SETTINGS.NUMBER_OF_CONCURENT_GAMES_PER_GPU);
        grp->neuralNetworkIds.push_back(std::make_shared<AlphaZeroNNId>(eng, gpu));
    }
    groups.push_back(grp);
    return grp;
}

// ---- search ------------------------------------------------------------------------------------------------------------
void AlphaZeroMCTS::clearNodes() { nn->engine->check(azr_mcts_clear(nn->engine->h), "clearNodes"); }
void AlphaZeroMCTS::trimNodes() { nn->engine->check(azr_mcts_trim(nn->engine->h), "trimNodes"); }
void AlphaZeroMCTS::simulate(const std::vector<State>& roots)
{
    Engine& e = *nn->engine;
    if ((int)roots.size() != e.games) throw std::invalid_argument("simulate: one root per engine game expected");
    std::vector<uint8_t> img((size_t)e.games * AZR_STATE_BYTES);
    for (int g = 0; g < e.games; g++) memcpy(img.data() + (size_t)g * AZR_STATE_BYTES, roots[g].data, AZR_STATE_BYTES);
    e.check(azr_engine_set_states(e.h, img.data()), "set_states");
    e.check(azr_mcts_simulate(e.h), "simulate");
}
std::vector<std::vector<float>> AlphaZeroMCTS::calculateMoveProbability()
{
    Engine& e = *nn->engine;
    std::vector<float> pi((size_t)e.games * AZR_MOVES);
    e.check(azr_mcts_policy(e.h, pi.data()), "policy");
    std::vector<std::vector<float>> out(e.games);
    for (int g = 0; g < e.games; g++) out[g].assign(pi.begin() + (size_t)g * AZR_MOVES, pi.begin() + (size_t)(g + 1) * AZR_MOVES);
    return out;
}
std::vector<uint8_t> AlphaZeroMCTS::pickHigestWeightedMove()
{
    std::vector<uint8_t> mv(nn->engine->games);
    nn->engine->check(azr_mcts_pick(nn->engine->h, 0, mv.data()), "pick");
    return mv;
}
std::vector<uint8_t> AlphaZeroMCTS::pickRandomWeightedMove()
{
    std::vector<uint8_t> mv(nn->engine->games);
    nn->engine->check(azr_mcts_pick(nn->engine->h, 1, mv.data()), "pick");
    return mv;
}

AlphaZeroPlayerGroup::AlphaZeroPlayerGroup(std::shared_ptr<AlphaZeroNNGroup> g) : nnGroup(g)
{
    for (size_t i = 0; i < g->size(); i++)
        for (int k = 0; k < SETTINGS.NUMBER_OF_CONCURENT_GAMES_PER_GPU; k++) players.push_back(std::make_shared<Player>());
}

void AlphaZeroPlayerGroup::takeTurns(int gpu, std::vector<State>& states, int8_t playerIndexTurn, std::vector<NNTrainDataStorage>* storages)
{
    auto nn = nnGroup->getNN(gpu);
    Engine& e = *nn->engine;
    AlphaZeroMCTS mcts(nn);
    const int G = e.games;
    if ((int)states.size() != G) throw std::invalid_argument("takeTurns: one State per engine game expected");
    std::vector<uint8_t> img((size_t)G * AZR_STATE_BYTES);
    std::vector<int8_t> status(G);
    mcts.trimNodes();  // AlphaZeroPlayer::takeTurn's own trim (alphazero_player.cpp:5)
    for (;;) {
        for (int g = 0; g < G; g++) memcpy(img.data() + (size_t)g * AZR_STATE_BYTES, states[g].data, AZR_STATE_BYTES);
        e.check(azr_engine_set_states(e.h, img.data()), "set_states");
        e.check(azr_engine_status(e.h, status.data()), "status");
        bool any = false;
        for (int g = 0; g < G; g++) any |= status[g] == -1 && states[g].getCurrentPlayerTurn() == playerIndexTurn;
        if (!any) break;
        e.check(azr_mcts_simulate(e.h), "simulate");
        std::vector<uint8_t> mv = mcts.pickHigestWeightedMove();
        for (int g = 0; g < G; g++)
            if (!(status[g] == -1 && states[g].getCurrentPlayerTurn() == playerIndexTurn)) mv[g] = 255;
        if (storages) {  // alphazero_player.cpp:15-18
            std::vector<uint8_t> in88((size_t)G * AZR_INPUT_BYTES);
            std::vector<float> pi((size_t)G * AZR_MOVES);
            e.check(azr_engine_encode(e.h, in88.data()), "encode");
            e.check(azr_mcts_policy(e.h, pi.data()), "policy");
            for (int g = 0; g < G; g++) {
                if (mv[g] == 255) continue;
                NNTrainData d;
                d.playerIndex = playerIndexTurn;
                memcpy(d.in.bytes, in88.data() + (size_t)g * AZR_INPUT_BYTES, AZR_INPUT_BYTES);
                d.out.policy.assign(pi.begin() + (size_t)g * AZR_MOVES, pi.begin() + (size_t)(g + 1) * AZR_MOVES);
                (*storages)[g].data.push_back(std::move(d));
            }
        }
        e.check(azr_engine_make_moves(e.h, mv.data(), nullptr), "makeMove");
        e.check(azr_engine_get_states(e.h, img.data()), "get_states");
        for (int g = 0; g < G; g++) memcpy(states[g].data, img.data() + (size_t)g * AZR_STATE_BYTES, AZR_STATE_BYTES);
    }
}

// ---- trainer ------------------------------------------------------------------------------------------------------------
SelfPlayReport AlphaZeroTrainer::generateTrainData(std::shared_ptr<AlphaZeroNNGroup> generate)
{
    const int P = (int)generate->size();
    const uint64_t target = (uint64_t)SETTINGS.TRAIN_ITERATION_GAMES;
    printf("Generating training data current sample count %d\n", int(trainStorage.data.size()));
    std::vector<NNTrainDataStorage> storageGroup(P);
    std::vector<SelfPlayReport> rep(P);
    // GPU i's seed stream: base + i * 2^24 + (games this GPU has started in earlier iterations) — no game of any
    // (iteration, GPU) pair is ever replayed.  Shares, seeds and the stream positions are fixed HERE, by the parent, before
    // any thread exists: the threads only read their own copies.
    if (selfPlayStarted.size() < (size_t)P) selfPlayStarted.resize(P, 0);
    std::vector<uint64_t> shares(P);
    std::vector<uint32_t> seeds(P);
    for (int i = 0; i < P; i++) {
        shares[i] = target / P + ((uint64_t)i < target % P ? 1 : 0);
        seeds[i] = SETTINGS.BASE_SEED + (uint32_t)i * (1u << 24) + (uint32_t)selfPlayStarted[i];
        selfPlayStarted[i] += shares[i];
    }
    auto t0 = std::chrono::steady_clock::now();
    forEachGpu(P, "self-play", [&](int i) {  // one self-play thread per GPU (alphazero_trainer.cpp:48-57)
        const uint64_t share = shares[i];
        const uint32_t seed = seeds[i];
        if (share == 0) return;
        Engine& e = *generate->getNN(i)->engine;
        // exactly `share` games are started and every one is played to its end (Counter::hasNext over
        // TRAIN_ITERATION_GAMES, alphazero_trainer.cpp:83)
        e.check(azr_selfplay_start_games(e.h, seed, share), "selfplay_start_games");
        azr_counters c{};
        std::vector<uint8_t> buf((size_t)e.games * 512 * AZR_RECORD_BYTES);
        while (c.games_finished + c.errors < share) {
            e.check(azr_selfplay_run(e.h, 4 * (SETTINGS.MCTS_SIMULATIONS + 2)), "selfplay_run");
            e.check(azr_selfplay_counters(e.h, &c), "counters");
            if (c.records_dropped) throw std::runtime_error("self-play records were dropped (sample_capacity too small)");
            for (size_t n = buf.size() / AZR_RECORD_BYTES; n == buf.size() / AZR_RECORD_BYTES;) {  // a partial drain keeps the rest
                e.check(azr_samples_drain(e.h, buf.data(), buf.size() / AZR_RECORD_BYTES, &n), "drain");
                storageGroup[i].appendPacked(buf.data(), n);
            }
            printf("\r[gpu %d] games %llu/%llu  decisions %llu  simulations %llu", i, (unsigned long long)c.games_finished,
                   (unsigned long long)share, (unsigned long long)c.decisions, (unsigned long long)c.simulations);
            fflush(stdout);
        }
        rep[i].games = c.games_finished; rep[i].decisions = c.decisions; rep[i].simulations = c.simulations;
        rep[i].samples = storageGroup[i].data.size(); rep[i].errors = c.errors;
    });
    SelfPlayReport tot;
    tot.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const size_t before = trainStorage.data.size();
    for (int i = 0; i < P; i++) {  // the "all-gather": concatenation in GPU order (alphazero_trainer.cpp:59-62)
        trainStorage.extend(storageGroup[i]);
        tot.games += rep[i].games; tot.decisions += rep[i].decisions; tot.simulations += rep[i].simulations;
        tot.samples += rep[i].samples; tot.errors += rep[i].errors;
    }
    printf("\nGenerated %d new samples for total %d\n", int(trainStorage.data.size() - before), int(trainStorage.data.size()));
    return tot;
}

void AlphaZeroTrainer::train(std::shared_ptr<AlphaZeroNNGroup> trainGroup, std::shared_ptr<AlphaZeroNNGroup> generateGroup)
{
    trainStorage.loadTrainingSamples(SETTINGS.DEFAULT_SAMPLES);
    printf("Started training\n");
    for (trainIteration = 0; trainIteration < SETTINGS.TRAIN_ITERATIONS; trainIteration++) {
        printf("Train iteration %ld\n", trainIteration);
        SelfPlayReport r = generateTrainData(generateGroup);
        printf("Self-play: %llu games, %llu decisions, %llu simulations in %.2f s  =>  %.0f simulations/s, %.2f games/s\n",
               (unsigned long long)r.games, (unsigned long long)r.decisions, (unsigned long long)r.simulations, r.seconds,
               r.simulations / r.seconds, r.games / r.seconds);
        trainStorage.trimOldExamples();
        trainGroup->train(trainStorage.data, SETTINGS.EPOCHS);
        if (updateIfImprovement(trainGroup, generateGroup, true)) trainStorage.updateOldGamesIndex();
    }
    trainStorage.saveTrainingSamples(SETTINGS.DEFAULT_SAMPLES);
}

// ---- game driver -----------------------------------------------------------------------------------------------------
void GameResults::add(const GameResults& o)
{
    count += o.count; draw += o.draw;
    for (int p = 0; p < 2; p++) { players[p].win += o.players[p].win; players[p].winAndStartedGame += o.players[p].winAndStartedGame; }
}
void GameResults::addGame(int gameStatus, int startingPlayer)  // game.cpp:193-213
{
    count++;
    if (gameStatus == State::DRAW) { draw++; return; }
    players[gameStatus].win++;
    if (gameStatus == startingPlayer) players[gameStatus].winAndStartedGame++;
}
std::ostream& operator<<(std::ostream& os, const GameResults& gr)  // game.cpp:227-235
{
    return os << gr.draw << ", " << gr.players[0].win << "/" << gr.players[0].winAndStartedGame << ", " << gr.players[1].win << "/"
              << gr.players[1].winAndStartedGame;
}

// every playGames call of a run plays other games: the reference draws from its process-global RNG, here a call counter
// moves the base seed (a compare / benchmark round must not replay the previous iteration's deals)
static std::atomic<uint32_t> arenaCallCounter{0};

GameResults GameGroup::playGames(AlphaZeroPlayerGroup& pg1, AlphaZeroPlayerGroup& pg2, int games, NNTrainDataStorage* tds)
{
    const uint32_t arenaCallsBase = ++arenaCallCounter;
    // Device-resident arena: pg1's engine of GPU i runs the G slots, pg2's network of the same GPU plays
    // AZR_PLAYER_ALPHAZERO_B (own tree per slot, evaluated on its own leaves).  Every slot is one of the reference's
    // threadPlayGame threads taking mirrored pairs from the shared counter (game.cpp:238-254).
    const int P = (int)pg1.nnGroup->size();
    printf("Playing games %d\n", games);
    std::vector<azr_game_results> res(P);
    std::vector<NNTrainDataStorage> st(P);
    const int pairs = games / 2;  // Counter::hasNext(2): whole pairs only
    const int mirror = SETTINGS.arenaMirrorMode();
    forEachGpu(P, "compare games", [&](int i) {
        Engine& e = *pg1.nnGroup->getNN(i)->engine;
        Engine& o = *pg2.nnGroup->getNN(i)->engine;
        memset(&res[i], 0, sizeof res[i]);
        const int share = 2 * (pairs / P + (i < pairs % P ? 1 : 0));
        if (share == 0) return;
        e.check(azr_arena_set_opponent_net(e.h, o.h), "arena_set_opponent_net");
        e.check(azr_arena_collect_samples(e.h, tds ? 1 : 0), "arena_collect_samples");
        e.check(azr_arena_start(e.h, AZR_PLAYER_ALPHAZERO, AZR_PLAYER_ALPHAZERO_B, share, 0, mirror,
                                SETTINGS.BASE_SEED + 7919u * arenaCallsBase + (uint32_t)i * (1u << 24)), "arena_start");
        int fin = 0;
        std::vector<uint8_t> buf;
        while (!fin) {
            e.check(azr_arena_run(e.h, 4 * (SETTINGS.MCTS_SIMULATIONS + 2), &fin), "arena_run");
            e.check(azr_arena_results(e.h, &res[i]), "arena_results");
            if (tds) {
                buf.resize((size_t)e.games * 512 * AZR_RECORD_BYTES);
                for (size_t n = buf.size() / AZR_RECORD_BYTES; n == buf.size() / AZR_RECORD_BYTES;) {
                    e.check(azr_samples_drain(e.h, buf.data(), buf.size() / AZR_RECORD_BYTES, &n), "drain");
                    st[i].appendPacked(buf.data(), n);
                }
            }
            if (i == 0) {
                printf("\r%d/%d [Draw/P1,P2]: %d, %d/%d, %d/%d", res[i].count, share, res[i].draw, res[i].win[0], res[i].win_and_started[0],
                       res[i].win[1], res[i].win_and_started[1]);
                fflush(stdout);
            }
        }
        e.check(azr_arena_collect_samples(e.h, 0), "arena_collect_samples");
        e.check(azr_arena_set_opponent_net(e.h, nullptr), "arena_set_opponent_net");
    });
    printf("\n");
    GameResults all;
    for (auto& r : res) {
        all.count += r.count; all.draw += r.draw;
        for (int p = 0; p < 2; p++) { all.players[p].win += r.win[p]; all.players[p].winAndStartedGame += r.win_and_started[p]; }
    }
    if (tds) for (auto& s : st) tds->extend(s);
    return all;
}

GameResults GameGroup::playGames(AlphaZeroPlayerGroup& pg1, int otherKind, int games)
{
    const uint32_t arenaCallsBase = ++arenaCallCounter;
    const int P = (int)pg1.nnGroup->size();
    printf("Playing games %d\n", games);
    std::vector<azr_game_results> res(P);
    const int pairs = games / 2;
    const int mirror = SETTINGS.arenaMirrorMode();
    forEachGpu(P, "benchmark games", [&](int i) {
        Engine& e = *pg1.nnGroup->getNN(i)->engine;
        const int share = 2 * (pairs / P + (i < pairs % P ? 1 : 0));
        memset(&res[i], 0, sizeof res[i]);
        if (share == 0) return;
        e.check(azr_arena_start(e.h, AZR_PLAYER_ALPHAZERO, otherKind, share, 0, mirror,
                                SETTINGS.BASE_SEED + 104729u * arenaCallsBase + (uint32_t)i * (1u << 24)), "arena_start");
        int fin = 0;
        while (!fin) e.check(azr_arena_run(e.h, 4 * (SETTINGS.MCTS_SIMULATIONS + 2), &fin), "arena_run");
        e.check(azr_arena_results(e.h, &res[i]), "arena_results");
    });
    GameResults all;
    for (auto& r : res) {
        all.count += r.count; all.draw += r.draw;
        for (int p = 0; p < 2; p++) { all.players[p].win += r.win[p]; all.players[p].winAndStartedGame += r.win_and_started[p]; }
    }
    return all;
}

// ---- trainer: model selection (alphazero_trainer.cpp:121-198) ----------------------------------------------------------
void AlphaZeroTrainer::benchmark(AlphaZeroPlayerGroup& azpg)
{
    printf("Playing benchmark games with random\n");
    GameResults rGR = GameGroup::playGames(azpg, AZR_PLAYER_RANDOM, SETTINGS.BENCHMARK_GAMES_RANDOM);
    printf("Model benchmark games played: %d \t Model: %d/%d \t Random: %d/%d\n", rGR.count, rGR.players[0].win,
           rGR.players[0].winAndStartedGame, rGR.players[1].win, rGR.players[1].winAndStartedGame);
    printf("Playing benchmark games with script\n");
    GameResults sGR = GameGroup::playGames(azpg, AZR_PLAYER_SCRIPT, SETTINGS.BENCHMARK_GAMES_SCRIPT);
    printf("Model benchmark games played: %d \t Model: %d/%d \t Script: %d/%d\n", sGR.count, sGR.players[0].win,
           sGR.players[0].winAndStartedGame, sGR.players[1].win, sGR.players[1].winAndStartedGame);
    logFile("log/azr-benchmark-log.txt") << trainIteration << ',' << rGR << ", " << sGR << std::endl;
}

bool AlphaZeroTrainer::isModelImproved(const GameResults& gr)
{
    // int >= int * float: evaluated in float like the reference's expression (alphazero_trainer.cpp:197)
    return (float)gr.players[0].win >= (float)(gr.players[0].win + gr.players[1].win) * SETTINGS.COMPARE_TRESHOLD;
}

bool AlphaZeroTrainer::updateIfImprovement(std::shared_ptr<AlphaZeroNNGroup> trainGroup, std::shared_ptr<AlphaZeroNNGroup> generateGroup,
                                           bool doBenchmark)
{
    const std::string iterCkpt = SETTINGS.DEFAULT_CHECKPOINT_DIR + "/checkpoint-iter-" + std::to_string(trainIteration) + ".bin";
    if (SETTINGS.COMPARE_GAMES > 0) {
        const size_t samples = trainStorage.data.size();
        AlphaZeroPlayerGroup trainAZPG(trainGroup), generateAZPG(generateGroup);
        printf("Playing comparison games betweean new and old model\n");
        GameResults gr;
        if (SETTINGS.INCLUDE_COMPARE_GAMES_TRAIN_SAMPLES) {
            gr = GameGroup::playGames(trainAZPG, generateAZPG, SETTINGS.COMPARE_GAMES, &trainStorage);
            printf("New samples generated from compare games %d\n", int(trainStorage.data.size() - samples));
        } else {
            gr = GameGroup::playGames(trainAZPG, generateAZPG, SETTINGS.COMPARE_GAMES);
        }
        logFile("log/azr-improvement-log.txt") << trainIteration << ',' << gr << std::endl;
        if (isModelImproved(gr)) {
            printf("Model improved\n");
            trainGroup->saveCheckpoint(SETTINGS.DEFAULT_BEST_CHECKPOINT);
            trainGroup->saveCheckpoint(iterCkpt);
            generateGroup->loadCheckpoint(SETTINGS.DEFAULT_BEST_CHECKPOINT);
            if (doBenchmark) benchmark(generateAZPG);
            return true;
        }
        printf("Model did not improve\n");
        if (SETTINGS.TRAINING_REVERT_MODEL) {
            printf("Model reverted back old\n");
            trainGroup->loadCheckpoint(SETTINGS.DEFAULT_LATEST_CHECKPOINT);
        }
        return false;
    }
    printf("Model improved (No compare games set)\n");
    trainGroup->saveCheckpoint(SETTINGS.DEFAULT_BEST_CHECKPOINT);
    trainGroup->saveCheckpoint(iterCkpt);
    generateGroup->loadCheckpoint(SETTINGS.DEFAULT_BEST_CHECKPOINT);
    AlphaZeroPlayerGroup generateAZPG(generateGroup);
    if (doBenchmark) benchmark(generateAZPG);
    return true;
}

}  // namespace azrhost
