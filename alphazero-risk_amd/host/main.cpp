// AlphaZero_Risk_hip — CLI of the MI355X-native build; flags and modes as src/alphazero_risk.cpp:160-199 /
// src/settings.h:91-137 (`-m learn` = `-m train`).
#include <cstdio>
#include <thread>

#include "azr_host.hpp"

using namespace azrhost;

static void executeTrain()
{
    auto cluster = std::make_shared<AlphaZeroCluster>();
    cluster->initGpus(SETTINGS.NUMBER_OF_GPUS);
    auto trainGroup = cluster->initPlayerGroup("az_train", SETTINGS.GRAPH_DEF_PB_1);
    trainGroup->loadCheckpoint(SETTINGS.DEFAULT_LATEST_CHECKPOINT);
    auto generateGroup = cluster->initPlayerGroup("az_generate", SETTINGS.GRAPH_DEF_PB_1);
    generateGroup->loadCheckpoint(SETTINGS.DEFAULT_LATEST_CHECKPOINT);
    AlphaZeroTrainer trainer;
    trainer.train(trainGroup, generateGroup);
}

// `-m play` (src/alphazero_risk.cpp:4-47): GameGroup::playGames(group1, group2, COMPARE_GAMES) on the device arena.
// az vs sp / rp (either side) and sp / rp among themselves run through azr_arena_*; az vs az goes through the batched
// Player seam (AlphaZeroPlayerGroup::takeTurns) with one shared net.
static int playerKind(const std::string& p)
{
    if (p == "az") return AZR_PLAYER_ALPHAZERO;
    if (p == "sp") return AZR_PLAYER_SCRIPT;
    if (p == "rp") return AZR_PLAYER_RANDOM;
    throw std::invalid_argument("unknown player '" + p + "' (az/sp/rp)");
}

static void executePlayAzVsAz(std::shared_ptr<AlphaZeroNNGroup> group)
{
    AlphaZeroPlayerGroup players(group);
    Engine& e = *group->getNN(0)->engine;
    const int G = e.games;
    int wins[2] = {0, 0}, draws = 0, count = 0;
    uint32_t next_seed = SETTINGS.BASE_SEED;
    while (count < SETTINGS.COMPARE_GAMES) {
        std::vector<uint32_t> seeds(G);
        for (int g = 0; g < G; g++) seeds[g] = next_seed++;
        e.check(azr_engine_new_games(e.h, seeds.data()), "new_games");
        std::vector<uint8_t> img((size_t)G * AZR_STATE_BYTES);
        e.check(azr_engine_get_states(e.h, img.data()), "get_states");
        std::vector<State> states(G);
        for (int g = 0; g < G; g++) memcpy(states[g].data, img.data() + (size_t)g * AZR_STATE_BYTES, AZR_STATE_BYTES);
        std::vector<int8_t> status(G, -1);
        for (;;) {
            players.takeTurns(0, states, 0);
            players.takeTurns(0, states, 1);
            e.check(azr_engine_status(e.h, status.data()), "status");
            bool running = false;
            for (int g = 0; g < G; g++) running |= status[g] == -1;
            if (!running) break;
        }
        for (int g = 0; g < G && count < SETTINGS.COMPARE_GAMES; g++, count++) {
            if (status[g] == State::DRAW) draws++;
            else wins[status[g]]++;
        }
        printf("\rGames: %d", count);
        fflush(stdout);
    }
    printf("\nGames: %d\nDraws:%d\nPlayer 1:%d\nPlayer 2:%d\n", count, draws, wins[0], wins[1]);
}

static void executePlay()
{
    const int k1 = playerKind(SETTINGS.PLAYER_1), k2 = playerKind(SETTINGS.PLAYER_2);
    auto cluster = std::make_shared<AlphaZeroCluster>();
    cluster->initGpus(SETTINGS.NUMBER_OF_GPUS);
    auto group = cluster->initPlayerGroup("az1", SETTINGS.GRAPH_DEF_PB_1);
    if (k1 == AZR_PLAYER_ALPHAZERO) group->loadCheckpoint(SETTINGS.CHECKPOINT_1);
    else if (k2 == AZR_PLAYER_ALPHAZERO) group->loadCheckpoint(SETTINGS.CHECKPOINT_2);
    if (k1 == AZR_PLAYER_ALPHAZERO && k2 == AZR_PLAYER_ALPHAZERO) { executePlayAzVsAz(group); return; }
    // one host thread per GPU, the game quota split over the GPUs (the reference shares one Counter)
    const int P = (int)group->size();
    std::vector<azr_game_results> res(P);
    std::vector<std::thread> threads;
    printf("Playing games %d\n", SETTINGS.COMPARE_GAMES);
    for (int i = 0; i < P; i++)
        threads.emplace_back([&, i]() {
            Engine& e = *group->getNN(i)->engine;
            const int share = SETTINGS.COMPARE_GAMES / P + (i < SETTINGS.COMPARE_GAMES % P ? 1 : 0);
            e.check(azr_arena_start(e.h, k1, k2, share, 0, SETTINGS.MIRROR_GAMES, SETTINGS.BASE_SEED + (uint32_t)i * (1u << 24)),
                    "arena_start");
            int fin = 0;
            while (!fin) {
                e.check(azr_arena_run(e.h, 4 * (SETTINGS.MCTS_SIMULATIONS + 2), &fin), "arena_run");
                e.check(azr_arena_results(e.h, &res[i]), "arena_results");
                if (i == 0) {
                    printf("\r%d/%d [Draw/P1,P2]: %d, %d/%d, %d/%d", res[i].count, share, res[i].draw, res[i].win[0],
                           res[i].win_and_started[0], res[i].win[1], res[i].win_and_started[1]);
                    fflush(stdout);
                }
            }
        });
    for (auto& t : threads) t.join();
    azr_game_results gr{};
    for (auto& r : res) {
        gr.count += r.count; gr.draw += r.draw;
        for (int p = 0; p < 2; p++) { gr.win[p] += r.win[p]; gr.win_and_started[p] += r.win_and_started[p]; }
    }
    printf("\nGames: %d\nDraws:%d\nPlayer 1:%d\nPlayer 2:%d\n", gr.count, gr.draw, gr.win[0], gr.win[1]);
}

int main(int argc, char* argv[])
{
    SETTINGS.init(argc, argv);
    printf("===> Starting program with %s\n", SETTINGS.describe().c_str());
    try {
        if (SETTINGS.MODE == "train") executeTrain();
        else if (SETTINGS.MODE == "play") executePlay();
        else printf("Mode '%s' is outside this round's hot-path scope (SURVEY §8f)\n", SETTINGS.MODE.c_str());
    } catch (const std::exception& ex) {
        fprintf(stderr, "fatal: %s\n", ex.what());
        return 1;
    }
    return 0;
}
