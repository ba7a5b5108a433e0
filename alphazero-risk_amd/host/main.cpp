// AlphaZero_Risk_hip — CLI of the MI355X-native build; flags and modes as src/alphazero_risk.cpp:160-199 /
// src/settings.h:91-137 (`-m learn` = `-m train`).
#include <cstdio>

#include "azr_host.hpp"

using namespace azrhost;

static void executeTrain()
{
    auto cluster = std::make_shared<AlphaZeroCluster>();
    cluster->initGpus(SETTINGS.NUMBER_OF_GPUS);
    auto generateGroup = cluster->initPlayerGroup("az_generate", SETTINGS.GRAPH_DEF_PB_1);
    generateGroup->loadCheckpoint(SETTINGS.DEFAULT_LATEST_CHECKPOINT);
    AlphaZeroTrainer trainer;
    trainer.train(generateGroup, generateGroup);
}

// `-m play` with both sides "az": AlphaZeroPlayerGroup vs itself through the batched Player seam.
// ScriptPlayer / RandomPlayer opponents are SURVEY §8(f-3) "next" rows.
static void executePlay()
{
    if (SETTINGS.PLAYER_1 != "az" || SETTINGS.PLAYER_2 != "az") {
        printf("This round builds the AlphaZero player only (--p1=az --p2=az); ScriptPlayer/RandomPlayer are SURVEY §8f-3.\n");
        return;
    }
    auto cluster = std::make_shared<AlphaZeroCluster>();
    cluster->initGpus(SETTINGS.NUMBER_OF_GPUS);
    auto group = cluster->initPlayerGroup("az1", SETTINGS.GRAPH_DEF_PB_1);
    group->loadCheckpoint(SETTINGS.CHECKPOINT_1);
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
