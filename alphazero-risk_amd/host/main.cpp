// AlphaZero_Risk_hip — CLI of the MI355X-native build; flags and modes as src/alphazero_risk.cpp:160-199 /
// src/settings.h:91-137 (`-m learn` = `-m train`).
#include <cstdio>
#include <cstring>

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
// Every pairing runs through azr_arena_*; az vs az = two AlphaZeroPlayers with their own trees and checkpoints
// (AZR_PLAYER_ALPHAZERO vs AZR_PLAYER_ALPHAZERO_B).
static int playerKind(const std::string& p)
{
    if (p == "az") return AZR_PLAYER_ALPHAZERO;
    if (p == "sp") return AZR_PLAYER_SCRIPT;
    if (p == "rp") return AZR_PLAYER_RANDOM;
    throw std::invalid_argument("unknown player '" + p + "' (az/sp/rp)");
}

static void executePlayAzVsAz(std::shared_ptr<AlphaZeroNNGroup> group1, std::shared_ptr<AlphaZeroNNGroup> group2)
{
    AlphaZeroPlayerGroup p1(group1), p2(group2);
    GameResults gr = GameGroup::playGames(p1, p2, SETTINGS.COMPARE_GAMES);
    printf("Games: %d\nDraws:%d\nPlayer 1:%d\nPlayer 2:%d\n", gr.count, gr.draw, gr.players[0].win, gr.players[1].win);
}

static void executePlay()
{
    const int k1 = playerKind(SETTINGS.PLAYER_1), k2 = playerKind(SETTINGS.PLAYER_2);
    auto cluster = std::make_shared<AlphaZeroCluster>();
    cluster->initGpus(SETTINGS.NUMBER_OF_GPUS);
    auto group = cluster->initPlayerGroup("az1", SETTINGS.GRAPH_DEF_PB_1);
    if (k1 == AZR_PLAYER_ALPHAZERO) group->loadCheckpoint(SETTINGS.CHECKPOINT_1);
    else if (k2 == AZR_PLAYER_ALPHAZERO) group->loadCheckpoint(SETTINGS.CHECKPOINT_2);
    if (k1 == AZR_PLAYER_ALPHAZERO && k2 == AZR_PLAYER_ALPHAZERO) {  // two AlphaZeroPlayers, each with its checkpoint
        auto group2 = cluster->initPlayerGroup("az2", SETTINGS.GRAPH_DEF_PB_2);
        group2->loadCheckpoint(SETTINGS.CHECKPOINT_2);
        executePlayAzVsAz(group, group2);
        return;
    }
    // one host thread per GPU, the game quota split over the GPUs (the reference shares one Counter)
    const int P = (int)group->size();
    std::vector<azr_game_results> res(P);
    printf("Playing games %d\n", SETTINGS.COMPARE_GAMES);
    const int pairs = SETTINGS.COMPARE_GAMES / 2;   // Counter::hasNext(2): whole pairs only (game.cpp:14-26)
    forEachGpu(P, "play", [&](int i) {
        Engine& e = *group->getNN(i)->engine;
        memset(&res[i], 0, sizeof res[i]);
        const int share = 2 * (pairs / P + (i < pairs % P ? 1 : 0));
        if (share == 0) return;
        e.check(azr_arena_start(e.h, k1, k2, share, 0, SETTINGS.arenaMirrorMode(), SETTINGS.BASE_SEED + (uint32_t)i * (1u << 24)),
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
