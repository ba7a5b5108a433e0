// TEST INFRASTRUCTURE (tests/test_host_threads.py): azrhost::forEachGpu — the one-host-thread-per-GPU fan-out of the C++ host
// (GameGroup::playGames, AlphaZeroTrainer::generateTrainData, `-m play`) — must carry an exception thrown inside a thread body out to
// the caller after every thread has been joined, instead of ending the process in std::terminate.  No GPU is touched.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>

#include "../../alphazero-risk_amd/host/azr_host.hpp"

int main()
{
    std::atomic<int> finished{0};
    // 1. every body runs, on its own thread, and the call returns when all are done
    azrhost::forEachGpu(4, "probe", [&](int) { std::this_thread::sleep_for(std::chrono::milliseconds(20)); finished++; });
    if (finished != 4) { printf("FAIL: %d of 4 bodies ran\n", finished.load()); return 2; }
    // 2. a logic_error (Engine::check's mapping of AZR_E_LOGIC) in thread 2 and an unknown exception in thread 3: the first failing
    //    GPU is reported, the slow healthy threads have been joined before the throw reaches the caller
    finished = 0;
    try {
        azrhost::forEachGpu(4, "compare games", [&](int i) {
            if (i == 2) throw std::logic_error("arena_run: engine said no");
            if (i == 3) throw 42;
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
            finished++;
        });
        printf("FAIL: no exception reached the caller\n");
        return 3;
    } catch (const std::runtime_error& ex) {
        printf("caught: %s\n", ex.what());
        if (finished != 2) { printf("FAIL: threads were not joined before the throw (%d)\n", finished.load()); return 4; }
    }
    printf("OK\n");
    return 0;
}
