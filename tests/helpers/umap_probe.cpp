// TEST HELPER: prints the iteration order of the host libstdc++'s std::unordered_map after inserting
// the set bits of each mask (stdin, one hex mask per line) in ascending order with operator[] — the
// way StateSimulations' ctor fills moveValues (reference alphazero_mcts.cpp:32-41).  Used to validate
// oracle/azr_oracle.c:orc_umap_order and the device emulation against the real library.
#include <cstdint>
#include <cstdio>
#include <unordered_map>

enum class Key : uint8_t { A = 0 };
struct Val { float q, p; uint32_t n; uint8_t a; };

int main()
{
    unsigned long long mask;
    while (std::scanf("%llx", &mask) == 1) {
        std::unordered_map<Key, Val> m;
        for (int i = 0; i < 43; i++)
            if (mask & (1ULL << i)) m[static_cast<Key>(i)] = Val{0, 0, 0, 0};
        bool first = true;
        for (auto e : m) { std::printf(first ? "%d" : " %d", (int)e.first); first = false; }
        std::printf("\n");
    }
}
