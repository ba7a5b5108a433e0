// prints std::shuffle's permutation of 0..n-1 under a minstd_rand0 seeded with a raw state, then the engine state —
// the libstdc++ algorithm AlphaZeroNN::train uses (alphazero_nn.cpp:372); reference for tests/test_gpu_train.py
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <random>
#include <vector>
int main(int argc, char** argv)
{
    const int n = atoi(argv[1]);
    const unsigned state = (unsigned)strtoul(argv[2], nullptr, 10);
    const int rounds = argc > 3 ? atoi(argv[3]) : 1;
    std::minstd_rand0 eng(state);
    std::vector<int> v(n);
    for (int i = 0; i < n; i++) v[i] = i;
    for (int r = 0; r < rounds; r++) {
        std::shuffle(v.begin(), v.end(), eng);
        for (int i = 0; i < n; i++) printf("%d ", v[i]);
        printf("\n");
    }
    std::cout << eng << "\n";
    return 0;
}
