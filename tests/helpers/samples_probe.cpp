// TEST INFRASTRUCTURE: drives the host side's NNTrainDataStorage (alphazero-risk_amd/host/azr_host.cpp) from the command line so
// that tests/test_samples_io.py can compare it with what the reference's own storage did (tests/golden/samples_io.npz).
//   samples_probe save <records.bin> <n> <out path>      appendPacked + saveTrainingSamples
//   samples_probe load <path> <out records.bin>          loadTrainingSamples + packed(); prints the count
//   samples_probe trim <n> <old> <smin> <smax>           trimOldExamples on n records marked z = i; prints n' first old'
//   samples_probe extend <a.bin> <na> <b.bin> <nb> <out> extend + updateOldGamesIndex; prints n old
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "azr_host.hpp"

using namespace azrhost;

static std::vector<uint8_t> slurp(const char* p)
{
    std::ifstream in(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    const std::string cmd = argv[1];
    if (cmd == "save" && argc == 5) {
        std::vector<uint8_t> r = slurp(argv[2]);
        NNTrainDataStorage st;
        st.appendPacked(r.data(), (size_t)atoi(argv[3]));
        st.saveTrainingSamples(argv[4]);
        return 0;
    }
    if (cmd == "load" && argc == 4) {
        NNTrainDataStorage st;
        st.loadTrainingSamples(argv[2]);
        std::vector<uint8_t> p = st.packed();
        std::ofstream(argv[3], std::ios::binary).write((const char*)p.data(), (std::streamsize)p.size());
        printf("count %zu\n", st.data.size());
        return 0;
    }
    if (cmd == "trim" && argc == 6) {
        SETTINGS.SAMPLES_STORAGE_MIN = atoi(argv[4]);
        SETTINGS.SAMPLES_STORAGE_MAX = atoi(argv[5]);
        NNTrainDataStorage st;
        const int n = atoi(argv[2]);
        std::vector<uint8_t> rec((size_t)n * AZR_RECORD_BYTES, 0);
        for (int i = 0; i < n; i++) { float z = (float)i; memcpy(&rec[(size_t)i * AZR_RECORD_BYTES + 89], &z, 4); }
        st.appendPacked(rec.data(), (size_t)n);
        st.oldGameIndex = (size_t)atol(argv[3]);
        st.trimOldExamples();
        printf("result %zu %d %zu\n", st.data.size(), st.data.empty() ? -1 : (int)st.data.front().out.value, st.oldGameIndex);
        return 0;
    }
    if (cmd == "extend" && argc == 7) {
        std::vector<uint8_t> a = slurp(argv[2]), b = slurp(argv[4]);
        NNTrainDataStorage sa, sb;
        sa.appendPacked(a.data(), (size_t)atoi(argv[3]));
        sb.appendPacked(b.data(), (size_t)atoi(argv[5]));
        sa.extend(sb);
        sa.updateOldGamesIndex();
        std::vector<uint8_t> p = sa.packed();
        std::ofstream(argv[6], std::ios::binary).write((const char*)p.data(), (std::streamsize)p.size());
        printf("result %zu %zu\n", sa.data.size(), sa.oldGameIndex);
        return 0;
    }
    if (cmd == "loglines" && argc == 14) {   // it | d w0 s0 w1 s1 (x2: second result for the benchmark line) | lp lv: the three log lines
        auto gr = [&](int o) { GameResults g; g.draw = atoi(argv[o]); g.players[0].win = atoi(argv[o + 1]); g.players[0].winAndStartedGame = atoi(argv[o + 2]);
                               g.players[1].win = atoi(argv[o + 3]); g.players[1].winAndStartedGame = atoi(argv[o + 4]); return g; };
        GameResults a = gr(3), b = gr(8);
        std::cout << atoi(argv[2]) << ',' << a << std::endl;                 // log/azr-improvement-log.txt (updateIfImprovement)
        std::cout << atoi(argv[2]) << ',' << a << ", " << b << std::endl;    // log/azr-benchmark-log.txt (benchmark)
        std::cout << (float)atof(argv[13]) << ", " << (float)atof(argv[13]) / 3 << ", " << std::endl;   // log/azr-nn-training-log.txt
        return 0;
    }
    return 2;
}
