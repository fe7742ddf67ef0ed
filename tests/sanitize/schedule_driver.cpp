// Sanitizer driver for the library's host-only code (csrc/bg_schedule.h), built by tests/test_sanitizers_cpu.py with
// g++ -fsanitize=address,undefined.  Deals random rounds to 1 .. n slots and checks what the streamed replay relies on:
// every game with a positive length sits in exactly one slot's queue, the offsets are monotone and end at the game count,
// and the reported step count is the largest slot load.
#include <cstdio>
#include <cstdlib>
#include "bg_schedule.h"

int main()
{
    uint64_t x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    long long checked = 0;
    for (int round = 0; round < 300; ++round) {
        const int64_t n_lanes = 1 + (int64_t)(rnd() % 700);
        const int64_t n_slots = 1 + (int64_t)(rnd() % 96);
        std::vector<int32_t> len((size_t)n_lanes), queue((size_t)n_lanes + 1, -7), off((size_t)n_slots + 1, -7);
        for (auto &l : len) l = (rnd() % 5 == 0) ? 0 : (int32_t)(1 + rnd() % 300);
        int64_t n_games = -1, n_steps = -1;
        if (bg::td_stream_schedule(len.data(), n_lanes, n_slots, queue.data(), off.data(), &n_games, &n_steps)) return 2;
        std::vector<int> seen((size_t)n_lanes, 0);
        int64_t want_games = 0, max_load = 0;
        for (auto l : len) want_games += l > 0;
        if (n_games != want_games || off[0] != 0 || off[(size_t)n_slots] != n_games) return 3;
        for (int64_t s = 0; s < n_slots; ++s) {
            if (off[(size_t)s] > off[(size_t)s + 1]) return 4;
            int64_t load = 0;
            for (int32_t q = off[(size_t)s]; q < off[(size_t)s + 1]; ++q) {
                const int32_t g = queue[(size_t)q];
                if (g < 0 || g >= n_lanes || len[(size_t)g] <= 0 || seen[(size_t)g]++) return 5;
                load += len[(size_t)g];
            }
            if (load > max_load) max_load = load;
        }
        if (max_load != n_steps || queue[(size_t)n_lanes] != -7) return 6;
        checked += n_games;
    }
    // invalid arguments are refused, nothing is written
    int32_t q = 1, o = 1;
    if (!bg::td_stream_schedule(nullptr, 1, 1, &q, &o, nullptr, nullptr) || !bg::td_stream_schedule(&q, 1, 0, &q, &o, nullptr, nullptr)) return 7;
    std::printf("OK %lld games dealt\n", checked);
    return 0;
}
