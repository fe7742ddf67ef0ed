// bg_schedule.h -- host-only part of the streamed TD(lambda) replay: dealing a round's games to the replay slots.
// Plain C++ (no HIP): bgamd.hip wraps it as bgamd_td_stream_schedule, tests/test_sanitizers_cpu.py compiles it under
// AddressSanitizer + UBSan.  The reference applies a round's games one after another
// (pysrc/TD(lambda) model/train.py:536-547); k slots replay them k at a time.
#pragma once
#include <algorithm>
#include <cstdint>
#include <functional>
#include <queue>
#include <utility>
#include <vector>

namespace bg {

// Longest game first, each to the slot with the fewest turns so far (ties: the lower slot); every slot then plays its share in
// a fixed pseudo-random order (a multiplicative hash of the lane).  h_queue[n_lanes], h_queue_offsets[n_slots + 1].
// Returns 0, or 1 for invalid arguments.
inline int td_stream_schedule(const int32_t *h_length, int64_t n_lanes, int64_t n_slots, int32_t *h_queue, int32_t *h_queue_offsets,
                              int64_t *h_n_games, int64_t *h_n_steps)
{
    if (!h_length || !h_queue || !h_queue_offsets || n_lanes < 0 || n_lanes > 0x7FFFFFFF || n_slots < 1) return 1;
    std::vector<int32_t> games;
    for (int64_t l = 0; l < n_lanes; ++l) if (h_length[l] > 0) games.push_back((int32_t)l);
    std::stable_sort(games.begin(), games.end(), [&](int32_t a, int32_t b) { return h_length[a] > h_length[b]; });
    const int64_t n_games = (int64_t)games.size();
    const int64_t k = n_games < n_slots ? n_games : n_slots;
    typedef std::pair<int64_t, int32_t> Load;                    // (turns so far, slot)
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (int64_t i = 0; i < k; ++i) heap.push(Load(0, (int32_t)i));
    std::vector<std::vector<int32_t>> share((size_t)k);
    int64_t n_steps = 0;
    for (int32_t g : games) {
        Load top = heap.top(); heap.pop();
        share[(size_t)top.second].push_back(g);
        top.first += h_length[g];
        if (top.first > n_steps) n_steps = top.first;
        heap.push(top);
    }
    auto key = [](int32_t lane) { return (uint32_t)((uint64_t)lane * 2654435761ull + 0x9E3779B9ull); };
    int64_t pos = 0;
    for (int64_t i = 0; i < k; ++i) {
        std::vector<int32_t> &sh = share[(size_t)i];
        std::sort(sh.begin(), sh.end(), [&](int32_t a, int32_t b) { return key(a) < key(b); });
        h_queue_offsets[i] = (int32_t)pos;
        for (int32_t g : sh) h_queue[pos++] = g;
    }
    for (int64_t i = k; i <= n_slots; ++i) h_queue_offsets[i] = (int32_t)pos;
    if (h_n_games) *h_n_games = n_games;
    if (h_n_steps) *h_n_steps = n_steps;
    return 0;
}

}  // namespace bg
