// Checks that the fast-tree builder emits the same nodes whether one thread builds the whole tree or the
// top splits are shared between threads.  Prints "ok <nodes>" per case; exits non-zero on a difference.
#include "../../raytracing_amd/csrc/fast_bvh.hpp"

#include <cstdio>
#include <cstring>
#include <random>

static std::vector<fastbvh::Item> soup(uint32_t n, uint32_t seed, bool clustered) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<fastbvh::Item> items(n);
    for (uint32_t i = 0; i < n; i++) {
        // clustered: a few dense clumps far apart, so that the top splits are very lopsided
        const float scale = clustered ? (i % 7 == 0 ? 1000.f : 1.f) : 10.f;
        const float off = clustered ? (float)(i % 3) * 5000.f : 0.f;
        for (int k = 0; k < 3; k++) {
            const float c = u(rng) * scale + off, h = (u(rng) + 1.f) * 0.01f * scale;
            items[i].bmin[k] = c - h;
            items[i].bmax[k] = c + h;
        }
        items[i].id = i;
    }
    if (clustered) // a run of identical boxes: every centroid coincides and the split falls back to the middle
        for (uint32_t i = 0; i < n / 3; i++)
            items[i] = items[0], items[i].id = i;
    return items;
}

int main(int argc, char **) {
    // any argument: the small cases only (for a run under the thread sanitizer)
    for (uint32_t n : {1u, 2u, 40000u, 300001u}) {
        if (argc > 1 && n > 100000u)
            continue;
        for (int clustered = 0; clustered < 2; clustered++) {
            const auto items = soup(n, 17u + n, clustered != 0);
            std::vector<fastbvh::Node> serial, shared;
            fastbvh::detail::Builder a(items), b(items);
            a.idx.resize(n);
            b.idx.resize(n);
            for (uint32_t i = 0; i < n; i++)
                a.idx[i] = b.idx[i] = i;
            a.build(0, n, serial);
            b.build_shared(0, n, shared, fastbvh::detail::kParallelDepth);
            std::vector<fastbvh::Node> entry;
            fastbvh::build(items, entry);
            if (serial.size() != 2u * n - 1u || serial.size() != shared.size() || serial.size() != entry.size() ||
                std::memcmp(serial.data(), shared.data(), serial.size() * sizeof(fastbvh::Node)) != 0 ||
                std::memcmp(serial.data(), entry.data(), serial.size() * sizeof(fastbvh::Node)) != 0) {
                std::printf("MISMATCH n=%u clustered=%d\n", n, clustered);
                return 1;
            }
            std::printf("ok %zu\n", serial.size());
        }
    }
    return 0;
}
