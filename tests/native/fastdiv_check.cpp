// Exhaustive-at-the-edges check of fastdiv.hpp against the machine's own divide.
#include "../../raytracing_amd/csrc/fastdiv.hpp"

#include <cstdio>
#include <random>

int main() {
    std::mt19937 rng(3);
    const uint32_t divisors[] = {1u, 2u, 3u, 5u, 7u, 64u, 240u, 480u, 1024u, 1472u, 4096u, 4097u, 65535u, 65536u, 1000003u, 0x7FFFFFFFu, 0x80000000u, 0x80000001u, 0xFFFFFFFEu, 0xFFFFFFFFu};
    unsigned long long checked = 0;
    auto check = [&](uint32_t x, uint32_t d, const FastDiv &f) {
        if (fastdiv(x, f) != x / d) {
            std::printf("MISMATCH %u / %u: %u != %u\n", x, d, fastdiv(x, f), x / d);
            return false;
        }
        checked++;
        return true;
    };
    auto run = [&](uint32_t d) {
        const FastDiv f = make_fastdiv(d);
        for (uint32_t x = 0; x < 5000u; x++)
            if (!check(x, d, f) || !check(0xFFFFFFFFu - x, d, f))
                return false;
        for (int k = -3; k <= 3; k++) // around the multiples of d
            for (uint32_t q = 0; q < 500u; q++) {
                const unsigned long long v = (unsigned long long)q * d + (long long)k;
                if (v <= 0xFFFFFFFFull && !check((uint32_t)v, d, f))
                    return false;
            }
        for (int i = 0; i < 20000; i++)
            if (!check(rng(), d, f))
                return false;
        return true;
    };
    for (uint32_t d : divisors)
        if (!run(d))
            return 1;
    for (uint32_t d = 1; d <= 3000u; d++)
        if (!run(d))
            return 1;
    for (int i = 0; i < 100; i++) {
        const uint32_t a = rng() | 1u, b = (rng() >> (rng() & 31u)) | 1u;
        if (!run(a) || !run(b))
            return 1;
    }
    std::printf("ok %llu\n", checked);
    return 0;
}
