// fastdiv.hpp -- exact unsigned 32-bit division by a divisor fixed at launch time, as one multiply-high, one subtract,
// one add and two shifts (Granlund & Montgomery's round-up method with a 33-bit multiplier kept as 32 bits + an add).
// A 32-bit divide by a run-time value costs the GPU some thirty instructions; the camera-ray generator needs two per ray.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define RT_FASTDIV_HD __host__ __device__ __forceinline__
#else
#define RT_FASTDIV_HD inline
#endif

struct FastDiv {
    uint32_t m, sh1, sh2;
};

// d >= 1
inline FastDiv make_fastdiv(uint32_t d) {
    uint32_t L = 0; // ceil(log2 d)
    while (L < 32 && (1ull << L) < d)
        L++;
    FastDiv f;
    f.m = (uint32_t)(((1ull << 32) * ((1ull << L) - d)) / d + 1ull);
    f.sh1 = L < 1 ? L : 1;
    f.sh2 = L < 1 ? 0 : L - 1;
    return f;
}

RT_FASTDIV_HD uint32_t fastdiv_mulhi(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

RT_FASTDIV_HD uint32_t fastdiv(uint32_t x, const FastDiv f) {
    const uint32_t q = fastdiv_mulhi(f.m, x);
    const uint32_t t = q + ((x - q) >> f.sh1);
    return t >> f.sh2;
}
