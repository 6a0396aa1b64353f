// ta_probe.hip -- what a compute unit's vector-memory path delivers for the access pattern of a stackless tree walk:
// every lane of a wave reads one small record at an address of its own (no two lanes in one cache line, or a few).
// Prints lane-loads per clock and CU for 4- / 8- / 16-byte records, for tables that fit the L1 (16 KB), the L2s (1 MB)
// and neither (64 MB), for lanes that share lines in groups of 2 / 4 / 8, and the same pattern out of LDS.
// Build: hipcc --offload-arch=gfx950 -O3 -o ta_probe ta_probe.hip      Run: ./ta_probe [clock_ghz]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                        \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

constexpr int kIters = 256; // rounds per lane
constexpr int kFlight = 4;  // independent loads per round

// T = uint (4 B), uint2 (8 B), uint4 (16 B).  `share`: lanes l, l^1, .. of a group of `share` read the same 128-byte line
// (different records of it).  mask: table size in records - 1.
template <class T>
__global__ void __launch_bounds__(256) k_gather(const T *__restrict__ table, unsigned mask, unsigned share, unsigned *out) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned group = tid / share; // lanes of a group draw the same line
    const unsigned per_line = 128u / sizeof(T);
    const unsigned line_sets = ((mask + 1u) / per_line) / 64u; // the table as sets of 64 lines: a wave's groups read different lines of one set
    unsigned h = group * 2654435761u + 12345u, hl = tid * 40503u + 977u;
    unsigned acc = 0;
    for (int i = 0; i < kIters; i++) {
        T v[kFlight];
#pragma unroll
        for (int k = 0; k < kFlight; k++) {
            h = h * 1664525u + 1013904223u;
            hl = hl * 22695477u + 1u;
            const unsigned line = ((h >> 9) & (line_sets - 1u)) * 64u + lane / share;
            const unsigned rec = line * per_line + (share > 1u ? (lane % share) % per_line : (hl >> 20) & (per_line - 1u));
            v[k] = table[rec & mask];
        }
#pragma unroll
        for (int k = 0; k < kFlight; k++) {
            if constexpr (sizeof(T) == 4)
                acc ^= v[k];
            else if constexpr (sizeof(T) == 8)
                acc ^= v[k].x ^ v[k].y;
            else
                acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        }
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}

// the same pattern out of LDS: 16-byte records at random places of a 32 KB array per block
__global__ void __launch_bounds__(256) k_lds(unsigned *out) {
    __shared__ uint4 s[2048];
    for (unsigned i = threadIdx.x; i < 2048u; i += 256u)
        s[i] = make_uint4(i, i * 3u, i * 5u, i * 7u);
    __syncthreads();
    unsigned h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    unsigned acc = 0;
    for (int i = 0; i < kIters; i++) {
        uint4 v[kFlight];
#pragma unroll
        for (int k = 0; k < kFlight; k++) {
            h = h * 1664525u + 1013904223u;
            v[k] = s[(h >> 7) & 2047u];
        }
#pragma unroll
        for (int k = 0; k < kFlight; k++)
            acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}

int main(int argc, char **argv) {
    const double ghz = argc > 1 ? std::atof(argv[1]) : 2.4;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int waves_per_simd = argc > 2 ? std::atoi(argv[2]) : 4;
    const int blocks = cus * waves_per_simd; // blocks of four waves: one wave per SIMD each
    const size_t bytes = 64u << 20;
    void *table = nullptr;
    unsigned *out = nullptr;
    CHECK(hipMalloc(&table, bytes));
    CHECK(hipMalloc((void **)&out, 256));
    {
        std::vector<unsigned> host(bytes / 4);
        for (size_t i = 0; i < host.size(); i++)
            host[i] = (unsigned)(i * 2654435761u);
        CHECK(hipMemcpy(table, host.data(), bytes, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::printf("device %s, %d CUs, %d blocks of 256 threads (%d waves per SIMD), %.2f GHz assumed\n", prop.name, cus, blocks,
                waves_per_simd, ghz);
    const size_t sizes[3] = {16u << 10, 1u << 20, 64u << 20};
    const char *size_name[3] = {"16 KB (L1)", "1 MB (L2)", "64 MB (Infinity Cache / HBM)"};
    const unsigned shares[4] = {1u, 2u, 4u, 8u};
    for (int rec = 4; rec <= 16; rec *= 2)
        for (int sz = 0; sz < 3; sz++)
            for (int sh = 0; sh < 4; sh++) {
                if (sz == 2 && sh > 0)
                    continue;
                const unsigned mask = (unsigned)(sizes[sz] / (size_t)rec) - 1u;
                float ms = 0.0f;
                for (int rep = 0; rep < 3; rep++) {
                    CHECK(hipEventRecord(e0, nullptr));
                    if (rec == 4)
                        hipLaunchKernelGGL(k_gather<unsigned>, dim3(blocks), dim3(256), 0, nullptr, (const unsigned *)table, mask, shares[sh], out);
                    else if (rec == 8)
                        hipLaunchKernelGGL(k_gather<uint2>, dim3(blocks), dim3(256), 0, nullptr, (const uint2 *)table, mask, shares[sh], out);
                    else
                        hipLaunchKernelGGL(k_gather<uint4>, dim3(blocks), dim3(256), 0, nullptr, (const uint4 *)table, mask, shares[sh], out);
                    CHECK(hipEventRecord(e1, nullptr));
                    CHECK(hipEventSynchronize(e1));
                    CHECK(hipGetLastError());
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                }
                const double lane_loads = (double)blocks * 256.0 * kIters * kFlight;
                std::printf("record %2d B, table %-28s lanes per line %u: %7.3f ms, %.3f lane-loads per clock and CU\n", rec, size_name[sz],
                            shares[sh], ms, lane_loads / (ms * 1e-3) / (ghz * 1e9) / cus);
            }
    float ms = 0.0f;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0, nullptr));
        hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(256), 0, nullptr, out);
        CHECK(hipEventRecord(e1, nullptr));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::printf("LDS, record 16 B, 32 KB per block: %7.3f ms, %.3f lane-loads per clock and CU\n", ms,
                (double)blocks * 256.0 * kIters * kFlight / (ms * 1e-3) / (ghz * 1e9) / cus);
    (void)hipFree(table);
    (void)hipFree(out);
    return 0;
}
