// valu_probe: how many wave64 fp32 vector instructions a SIMD of this GPU issues per clock, measured.
// Every thread runs a loop of independent multiply-adds (or of the wide walk's instruction mix: byte conversion,
// multiply-add, three-way maximum); waves per SIMD and chains per thread are swept.  Prints wave-instructions per SIMD and
// clock at the clock rate given on the command line (GHz), and lane-operations per second chip-wide.
// valu_probe <GHz> <lanes> [mix]: one configuration with only <lanes> lanes of every wave at work -- run under the hardware
// counters, it says what the counters report for a known number of active lanes (tools/lane_calibration.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int CHAINS, int MIX>
__global__ void __launch_bounds__(256) k_valu(float *out, int iters, float a, float b, unsigned lanes) {
    float x[CHAINS];
    if ((threadIdx.x & 63u) >= lanes) // the other lanes of the wave sit the loop out
        return;
    unsigned w = threadIdx.x * 2654435761u;
#pragma unroll
    for (int c = 0; c < CHAINS; c++)
        x[c] = (float)(threadIdx.x + c);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (MIX == 0) {
                x[c] = __builtin_fmaf(x[c], a, b);
            } else if (MIX == 1) { // conversion of a byte, multiply-add, three-way maximum: three instructions
                const float p = (float)((w >> (8 * (c & 3))) & 0xFFu);
                const float t = __builtin_fmaf(p, a, b);
                x[c] = fmaxf(fmaxf(x[c], t), b);
            } else if (MIX == 4) { // packed multiply-add: two results per instruction (x[c] holds the sum of the pair)
                typedef float v2f __attribute__((ext_vector_type(2)));
                static_assert(CHAINS % 2 == 0, "pairs");
                if (c % 2 == 0) {
                    v2f p = {x[c], x[c + 1]};
                    p = __builtin_elementwise_fma(p, (v2f){a, a}, (v2f){b, b});
                    x[c] = p.x;
                    x[c + 1] = p.y;
                }
            } else if (MIX == 3) { // a quarter-rate instruction (reciprocal) and a multiply-add
                x[c] = __builtin_fmaf(__builtin_amdgcn_rcpf(x[c]), a, b);
            } else { // mul + add, not fused
                x[c] = x[c] * a;
                asm volatile("" : "+v"(x[c]));
                x[c] = x[c] + b;
            }
        }
        w = w * 1664525u + 1013904223u;
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; c++)
        s += x[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CHAINS, int MIX>
double run(int blocks_per_cu, int cus, int iters, float *d_out, unsigned lanes = 64u) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int blocks = blocks_per_cu * cus;
    hipLaunchKernelGGL((k_valu<CHAINS, MIX>), dim3(blocks), dim3(256), 0, 0, d_out, iters / 10, 1.0001f, 0.5f, lanes);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_valu<CHAINS, MIX>), dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0001f, 0.5f, lanes);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}

int main(int argc, char **argv) {
    const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float *d_out;
    (void)hipMalloc(&d_out, (size_t)cus * 8 * 256 * sizeof(float));
    const int iters = 20000;
    if (argc > 2) { // counter calibration: one configuration, `lanes` lanes of every wave at work, under rocprofv3 --pmc
        const unsigned lanes = (unsigned)atoi(argv[2]);
        const int mix = argc > 3 ? atoi(argv[3]) : 0;
        const double secs = mix == 1   ? run<8, 1>(4, cus, iters, d_out, lanes)
                            : mix == 3 ? run<8, 3>(4, cus, iters, d_out, lanes)
                                       : run<8, 0>(4, cus, iters, d_out, lanes);
        printf("calibration: %u lanes per wave, mix %d, 4 waves/SIMD, %.3f ms\n", lanes, mix, secs * 1e3);
        return 0;
    }
    printf("%d CUs, clock taken as %.2f GHz; wave-instructions per SIMD and clock (1 / this = clocks per instruction)\n", cus, ghz);
    for (int bpc : {1, 2, 4, 8}) {
        struct { const char *name; double secs; double per_iter; } rows[] = {
            {"fma x8 chains", run<8, 0>(bpc, cus, iters, d_out), 8.0},
            {"fma x2 chains", run<2, 0>(bpc, cus, iters, d_out), 2.0},
            {"cvt+fma+max3 x8", run<8, 1>(bpc, cus, iters, d_out), 24.0 + 2.0},
            {"mul,add x8", run<8, 2>(bpc, cus, iters, d_out), 16.0},
            {"pk_fma x4 (8 results)", run<8, 4>(bpc, cus, iters, d_out), 4.0},
        };
        for (auto &r : rows) {
            const double wave_instr = (double)bpc * cus * 4 * iters * r.per_iter; // 4 waves per block
            const double per_simd_clock = wave_instr / (cus * 4.0) / (r.secs * ghz * 1e9);
            printf("  %d waves/SIMD  %-18s %8.3f ms  %.3f wave-instr/SIMD/clk  %.1f T lane-ops/s\n", bpc, r.name, r.secs * 1e3, per_simd_clock,
                   wave_instr * 64 / r.secs / 1e12);
        }
    }
    return 0;
}
