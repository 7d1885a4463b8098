// Micro-benchmark: what does the vector L1 (TA/TCP) charge for the BVH node gather?
// Each group of G lanes fetches one NODE-byte record per step from a pseudo-random slot of a table (the next slot
// depends on the loaded data, like a traversal), 16 B per lane per instruction.  Prints node fetches per second
// for table sizes that sit in L1, L2, the Infinity Cache and beyond.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_gather tools/ubench_gather.hip && ./ubench_gather
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__device__ inline uint32_t mixu(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

template <int G, int NODE>
__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ table, uint32_t mask, int iters,
                                                uint32_t* __restrict__ out) {
    constexpr int PER_LANE = NODE / G / 16;  // dwordx4 loads per lane per step
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sub = tid % G;
    uint32_t slot = mixu(tid / G);
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const uint4* p = table + (size_t)(slot & mask) * (NODE / 16) + sub * PER_LANE;
        uint4 v[PER_LANE];
#pragma unroll
        for (int j = 0; j < PER_LANE; j++) v[j] = p[j];
        uint32_t x = 0;
#pragma unroll
        for (int j = 0; j < PER_LANE; j++) x ^= v[j].x + v[j].y + v[j].z + v[j].w;
        // combine over the group so every lane of a group follows the same slot
        for (int d = 1; d < G; d <<= 1) x ^= __shfl_xor(x, d);
        acc += x;
        slot = mixu(slot + x + it);
    }
    if (acc == 0x12345678u) out[tid] = acc;
}

template <int G, int NODE>
int run(const uint4* d_table, size_t table_bytes, uint32_t* d_out, int waves_per_simd) {
    const uint32_t n_nodes = (uint32_t)(table_bytes / NODE);
    uint32_t mask = 1;
    while (mask * 2 <= n_nodes) mask *= 2;
    mask -= 1;
    const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
    const int iters = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    k_gather<G, NODE><<<blocks, 256>>>(d_table, mask, 200, d_out);
    CK(hipEventRecord(e0));
    k_gather<G, NODE><<<blocks, 256>>>(d_table, mask, iters, d_out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double fetches = (double)blocks * 256 / G * iters;
    const double wave_steps = (double)blocks * 4 * iters;
    printf("G=%d node=%3dB table=%8.2f MB waves/simd=%d : %7.2f Gfetch/s  %6.2f TB/s  %6.1f ns per wave-step per CU\n", G,
           NODE, (mask + 1.0) * NODE / 1e6, waves_per_simd, fetches / ms * 1e-6, fetches * NODE / ms * 1e-9,
           ms * 1e6 / (wave_steps / 256));
    fflush(stdout);
    return 0;
}

// `ubench_gather calib`: the FETCH_SIZE calibration VERDICT r1 asked for.  Dependent random gathers of the trace
// kernel's shape (a pair reads a 64-byte record with 2 x dwordx4 per lane; and the 128-byte variant, 4 per lane) from
// a 1 GiB table — 32x the L2s and 4x the Infinity Cache, so all but ~3 % of the fetches leave the L2.  The byte count
// is known (printed); run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and divide (tools/calibrate_fetch.sh).
template <int G, int NODE> int calib(const uint4* d_table, uint32_t* d_out, const char* name) {
    const uint32_t n_nodes = (uint32_t)(((size_t)1 << 30) / NODE);
    const uint32_t mask = n_nodes - 1;  // a power of two
    const int blocks = 256 * 6, iters = 4000;
    k_gather<G, NODE><<<blocks, 256>>>(d_table, mask, iters, d_out);
    CK(hipDeviceSynchronize());
    const double fetches = (double)blocks * 256 / G * iters;
    printf("CALIB %s fetches %.0f record_bytes %d requested_bytes %.0f lines128_bytes %.0f\n", name, fetches, NODE, fetches * NODE,
           fetches * 128.0);
    return 0;
}

int main(int argc, char** argv) {
    const size_t max_bytes = (size_t)1 << 30;
    uint4* d_table;
    uint32_t* d_out;
    CK(hipMalloc(&d_table, max_bytes));
    CK(hipMalloc(&d_out, (size_t)256 * 8 * 256 * 4));
    std::vector<uint32_t> h(max_bytes / 4);
    uint32_t s = 12345;
    for (auto& w : h) {
        s = s * 1664525u + 1013904223u;
        w = s >> 8;
    }
    CK(hipMemcpy(d_table, h.data(), max_bytes, hipMemcpyHostToDevice));
    if (argc > 1 && std::string(argv[1]) == "calib") {
        if (calib<2, 64>(d_table, d_out, "pair_64B")) return 1;
        if (calib<2, 128>(d_table, d_out, "pair_128B")) return 1;
        return 0;
    }
    const size_t sizes[] = {16u << 10, 2u << 20, 32u << 20, 128u << 20, 1u << 30};
    for (size_t sz : sizes) {
        for (int w : {6, 8}) {
            if (run<2, 128>(d_table, sz, d_out, w)) return 1;
            if (run<2, 64>(d_table, sz, d_out, w)) return 1;
            if (run<4, 128>(d_table, sz, d_out, w)) return 1;
            if (run<4, 64>(d_table, sz, d_out, w)) return 1;
            if (run<1, 64>(d_table, sz, d_out, w)) return 1;
            if (run<8, 128>(d_table, sz, d_out, w)) return 1;
            if (run<1, 32>(d_table, sz, d_out, w)) return 1;
            if (run<2, 32>(d_table, sz, d_out, w)) return 1;
        }
        printf("\n");
    }
    return 0;
}
