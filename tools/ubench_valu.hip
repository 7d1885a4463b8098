// ubench_valu.hip — DEVELOPMENT TOOL: issue cost of the vector instructions the trace kernel's steps are made of, on
// gfx950, at 1 / 2 / 4 waves per SIMD.  Each wave runs REPS x 16 x 4 independent copies of one instruction between
// two s_memtime stamps; reported: shader cycles per wave-instruction seen by one wave, and per SIMD (divided by the
// waves sharing it) — the second is what the instruction costs when enough waves are there to fill the SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_valu.hip -o tools/ubench_valu && tools/ubench_valu
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

#define REPS 64
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)

typedef float f2 __attribute__((ext_vector_type(2)));

#define PROLOGUE                                                                                \
    float a0 = seed + threadIdx.x, a1 = a0 * 1.5f, a2 = a0 + 3.f, a3 = a0 * 0.25f;              \
    float d0 = 0, d1 = 0, d2 = 0, d3 = 0;                                                       \
    unsigned u0 = threadIdx.x * 2654435761u, u1 = u0 ^ 0x5bd1e995u, u2 = 0x0005040Cu;           \
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a3}, q0 = {0, 0}, q1 = {0, 0}, q2 = {0, 0}, q3 = {0, 0}; \
    double f0 = a0, f1 = a1, f2_ = a2, g0 = 0, g1 = 0, g2 = 0, g3 = 0;                          \
    (void)u2;                                                                                   \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define EPILOGUE                                                                                                   \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                          \
    if (d0 + d1 + d2 + d3 + q0.x + q1.y + q2.x + q3.y + (float)(g0 + g1 + g2 + g3) == 12345.678f) out[1] = 1;      \
    if ((threadIdx.x & 63) == 0) out[2 + blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;

// 32-bit: four independent destinations d0..d3 (%0..%3), sources a0..a3 (%4..%7), u0..u2 (%8..%10)
#define K32(name, body)                                                                                    \
    __global__ void __launch_bounds__(1024) name(unsigned long long *out, float seed) {                     \
        PROLOGUE                                                                                           \
        for (int r = 0; r < REPS; r++)                                                                     \
            asm volatile(R16(body) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)                                \
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(u0), "v"(u1), "v"(u2) : "vcc", "s20", "s21", "s22", "s23"); \
        EPILOGUE                                                                                           \
    }
// packed: destinations q0..q3 (%0..%3), sources p0..p2 (%4..%6)
#define KPK(name, body)                                                                                    \
    __global__ void __launch_bounds__(1024) name(unsigned long long *out, float seed) {                     \
        PROLOGUE                                                                                           \
        for (int r = 0; r < REPS; r++)                                                                     \
            asm volatile(R16(body) : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(p0), "v"(p1), "v"(p2));  \
        EPILOGUE                                                                                           \
    }
// f64: destinations g0..g3 (%0..%3), sources f0..f2 (%4..%6)
#define K64(name, body)                                                                                    \
    __global__ void __launch_bounds__(1024) name(unsigned long long *out, float seed) {                     \
        PROLOGUE                                                                                           \
        for (int r = 0; r < REPS; r++)                                                                     \
            asm volatile(R16(body) : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3) : "v"(f0), "v"(f1), "v"(f2_)); \
        EPILOGUE                                                                                           \
    }
// f64 -> f32 conversions: destinations d0..d3, sources f0..f2 (%4..%6)
#define K6432(name, body)                                                                                  \
    __global__ void __launch_bounds__(1024) name(unsigned long long *out, float seed) {                     \
        PROLOGUE                                                                                           \
        for (int r = 0; r < REPS; r++)                                                                     \
            asm volatile(R16(body) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(f0), "v"(f1), "v"(f2_)); \
        EPILOGUE                                                                                           \
    }

K32(k_fma, "v_fma_f32 %0, %4, %5, %6\n v_fma_f32 %1, %5, %6, %7\n v_fma_f32 %2, %4, %6, %7\n v_fma_f32 %3, %4, %5, %7\n")
K32(k_add, "v_add_f32 %0, %4, %5\n v_add_f32 %1, %5, %6\n v_add_f32 %2, %4, %6\n v_add_f32 %3, %4, %7\n")
K32(k_max3, "v_max3_f32 %0, %4, %5, %6\n v_max3_f32 %1, %5, %6, %7\n v_max3_f32 %2, %4, %6, %7\n v_max3_f32 %3, %4, %5, %7\n")
K32(k_perm, "v_perm_b32 %0, %8, %9, %10\n v_perm_b32 %1, %9, %8, %10\n v_perm_b32 %2, %8, %10, %9\n v_perm_b32 %3, %9, %10, %8\n")
K32(k_alignbit, "v_alignbit_b32 %0, %8, %8, %10\n v_alignbit_b32 %1, %9, %9, %10\n v_alignbit_b32 %2, %8, %9, %10\n v_alignbit_b32 %3, %9, %8, %10\n")
K32(k_cvt, "v_cvt_f32_u32 %0, %8\n v_cvt_f32_u32 %1, %9\n v_cvt_f32_u32 %2, %10\n v_cvt_f32_u32 %3, %8\n")
K32(k_cvt_sdwa, "v_cvt_f32_u32_sdwa %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %1, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
                "v_cvt_f32_u32_sdwa %2, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_u32_sdwa %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n")
K32(k_cvt_ubyte, "v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %9\n v_cvt_f32_ubyte2 %2, %10\n v_cvt_f32_ubyte3 %3, %8\n")
K32(k_cndmask, "v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %6, vcc\n v_cndmask_b32 %2, %4, %6, vcc\n v_cndmask_b32 %3, %4, %7, vcc\n")
K32(k_cmp, "v_cmp_le_f32 s[20:21], %4, %5\n v_cmp_le_f32 s[22:23], %5, %6\n v_cmp_le_f32 s[20:21], %4, %6\n v_cmp_le_f32 s[22:23], %4, %7\n")
K32(k_and_or, "v_and_or_b32 %0, %8, %9, %10\n v_and_or_b32 %1, %9, %8, %10\n v_and_or_b32 %2, %8, %10, %9\n v_and_or_b32 %3, %9, %10, %8\n")
K32(k_add3, "v_add3_u32 %0, %8, %9, %10\n v_add3_u32 %1, %9, %8, %10\n v_add3_u32 %2, %8, %10, %9\n v_add3_u32 %3, %9, %10, %8\n")
K32(k_mad_u24, "v_mad_u32_u24 %0, %8, %9, %10\n v_mad_u32_u24 %1, %9, %8, %10\n v_mad_u32_u24 %2, %8, %10, %9\n v_mad_u32_u24 %3, %9, %10, %8\n")
K32(k_mul_lo, "v_mul_lo_u32 %0, %8, %9\n v_mul_lo_u32 %1, %9, %10\n v_mul_lo_u32 %2, %8, %10\n v_mul_lo_u32 %3, %9, %8\n")
K32(k_sub_u32, "v_sub_u32 %0, %8, %9\n v_sub_u32 %1, %9, %10\n v_sub_u32 %2, %8, %10\n v_sub_u32 %3, %9, %8\n")
K32(k_min3_u32, "v_min3_u32 %0, %8, %9, %10\n v_min3_u32 %1, %9, %8, %10\n v_min3_u32 %2, %8, %10, %9\n v_min3_u32 %3, %9, %10, %8\n")
K32(k_rcp, "v_rcp_f32 %0, %4\n v_rcp_f32 %1, %5\n v_rcp_f32 %2, %6\n v_rcp_f32 %3, %7\n")
K32(k_bfe, "v_bfe_u32 %0, %8, 8, 15\n v_bfe_u32 %1, %9, 8, 15\n v_bfe_u32 %2, %10, 8, 15\n v_bfe_u32 %3, %8, 16, 15\n")
KPK(k_pk_fma, "v_pk_fma_f32 %0, %4, %5, %6\n v_pk_fma_f32 %1, %5, %6, %4\n v_pk_fma_f32 %2, %4, %6, %5\n v_pk_fma_f32 %3, %6, %5, %4\n")
KPK(k_pk_mul, "v_pk_mul_f32 %0, %4, %5\n v_pk_mul_f32 %1, %5, %6\n v_pk_mul_f32 %2, %4, %6\n v_pk_mul_f32 %3, %6, %5\n")
KPK(k_pk_add, "v_pk_add_f32 %0, %4, %5\n v_pk_add_f32 %1, %5, %6\n v_pk_add_f32 %2, %4, %6\n v_pk_add_f32 %3, %6, %5\n")
K64(k_fma64, "v_fma_f64 %0, %4, %5, %6\n v_fma_f64 %1, %5, %6, %4\n v_fma_f64 %2, %4, %6, %5\n v_fma_f64 %3, %6, %5, %4\n")
K64(k_mul64, "v_mul_f64 %0, %4, %5\n v_mul_f64 %1, %5, %6\n v_mul_f64 %2, %4, %6\n v_mul_f64 %3, %6, %5\n")
K64(k_add64, "v_add_f64 %0, %4, %5\n v_add_f64 %1, %5, %6\n v_add_f64 %2, %4, %6\n v_add_f64 %3, %6, %5\n")
K64(k_rcp64, "v_rcp_f64 %0, %4\n v_rcp_f64 %1, %5\n v_rcp_f64 %2, %6\n v_rcp_f64 %3, %4\n")
K6432(k_cvt3264, "v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %4\n")

struct Case {
    const char *name;
    void (*fn)(unsigned long long *, float);
};

int main() {
    std::vector<Case> cases = {{"v_fma_f32", k_fma},       {"v_add_f32", k_add},         {"v_max3_f32", k_max3},       {"v_perm_b32", k_perm},
                               {"v_alignbit_b32", k_alignbit}, {"v_cvt_f32_u32", k_cvt},  {"v_cvt_f32_u32 sdwa", k_cvt_sdwa}, {"v_cvt_f32_ubyteN", k_cvt_ubyte},
                               {"v_cndmask_b32", k_cndmask}, {"v_cmp_le_f32 -> sgpr", k_cmp}, {"v_and_or_b32", k_and_or}, {"v_add3_u32", k_add3},
                               {"v_mad_u32_u24", k_mad_u24}, {"v_mul_lo_u32", k_mul_lo}, {"v_sub_u32", k_sub_u32}, {"v_min3_u32", k_min3_u32},
                               {"v_rcp_f32", k_rcp}, {"v_bfe_u32", k_bfe}, {"v_pk_fma_f32", k_pk_fma}, {"v_pk_mul_f32", k_pk_mul}, {"v_pk_add_f32", k_pk_add},
                               {"v_fma_f64", k_fma64}, {"v_mul_f64", k_mul64}, {"v_add_f64", k_add64}, {"v_rcp_f64", k_rcp64}, {"v_cvt_f32_f64", k_cvt3264}};
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned long long *d = nullptr;
    (void)hipMalloc(&d, sizeof(unsigned long long) * (2 + cus * 16 + 64));
    std::printf("%-22s %28s %28s %28s\n", "instruction", "1 wave/SIMD (wave | SIMD)", "2 waves/SIMD", "4 waves/SIMD");
    const double n_inst = (double)REPS * 16 * 4;
    for (auto &c : cases) {
        std::printf("%-22s", c.name);
        for (int wps : {1, 2, 4}) {
            const int threads = 256 * wps;  // one block per CU: wps waves on each of the 4 SIMDs
            hipLaunchKernelGGL(c.fn, dim3(cus), dim3(threads), 0, 0, d, 1.0f);  // warm-up
            hipLaunchKernelGGL(c.fn, dim3(cus), dim3(threads), 0, 0, d, 1.0f);
            (void)hipDeviceSynchronize();
            std::vector<unsigned long long> h(2 + cus * 16);
            (void)hipMemcpy(h.data(), d, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
            std::vector<double> t;
            for (int i = 0; i < cus * threads / 64; i++) t.push_back((double)h[2 + i]);
            std::sort(t.begin(), t.end());
            const double med = t[t.size() / 2];
            std::printf("   %10.2f | %10.2f   ", med / n_inst, med / n_inst / wps);
        }
        std::printf("\n");
    }
    (void)hipFree(d);
    return 0;
}
