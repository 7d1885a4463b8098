// tk_common.h — scalar/vector vocabulary of the HIP path-tracing core (gfx950).
//
// Everything here is usable from device code; the same headers are compiled for the host by the
// BVH builder and by tests/hostsim (a serial re-execution of the kernel bodies used to debug the
// device logic without a GPU — test infrastructure, never a product fallback).
//
// Floating-point contract: the device code is compiled with -ffp-contract=off, so every +,-,*,/
// and sqrt below is one IEEE-754 operation in source order (hipcc's fp32 divide/sqrt are correctly
// rounded by default).  Where a fused multiply-add is wanted (box tests) it is written tk_fma().
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TK_HD __host__ __device__ __forceinline__
#define TK_D __device__ __forceinline__
#else
#define TK_HD inline
#define TK_D inline
#endif

namespace tk {

template <class R> struct Const;
template <> struct Const<float> {
    static constexpr float DET_EPS = 1e-7f;  // c_EPSILON in its determinant-cull role (reference src/shape.cpp:58)
    static constexpr float PI = 3.14159265358979323846f;
    static constexpr float INVPI = 1.0f / PI;
    static constexpr float TWOPI = 2.0f * PI;
    static constexpr float INVTWOPI = 1.0f / TWOPI;
    static constexpr float BOX_GROW = 1.0000004f;  // 1 + 3 ulp: far-plane widening of the slab test
    static constexpr float BOX_SHRINK = 0.9999996f;
    static TK_HD float inf() { return __builtin_huge_valf(); }
};
template <> struct Const<double> {
    static constexpr double DET_EPS = 1e-7;
    static constexpr double PI = 3.14159265358979323846;
    static constexpr double INVPI = 1.0 / PI;
    static constexpr double TWOPI = 2.0 * PI;
    static constexpr double INVTWOPI = 1.0 / TWOPI;
    static constexpr double BOX_GROW = 1.0000000000000007;
    static constexpr double BOX_SHRINK = 0.9999999999999993;
    static TK_HD double inf() { return __builtin_huge_val(); }
};

// ---- scalar math: one name per operation, float/double overloads
TK_HD float tk_sqrt(float x) { return sqrtf(x); }
TK_HD double tk_sqrt(double x) { return sqrt(x); }
TK_HD float tk_fmin(float a, float b) { return fminf(a, b); }
TK_HD double tk_fmin(double a, double b) { return fmin(a, b); }
TK_HD float tk_fmax(float a, float b) { return fmaxf(a, b); }
TK_HD double tk_fmax(double a, double b) { return fmax(a, b); }
TK_HD float tk_fma(float a, float b, float c) { return fmaf(a, b, c); }
TK_HD double tk_fma(double a, double b, double c) { return fma(a, b, c); }
TK_HD float tk_sin(float x) { return sinf(x); }
TK_HD double tk_sin(double x) { return sin(x); }
TK_HD float tk_cos(float x) { return cosf(x); }
TK_HD double tk_cos(double x) { return cos(x); }
TK_HD float tk_tan(float x) { return tanf(x); }
TK_HD double tk_tan(double x) { return tan(x); }
TK_HD float tk_pow(float x, float y) { return powf(x, y); }
TK_HD double tk_pow(double x, double y) { return pow(x, y); }
TK_HD float tk_log(float x) { return logf(x); }
TK_HD double tk_log(double x) { return log(x); }
TK_HD float tk_acos(float x) { return acosf(x); }
TK_HD double tk_acos(double x) { return acos(x); }
TK_HD float tk_atan2(float y, float x) { return atan2f(y, x); }
TK_HD double tk_atan2(double y, double x) { return atan2(y, x); }
TK_HD float tk_floor(float x) { return floorf(x); }
TK_HD double tk_floor(double x) { return floor(x); }
TK_HD float tk_fabs(float x) { return fabsf(x); }
TK_HD double tk_fabs(double x) { return fabs(x); }
TK_HD float tk_fmod(float a, float b) { return fmodf(a, b); }
TK_HD double tk_fmod(double a, double b) { return fmod(a, b); }
TK_HD bool tk_isinf(float x) { return __builtin_isinf(x); }
TK_HD bool tk_isinf(double x) { return __builtin_isinf(x); }
template <class R> TK_HD R tk_clamp(R v, R lo, R hi) { return v < lo ? lo : (hi < v ? hi : v); }  // std::clamp
// x^5 as the reference's pow(x, 5) resolves for floating x: a libm pow with exponent 5
template <class R> TK_HD R tk_pow5(R x) { return tk_pow(x, R(5)); }

// ---- 3-vectors with the reference's operator semantics (src/vector.h): v / s multiplies by 1/s
template <class R> struct Vec3 {
    R x, y, z;
};
template <class R> struct Vec2 {
    R x, y;
};
template <class R> TK_HD Vec3<R> mk3(R x, R y, R z) { return Vec3<R>{x, y, z}; }
template <class R> TK_HD Vec3<R> operator+(Vec3<R> a, Vec3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class R> TK_HD Vec3<R> operator-(Vec3<R> a, Vec3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class R> TK_HD Vec3<R> operator-(Vec3<R> a) { return {-a.x, -a.y, -a.z}; }
template <class R> TK_HD Vec3<R> operator*(Vec3<R> a, Vec3<R> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <class R> TK_HD Vec3<R> operator*(R s, Vec3<R> v) { return {s * v.x, s * v.y, s * v.z}; }
template <class R> TK_HD Vec3<R> operator*(Vec3<R> v, R s) { return {v.x * s, v.y * s, v.z * s}; }
template <class R> TK_HD Vec3<R> operator/(Vec3<R> v, R s) {
    R inv = R(1) / s;
    return {v.x * inv, v.y * inv, v.z * inv};
}
template <class R> TK_HD Vec3<R> one_minus(Vec3<R> v) { return {R(1) - v.x, R(1) - v.y, R(1) - v.z}; }
template <class R> TK_HD R dot(Vec3<R> a, Vec3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class R> TK_HD Vec3<R> cross(Vec3<R> a, Vec3<R> b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class R> TK_HD R length(Vec3<R> v) { return tk_sqrt(dot(v, v)); }
template <class R> TK_HD Vec3<R> normalize(Vec3<R> v) {
    R l = length(v);
    if (l <= R(0)) return {R(0), R(0), R(0)};
    return v / l;
}
template <class R> TK_HD Vec2<R> operator+(Vec2<R> a, Vec2<R> b) { return {a.x + b.x, a.y + b.y}; }
template <class R> TK_HD Vec2<R> operator*(R s, Vec2<R> v) { return {s * v.x, s * v.y}; }

// Frisvad orthonormal basis (reference src/vector.h:314-326)
template <class R> TK_HD Vec3<R> to_world(Vec3<R> n, Vec3<R> v) {
    Vec3<R> x, y;
    if (n.z < R(-1 + 1e-6)) {
        x = {R(0), R(-1), R(0)};
        y = {R(-1), R(0), R(0)};
    } else {
        R a = R(1) / (R(1) + n.z);
        R b = -n.x * n.y * a;
        x = {R(1) - n.x * n.x * a, b, -n.x};
        y = {b, R(1) - n.y * n.y * a, -n.y};
    }
    return x * v.x + y * v.y + n * v.z;
}

// ---- whole-record loads (device).  A record read as plain C++ is loaded field by field where each branch needs it: more
// requests per lane, and dependent ones.  load_record issues sizeof(T) / 16 16-byte loads back to back and waits once —
// inline asm, because the compiler narrows C++ loads back to the fields that are used.  T: 16-byte multiple, 4-byte
// aligned; the shade kernels use it for the primitive record of a hit.
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> __device__ __forceinline__ void load_record(const T *src, T &dst) {
    constexpr int N = (int)(sizeof(T) / 16);
    static_assert(sizeof(T) % 16 == 0 && (N == 4 || N == 6 || N == 7), "record sizes in use: 64, 96, 112 bytes");
    uint4 q0, q1, q2, q3, q4, q5, q6;
    const char *p = (const char *)src;
    if constexpr (N == 4) {
        asm volatile(
            "global_load_dwordx4 %0, %4, off\n\t"
            "global_load_dwordx4 %1, %4, off offset:16\n\t"
            "global_load_dwordx4 %2, %4, off offset:32\n\t"
            "global_load_dwordx4 %3, %4, off offset:48\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
            : "v"(p)
            : "memory");
    } else if constexpr (N == 6) {
        asm volatile(
            "global_load_dwordx4 %0, %6, off\n\t"
            "global_load_dwordx4 %1, %6, off offset:16\n\t"
            "global_load_dwordx4 %2, %6, off offset:32\n\t"
            "global_load_dwordx4 %3, %6, off offset:48\n\t"
            "global_load_dwordx4 %4, %6, off offset:64\n\t"
            "global_load_dwordx4 %5, %6, off offset:80\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5)
            : "v"(p)
            : "memory");
    } else {
        asm volatile(
            "global_load_dwordx4 %0, %7, off\n\t"
            "global_load_dwordx4 %1, %7, off offset:16\n\t"
            "global_load_dwordx4 %2, %7, off offset:32\n\t"
            "global_load_dwordx4 %3, %7, off offset:48\n\t"
            "global_load_dwordx4 %4, %7, off offset:64\n\t"
            "global_load_dwordx4 %5, %7, off offset:80\n\t"
            "global_load_dwordx4 %6, %7, off offset:96\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6)
            : "v"(p)
            : "memory");
    }
    uint4 *d = (uint4 *)&dst;
    d[0] = q0, d[1] = q1, d[2] = q2, d[3] = q3;
    if constexpr (N >= 6) d[4] = q4, d[5] = q5;
    if constexpr (N >= 7) d[6] = q6;
}
#endif

// ---- counter-based random stream (specification: DESIGN.md §RNG; pinned by tests against
// oracle_counter_words).  One 64-bit word per random_real(); f32 keeps the top 24 bits, f64 the top 53.
struct Rng {
    uint64_t key;
    uint32_t ctr;
};
TK_HD uint64_t rng_mix(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
TK_HD uint64_t rng_key(uint64_t seed, uint64_t pixel, uint64_t sample) {
    return rng_mix(rng_mix(seed + 0x9E3779B97F4A7C15ull * (pixel + 1)) + 0xD1B54A32D192ED03ull * (sample + 1));
}
TK_HD uint64_t rng_word(Rng &r) { return rng_mix(r.key + 0x9E3779B97F4A7C15ull * (uint64_t)(r.ctr++)); }
TK_HD void rng_real(Rng &r, float &out) { out = (float)(rng_word(r) >> 40) * (1.0f / 16777216.0f); }
TK_HD void rng_real(Rng &r, double &out) { out = (double)(rng_word(r) >> 11) * (1.0 / 9007199254740992.0); }
// test-hook generator: the draws come from a table (take_hip_debug_table feeds the reference's own mt19937
// draws to the device functions so that they can be compared with the reference's golden tables)
struct TableRng {
    const double *v;
    uint32_t ctr;
};
TK_HD void rng_real(TableRng &r, float &out) { out = (float)r.v[r.ctr++]; }
TK_HD void rng_real(TableRng &r, double &out) { out = r.v[r.ctr++]; }
template <class R, class G> TK_HD R random_real(G &r) {
    R v;
    rng_real(r, v);
    return v;
}

}  // namespace tk
