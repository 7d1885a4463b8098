// tk_kernels.h — the HIP kernels of the wavefront path tracer (gfx950, wave64).
//
//   k_generate        camera rays for one batch (src/render.cpp:69-75)
//   k_trace_group     (tk_trace_quad.h) closest hit for the extend queue (scene_intersect, src/scene.cpp:25), first
//                     hit for the shadow queue + radiance add (scene_occluded, src/scene.cpp:49), and — with the
//                     HookIo policy — the C-ABI trace hooks (AoS rays in, hit records out)
//   k_shade<TAG>      one integrator round per path, one instance per material tag, block-aggregated compaction
//                     into the next queues
//   k_sort_*          counting sort of the extend queue by the material tag of the hit
//   k_accumulate      per-pixel sum of the batch's samples, in sample order (src/render.cpp:68-77)
//   k_resolve         divide by spp, vertical flip (src/render.cpp:78)
//
// Launch shape: the trace kernel (tk_trace_quad.h) is persistent (grid = CUs x resident blocks) and pulls work from
// a device-side head counter, so a round needs no host read-back of the queue length.
#pragma once

#include <hip/hip_runtime.h>

#include "tk_integrate.h"
#include "tk_trace_quad.h"

namespace tk {

constexpr int BLOCK = 256;  // 4 waves
constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
// rank of this lane among the set bits of `mask` below it
__device__ __forceinline__ int mask_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}
// Wave-aggregated append: every lane of the wave calls it (convergent); lanes with `want` get a slot.
__device__ __forceinline__ void wave_append(bool want, int32_t value, int32_t *counter, int32_t *queue) {
    const uint64_t mask = __ballot(want);
    if (mask == 0) return;
    const int leader = __ffsll((unsigned long long)mask) - 1;
    int32_t base = 0;
    if (lane_id() == leader) base = atomicAdd(counter, (int32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (want) queue[base + mask_rank(mask)] = value;
}

template <class R>
__global__ void __launch_bounds__(BLOCK)
k_generate(DeviceScene<R> sc, RenderParams<R> rp, PathState<R> st, int32_t *queue, int64_t n) {
    for (int64_t s = (int64_t)blockIdx.x * BLOCK + threadIdx.x; s < n; s += (int64_t)gridDim.x * BLOCK) {
        generate_path(sc, rp, st, s);
        queue[s] = (int32_t)s;
    }
}

// Round 0 without a generate pass: the closest-hit launch of the camera rays makes each ray itself (two random numbers
// and a normalise) instead of reading it from a record that a kernel before it wrote — k_generate moved 100 B per path
// through HBM for nothing (35 ms of a 6.7 s mixed step in f64, 18 ms in f32).  The ray goes into the path record with
// the hit (same cache line), where the shade round and — in two-level scenes — reload() find it; the work list is the
// identity, so no queue is read.  Everything else of the initial record (throughput 1, radiance 0, stream counter) is
// implied by k == 0 in shade_path.
template <class R> struct CameraIo : PathIo<R> {
    CameraRec<R> cam;
    RenderParams<R> rp;
    template <bool SHADOW> __device__ __forceinline__ void load(int32_t i, RayT<R> &ray, int32_t &tag) const {
        const int64_t slot = i;
        tag = i;
        const Vec3<R> d = camera_dir(cam, rp, slot);
        const PathState<R> &st = this->st;
        st.R_(S_OX, slot) = cam.lookfrom[0], st.R_(S_OY, slot) = cam.lookfrom[1], st.R_(S_OZ, slot) = cam.lookfrom[2];
        st.R_(S_DX, slot) = d.x, st.R_(S_DY, slot) = d.y, st.R_(S_DZ, slot) = d.z;
        ray = make_ray(cam.lookfrom[0], cam.lookfrom[1], cam.lookfrom[2], d.x, d.y, d.z, this->eps, Const<R>::inf());
    }
};
__global__ void __launch_bounds__(BLOCK) k_iota(int32_t *queue, int64_t n) {
    for (int64_t s = (int64_t)blockIdx.x * BLOCK + threadIdx.x; s < n; s += (int64_t)gridDim.x * BLOCK) queue[s] = (int32_t)s;
}

// One integrator round.  Requests are compacted into the next extend queue and the shadow queue with one
// atomic per wave and queue (ballot + mbcnt prefix).
// TAG: the kernel is instantiated per material tag (+ TAG_MISS).  With a sorted queue (`tag_count` != null) an
// instance walks only its tag's segment [sum(tag_count[0..TAG)), +tag_count[TAG]); with an unsorted queue
// (single-tag scenes) the one instance of the scene's tag walks the whole queue.
// ALT: the reference's other integrators (shade_path_alt, tk_integrate.h; rp.integrator 1..3) — separate instances, so
// that the default integrator's register allocation is untouched.
#ifndef TK_SHADE_RECORD
#define TK_SHADE_RECORD 1  // the shade kernels work on a register copy of the path record (0: in memory)
#endif
// one path's record held in registers: the accessors of PathState on a local copy
template <class R> struct RecordView {
    union {
        mutable uint4 q[PATH_REC * sizeof(R) / 16];
        mutable R w[PATH_REC];
    };
    __device__ __forceinline__ RecordView() {}
    __device__ __forceinline__ R &R_(int c, int64_t) const { return w[c]; }
    __device__ __forceinline__ int32_t &I_(int c, int64_t) const { return *reinterpret_cast<int32_t *>(&w[c]); }
};
#ifndef TK_SHADE_WAVES_F32
#define TK_SHADE_WAVES_F32 5  // register cap of the shade kernels in waves per SIMD (0: none).  The f32 Diffuse instance sits at 96 VGPRs = the last count that gives five waves; 97 is four, and the fifth wave is worth 4 % of its time: the cap keeps an innocent edit from costing it
#endif
#ifndef TK_SHADE_WAVES_F64
#define TK_SHADE_WAVES_F64 0
#endif
// (the f32 cap applies to the Diffuse instance of the default integrator — the one the headline workload runs — and to
// the Disney stubs that clone it: 97 VGPRs without it, 96 and no scratch with it; the other tags would pay with scratch)
template <class R, int TAG, bool ALT> constexpr int shade_waves() {
    const bool lambert = TAG == 0 || TAG == 7 || TAG == 8 || TAG == 10 || TAG == 11;  // Diffuse and the Disney stubs that clone it
    const int w = sizeof(R) == 4 ? ((lambert && !ALT) ? TK_SHADE_WAVES_F32 : 0) : TK_SHADE_WAVES_F64;
    return w > 0 ? w : 1;
}
template <class R, int TAG, bool ALT = false>
__global__ void __launch_bounds__(BLOCK, (shade_waves<R, TAG, ALT>()))
k_shade(DeviceScene<R> sc, RenderParams<R> rp, PathState<R> st, const int32_t *__restrict__ queue,
        const int32_t *__restrict__ n_ptr, const int32_t *__restrict__ tag_count, int32_t *next_queue,
        int32_t *n_next, int32_t *shadow_queue, int32_t *n_shadow, int k, unsigned long long *counters, float *to_f32) {
    int32_t begin = 0, n = *n_ptr;
    if (tag_count) {
        for (int t = 0; t < TAG; t++) begin += tag_count[t];
        n = tag_count[TAG];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&counters[C_BOUNCES], (unsigned long long)n);
    // one path per thread, no grid-stride loop: the grid covers an upper bound of the queue length the host knows
    // (queues only shrink), so nothing loop-invariant is kept live across iterations (fewer SGPR/VGPR spills)
    const int32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (blockIdx.x * BLOCK >= n) return;  // whole block beyond the queue (uniform)
    uint32_t req = 0;
    int32_t slot = 0;
    if (i < n) {
        slot = queue[begin + i];
#if TK_SHADE_RECORD
        if constexpr (sizeof(R) == 4) {
            // the whole 128-byte record in registers: eight 16-byte loads up front and eight 16-byte stores at the end
            // instead of field-wise accesses spread over the function, each a request of its own per lane and a
            // dependency of its own (measured: shade kernel -12 %; 98 instead of 68 VGPRs, 5 waves instead of 7).
            // Inline asm: written as C++ loads the compiler narrows them back to the fields each branch reads.
            RecordView<R> rv;
            const char *rec = (const char *)(st.r + (int64_t)slot * PATH_REC);
            asm volatile(
                "global_load_dwordx4 %0, %8, off\n\t"
                "global_load_dwordx4 %1, %8, off offset:16\n\t"
                "global_load_dwordx4 %2, %8, off offset:32\n\t"
                "global_load_dwordx4 %3, %8, off offset:48\n\t"
                "global_load_dwordx4 %4, %8, off offset:64\n\t"
                "global_load_dwordx4 %5, %8, off offset:80\n\t"
                "global_load_dwordx4 %6, %8, off offset:96\n\t"
                "global_load_dwordx4 %7, %8, off offset:112\n\t"
                "s_waitcnt vmcnt(0)"
                : "=&v"(rv.q[0]), "=&v"(rv.q[1]), "=&v"(rv.q[2]), "=&v"(rv.q[3]), "=&v"(rv.q[4]), "=&v"(rv.q[5]), "=&v"(rv.q[6]),
                  "=&v"(rv.q[7])
                : "v"(rec)
                : "memory");
            req = ALT ? shade_path_alt<R, TAG, RecordView<R>>(sc, rp, rv, (int64_t)slot, k)
                      : shade_path<R, TAG, RecordView<R>>(sc, rp, rv, (int64_t)slot, k);
            uint4 *out = (uint4 *)(st.r + (int64_t)slot * PATH_REC);
#pragma unroll
            for (int c = 0; c < PATH_REC / 4; c++) out[c] = rv.q[c];
        } else
#endif
#if TK_SHADE_RECORD
        if constexpr (sizeof(R) == 8) {  // the 256-byte f64 record: 64 registers (measured: shade kernel -11 %)
            RecordView<R> rv;
            const char *rec = (const char *)(st.r + (int64_t)slot * PATH_REC);
            asm volatile(
                "global_load_dwordx4 %0, %16, off\n\t"
                "global_load_dwordx4 %1, %16, off offset:16\n\t"
                "global_load_dwordx4 %2, %16, off offset:32\n\t"
                "global_load_dwordx4 %3, %16, off offset:48\n\t"
                "global_load_dwordx4 %4, %16, off offset:64\n\t"
                "global_load_dwordx4 %5, %16, off offset:80\n\t"
                "global_load_dwordx4 %6, %16, off offset:96\n\t"
                "global_load_dwordx4 %7, %16, off offset:112\n\t"
                "global_load_dwordx4 %8, %16, off offset:128\n\t"
                "global_load_dwordx4 %9, %16, off offset:144\n\t"
                "global_load_dwordx4 %10, %16, off offset:160\n\t"
                "global_load_dwordx4 %11, %16, off offset:176\n\t"
                "global_load_dwordx4 %12, %16, off offset:192\n\t"
                "global_load_dwordx4 %13, %16, off offset:208\n\t"
                "global_load_dwordx4 %14, %16, off offset:224\n\t"
                "global_load_dwordx4 %15, %16, off offset:240\n\t"
                "s_waitcnt vmcnt(0)"
                : "=&v"(rv.q[0]), "=&v"(rv.q[1]), "=&v"(rv.q[2]), "=&v"(rv.q[3]), "=&v"(rv.q[4]), "=&v"(rv.q[5]), "=&v"(rv.q[6]),
                  "=&v"(rv.q[7]), "=&v"(rv.q[8]), "=&v"(rv.q[9]), "=&v"(rv.q[10]), "=&v"(rv.q[11]), "=&v"(rv.q[12]),
                  "=&v"(rv.q[13]), "=&v"(rv.q[14]), "=&v"(rv.q[15])
                : "v"(rec)
                : "memory");
            req = ALT ? shade_path_alt<R, TAG, RecordView<R>>(sc, rp, rv, (int64_t)slot, k)
                      : shade_path<R, TAG, RecordView<R>>(sc, rp, rv, (int64_t)slot, k);
            // mixed precision, last exact round (to_f32 = the f32 records): a path that goes on does so in the f32
            // record of its slot — ray, throughput, radiance so far, pending BSDF sample, stream counter, flags — written
            // here, from the registers that hold them, instead of by a pass of its own over both record sets.  The
            // radiance moves with the path: the f64 record keeps only what this round's shadow ray still adds.
            if (to_f32 != nullptr && (req & REQ_EXTEND)) {
                RecordView<float> fv;
#pragma unroll
                for (int c = 0; c < PATH_REC / 4; c++) fv.q[c] = make_uint4(0, 0, 0, 0);
                constexpr int WORDS[] = {S_OX, S_OY, S_OZ, S_DX, S_DY, S_DZ, S_PDF, S_TX, S_TY, S_TZ, S_LX, S_LY, S_LZ, S_FX, S_FY, S_FZ};
#pragma unroll
                for (int w = 0; w < 16; w++) fv.w[WORDS[w]] = (float)rv.w[WORDS[w]];
                fv.I_(S_CTR, 0) = rv.I_(S_CTR, 0);
                fv.I_(S_FLAGS, 0) = rv.I_(S_FLAGS, 0);
                rv.w[S_LX] = 0.0, rv.w[S_LY] = 0.0, rv.w[S_LZ] = 0.0;
                rv.I_(S_CONV, 0) = 1;
                uint4 *out32 = (uint4 *)(to_f32 + (int64_t)slot * PATH_REC);
#pragma unroll
                for (int c = 0; c < PATH_REC / 4; c++) out32[c] = fv.q[c];
            }
            uint4 *out = (uint4 *)(st.r + (int64_t)slot * PATH_REC);
#pragma unroll
            for (int c = 0; c < 16; c++) out[c] = rv.q[c];
        } else
#endif
        req = ALT ? shade_path_alt<R, TAG>(sc, rp, st, (int64_t)slot, k) : shade_path<R, TAG>(sc, rp, st, (int64_t)slot, k);
    }
    // Block-aggregated append to the two output queues: ballot + mbcnt inside each wave, wave counts combined in
    // LDS, ONE atomicAdd per block and queue.  (One atomic per wave on a single counter word was the limit of this
    // kernel: ~86 atomics/us, the rate MI355X_MICROARCH.md gives for one contended word.)
    __shared__ int32_t s_cnt[2][BLOCK / WAVE];
    __shared__ int32_t s_base[2];
    const bool want_e = (req & REQ_EXTEND) != 0, want_s = (req & REQ_SHADOW) != 0;
    const uint64_t me = __ballot(want_e), ms = __ballot(want_s);
    const int wave = threadIdx.x / WAVE;
    if (lane_id() == 0) {
        s_cnt[0][wave] = (int32_t)__popcll(me);
        s_cnt[1][wave] = (int32_t)__popcll(ms);
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        int32_t tot = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / WAVE; w++) tot += s_cnt[threadIdx.x][w];
        s_base[threadIdx.x] = tot ? atomicAdd(threadIdx.x == 0 ? n_next : n_shadow, tot) : 0;
    }
    __syncthreads();
    int32_t off_e = s_base[0], off_s = s_base[1];
    for (int w = 0; w < wave; w++) off_e += s_cnt[0][w], off_s += s_cnt[1][w];
    if (want_e) next_queue[off_e + mask_rank(me)] = slot;
    if (want_s) shadow_queue[off_s + mask_rank(ms)] = slot;
}

// ---- material sort of the extend queue (after trace_closest, before shade): a stable counting sort on the tag of
// the hit primitive (two dependent gathers per path, here and not in the trace kernel, where a load at the end of a
// ray stalls the wave; misses sort last), with no atomics at all —
// the first version's per-wave atomics on 13 cursor words cost 39 % of a mixed-material step.
//   k_sort_count    every WAVE owns a contiguous range of the queue: it reads the hit words, writes one key byte
//                   per entry and counts each key with ballots; hist[key][wave]
//   k_sort_scan     one block: exclusive scan of hist in (key, wave) order -> base[key][wave]; tag_count[key]
//   k_sort_scatter  same ranges: position = base[key][wave] + entries of that key seen so far in the wave
// The sorted order is deterministic (ranges and ranks do not depend on timing).
constexpr int N_SORT_KEYS = TAKE_MAT_COUNT + 1;
constexpr int SORT_SCAN_THREADS = 1024;
template <class R> __device__ __forceinline__ int sort_key(const PrimRec<R> *__restrict__ prims, const InstShade<R> *__restrict__ inst_shade,
                                                       const PathState<R> &st, int32_t slot) {
    const int32_t prim = st.I_(S_HIT, slot);
    if (prim < 0) return N_SORT_KEYS - 1;
    if (inst_shade) {  // two-level scenes: the material of a placement overrides the prototype's
        const int32_t inst = st.I_(S_INST, slot);
        if (inst >= 0) return inst_shade[inst].tag;
    }
    return (prims[prim].meta >> 8) & 0xff;
}
// range of queue entries owned by global wave `w` of `n_waves`: multiples of 64, contiguous, covering [0, n)
__device__ __forceinline__ void sort_range(int32_t n, int32_t w, int32_t n_waves, int32_t &begin, int32_t &end) {
    const int32_t per = ((n + n_waves - 1) / n_waves + WAVE - 1) / WAVE * WAVE;
    const int64_t b = (int64_t)w * per;
    begin = (int32_t)(b < n ? b : n);
    end = (int32_t)(b + per < n ? b + per : n);
}
template <class R>
__global__ void __launch_bounds__(BLOCK)
k_sort_count(const PrimRec<R> *__restrict__ prims, const InstShade<R> *__restrict__ inst_shade, PathState<R> st,
             const int32_t *__restrict__ queue, const int32_t *__restrict__ n_ptr, uint8_t *keys, int32_t *hist) {
    const int32_t n = *n_ptr, n_waves = gridDim.x * (BLOCK / WAVE);
    const int32_t w = blockIdx.x * (BLOCK / WAVE) + threadIdx.x / WAVE;
    int32_t begin, end;
    sort_range(n, w, n_waves, begin, end);
    int32_t cnt[N_SORT_KEYS];
#pragma unroll
    for (int t = 0; t < N_SORT_KEYS; t++) cnt[t] = 0;
    for (int32_t i = begin + lane_id(); i < end + lane_id(); i += WAVE) {  // uniform trip count
        const bool valid = i < end;
        const int key = valid ? sort_key(prims, inst_shade, st, queue[i]) : -1;
        if (valid) keys[i] = (uint8_t)key;
#pragma unroll
        for (int t = 0; t < N_SORT_KEYS; t++) cnt[t] += (int32_t)__popcll(__ballot(key == t));
    }
    int32_t mine = 0;
#pragma unroll
    for (int t = 0; t < N_SORT_KEYS; t++) mine = lane_id() == t ? cnt[t] : mine;
    if (lane_id() < N_SORT_KEYS) hist[lane_id() * n_waves + w] = mine;
}
__global__ void __launch_bounds__(SORT_SCAN_THREADS)
k_sort_scan(const int32_t *__restrict__ hist, int32_t *base, int32_t *tag_count, int32_t n_waves) {
    __shared__ int32_t s_sum[SORT_SCAN_THREADS];
    const int32_t total = N_SORT_KEYS * n_waves;
    const int32_t per = (total + SORT_SCAN_THREADS - 1) / SORT_SCAN_THREADS;
    const int32_t lo = min((int32_t)threadIdx.x * per, total), hi = min(lo + per, total);
    int32_t sum = 0;
    for (int32_t i = lo; i < hi; i++) sum += hist[i];
    s_sum[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < SORT_SCAN_THREADS; off <<= 1) {  // Hillis-Steele inclusive scan of the per-thread sums
        const int32_t v = threadIdx.x >= off ? s_sum[threadIdx.x - off] : 0;
        __syncthreads();
        s_sum[threadIdx.x] += v;
        __syncthreads();
    }
    int32_t run = s_sum[threadIdx.x] - sum;  // exclusive prefix of this thread's run
    for (int32_t i = lo; i < hi; i++) {
        base[i] = run;
        run += hist[i];
    }
    __syncthreads();
    if (threadIdx.x < N_SORT_KEYS) {  // totals per key = base of the next key's first wave - base of this key's
        const int32_t first = threadIdx.x * n_waves, next = first + n_waves;
        const int32_t end = next < total ? base[next] : s_sum[SORT_SCAN_THREADS - 1];
        tag_count[threadIdx.x] = end - base[first];
    }
}
__global__ void __launch_bounds__(BLOCK)
k_sort_scatter(const int32_t *__restrict__ queue, const int32_t *__restrict__ n_ptr, const uint8_t *__restrict__ keys,
               const int32_t *__restrict__ base, int32_t *sorted) {
    const int32_t n = *n_ptr, n_waves = gridDim.x * (BLOCK / WAVE);
    const int32_t w = blockIdx.x * (BLOCK / WAVE) + threadIdx.x / WAVE;
    int32_t begin, end;
    sort_range(n, w, n_waves, begin, end);
    if (begin >= end) return;
    int32_t off[N_SORT_KEYS];
#pragma unroll
    for (int t = 0; t < N_SORT_KEYS; t++) off[t] = base[t * n_waves + w];
    for (int32_t i = begin + lane_id(); i < end + lane_id(); i += WAVE) {
        const bool valid = i < end;
        const int key = valid ? (int)keys[i] : -1;
        int32_t pos = 0;
#pragma unroll
        for (int t = 0; t < N_SORT_KEYS; t++) {
            const uint64_t m = __ballot(key == t);
            if (key == t) pos = off[t] + mask_rank(m);
            off[t] += (int32_t)__popcll(m);
        }
        if (valid) sorted[pos] = queue[i];
    }
}

// src/render.cpp:68-77: color += sample, in sample order.
template <class R>
__global__ void __launch_bounds__(BLOCK)
k_accumulate(PathState<R> st, R *accum, int32_t npix, int32_t spb) {
    for (int32_t p = blockIdx.x * BLOCK + threadIdx.x; p < npix; p += gridDim.x * BLOCK) {
        R r = accum[3 * (int64_t)p], g = accum[3 * (int64_t)p + 1], b = accum[3 * (int64_t)p + 2];
        for (int s = 0; s < spb; s++) {
            const int64_t slot = (int64_t)s * npix + p;
            r = r + st.R_(S_LX, slot);
            g = g + st.R_(S_LY, slot);
            b = b + st.R_(S_LZ, slot);
        }
        accum[3 * (int64_t)p] = r;
        accum[3 * (int64_t)p + 1] = g;
        accum[3 * (int64_t)p + 2] = b;
    }
}
// Mixed precision (TAKE_PRECISION_MIXED): the paths of `queue` (alive after the exact rounds) continue on f32 records
// of the same slots: ray, throughput, radiance so far, pending BSDF sample, random-stream counter, flags.  The radiance
// moves with the path (the f64 record's is cleared), so that a sample's value is the sum of the two records' radiance
// whichever round its path ended in (k_accumulate_mixed; S_CONV in the f64 record says whether the f32 record counts).
// The render loop does this inside the last exact shade round (k_shade, to_f32); this kernel is the stand-alone form
// for builds without the register copy of the record (TK_SHADE_RECORD = 0).
__global__ void __launch_bounds__(BLOCK)
k_convert_state(PathState<double> a, PathState<float> b, const int32_t *__restrict__ queue, const int32_t *__restrict__ n_ptr) {
    const int32_t n = *n_ptr;
    for (int32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
        const int64_t s = queue[i];
        constexpr int WORDS[] = {S_OX, S_OY, S_OZ, S_DX, S_DY, S_DZ, S_PDF, S_TX, S_TY, S_TZ, S_LX, S_LY, S_LZ, S_FX, S_FY, S_FZ};
#pragma unroll
        for (int w = 0; w < 16; w++) b.R_(WORDS[w], s) = (float)a.R_(WORDS[w], s);
        b.I_(S_CTR, s) = a.I_(S_CTR, s);
        b.I_(S_FLAGS, s) = a.I_(S_FLAGS, s);
        a.R_(S_LX, s) = 0.0, a.R_(S_LY, s) = 0.0, a.R_(S_LZ, s) = 0.0;
        a.I_(S_CONV, s) = 1;
    }
}
__global__ void __launch_bounds__(BLOCK)
k_accumulate_mixed(PathState<double> a, PathState<float> b, double *accum, int32_t npix, int32_t spb) {
    for (int32_t p = blockIdx.x * BLOCK + threadIdx.x; p < npix; p += gridDim.x * BLOCK) {
        double r = accum[3 * (int64_t)p], g = accum[3 * (int64_t)p + 1], bl = accum[3 * (int64_t)p + 2];
        for (int s = 0; s < spb; s++) {
            const int64_t slot = (int64_t)s * npix + p;
            // (S_CONV: the f32 record of a slot holds this batch's path only if the path was converted)
            // (loaded whatever the flag says, then selected: a load that waits for the flag halves the loads in flight)
            const bool conv = a.I_(S_CONV, slot) != 0;
            const float bx = b.R_(S_LX, slot), by = b.R_(S_LY, slot), bz = b.R_(S_LZ, slot);
            r = r + (a.R_(S_LX, slot) + (conv ? (double)bx : 0.0));
            g = g + (a.R_(S_LY, slot) + (conv ? (double)by : 0.0));
            bl = bl + (a.R_(S_LZ, slot) + (conv ? (double)bz : 0.0));
        }
        accum[3 * (int64_t)p] = r;
        accum[3 * (int64_t)p + 1] = g;
        accum[3 * (int64_t)p + 2] = bl;
    }
}
// src/render.cpp:78: img(x, height - y - 1) = color / spp — the local rows come out in increasing image row.
template <class R>
__global__ void __launch_bounds__(BLOCK)
k_resolve(const R *__restrict__ accum, R *out, int32_t width, int32_t n_local_rows, int32_t spp) {
    const int32_t npix = width * n_local_rows;
    const R inv = R(1) / R(spp);
    for (int32_t p = blockIdx.x * BLOCK + threadIdx.x; p < npix; p += gridDim.x * BLOCK) {
        const int lr = p / width, x = p % width;
        const int64_t o = 3 * ((int64_t)(n_local_rows - 1 - lr) * width + x);
        out[o] = accum[3 * (int64_t)p] * inv;
        out[o + 1] = accum[3 * (int64_t)p + 1] * inv;
        out[o + 2] = accum[3 * (int64_t)p + 2] * inv;
    }
}

// ---- egress on the device: the conversion half of the reference's imwrite("image.exr")
// (src/image.cpp:155-176: Image3 double -> float, then tinyexr's SaveEXR(.., components 3, fp16) -> per scanline the
// channels B, G, R as 16-bit halves).  One thread per pixel and channel; out = uint16 [height][3][width] in the byte
// order of an EXR scanline block before its ZIP pre-filter, so that the host only has to deflate and frame it.
// float -> half as that writer rounds (pinned on files it wrote, tests/golden/egress): the magnitude is rounded
// HALF UP on the first dropped mantissa bit — not to nearest-even; a carry may run into the exponent (up to inf);
// float denormals become zero; anything at or beyond 2^16 becomes inf; below the half normal range the same rule
// produces half denormals.
__device__ __forceinline__ uint16_t exr_half_bits(float f) {
    const uint32_t u = __float_as_uint(f);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const int32_t exp = (int32_t)((u >> 23) & 0xFFu);
    const uint32_t man = u & 0x7FFFFFu;
    const int32_t newexp = exp - 112;
    uint32_t out = 0;
    if (exp == 255) out = 0x7C00u | (man != 0 ? 0x200u : 0u);
    else if (exp != 0 && newexp >= 31) out = 0x7C00u;
    else if (exp != 0 && newexp > 0) out = (((uint32_t)newexp << 10) | (man >> 13)) + ((man >> 12) & 1u);
    else if (exp != 0 && 14 - newexp <= 24) {
        const uint32_t full = man | 0x800000u;
        const int sh = 14 - newexp;
        out = (full >> sh) + ((full >> (sh - 1)) & 1u);
    }
    return (uint16_t)(out | sign);
}
template <class R>
__global__ void __launch_bounds__(BLOCK) k_pack_exr(const R *__restrict__ rgb, int32_t width, int32_t height, uint16_t *out) {
    const int64_t total = (int64_t)width * height * 3;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) {
        const int x = (int)(i % width);
        const int c = (int)((i / width) % 3);  // position in the scanline: 0 = B, 1 = G, 2 = R
        const int64_t y = i / ((int64_t)3 * width);
        out[i] = exr_half_bits((float)rgb[(y * width + x) * 3 + (2 - c)]);
    }
}

}  // namespace tk

// ---- take_hip_debug_table: the device shading functions on the rows of the reference's golden tables
// (tests/golden/tables; column layouts of oracle/ref_harness.cpp).  `rnd` holds, per row, the first draws of the
// mt19937 stream the reference used, so sampling can be compared value for value.
namespace tk {
enum DebugTable { TAB_MATERIAL = 0, TAB_LIGHT = 1, TAB_TEXTURE = 2, TAB_TO_WORLD = 3, TAB_HEMICOS = 4, TAB_BURLEY = 5 };
constexpr int TAB_RND = 8;
template <class R>
__global__ void __launch_bounds__(BLOCK)
k_debug_table(DeviceScene<R> sc, int kind, const double *__restrict__ in, const double *__restrict__ rnd, int64_t n,
              double *out) {
    const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n) return;
    TableRng rng{rnd + TAB_RND * r, 0};
    auto V = [](const double *p) { return Vec3<R>{R(p[0]), R(p[1]), R(p[2])}; };
    auto put = [](double *o, Vec3<R> v) { o[0] = (double)v.x, o[1] = (double)v.y, o[2] = (double)v.z; };
    if (kind == TAB_MATERIAL) {
        const double *p = in + 27 * r;
        double *o = out + 14 * r;
        MaterialRec<R> m{};
        m.tag = (int)p[0];
        m.tex_kind = p[22] != 0 ? 1 : 0;
        m.tex_image = 0;
        for (int a = 0; a < 3; a++) m.color[a] = R(p[1 + a]);
        m.uscale = m.tex_kind ? R(p[23]) : R(1), m.vscale = m.tex_kind ? R(p[24]) : R(1);
        m.uoffset = m.tex_kind ? R(p[25]) : R(0), m.voffset = m.tex_kind ? R(p[26]) : R(0);
        m.p0 = R(p[4]), m.p1 = R(p[5]);
        Isect<R> v{};
        v.gn = V(p + 6), v.sn = V(p + 9);
        v.uv = {R(p[12]), R(p[13])};
        const Vec3<R> dir_in = V(p + 14), dir_out = V(p + 17);
        for (int a = 0; a < 14; a++) o[a] = 0.0;
        BsdfSample<R> rec;
        const bool ok = sample_bsdf(m, dir_in, v, rng, rec);
        o[5] = (double)random_real<R>(rng);
        if (ok) {
            o[0] = 1.0;
            put(o + 1, rec.dir_out);
            o[4] = (double)rec.pdf;
            put(o + 6, eval_bsdf(sc, m, dir_in, rec.dir_out, rec.pdf, v));
            o[13] = (double)bsdf_pdf(m, dir_in, rec.dir_out, v);
        }
        o[9] = (double)bsdf_pdf(m, dir_in, dir_out, v);
        put(o + 10, eval_bsdf(sc, m, dir_in, dir_out, R(p[20]), v));
    } else if (kind == TAB_BURLEY) {
        // tags 12..16 (rows of oracle_tab_burley): tag, colour[3], param[12], gn[3], sn[3], dir_in[3], dir_out[3],
        // seed, back -> the material table's 14 output columns
        const double *p = in + 30 * r;
        double *o = out + 14 * r;
        MaterialRec<R> m{};
        m.tag = (int)p[0];
        for (int a = 0; a < 3; a++) m.color[a] = R(p[1 + a]);
        m.uscale = m.vscale = R(1);
        for (int a = 0; a < TAKE_MATERIAL_PARAMS; a++) m.p[a] = R(p[4 + a]);
        m.p0 = m.p[0], m.p1 = m.p[1];
        Isect<R> v{};
        v.gn = V(p + 16), v.sn = V(p + 19);
        v.back = p[29] != 0;
        const Vec3<R> dir_in = V(p + 22), dir_out = V(p + 25);
        for (int a = 0; a < 14; a++) o[a] = 0.0;
        BsdfSample<R> rec;
        const bool ok = sample_bsdf(m, dir_in, v, rng, rec);
        o[5] = (double)random_real<R>(rng);
        if (ok) {
            o[0] = 1.0;
            put(o + 1, rec.dir_out);
            o[4] = (double)rec.pdf;
            put(o + 6, eval_bsdf(sc, m, dir_in, rec.dir_out, rec.pdf, v));
            o[13] = (double)bsdf_pdf(m, dir_in, rec.dir_out, v);
        }
        o[9] = (double)bsdf_pdf(m, dir_in, dir_out, v);
        put(o + 10, eval_bsdf(sc, m, dir_in, dir_out, R(0), v));
    } else if (kind == TAB_LIGHT) {
        const double *p = in + 30 * r;
        double *o = out + 9 * r;
        LightRec<R> l{};
        l.kind = 1;
        l.is_sphere = p[0] == 0 ? 1 : 0;
        for (int a = 0; a < 9; a++) l.v[a] = R(p[1 + a]), l.n[a] = R(p[10 + a]);
        const Vec3<R> ref = V(p + 19);
        const LightSample<R> s = sample_light_point(l, ref, rng);
        put(o, s.pos);
        put(o + 3, s.n);
        o[6] = (double)random_real<R>(rng);
        o[7] = (double)light_pdf_area(l, s.pos, ref);
        o[8] = (double)light_pdf_area(l, V(p + 24), ref);
    } else if (kind == TAB_TEXTURE) {
        const double *p = in + 6 * r;
        MaterialRec<R> m{};
        m.tex_kind = 1, m.tex_image = 0;
        m.uscale = R(p[2]), m.vscale = R(p[3]), m.uoffset = R(p[4]), m.voffset = R(p[5]);
        put(out + 3 * r, eval_texture(sc, m, Vec2<R>{R(p[0]), R(p[1])}));
    } else if (kind == TAB_TO_WORLD) {
        put(out + 3 * r, to_world(V(in + 6 * r), V(in + 6 * r + 3)));
    } else if (kind == TAB_HEMICOS) {
        put(out + 4 * r, hemisphere_cos<R>(rng));
        out[4 * r + 3] = (double)random_real<R>(rng);
    }
}
}  // namespace tk
