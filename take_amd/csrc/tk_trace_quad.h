// tk_trace_quad.h — the trace kernel.  One template, G lanes per ray: G = 1 (one ray per lane) is the production
// setting since the nodes are 64 bytes; G = 2 ("pair") and G = 4 ("quad") are kept as instances of the same template
// for A/B builds (-DTQ_GROUP=2) and pass the same parity suite.
//
// History of the choice (profiles/): with full-width 128-byte nodes (round 1) one ray per lane meant eight scattered
// 16-byte requests per lane and node, and lanes sharing a ray won (quad 143.6, pair 121.5, one ray per lane 128.8 ms
// per 8.3 M samples).  With the 64-byte compressed node (QN) a lane reads its whole node with 4 x dwordx4 from one
// line, the per-step bookkeeping (ranking, stack, phase logic: more than half of a step's instructions) is paid once
// per 64 rays instead of once per 32, leaves hold one primitive, and one ray per lane is ahead: f32 65.8 -> 80.0,
// f64 44.7 -> 61.4 Msamples/s on the bench scene (83.3 / 63.5 with the whole-record loads of the shade kernels).
//   * node step: 24 x v_cvt_f32_u32 (SDWA word select) -> 12 x v_pk_fma_f32 -> entry / exit distances directly (the
//     slot words are rotated per axis so that the plane met first is the low half: no min / max per axis) -> max3 /
//     min3; the four entry distances become integer keys (float bits | 3 - slot) and are ranked with six comparisons;
//     the nearest child stays in a register, the others go far-to-near on a per-ray LDS stack (15 levels of 8-byte
//     entries, 24-bit-multiply addressing; deeper levels in a global spill area behind an out-of-line call); popped
//     entries beyond the closest hit are dropped; shadow rays skip ranking and keys;
//   * a leaf is one primitive (host and device builders), tested by the ray's lane; hit attributes stay there;
//   * node steps and leaf steps run in separate phases; a wave keeps a pool of 64 queue indices (one atomic per
//     64 rays) and refills idle ray slots from it, so lanes do not idle behind the longest ray of the wave;
//   * nothing that loads sits at the end of a ray: that point is on the critical path of the whole wave;
//   * G >= 2 only: the lanes of a group split the child slots, keys are exchanged by DPP, a leaf's primitives are
//     dealt to the lanes.
// Same conservative box tests and the same primitive tests as tk_traverse.h: results are bit-identical to the
// per-lane traversal and to the oracle for every G; exact ties in t are resolved on (u, v), not on visiting order.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>

#include "tk_traverse.h"

namespace tk {

// tuning knobs (overridable with -D for experiments)
#ifndef TQ_GROUP
#define TQ_GROUP 1  // lanes per ray of the instantiated kernels: 1 = one ray per lane (production), 2 = pair, 4 = quad
#endif
#ifndef TQ_MIN_WAVES
// register cap in waves per SIMD.  One ray per lane: 92 VGPRs = 5 waves (6 caps at 80 and changes nothing measurable);
// pair: 6 (measured +4 % over 5; 7 and 8 spill and lose)
#define TQ_MIN_WAVES (TQ_GROUP == 1 ? 5 : 6)
#endif
#ifndef TQ_INST_WAVES
#define TQ_INST_WAVES (TQ_MIN_WAVES - 1)  // f32 two-level instances: room for the saved world-space ray
#endif
#ifndef TQ_F64_WAVES
#define TQ_F64_WAVES 4  // register cap of the f64 instances (waves per SIMD)
#endif
#ifndef TQ_W8_WAVES
#define TQ_W8_WAVES 4       // register caps of the 8-wide instances (waves per SIMD): f32, f32 two-level, f64
#endif
#ifndef TQ_W8_INST_WAVES
#define TQ_W8_INST_WAVES 4
#endif
#ifndef TQ_W8_F64_WAVES
#define TQ_W8_F64_WAVES 4
#endif
#ifndef TQ_W8_LEVELS
#define TQ_W8_LEVELS 19     // LDS stack levels of the 8-wide instances: 19 x 260 x 8 B = 39.5 KB per block = four blocks per CU
#endif
#ifndef TQ_G1_LEVELS
// stack levels in LDS with one ray per lane: 15 x 260 entries x 8 B = 31 KB per block = the five blocks per CU the
// registers allow (measured: 8 levels -13 %, 12 and 15 equal, 16 at four blocks -5 %)
#define TQ_G1_LEVELS 15
#endif
#ifndef TQ_PAIR_LEVELS
#define TQ_PAIR_LEVELS 24
#endif
#ifndef TQ_SWITCH_LANES
// leave the node phase when fewer lanes than this (of 64) are at interior nodes (one ray per lane: 16 / 24 / 32 / 40
// within 1 %; pair: 32)
#define TQ_SWITCH_LANES (TQ_GROUP == 1 ? 24 : 32)
#endif
#ifndef TQ_REFILL_DIV
#define TQ_REFILL_DIV 8     // refill when at least 1/TQ_REFILL_DIV of the wave's ray slots are idle (4: -2 %, measured)
#endif
#ifndef TQ_NODE_ITERS_DEF
#define TQ_NODE_ITERS_DEF 6
#endif
#ifndef TQ_TIEBREAK
#define TQ_TIEBREAK 1  // 0: experiments only — exact ties in t go to the last candidate seen (tree-dependent)
#endif
constexpr int TQ_BLOCK = 256;                     // 4 waves
constexpr int TQ_NODE_ITERS = TQ_NODE_ITERS_DEF;  // at most this many node steps before the next leaf phase / refill check
constexpr uint32_t TQ_KEY_INVALID = 0x7FFFFFFFu;  // above every finite non-negative float's bit pattern, below 2^31

typedef unsigned long long tq_entry;          // low word: child word, high word: order key (float bits | lane)

// wave-wide vote straight from the compare (HIP's __ballot goes through an integer 0/1 first)
__device__ __forceinline__ uint64_t tq_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// 1 if a < b for keys below 2^31 (no carry/SGPR traffic: one subtract, one shift)
__device__ __forceinline__ int tq_less(uint32_t a, uint32_t b) { return (int)((a - b) >> 31); }

// ---- cross-lane helpers inside a group (DPP quad_perm: no LDS traffic)
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
}
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(dpp_i<CTRL>(__float_as_int(v)));
}
template <int CTRL> __device__ __forceinline__ double dpp_f(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(dpp_i<CTRL>(hi), dpp_i<CTRL>(lo));
}
constexpr int QP_B0 = 0x00, QP_B1 = 0x55, QP_B2 = 0xAA, QP_B3 = 0xFF;  // broadcast lane i of the quad
constexpr int QP_X1 = 0xB1;                                            // [1,0,3,2]
constexpr int QP_EVEN = 0xA0, QP_ODD = 0xF5;                           // [0,0,2,2] / [1,1,3,3]: one lane of each pair
constexpr int QP_X2 = 0x4E;                                            // [2,3,0,1]
struct StackSpill {
    tq_entry *base;  // [level][ray group of the persistent grid]
    int64_t stride;  // ray groups in the persistent grid
};
// The overflow path of the stack is kept out of line so that the common path compiles to plain ds_write_b64 /
// ds_read_b64 (a pointer select between LDS and global memory turns every access into a flat_* instruction).
__device__ __noinline__ void tq_spill_store(tq_entry *p, tq_entry e) { *p = e; }
__device__ __noinline__ tq_entry tq_spill_load(const tq_entry *p) { return *p; }

// IO policies: where rays come from and where results go.
//   load(i, ray, tag): ray i of the work list;  store_hit(tag, prim, t, u, v) / store_occlusion(tag, occ):
//   called by ONE lane of the group.
template <class R> struct PathIo {  // the render loop: rays in the path state, indexed through a queue
    static constexpr bool UNIFORM_TMIN = true;  // every ray starts at the same tmin (the ray epsilon): a scalar, not per-lane state
    const PrimRec<R> *prims;
    PathState<R> st;
    const int32_t *queue;
    R eps;
    __device__ __forceinline__ R tmin() const { return eps; }
    template <bool SHADOW> __device__ __forceinline__ void load(int32_t i, RayT<R> &ray, int32_t &tag) const {
        const int64_t slot = queue[i];
        tag = (int32_t)slot;
        reload<SHADOW>(slot, ray);
    }
    // the ray of path `slot` again (two-level scenes: a ray that leaves an instance gets its world-space form back)
    template <bool SHADOW> __device__ __forceinline__ void reload(int64_t slot, RayT<R> &ray) const {
        if (!SHADOW)
            ray = make_ray(st.R_(S_OX, slot), st.R_(S_OY, slot), st.R_(S_OZ, slot), st.R_(S_DX, slot), st.R_(S_DY, slot),
                           st.R_(S_DZ, slot), eps, Const<R>::inf());
        else
            ray = make_ray(st.R_(S_OX, slot), st.R_(S_OY, slot), st.R_(S_OZ, slot), st.R_(S_SX, slot), st.R_(S_SY, slot),
                           st.R_(S_SZ, slot), eps, st.R_(S_ST, slot));
    }
    // (no loads here: the end of a ray sits on the critical path of the whole wave — one dependent load in this
    // function cost the closest-hit kernel 6 %)
    __device__ __forceinline__ void store_hit(int64_t slot, int32_t prim, R t, R u, R v) const {
        st.I_(S_HIT, slot) = prim;
        st.R_(S_HT, slot) = t;
        st.R_(S_HU, slot) = u;
        st.R_(S_HV, slot) = v;
    }
    __device__ __forceinline__ void store_instance(int64_t slot, int32_t inst) const { st.I_(S_INST, slot) = inst; }
    // counting mode (the instrumented instances): "would the previous occluder of this path slot have stopped this
    // shadow ray too?" — the measurement behind DESIGN.md §7's answer to an occluder cache
    static constexpr bool OCC_PROBE = true;
    __device__ __forceinline__ int32_t last_occluder(int64_t slot) const { return st.I_(S_OCC, slot); }
    __device__ __forceinline__ void set_last_occluder(int64_t slot, int32_t prim) const { st.I_(S_OCC, slot) = prim; }
    __device__ __forceinline__ void store_occlusion(int64_t slot, bool occluded) const {
        if (!occluded) {  // the NEE term of this iteration reaches the light (path_tracing.h:53-58)
            st.R_(S_LX, slot) = st.R_(S_LX, slot) + st.R_(S_CX, slot);
            st.R_(S_LY, slot) = st.R_(S_LY, slot) + st.R_(S_CY, slot);
            st.R_(S_LZ, slot) = st.R_(S_LZ, slot) + st.R_(S_CZ, slot);
        }
    }
};
template <class R> struct RayAoS {
    R org[3], tmin, dir[3], tmax;
};
template <class R> struct HitAoS;
template <> struct HitAoS<float> {
    int32_t shape_id;
    float t, u, v;
};
template <> struct HitAoS<double> {
    int32_t shape_id, reserved;
    double t, u, v;
};
template <class R> struct HookIo {  // the C-ABI trace hooks: AoS rays in, hit records out
    static constexpr bool UNIFORM_TMIN = false;
    static constexpr bool OCC_PROBE = false;
    __device__ __forceinline__ int32_t last_occluder(int64_t) const { return -1; }
    __device__ __forceinline__ void set_last_occluder(int64_t, int32_t) const {}
    __device__ __forceinline__ R tmin() const { return R(0); }
    const PrimRec<R> *prims;
    const RayAoS<R> *rays;
    HitAoS<R> *hits;
    int32_t *occluded;
    const InstShade<R> *inst_shade;  // two-level scenes: shape id of an instanced hit = shape_base + local face
    template <bool SHADOW> __device__ __forceinline__ void load(int32_t i, RayT<R> &ray, int32_t &tag) const {
        tag = i;
        reload<SHADOW>(i, ray);
    }
    template <bool SHADOW> __device__ __forceinline__ void reload(int64_t i, RayT<R> &ray) const {
        const RayAoS<R> q = rays[i];
        // entry distances are ordered through their bit patterns, which needs tmin >= +0 (include/take_hip.h: the
        // host entry points refuse a negative tmin; device-resident rays are clamped here: -0.0, negatives and NaN
        // start at 0)
        ray = make_ray(q.org[0], q.org[1], q.org[2], q.dir[0], q.dir[1], q.dir[2], q.tmin > R(0) ? q.tmin : R(0), q.tmax);
    }
    // called after store_hit by the same lane
    __device__ __forceinline__ void store_instance(int64_t i, int32_t inst) const {
        if (inst >= 0 && hits[i].shape_id >= 0) hits[i].shape_id += inst_shade[inst].shape_base;
    }
    __device__ __forceinline__ void store_hit(int64_t i, int32_t prim, R t, R u, R v) const {
        HitAoS<R> h{};
        h.shape_id = prim >= 0 ? prims[prim].shape_id : -1;
        h.t = prim >= 0 ? t : R(0);
        h.u = u;
        h.v = v;
        hits[i] = h;
    }
    __device__ __forceinline__ void store_occlusion(int64_t i, bool occ) const { occluded[i] = occ ? 1 : 0; }
};

// Geometry of a ray group: G lanes cooperate on one ray (G = 4: quad, one child slot per lane; G = 2: pair, two
// slots per lane; G = 1: one ray per lane, four slots per lane).  More lanes per ray = better coalescing and less
// divergence, fewer lanes per ray = fewer instructions per box test (the per-step overhead is shared by more boxes).
template <int G, int W = 4> struct GroupGeom {
    static constexpr int LOG2 = G == 4 ? 2 : (G == 2 ? 1 : 0);
    static constexpr int CPL = 4 / G;                    // child slots per lane (4-wide nodes)
    static constexpr int GROUPS = TQ_BLOCK / G;          // rays in flight per block
    static constexpr int PER_WAVE = 64 / G;
    static constexpr int LEVELS = W == 8 ? TQ_W8_LEVELS : (G == 4 ? 32 : (G == 2 ? TQ_PAIR_LEVELS : TQ_G1_LEVELS));  // stack levels in LDS
    static constexpr int STRIDE = GROUPS + 4;            // entries per level (+4: 32 B skew between levels)
    static constexpr int SPILL = (W == 8 ? MAX_STACK_ENTRIES_W8 : MAX_STACK_ENTRIES) + 4 - LEVELS;  // deeper levels in global memory (the builders cap the depth)
};
template <int G, class T> __device__ __forceinline__ T group_min(T v) {
    if (G >= 2) v = tk_fmin(v, dpp_f<QP_X1>(v));
    if (G >= 4) v = tk_fmin(v, dpp_f<QP_X2>(v));
    return v;
}
template <int G> __device__ __forceinline__ int group_max_i(int v) {
    if (G >= 2) v = max(v, dpp_i<QP_X1>(v));
    if (G >= 4) v = max(v, dpp_i<QP_X2>(v));
    return v;
}

// QN: traverse the 64-byte compressed nodes (sc.qnodes; pairs only) instead of the full-width ones.
// INST: two-level scenes (TakeInstance).  A leaf of the top-level tree may be an instance word: the ray is moved into
// the prototype's object space (t is the same number in both spaces), a return marker goes on the stack, traversal
// continues at the prototype's root; when the marker is popped the ray gets its world-space form back.  A separate
// instance of the kernel, so that one-level scenes pay nothing.
// W: node width of the compressed tree (QN): 4 = QNode4 (sc.qnodes), 8 = QNode8 (sc.qnodes8; one ray per lane only).
template <class R, int G, bool ANY_HIT, bool COUNT, class Io, bool QN = false, bool INST = false, int W = 4>
__global__ void __launch_bounds__(TQ_BLOCK, W == 8 ? (sizeof(R) == 8 ? TQ_W8_F64_WAVES : (INST ? TQ_W8_INST_WAVES : TQ_W8_WAVES))
                                                   : (sizeof(R) == 8 ? (INST ? TQ_F64_WAVES - 1 : TQ_F64_WAVES) : (INST ? TQ_INST_WAVES : TQ_MIN_WAVES)))  // f64, two-level: room for the wider state
k_trace_group(DeviceScene<R> sc, Io io, const int32_t *__restrict__ n_ptr, int32_t n_direct, int32_t *head,
              unsigned long long *counters, int counter_word, StackSpill spill) {
    using GG = GroupGeom<G, W>;
    constexpr int CPL = GG::CPL;
    static_assert(W == 4 || (W == 8 && QN && G == 1), "8-wide nodes: compressed, one ray per lane");
    constexpr uint32_t KEY_SLOT_MASK = W == 8 ? 7u : 3u;  // low bits of an order key hold the slot
    __shared__ tq_entry s_stack[GG::LEVELS * GG::STRIDE];
    typedef __attribute__((address_space(3))) tq_entry lds_entry;
    const int lane = threadIdx.x & 63;
    const int gl = lane & (G - 1);            // lane within the group
    const int grp = threadIdx.x >> GG::LOG2;  // group within the block
    lds_entry *const stk = (lds_entry *)&s_stack[grp];  // this group's column; level l at stk[l * STRIDE]
    tq_entry *const spl = spill.base + ((int64_t)blockIdx.x * GG::GROUPS + grp);
    auto spill_at = [&](int level) -> tq_entry * { return spl + (int64_t)(level - GG::LEVELS) * spill.stride; };
    const int32_t n = n_ptr ? *n_ptr : n_direct;
    if (blockIdx.x == 0 && threadIdx.x == 0 && counter_word >= 0)
        atomicAdd(&counters[counter_word], (unsigned long long)n);
    static_assert(!QN || G <= 2, "compressed nodes: one ray per lane (production) and the pair kernel");
    const char *const node_base = QN ? (W == 8 ? (const char *)sc.qnodes8 : (const char *)sc.qnodes) : (const char *)sc.nodes;
    const char *const prim_base = (const char *)sc.prims;

    // wave-local pool of queue indices [pool_next, pool_end), refilled 64 at a time
    int32_t pool_next = 0, pool_end = 0;
    bool exhausted = false;
    // per-group traversal state (identical in the G lanes unless noted)
    RayT<R> ray{};
    R idx = R(0), idy = R(0), idz = R(0), tbest = R(0);
    QRay qr{};  // QN: the ray in grid space (replaces idx/idy/idz, which are dead then)
    // QN in f64: the box tests run in f32 on the compressed nodes (conservative, so precision is not at stake; hits are
    // decided by the double-precision primitive tests) — the limits enter them rounded outwards
    float tmin_f = 0.0f, tbest_f = 0.0f;
    auto lim_lo = [&]() -> float {
        if constexpr (Io::UNIFORM_TMIN) return stack_key(io.tmin());  // uniform: stays in a scalar register
        else if constexpr (sizeof(R) == 4) return (float)ray.tmin;
        else return tmin_f;
    };
    auto lim_hi = [&]() -> float { if constexpr (sizeof(R) == 4) return (float)tbest; else return tbest_f; };
    int32_t tag = 0;            // path slot (render) / ray index (trace hooks) of the ray in this slot
    int sp = 0;                 // entries on this group's stack
    int32_t cur = CHILD_EMPTY;  // >= 0: at an interior node; < 0: at a leaf; CHILD_EMPTY: the slot holds no ray
    // per-lane best candidate (differs between the lanes of a group)
    R my_t = Const<R>::inf(), my_u = R(0), my_v = R(0);
    int32_t my_prim = -1;
    int32_t inst = -1, my_inst = -1;  // INST: instance the ray is inside of / the candidate was found in
    Vec3<R> w_o{}, w_d{};             // INST: the world-space ray while the lanes are inside an instance
    QRay w_qr{};
    R w_idx = R(0), w_idy = R(0), w_idz = R(0);
    uint32_t cnt_nodes = 0, cnt_prims = 0, cnt_leaves = 0, cnt_wnode = 0, cnt_wleaf = 0, cnt_wait = 0, cnt_idle = 0;

    // The trace kernels are bound by VALU issue (profiles/): stack addressing uses 24-bit multiplies
    // (full rate; the 32x32 and 64-bit forms the compiler picks for plain indexing are quarter rate).
    typedef __attribute__((address_space(3))) char lds_char;
    constexpr uint32_t LEVEL_BYTES = (uint32_t)GG::STRIDE * 8u;
    auto lds_level = [&](int level) -> lds_entry * { return (lds_entry *)((lds_char *)stk + __umul24((uint32_t)level, LEVEL_BYTES)); };
    auto push_entry = [&](int level, tq_entry e) {
        if (level < GG::LEVELS)
            *lds_level(level) = e;
        else
            tq_spill_store(spill_at(level), e);
    };
    // Next subtree that can still hold a closer hit (entries whose entry distance is beyond the closest hit are
    // dropped).  Returns true when the stack is empty: the ray is finished.
    auto advance = [&]() -> bool {
        for (;;) {
            if (sp == 0) return true;
            --sp;
            const tq_entry e = (sp < GG::LEVELS) ? *lds_level(sp) : tq_spill_load(spill_at(sp));
            cur = (int32_t)(uint32_t)e;
            if (INST && cur == CHILD_EMPTY) {  // the return marker of an instance: back to world space
                ray.o = w_o, ray.d = w_d;
                if constexpr (QN) qr = w_qr;
                else idx = w_idx, idy = w_idy, idz = w_idz;
                inst = -1;
                continue;
            }
            if (ANY_HIT) return false;  // the limit of a shadow ray never shrinks: nothing to cull
            const float key = __uint_as_float((uint32_t)(e >> 32) & ~KEY_SLOT_MASK);
            if ((R)key <= tbest) return false;
        }
    };
    // ONE lane of the group writes the result; the slot becomes idle
    auto finish = [&]() {
        if (ANY_HIT) {
            const bool occ = group_max_i<G>(my_prim) >= 0;
            if (gl == 0) io.store_occlusion(tag, occ);
            if constexpr (COUNT && Io::OCC_PROBE && G == 1 && !INST) {
                if (occ) io.set_last_occluder(tag, my_prim);
            }
        } else {
            // the lane holding the closest candidate writes it.  At exactly equal distances (a ray through an edge
            // shared by two triangles) the candidate with the larger (u, v) wins, here and in the leaf phase — then, for
            // exactly coincident primitives, the larger shape id: a rule on values every tree produces identically,
            // so the image does not depend on the builder — and one that needs no load on the common path (the end
            // of a ray is on the critical path of the wave).
            bool mine = my_prim >= 0 && my_t == tbest;
            bool any = mine;
            if (!TQ_TIEBREAK) {
                const int win = group_max_i<G>(mine ? gl : -1);
                any = win >= 0, mine = gl == win;
            } else if (G >= 2) {
                const bool pm = dpp_i<QP_X1>((int)mine) != 0;
                const R pu = dpp_f<QP_X1>(my_u), pv = dpp_f<QP_X1>(my_v);
                any = any || pm;
                if (mine && pm) {
                    // (t, u, v) all equal = exactly coincident primitives: the larger (instance, primitive index) wins.
                    // Both builders lay coincident primitives out with ascending shape ids (tk_host_scene.h:
                    // order_coincident; the device build's stable Morton sort), so this is "the larger shape id" —
                    // a rule every tree applies alike — decided on registers, without a load.
                    const int32_t pp = dpp_i<QP_X1>(my_prim);
                    bool lose = pu > my_u || (pu == my_u && (pv > my_v || (pv == my_v && pp > my_prim)));
                    if (INST) {
                        const int32_t pi = dpp_i<QP_X1>(my_inst);
                        if (pu == my_u && pv == my_v && pi != my_inst) lose = pi > my_inst;
                    }
                    if (lose) mine = false;
                }
            }
            if (TQ_TIEBREAK && G >= 4) {  // winners of the two pairs against each other
                const bool pm = dpp_i<QP_X2>((int)mine) != 0;
                const R pu = dpp_f<QP_X2>(my_u), pv = dpp_f<QP_X2>(my_v);
                const bool pany = dpp_i<QP_X2>((int)any) != 0;
                // the partner lane (gl ^ 2) need not be its pair's winner: look at both lanes of the other pair
                const bool qm = dpp_i<QP_X1>(dpp_i<QP_X2>((int)mine)) != 0;
                const R qu = dpp_f<QP_X1>(dpp_f<QP_X2>(my_u)), qv = dpp_f<QP_X1>(dpp_f<QP_X2>(my_v));
                const bool lose_p = pm && (pu > my_u || (pu == my_u && (pv > my_v || (pv == my_v && (gl & 2) == 0))));
                const bool lose_q = qm && (qu > my_u || (qu == my_u && (qv > my_v || (qv == my_v && (gl & 2) == 0))));
                if (mine && (lose_p || lose_q)) mine = false;
                any = any || pany;
            }
            if (!any) {
                if (gl == 0) {
                    io.store_hit(tag, -1, tbest, R(0), R(0));  // (no hit: tbest is still the ray's tmax — which need not stay live)
                    if (INST) io.store_instance(tag, -1);
                }
            } else if (mine) {
                io.store_hit(tag, my_prim, my_t, my_u, my_v);
                if (INST) io.store_instance(tag, my_inst);
            }
        }
        cur = CHILD_EMPTY;
    };

    for (;;) {
        // ------------------------------------------------------------------ refill idle groups from the pool
        {
            const uint64_t idle0 = tq_ballot(cur == CHILD_EMPTY);
            const int n_idle = __builtin_popcountll(idle0) >> GG::LOG2;
            if (n_idle * TQ_REFILL_DIV >= GG::PER_WAVE) {  // enough of the wave's ray slots are idle
#pragma unroll 1
                for (int pass = 0; pass < 2; ++pass) {
                    const uint64_t idle = tq_ballot(cur == CHILD_EMPTY);
                    if (idle == 0) break;
                    int32_t avail = pool_end - pool_next;
                    if (avail == 0) {
                        if (exhausted) break;
                        int32_t base = 0;
                        if (lane == 0) base = atomicAdd(head, 64);
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (base >= n) {
                            exhausted = true;
                            break;
                        }
                        pool_next = base;
                        pool_end = min(base + 64, n);
                        avail = pool_end - pool_next;
                    }
                    const int my_rank = __builtin_popcountll(idle & ((1ull << (lane & ~(G - 1))) - 1ull)) >> GG::LOG2;
                    if (cur == CHILD_EMPTY && my_rank < avail) {
                        io.template load<ANY_HIT>(pool_next + my_rank, ray, tag);
                        idx = safe_inv(ray.d.x), idy = safe_inv(ray.d.y), idz = safe_inv(ray.d.z);
                        if constexpr (QN) qr = qray_make(sc.grid_lo, sc.grid_step, ray.o, idx, idy, idz);
                        tbest = ray.tmax;
                        if constexpr (QN && sizeof(R) == 8) tmin_f = stack_key(ray.tmin), tbest_f = float_above(tbest);
                        sp = 0;
                        my_t = Const<R>::inf();
                        my_prim = -1;
                        my_u = my_v = R(0);
                        if (INST) inst = -1, my_inst = -1;
                        cur = sc.root_child;
                        if (sc.root_child == CHILD_EMPTY) finish();  // empty scene: a miss, the slot stays idle
                        if constexpr (COUNT && ANY_HIT && Io::OCC_PROBE && G == 1 && !INST) {
                            // instrument (counting instances only; the traversal goes on regardless): test the
                            // primitive that occluded this slot's previous shadow ray against the new one
                            const int32_t prev = io.last_occluder(tag);
                            if (prev >= 0) {
                                const PrimRec<R> &p = sc.prims[prev];
                                R a9[9];
                                for (int q9 = 0; q9 < 9; q9++) a9[q9] = p.a[q9];
                                RayT<R> r2 = ray;
                                if constexpr (Io::UNIFORM_TMIN) r2.tmin = io.tmin();
                                R t2, u2 = R(0), v2 = R(0);
                                const bool hit2 = ((p.meta & 0xff) == PRIM_TRIANGLE) ? tri_test(a9, r2, ray.tmax, t2, u2, v2) : sphere_test(a9, r2, ray.tmax, t2);
                                if (hit2) atomicAdd(&counters[C_OCC_CACHE_HITS], 1ull);
                            }
                        }
                    }
                    pool_next += min(avail, (int32_t)(__builtin_popcountll(idle) >> GG::LOG2));
                }
            }
        }
        if (tq_ballot(cur != CHILD_EMPTY) == 0) {
            if (exhausted) break;
            continue;
        }
        // ------------------------------------------------------------------ node phase
        // Interior-node steps only; a group that reaches a leaf waits.  The leaf code (triangle / sphere tests) is
        // ~2x the node code, and some group of the wave is at a leaf in almost every step: running the two in
        // separate phases keeps the leaf code out of the node steps.
#pragma unroll 1
        for (int it = 0; it < TQ_NODE_ITERS; ++it) {
            const bool at_node = cur >= 0;
            const uint64_t m_node = tq_ballot(at_node);
            if (m_node == 0) break;
            const uint64_t m_leaf = tq_ballot((uint32_t)cur > (uint32_t)CHILD_EMPTY);
            // few groups left at interior nodes and some waiting at a leaf: switch to the leaf phase
            if (__builtin_popcountll(m_node) < TQ_SWITCH_LANES && m_leaf != 0) break;
            if (COUNT) {
                const int nw = __builtin_popcountll(m_leaf), ni = __builtin_popcountll(tq_ballot(cur == CHILD_EMPTY));
                if (lane == 0) cnt_wnode++, cnt_wait += (uint32_t)(nw >> GG::LOG2), cnt_idle += (uint32_t)(ni >> GG::LOG2);
            }
            if constexpr (W == 8) {
              if (at_node) {
                // ---- 8-wide step.  The lane loads its ray's eight 16-byte slots in VISITING order: the slot of rank k
                // sits at byte (k << 4) ^ oct16 of the node's line (slot = rank ^ octant: front to back as far as one
                // permutation per octant can tell).  Closest hit: the nearest hit child (minimum of the eight keys =
                // entry distance bits | rank) is visited next, the others go on the stack far rank first, each with
                // its key for the pop-time culling — no ranking of eight distances.  Shadow rays: key = rank.
                if (COUNT) cnt_nodes++;
                const uint32_t off = ((uint32_t)cur * (uint32_t)sizeof(QNode8)) ^ qr.oct16;
                uint4 c[8];
#pragma unroll
                for (int k = 0; k < 8; k++) c[k] = *(const uint4 *)(node_base + (off ^ (uint32_t)(k << 4)));
                uint32_t key[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    float tn;
                    const bool ok = qbox_test(qr, c[k].x, c[k].y, c[k].z, (int32_t)c[k].w, lim_lo(), lim_hi(), tn);
                    if (ANY_HIT) key[k] = ok ? (uint32_t)k : TQ_KEY_INVALID;
                    else key[k] = ok ? ((__float_as_uint(TQ_PRESCALE ? tn : tn * Const<float>::BOX_SHRINK) & ~7u) | (uint32_t)k) : TQ_KEY_INVALID;
                }
                const uint32_t kmin = min(min(min(key[0], key[1]), min(key[2], key[3])), min(min(key[4], key[5]), min(key[6], key[7])));
                if (kmin != TQ_KEY_INVALID) {
                    int32_t cand = CHILD_EMPTY;
                    int cnt = 0;
                    if (sp + 7 <= GG::LEVELS) {  // all of this step's entries land in LDS: no per-entry check
#pragma unroll
                        for (int k = 7; k >= 0; k--) {
                            if (key[k] == kmin) cand = (int32_t)c[k].w;
                            const bool push = key[k] != TQ_KEY_INVALID && key[k] != kmin;  // (valid keys are distinct)
                            if (push) *lds_level(sp + cnt) = ((tq_entry)key[k] << 32) | (tq_entry)c[k].w;
                            cnt += push ? 1 : 0;
                        }
                    } else {
#pragma unroll
                        for (int k = 7; k >= 0; k--) {
                            if (key[k] == kmin) cand = (int32_t)c[k].w;
                            const bool push = key[k] != TQ_KEY_INVALID && key[k] != kmin;
                            if (push) push_entry(sp + cnt, ((tq_entry)key[k] << 32) | (tq_entry)c[k].w);
                            cnt += push ? 1 : 0;
                        }
                    }
                    sp += cnt;
                    cur = cand;
                } else if (advance()) {
                    finish();
                }
              }
            } else
            if (at_node) {
                if (COUNT && gl == 0) cnt_nodes++;
                uint32_t key[CPL];
                int32_t child[CPL];
                if constexpr (QN) {
                    // each lane loads its two 16-byte slots (planes on the scene grid + child word): one line
                    // look-up per ray per instruction, no exchange between the lanes
                    const uint32_t off = (uint32_t)cur * (uint32_t)sizeof(QNode4) + (uint32_t)(gl * CPL) * (uint32_t)sizeof(QChild);
#pragma unroll
                    for (int j = 0; j < CPL; j++) {
                        const uint4 c = *(const uint4 *)(node_base + off + j * (uint32_t)sizeof(QChild));
                        float tn;
                        const bool ok = qbox_test(qr, c.x, c.y, c.z, (int32_t)c.w, lim_lo(), lim_hi(), tn);
                        key[j] = ok ? ((__float_as_uint(TQ_PRESCALE ? tn : tn * Const<float>::BOX_SHRINK) & ~3u) | (uint32_t)(3 - (gl * CPL + j)))
                                    : TQ_KEY_INVALID;
                        child[j] = (int32_t)c.w;
                    }
                } else {
                    // CPL child slots per lane: 32-bit byte offset from the (scalar) node base
                    const uint32_t off = (uint32_t)cur * (uint32_t)sizeof(Node4<R>) +
                                         (uint32_t)(gl * CPL) * (uint32_t)sizeof(NodeChild<R>);
#pragma unroll
                    for (int j = 0; j < CPL; j++) {
                        const NodeChild<R> c = *(const NodeChild<R> *)(node_base + off + j * (uint32_t)sizeof(NodeChild<R>));
                        R tn;
                        const bool ok = box_test(c, ray.o, idx, idy, idz, ray.tmin, tbest, tn);
                        // order key: the (shrunk, hence conservative) entry distance as an integer — non-negative
                        // floats order like their bit patterns — with 3 - slot number in the two low bits, which
                        // makes the four keys of a node distinct and sends equal distances (every box that contains
                        // the ray origin enters at tmin) to the LAST slot first: slots are stored largest box first
                        // (tk_bvh.h, for the shadow rays), and the smallest box is the likeliest to give a near hit
                        key[j] = ok ? ((__float_as_uint(stack_key(tn * Const<R>::BOX_SHRINK)) & ~3u) | (uint32_t)(3 - (gl * CPL + j)))
                                    : TQ_KEY_INVALID;
                        child[j] = c.child;
                    }
                }
                if constexpr (ANY_HIT && G == 2) {
                    // shadow rays: any occluder ends the ray and the limit never shrinks, so the order of the
                    // children does not matter and nothing is culled at pop time — no ranking, no keys.  The hit
                    // slots are numbered lane 0 first; number 0 is visited next, the others are pushed.
                    const int v0 = key[0] != TQ_KEY_INVALID, v1 = key[CPL - 1] != TQ_KEY_INVALID;
                    const int mine = v0 + v1;
                    const int theirs = dpp_i<QP_X1>(mine);
                    const int nhit = mine + theirs;
                    if (nhit != 0) {
                        const int pos0 = gl ? theirs : 0, pos1 = pos0 + v0;
                        int32_t cand = CHILD_EMPTY;  // INT_MIN: below every child word
                        if (sp + nhit - 1 <= GG::LEVELS) {  // all of this step's entries land in LDS: no per-entry check
                            if (v0) {
                                if (pos0 == 0) cand = child[0];
                                else *lds_level(sp + pos0 - 1) = (tq_entry)(uint32_t)child[0];
                            }
                            if (v1) {
                                if (pos1 == 0) cand = child[CPL - 1];
                                else *lds_level(sp + pos1 - 1) = (tq_entry)(uint32_t)child[CPL - 1];
                            }
                        } else {
                            if (v0) {
                                if (pos0 == 0) cand = child[0];
                                else push_entry(sp + pos0 - 1, (tq_entry)(uint32_t)child[0]);
                            }
                            if (v1) {
                                if (pos1 == 0) cand = child[CPL - 1];
                                else push_entry(sp + pos1 - 1, (tq_entry)(uint32_t)child[CPL - 1]);
                            }
                        }
                        sp += nhit - 1;
                        cur = group_max_i<G>(cand);
                    } else if (advance()) {
                        finish();
                    }
                } else if constexpr (ANY_HIT && G == 1) {
                    // the same for one ray per lane: the first hit slot is visited next, the others are pushed
                    int nhit = 0;
#pragma unroll
                    for (int j = 0; j < CPL; j++) nhit += key[j] != TQ_KEY_INVALID;
                    if (nhit != 0) {
                        int32_t cand = CHILD_EMPTY;
                        int pos = 0;
#pragma unroll
                        for (int j = 0; j < CPL; j++) {
                            if (key[j] != TQ_KEY_INVALID) {
                                if (pos == 0) cand = child[j];
                                else push_entry(sp + pos - 1, (tq_entry)(uint32_t)child[j]);
                                pos++;
                            }
                        }
                        sp += nhit - 1;
                        cur = cand;
                    } else if (advance()) {
                        finish();
                    }
                } else {
                    // rank of each of my slots among the four keys of the node (keys of the other lanes come by DPP;
                    // a key never compares less than itself, so broadcasting all four is fine)
                    int rank[CPL], nhit = 0;
                    if (G == 4) {
                        const uint32_t k0 = (uint32_t)dpp_i<QP_B0>((int)key[0]), k1 = (uint32_t)dpp_i<QP_B1>((int)key[0]);
                        const uint32_t k2 = (uint32_t)dpp_i<QP_B2>((int)key[0]), k3 = (uint32_t)dpp_i<QP_B3>((int)key[0]);
                        rank[0] = tq_less(k0, key[0]) + tq_less(k1, key[0]) + tq_less(k2, key[0]) + tq_less(k3, key[0]);
                        nhit = tq_less(k0, TQ_KEY_INVALID) + tq_less(k1, TQ_KEY_INVALID) + tq_less(k2, TQ_KEY_INVALID) +
                               tq_less(k3, TQ_KEY_INVALID);
                    } else if (G == 2) {
                        const uint32_t p0 = (uint32_t)dpp_i<QP_X1>((int)key[0]), p1 = (uint32_t)dpp_i<QP_X1>((int)key[CPL - 1]);
                        const int lt01 = tq_less(key[0], key[CPL - 1]);  // my two slots against each other (keys distinct)
                        rank[0] = (1 - lt01) + tq_less(p0, key[0]) + tq_less(p1, key[0]);
                        rank[CPL - 1] = lt01 + tq_less(p0, key[CPL - 1]) + tq_less(p1, key[CPL - 1]);
                        const int mine = tq_less(key[0], TQ_KEY_INVALID) + tq_less(key[CPL - 1], TQ_KEY_INVALID);
                        nhit = mine + dpp_i<QP_X1>(mine);
                    } else {
                        // one ray per lane: six comparisons give all four ranks (valid keys are distinct, so
                        // [a < b] = 1 - [b < a] wherever it matters: an invalid key is never less than anything)
                        constexpr int I1 = CPL > 1 ? 1 : 0, I2 = CPL > 2 ? 2 : 0, I3 = CPL - 1;  // (in range for every G)
                        const int l01 = tq_less(key[0], key[I1]), l02 = tq_less(key[0], key[I2]), l03 = tq_less(key[0], key[I3]);
                        const int l12 = tq_less(key[I1], key[I2]), l13 = tq_less(key[I1], key[I3]), l23 = tq_less(key[I2], key[I3]);
                        rank[0] = (1 - l01) + (1 - l02) + (1 - l03);
                        rank[I1] = l01 + (1 - l12) + (1 - l13);
                        rank[I2] = l02 + l12 + (1 - l23);
                        rank[I3] = l03 + l13 + l23;
#pragma unroll
                        for (int j = 0; j < CPL; j++) nhit += tq_less(key[j], TQ_KEY_INVALID);
                    }
                    if (nhit != 0) {
                        // the nearest child is visited next and never touches the stack; the others are pushed
                        // far-to-near, so the next nearest ends on top
                        int32_t cand = CHILD_EMPTY;  // INT_MIN: below every child word
                        if (sp + nhit - 1 <= GG::LEVELS) {  // all of this step's entries land in LDS: no per-entry check
#pragma unroll
                            for (int j = 0; j < CPL; j++) {
                                if (rank[j] == 0) cand = child[j];  // rank 0 with nhit != 0 is a hit
                                if (key[j] != TQ_KEY_INVALID && rank[j] != 0)
                                    *lds_level(sp + nhit - 1 - rank[j]) = ((tq_entry)key[j] << 32) | (tq_entry)(uint32_t)child[j];
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < CPL; j++) {
                                if (rank[j] == 0) cand = child[j];
                                if (key[j] != TQ_KEY_INVALID && rank[j] != 0)
                                    push_entry(sp + nhit - 1 - rank[j], ((tq_entry)key[j] << 32) | (tq_entry)(uint32_t)child[j]);
                            }
                        }
                        sp += nhit - 1;
                        cur = group_max_i<G>(cand);
                    } else if (advance()) {
                        finish();
                    }
                }
            }
        }
        // ------------------------------------------------------------------ leaf phase: primitives dealt to the lanes
        {
            const int32_t leaf = cur;
            const bool at_leaf = (uint32_t)leaf > (uint32_t)CHILD_EMPTY;
            if (COUNT) {  // (the vote must be taken by all lanes: inside `lane == 0 &&` it would see lane 0 alone)
                const uint64_t m_at_leaf = tq_ballot(at_leaf);
                if (lane == 0 && m_at_leaf != 0) cnt_wleaf++;
            }
            if (INST && at_leaf && is_instance_word(leaf)) {
                // enter the instance (uniform over the lanes of the group: `cur` is)
                const InstTrace<R> &it = sc.inst_trace[instance_of_word(leaf)];
                push_entry(sp, (tq_entry)(uint32_t)CHILD_EMPTY);  // return marker (key 0: never culled)
                sp++;
                const Vec3<R> o = ray.o, d = ray.d;
                w_o = o, w_d = d;  // (instances do not nest: the saved ray is the world-space one)
                if constexpr (QN) w_qr = qr;
                else w_idx = idx, w_idy = idy, w_idz = idz;
                ray.o = {it.inv[0] * o.x + it.inv[1] * o.y + it.inv[2] * o.z + it.inv[3],
                         it.inv[4] * o.x + it.inv[5] * o.y + it.inv[6] * o.z + it.inv[7],
                         it.inv[8] * o.x + it.inv[9] * o.y + it.inv[10] * o.z + it.inv[11]};
                ray.d = {it.inv[0] * d.x + it.inv[1] * d.y + it.inv[2] * d.z, it.inv[4] * d.x + it.inv[5] * d.y + it.inv[6] * d.z,
                         it.inv[8] * d.x + it.inv[9] * d.y + it.inv[10] * d.z};
                idx = safe_inv(ray.d.x), idy = safe_inv(ray.d.y), idz = safe_inv(ray.d.z);
                if constexpr (QN) qr = qray_make(it.grid_lo, it.grid_step, ray.o, idx, idy, idz);
                inst = instance_of_word(leaf);
                cur = it.root_child;
            } else if (at_leaf) {
                const int first = leaf_first(leaf), cnt = leaf_count(leaf);
                if (COUNT && gl == 0) cnt_prims += (uint32_t)cnt, cnt_leaves++;
                if constexpr (Io::UNIFORM_TMIN) ray.tmin = io.tmin();  // (re-materialised from the scalar: not kept per lane)
#pragma unroll 1
                for (int k = gl; k < cnt; k += G) {
                    const uint32_t off = (uint32_t)(first + k) * (uint32_t)sizeof(PrimRec<R>);
                    // the test side of the record (a[9], shape id, meta) as whole 16-byte loads: three (f32) / five
                    // (f64) line look-ups instead of the five / seven the field-wise loads compile to — the vector L1's
                    // look-up rate is one of the kernel's limits
                    struct PrimTest {
                        R a[9];
                        int32_t shape_id, meta;
                    };
                    static_assert(sizeof(PrimTest) <= (sizeof(R) == 4 ? 48 : 80) && offsetof(PrimRec<R>, meta) == offsetof(PrimTest, meta),
                                  "the head of PrimRec");
                    constexpr int NQ = sizeof(R) == 4 ? 3 : 5;
                    union {
                        uint4 q[NQ];
                        PrimTest t;
                    } rec;
                    // (inline asm: written as C++ loads the compiler narrows them back to the fields each branch uses)
                    if constexpr (sizeof(R) == 4) {
                        asm volatile(
                            "global_load_dwordx4 %0, %3, %4\n\t"
                            "global_load_dwordx4 %1, %3, %4 offset:16\n\t"
                            "global_load_dwordx4 %2, %3, %4 offset:32\n\t"
                            "s_waitcnt vmcnt(0)"
                            : "=&v"(rec.q[0]), "=&v"(rec.q[1]), "=&v"(rec.q[2])
                            : "v"(off), "s"(prim_base)
                            : "memory");
                    } else {
                        asm volatile(
                            "global_load_dwordx4 %0, %5, %6\n\t"
                            "global_load_dwordx4 %1, %5, %6 offset:16\n\t"
                            "global_load_dwordx4 %2, %5, %6 offset:32\n\t"
                            "global_load_dwordx4 %3, %5, %6 offset:48\n\t"
                            "global_load_dwordx4 %4, %5, %6 offset:64\n\t"
                            "s_waitcnt vmcnt(0)"
                            : "=&v"(rec.q[0]), "=&v"(rec.q[1]), "=&v"(rec.q[2]), "=&v"(rec.q[3]), "=&v"(rec.q[NQ - 1])
                            : "v"(off), "s"(prim_base)
                            : "memory");
                    }
                    const PrimTest &p = rec.t;
                    R t, u = R(0), v = R(0);
                    // later primitives of this lane see the distance of its earlier hits; the tests accept t == limit,
                    // and an equal distance replaces the candidate only for a larger (u, v) (tree-independent ties)
                    const R tlim = tk_fmin(tbest, my_t);
                    const bool ok = ((p.meta & 0xff) == PRIM_TRIANGLE) ? tri_test(p.a, ray, tlim, t, u, v)
                                                                       : sphere_test(p.a, ray, tlim, t);
                    bool take = ok;
                    if (TQ_TIEBREAK && ok && t == my_t) {  // rare: an exact tie; coincident primitives: larger (instance, index)
                        take = u > my_u || (u == my_u && (v > my_v || (v == my_v && first + k > my_prim)));
                        if (INST && u == my_u && v == my_v && inst != my_inst) take = inst > my_inst;
                    }
                    if (take) {
                        my_t = t, my_u = u, my_v = v;
                        my_prim = first + k;
                        if (INST) my_inst = inst;
                    }
                }
                tbest = tk_fmin(tbest, group_min<G>(my_t));
                if constexpr (QN && sizeof(R) == 8) tbest_f = float_above(tbest);
                bool finished = ANY_HIT ? (group_max_i<G>(my_prim) >= 0) : false;
                if (!finished) finished = advance();
                if (finished) finish();
            }
        }
    }
    if (COUNT) {
        unsigned long long nn = cnt_nodes, pp = cnt_prims, ll = cnt_leaves;
        for (int off = 32; off > 0; off >>= 1) {
            nn += __shfl_down(nn, off);
            pp += __shfl_down(pp, off);
            ll += __shfl_down(ll, off);
        }
        if (lane == 0) {
            atomicAdd(&counters[C_NODE_VISITS], nn);
            atomicAdd(&counters[C_PRIM_TESTS], pp);
            atomicAdd(&counters[C_LEAF_VISITS], ll);
            atomicAdd(&counters[C_WAVE_NODE_STEPS], (unsigned long long)cnt_wnode);
            atomicAdd(&counters[C_WAVE_LEAF_STEPS], (unsigned long long)cnt_wleaf);
            atomicAdd(&counters[C_WAIT_SLOTS], (unsigned long long)cnt_wait);
            atomicAdd(&counters[C_IDLE_SLOTS], (unsigned long long)cnt_idle);
        }
    }
}

}  // namespace tk
