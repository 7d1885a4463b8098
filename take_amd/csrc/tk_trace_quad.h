// tk_trace_quad.h — quad traversal: 4 lanes cooperate on one ray (a wave carries 16 rays).
//
// Why (profiles/r01_a_perlane_bvh4_spp4.txt): with one ray per lane every node visit is eight scattered 16-byte
// requests per lane, 64 different lines per wave instruction, and a wave runs as long as its slowest of 64 rays
// (VALU lane efficiency ~14 %, 63 % of wave cycles waiting on memory).  Here
//   * lane j of a quad loads child slot j of the 128-byte node: one wave instruction reads 16 whole cache lines,
//     4 lanes per line (2 x dwordx4 per lane) — 4x fewer L1 line look-ups per node visit;
//   * the four child boxes are tested in parallel, ordered by entry distance with quad DPP broadcasts
//     (no LDS, no sorting network on one lane), pushed far-to-near on a per-quad LDS stack;
//   * a leaf (<= 4 primitives) is tested in one step, one primitive per lane; the closest distance is a 2-step
//     quad min; hit attributes stay in the lane that found them until the ray is finished;
//   * a wave keeps a pool of 64 queue indices (one atomic per 64 rays) and refills idle quads from it, so lanes do
//     not idle behind the longest ray of the wave.
// Same conservative box test and the same primitive tests as tk_traverse.h: results are bit-identical to the
// per-lane traversal (and to the oracle) up to exact ties.
#pragma once

#include <hip/hip_runtime.h>

#include "tk_traverse.h"

namespace tk {

constexpr int TQ_BLOCK = 256;                 // 4 waves = 64 quads
constexpr int TQ_QUADS = TQ_BLOCK / 4;
constexpr int TQ_LEVELS = 32;                 // per-quad stack levels in LDS (8 B each: 17 KB per block)
constexpr int TQ_STRIDE = TQ_QUADS + 4;       // level stride in entries: 544 B = 32 (mod 128) -> conflict-free quads
constexpr int TQ_SPILL = 68;                  // deeper levels in global memory (per quad); builder caps depth at 96
constexpr int TQ_NODE_ITERS = 12;             // at most this many node steps before the next leaf phase / refill check
constexpr int TQ_NODE_MIN_QUADS = 6;          // leave the node phase when fewer quads than this are at interior nodes
constexpr int TQ_REFILL_MIN = 4;              // refill when at least this many of the 16 quads are idle
constexpr uint32_t TQ_KEY_INVALID = 0xFFFFFFFFu;

typedef unsigned long long tq_entry;          // low word: child word, high word: order key (float bits | lane)

// ---- quad cross-lane helpers (DPP quad_perm: no LDS traffic)
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
constexpr int QP_X2 = 0x4E;                                            // [2,3,0,1]
template <class R> __device__ __forceinline__ R quad_min(R v) {
    v = tk_fmin(v, dpp_f<QP_X1>(v));
    return tk_fmin(v, dpp_f<QP_X2>(v));
}
__device__ __forceinline__ int quad_max_i(int v) {
    v = max(v, dpp_i<QP_X1>(v));
    return max(v, dpp_i<QP_X2>(v));
}

struct QuadSpill {
    tq_entry *base;  // [level][global quad]
    int64_t stride;  // quads in the persistent grid
};
// The overflow path of the stack is kept out of line so that the common path compiles to plain ds_write_b64 /
// ds_read_b64 (a pointer select between LDS and global memory turns every access into a flat_* instruction).
__device__ __noinline__ void tq_spill_store(tq_entry *p, tq_entry e) { *p = e; }
__device__ __noinline__ tq_entry tq_spill_load(const tq_entry *p) { return *p; }

// IO policies: where rays come from and where results go.
//   count(), load(i, ray, tag): ray i of the work list;  store_*(tag, ...): called by ONE lane of the quad.
template <class R> struct PathIo {  // the render loop: rays in the path state, indexed through a queue
    PathState<R> st;
    const int32_t *queue;
    R eps;
    template <bool SHADOW> __device__ __forceinline__ void load(int32_t i, RayT<R> &ray, int64_t &tag) const {
        const int64_t slot = queue[i];
        tag = slot;
        if (!SHADOW)
            ray = make_ray(st.R_(S_OX, slot), st.R_(S_OY, slot), st.R_(S_OZ, slot), st.R_(S_DX, slot), st.R_(S_DY, slot),
                           st.R_(S_DZ, slot), eps, Const<R>::inf());
        else
            ray = make_ray(st.R_(S_OX, slot), st.R_(S_OY, slot), st.R_(S_OZ, slot), st.R_(S_SX, slot), st.R_(S_SY, slot),
                           st.R_(S_SZ, slot), eps, st.R_(S_ST, slot));
    }
    __device__ __forceinline__ void store_hit(int64_t slot, int32_t prim, int32_t, R t, R u, R v) const {
        st.I_(S_HIT, slot) = prim;
        st.R_(S_HT, slot) = t;
        st.R_(S_HU, slot) = u;
        st.R_(S_HV, slot) = v;
    }
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
    const RayAoS<R> *rays;
    HitAoS<R> *hits;
    int32_t *occluded;
    template <bool SHADOW> __device__ __forceinline__ void load(int32_t i, RayT<R> &ray, int64_t &tag) const {
        const RayAoS<R> q = rays[i];
        tag = i;
        ray = make_ray(q.org[0], q.org[1], q.org[2], q.dir[0], q.dir[1], q.dir[2], q.tmin, q.tmax);
    }
    __device__ __forceinline__ void store_hit(int64_t i, int32_t prim, int32_t shape, R t, R u, R v) const {
        HitAoS<R> h{};
        h.shape_id = shape;
        h.t = prim >= 0 ? t : R(0);
        h.u = u;
        h.v = v;
        hits[i] = h;
    }
    __device__ __forceinline__ void store_occlusion(int64_t i, bool occ) const { occluded[i] = occ ? 1 : 0; }
};

template <class R, bool ANY_HIT, bool COUNT, class Io>
__global__ void __launch_bounds__(TQ_BLOCK)
k_trace_quad(DeviceScene<R> sc, Io io, const int32_t *__restrict__ n_ptr, int32_t n_direct, int32_t *head,
             unsigned long long *counters, int counter_word, QuadSpill spill) {
    __shared__ tq_entry s_stack[TQ_LEVELS * TQ_STRIDE];
    const int lane = threadIdx.x & 63;
    const int ql = lane & 3;
    const int quad = threadIdx.x >> 2;
    tq_entry *const spl = spill.base + ((int64_t)blockIdx.x * TQ_QUADS + quad);
    const int32_t n = n_ptr ? *n_ptr : n_direct;
    if (blockIdx.x == 0 && threadIdx.x == 0 && counter_word >= 0)
        atomicAdd(&counters[counter_word], (unsigned long long)n);
    const char *const node_base = (const char *)sc.nodes;
    const char *const prim_base = (const char *)sc.prims;

    // wave-local pool of queue indices [pool_next, pool_end), refilled 64 at a time
    int32_t pool_next = 0, pool_end = 0;
    bool exhausted = false;
    // per-quad traversal state (identical in the 4 lanes unless noted)
    bool active = false;
    RayT<R> ray{};
    R idx = R(0), idy = R(0), idz = R(0), tbest = R(0);
    int64_t tag = 0;
    int sp = 0;
    int32_t cur = CHILD_EMPTY;
    // per-lane best candidate (differs between the lanes of a quad)
    R my_t = Const<R>::inf(), my_u = R(0), my_v = R(0);
    int32_t my_prim = -1, my_shape = -1;
    uint32_t cnt_nodes = 0, cnt_prims = 0, cnt_leaves = 0, cnt_wnode = 0, cnt_wleaf = 0;

    // Next subtree that can still hold a closer hit (entries whose entry distance is beyond the closest hit are
    // dropped).  Returns true when the stack is empty: the ray is finished.
    auto advance = [&]() -> bool {
        for (;;) {
            if (sp == 0) return true;
            --sp;
            const tq_entry e = (sp < TQ_LEVELS) ? s_stack[sp * TQ_STRIDE + quad]
                                                : tq_spill_load(spl + (int64_t)(sp - TQ_LEVELS) * spill.stride);
            cur = (int32_t)(uint32_t)e;
            const float key = __uint_as_float((uint32_t)(e >> 32) & ~3u);
            if ((R)key <= tbest) return false;
        }
    };
    // ONE lane of the quad writes the result
    auto finish = [&]() {
        if (ANY_HIT) {
            const bool occ = quad_max_i(my_prim) >= 0;
            if (ql == 0) io.store_occlusion(tag, occ);
        } else {
            // the lane holding the closest candidate writes it (highest lane on an exact tie)
            const int win = quad_max_i((my_prim >= 0 && my_t == tbest) ? ql : -1);
            if (win < 0) {
                if (ql == 0) io.store_hit(tag, -1, -1, ray.tmax, R(0), R(0));
            } else if (ql == win) {
                io.store_hit(tag, my_prim, my_shape, my_t, my_u, my_v);
            }
        }
        active = false;
    };

    for (;;) {
        // ------------------------------------------------------------------ refill idle quads from the pool
        {
            const uint64_t idle0 = __ballot(!active);
            const int n_idle = (int)(__popcll(idle0) >> 2);
            if (n_idle >= TQ_REFILL_MIN) {
#pragma unroll 1
                for (int pass = 0; pass < 2; ++pass) {
                    const uint64_t idle = __ballot(!active);
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
                    const int my_rank = (int)(__popcll(idle & ((1ull << (lane & ~3)) - 1ull)) >> 2);
                    if (!active && my_rank < avail) {
                        io.template load<ANY_HIT>(pool_next + my_rank, ray, tag);
                        idx = safe_inv(ray.d.x), idy = safe_inv(ray.d.y), idz = safe_inv(ray.d.z);
                        tbest = ray.tmax;
                        sp = 0;
                        cur = sc.root_child;
                        my_t = Const<R>::inf();
                        my_prim = -1;
                        my_shape = -1;
                        my_u = my_v = R(0);
                        active = true;
                    }
                    pool_next += min(avail, (int32_t)(__popcll(idle) >> 2));
                }
            }
        }
        if (__ballot(active) == 0) {
            if (exhausted) break;
            continue;
        }
        // ------------------------------------------------------------------ node phase
        // Interior-node steps only; a quad that reaches a leaf waits.  The leaf code (triangle / sphere tests) is
        // ~2x the node code, and with one ray per quad some quad of the wave is at a leaf in almost every step:
        // running the two in separate phases keeps the leaf code out of the node steps.
#pragma unroll 1
        for (int it = 0; it < TQ_NODE_ITERS; ++it) {
            const bool at_node = active && cur >= 0;
            const int n_node = (int)__popcll(__ballot(at_node));
            if (n_node == 0) break;
            // few quads left at interior nodes and some waiting at a leaf: switch to the leaf phase
            if (n_node < 4 * TQ_NODE_MIN_QUADS && __ballot(active && cur < 0) != 0) break;
            if (COUNT && lane == 0) cnt_wnode++;
            if (at_node) {
                // one child slot per lane: 32-bit byte offset from the (scalar) node base
                const uint32_t off = (uint32_t)cur * (uint32_t)sizeof(Node4<R>) + (uint32_t)ql * (uint32_t)sizeof(NodeChild<R>);
                const NodeChild<R> c = *(const NodeChild<R> *)(node_base + off);
                if (COUNT && ql == 0) cnt_nodes++;
                R tn;
                const bool ok = box_test(c, ray.o, idx, idy, idz, ray.tmin, tbest, tn);
                // order key: the (shrunk, hence conservative) entry distance as an integer — non-negative floats
                // order like their bit patterns — with the lane in the two low bits to make the four keys distinct
                const uint32_t key =
                    ok ? ((__float_as_uint(stack_key(tn * Const<R>::BOX_SHRINK)) & ~3u) | (uint32_t)ql) : TQ_KEY_INVALID;
                const uint32_t k0 = (uint32_t)dpp_i<QP_B0>((int)key), k1 = (uint32_t)dpp_i<QP_B1>((int)key);
                const uint32_t k2 = (uint32_t)dpp_i<QP_B2>((int)key), k3 = (uint32_t)dpp_i<QP_B3>((int)key);
                const int rank = (int)(k0 < key) + (int)(k1 < key) + (int)(k2 < key) + (int)(k3 < key);
                const int nhit = (int)(k0 != TQ_KEY_INVALID) + (int)(k1 != TQ_KEY_INVALID) + (int)(k2 != TQ_KEY_INVALID) +
                                 (int)(k3 != TQ_KEY_INVALID);
                // far-to-near: the nearest child ends on top of the stack and is popped right below
                if (ok) {
                    const int level = sp + nhit - 1 - rank;
                    const tq_entry e = ((tq_entry)key << 32) | (tq_entry)(uint32_t)c.child;
                    if (level < TQ_LEVELS)
                        s_stack[level * TQ_STRIDE + quad] = e;
                    else
                        tq_spill_store(spl + (int64_t)(level - TQ_LEVELS) * spill.stride, e);
                }
                sp += nhit;
                if (advance()) finish();
            }
        }
        // ------------------------------------------------------------------ leaf phase: one primitive per lane
        {
            const bool at_leaf = active && cur < 0 && cur != CHILD_EMPTY;
            if (COUNT && lane == 0 && __ballot(at_leaf) != 0) cnt_wleaf++;
            if (at_leaf) {
                const int first = leaf_first(cur), cnt = leaf_count(cur);
                if (COUNT && ql == 0) cnt_prims += (uint32_t)cnt, cnt_leaves++;
                if (ql < cnt) {
                    const uint32_t off = (uint32_t)(first + ql) * (uint32_t)sizeof(PrimRec<R>);
                    const PrimRec<R> p = *(const PrimRec<R> *)(prim_base + off);
                    R t, u = R(0), v = R(0);
                    const bool ok = ((p.meta & 0xff) == PRIM_TRIANGLE) ? tri_test(p.a, ray, tbest, t, u, v)
                                                                       : sphere_test(p.a, ray, tbest, t);
                    if (ok) {
                        my_t = t, my_u = u, my_v = v;
                        my_prim = first + ql;
                        my_shape = p.shape_id;
                    }
                }
                tbest = tk_fmin(tbest, quad_min(my_t));
                bool finished = ANY_HIT ? (quad_max_i(my_prim) >= 0) : false;
                if (!finished) finished = advance();
                if (finished) finish();
            } else if (active && cur == CHILD_EMPTY) {
                if (advance()) finish();  // empty scene: nothing to test
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
        }
    }
}

}  // namespace tk
