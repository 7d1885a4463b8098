// tk_build_gpu.h — BVH construction on the device (f32 scenes): the GPU counterpart of `construct_bvh`
// (src/bvh.cpp:8-45) for scenes where the host SAH build is the wait (10M triangles: ~6 s on 16 cores).
//
//   k_prim_boxes     primitive AABBs (of the geometry the intersection tests see: v0, v0+e1, v0+e2 — one ulp wider)
//                    + scene bounds (wave reduce, ordered-int atomics)
//   k_morton         63-bit Morton code of the box centre (21 bits per axis: 30 bits leave whole clusters of a 10M-
//                    triangle scene in one cell); rocPRIM radix sort of (code, primitive) pairs
//   k_leaves         leaves = runs of `leaf_size` consecutive primitives in Morton order
//   k_hierarchy      Karras 2012: every internal node finds its key range and split in parallel
//   k_refit          bottom-up boxes, second arrival at a node computes it (agent-scope fences around the flag)
//   k_collapse       BVH2 -> 4-wide nodes, one launch per tree level, breadth-first numbering (same layout and the
//                    same "open the child with the largest area" rule as the host collapse, tk_bvh.h)
//   k_quantise       64-byte compressed nodes on the 15-bit scene grid (same rounding rules as quantise_nodes)
//   k_permute        primitive and shading records into leaf order
//
// The tree is an LBVH: built in milliseconds, but without the SAH its boxes overlap more, so traversal visits more
// nodes than with the host build (numbers in DESIGN.md).  Results do not depend on the tree (conservative box
// tests): the parity tests require bit-identical hit tables and images for both builders.
#pragma once

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <cmath>
#include <limits>

#include "tk_bvh.h"
#include "tk_scene.h"

namespace tk {
namespace lbvh {

constexpr int BLK = 256;
constexpr int MAX_LEVELS = (MAX_STACK_ENTRIES - 1) / 3;  // 4-wide levels the traversal stack can take

struct Box {
    float lo[3], hi[3];
};

__device__ __forceinline__ float f_below(float x) {  // the float just below x (x finite)
    uint32_t u = __float_as_uint(x);
    u = x > 0.0f ? u - 1u : (x < 0.0f ? u + 1u : 0x80000001u);
    return __uint_as_float(u);
}
__device__ __forceinline__ float f_above(float x) {
    uint32_t u = __float_as_uint(x);
    u = x > 0.0f ? u + 1u : (x < 0.0f ? u - 1u : 0x00000001u);
    return __uint_as_float(u);
}
// floats as integers with the same order (for atomicMin / atomicMax)
__host__ __device__ __forceinline__ int f2ord(float f) {
    int i;
    __builtin_memcpy(&i, &f, 4);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__host__ __device__ __forceinline__ float ord2f(int i) {
    i = i >= 0 ? i : i ^ 0x7fffffff;
    float f;
    __builtin_memcpy(&f, &i, 4);
    return f;
}
__device__ __forceinline__ Box box_union(const Box &a, const Box &b) {
    Box r;
#pragma unroll
    for (int k = 0; k < 3; k++) r.lo[k] = fminf(a.lo[k], b.lo[k]), r.hi[k] = fmaxf(a.hi[k], b.hi[k]);
    return r;
}
__device__ __forceinline__ float half_area(const Box &b) {
    const float x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
    return x * y + y * z + z * x;
}

// ---- primitive records straight from the caller's mesh arrays (parser -> device buffers).
// The same arithmetic as the host loop of tk_host_scene.h (positions rounded to float, e_k = v_k - v0 in float, no
// contraction), so the records — and with them every hit — are bit-identical to the host-built ones.
struct MeshSrc {
    int64_t pos_off;   // first vertex of this mesh in the concatenated double positions (units: vertices)
    int32_t fbase;     // first face in face_idx (units: faces)
    int32_t material;
    int32_t tag;       // material tag
    int32_t has_attr;  // vertex normals and/or uvs exist
};
struct SphereSrc {
    double c[3], r;
    int32_t material, tag;
};
__global__ void __launch_bounds__(BLK)
k_make_prims(const int32_t *__restrict__ kind, const int32_t *__restrict__ ref, const int32_t *__restrict__ face,
             const int32_t *__restrict__ area_light, const MeshSrc *__restrict__ meshes, const double *__restrict__ positions,
             const int32_t *__restrict__ face_idx, const SphereSrc *__restrict__ spheres, int n, PrimRec<float> *out) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= n) return;
    PrimRec<float> p{};
    p.shape_id = i;
    p.area_light = area_light[i];
    p.nidx = -1;
    if (kind[i] == 0) {
        const SphereSrc s = spheres[ref[i]];
        p.a[0] = (float)s.c[0], p.a[1] = (float)s.c[1], p.a[2] = (float)s.c[2], p.a[3] = (float)s.r;
        p.meta = PRIM_SPHERE | (s.tag << 8);
        p.material = s.material;
        p.mesh = -(1 + ref[i]);
    } else {
        const MeshSrc m = meshes[ref[i]];
        const int32_t *idx = face_idx + 3 * ((int64_t)m.fbase + face[i]);
        float v[3][3];
        for (int k = 0; k < 3; k++)
            for (int a = 0; a < 3; a++) v[k][a] = (float)positions[3 * (m.pos_off + idx[k]) + a];
        for (int a = 0; a < 3; a++) p.a[a] = v[0][a], p.a[3 + a] = v[1][a] - v[0][a], p.a[6 + a] = v[2][a] - v[0][a];
        p.meta = PRIM_TRIANGLE | (m.tag << 8);
        p.material = m.material;
        p.mesh = ref[i];
        if (m.has_attr) p.nidx = m.fbase + face[i], p.meta |= META_HAS_ATTR;
    }
    out[i] = p;
}

// scene_ord[0..2] = min of lo (ordered ints), [3..5] = max of hi; initialised to INT_MAX / INT_MIN by the caller
__global__ void __launch_bounds__(BLK) k_prim_boxes(const PrimRec<float> *__restrict__ prims, int n, Box *pb, int *scene_ord) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    Box b;
#pragma unroll
    for (int k = 0; k < 3; k++) b.lo[k] = __builtin_huge_valf(), b.hi[k] = -__builtin_huge_valf();
    if (i < n) {
        const PrimRec<float> p = prims[i];
        if ((p.meta & 0xff) == PRIM_TRIANGLE) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float v0 = p.a[k], v1 = p.a[k] + p.a[3 + k], v2 = p.a[k] + p.a[6 + k];
                b.lo[k] = f_below(fminf(v0, fminf(v1, v2)));  // the sums round by at most half an ulp
                b.hi[k] = f_above(fmaxf(v0, fmaxf(v1, v2)));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; k++) b.lo[k] = f_below(p.a[k] - p.a[3]), b.hi[k] = f_above(p.a[k] + p.a[3]);
        }
        pb[i] = b;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float lo = b.lo[k], hi = b.hi[k];
        for (int off = 32; off > 0; off >>= 1) lo = fminf(lo, __shfl_xor(lo, off)), hi = fmaxf(hi, __shfl_xor(hi, off));
        if ((threadIdx.x & 63) == 0 && lo <= hi) {
            atomicMin(&scene_ord[k], f2ord(lo));
            atomicMax(&scene_ord[3 + k], f2ord(hi));
        }
    }
}

__device__ __forceinline__ uint64_t expand21(uint64_t v) {  // 21 bits -> every third bit
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x001F00000000FFFFull;
    v = (v | (v << 16)) & 0x001F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
__global__ void __launch_bounds__(BLK) k_morton(const Box *__restrict__ pb, int n, const int *__restrict__ scene_ord, uint64_t *keys, uint32_t *vals) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= n) return;
    const Box b = pb[i];
    uint64_t q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float lo = ord2f(scene_ord[k]), hi = ord2f(scene_ord[3 + k]);
        const float ext = hi - lo;
        const float t = ext > 0.0f ? (0.5f * (b.lo[k] + b.hi[k]) - lo) / ext : 0.0f;
        q[k] = (uint64_t)fminf(fmaxf(t * 2097152.0f, 0.0f), 2097151.0f);
    }
    keys[i] = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
    vals[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(BLK) k_leaves(const Box *__restrict__ pb, const uint64_t *__restrict__ keys_sorted, const uint32_t *__restrict__ vals_sorted,
                                                 int n, int leaf_size, int n_leaves, Box *lbox, uint64_t *lkey) {
    const int l = blockIdx.x * BLK + threadIdx.x;
    if (l >= n_leaves) return;
    const int first = l * leaf_size, cnt = min(leaf_size, n - first);
    Box b = pb[vals_sorted[first]];
    for (int k = 1; k < cnt; k++) b = box_union(b, pb[vals_sorted[first + k]]);
    lbox[l] = b;
    lkey[l] = keys_sorted[first];
}

// common-prefix length of leaf keys i and j (ties broken by the leaf index), -1 outside the array
__device__ __forceinline__ int prefix_len(const uint64_t *__restrict__ k, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = k[i], b = k[j];
    return a == b ? 64 + __clz((uint32_t)(i ^ j)) : __clzll((long long)(a ^ b));
}
// child reference: >= 0 internal node, < 0 leaf ~l.  parent_i[0] = -1 (root = internal node 0).
__global__ void __launch_bounds__(BLK) k_hierarchy(const uint64_t *__restrict__ lkey, int n_leaves, int2 *child, int *parent_i, int *parent_l) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i >= n_leaves - 1) return;
    const int d = prefix_len(lkey, n_leaves, i, i + 1) - prefix_len(lkey, n_leaves, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = prefix_len(lkey, n_leaves, i, i - d);
    int lmax = 2;
    while (prefix_len(lkey, n_leaves, i, i + lmax * d) > dmin) lmax *= 2;
    int len = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (prefix_len(lkey, n_leaves, i, i + (len + t) * d) > dmin) len += t;
    const int j = i + len * d;
    const int dnode = prefix_len(lkey, n_leaves, i, j);
    int s = 0, t = len;
    do {
        t = (t + 1) >> 1;
        if (prefix_len(lkey, n_leaves, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int left = lo == gamma ? ~gamma : gamma, right = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    child[i] = make_int2(left, right);
    if (left >= 0) parent_i[left] = i; else parent_l[~left] = i;
    if (right >= 0) parent_i[right] = i; else parent_l[~right] = i;
    if (i == 0) parent_i[0] = -1;
}

// flag[] zeroed by the caller.  A box written on one CU is read on another: release before the flag, acquire after
// it (a CU's vector L1 is not refreshed by other CUs' stores, MI355X_MICROARCH.md).
__global__ void __launch_bounds__(BLK) k_refit(int n_leaves, const int2 *__restrict__ child, const int *__restrict__ parent_i, const int *__restrict__ parent_l,
                                                const Box *lbox, Box *ibox, int *flag) {
    const int l = blockIdx.x * BLK + threadIdx.x;
    if (l >= n_leaves) return;
    int p = parent_l[l];
    while (p >= 0) {
        __threadfence();
        if (atomicAdd(&flag[p], 1) == 0) return;  // the sibling subtree is not done: its last thread will come by
        __threadfence();
        const int2 c = child[p];
        const Box a = c.x < 0 ? lbox[~c.x] : ibox[c.x], b = c.y < 0 ? lbox[~c.y] : ibox[c.y];
        ibox[p] = box_union(a, b);
        p = parent_i[p];
    }
}

// One tree level of the collapse.  lvl[k] = number of wide nodes on level k (lvl[0] = 1 set by the caller);
// node index = (nodes on earlier levels) + position in the level's frontier; frontier entries are BVH2 node ids.
__global__ void __launch_bounds__(BLK) k_collapse(int level, const int *__restrict__ frontier_in, int *frontier_out, int *lvl, const int2 *__restrict__ child,
                                                   const Box *__restrict__ ibox, const Box *__restrict__ lbox, int leaf_size, int n_prims, Node4<float> *nodes) {
    int off_k = 0;
    for (int j = 0; j < level; j++) off_k += lvl[j];
    const int n_in = lvl[level], off_k1 = off_k + n_in;
    const int lane = threadIdx.x & 63;
    const int n_round = (n_in + 63) / 64 * 64;
    for (int idx = blockIdx.x * BLK + threadIdx.x; idx < n_round; idx += gridDim.x * BLK) {
        const bool valid = idx < n_in;
        int kids[4] = {0, 0, 0, 0}, nk = 0, n_int = 0;
        if (valid) {
            const int2 c = child[frontier_in[idx]];
            kids[0] = c.x, kids[1] = c.y, nk = 2;
            while (nk < 4) {  // open the interior child with the largest surface area
                int best = -1;
                float best_area = -1.0f;
                for (int i = 0; i < nk; i++)
                    if (kids[i] >= 0) {
                        const float a = half_area(ibox[kids[i]]);
                        if (a > best_area) best_area = a, best = i;
                    }
                if (best < 0) break;
                const int2 g = child[kids[best]];
                kids[best] = g.x;
                kids[nk++] = g.y;
            }
            for (int i = 0; i < nk; i++) n_int += kids[i] >= 0;
            // slot order = visiting order of the (unranked) shadow-ray traversal: largest box first (tk_bvh.h)
            float ar[4];
            for (int i = 0; i < nk; i++) ar[i] = half_area(kids[i] >= 0 ? ibox[kids[i]] : lbox[~kids[i]]);
            for (int i = 1; i < nk; i++)
                for (int j = i; j > 0 && ar[j] > ar[j - 1]; j--) {
                    const float ta = ar[j];
                    ar[j] = ar[j - 1], ar[j - 1] = ta;
                    const int tk = kids[j];
                    kids[j] = kids[j - 1], kids[j - 1] = tk;
                }
        }
        // wave-aggregated allocation of the interior children on the next level
        int incl = n_int;
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        const int total = __shfl(incl, 63);
        int base = 0;
        if (lane == 0 && total) base = atomicAdd(&lvl[level + 1], total);
        base = __shfl(base, 0);
        if (valid) {
            int next = base + incl - n_int;
            Node4<float> nd;
            for (int i = 0; i < 4; i++) {
                NodeChild<float> &o = nd.c[i];
                o.pad = 0;
                if (i < nk) {
                    const int k = kids[i];
                    const Box b = k >= 0 ? ibox[k] : lbox[~k];
                    for (int a = 0; a < 3; a++) o.bmin[a] = b.lo[a], o.bmax[a] = b.hi[a];
                    if (k >= 0) {
                        frontier_out[next] = k;
                        o.child = off_k1 + next;
                        next++;
                    } else {
                        const int first = (~k) * leaf_size;
                        o.child = make_leaf(first, min(leaf_size, n_prims - first));
                    }
                } else {
                    for (int a = 0; a < 3; a++) o.bmin[a] = __builtin_huge_valf(), o.bmax[a] = -__builtin_huge_valf();
                    o.child = CHILD_EMPTY;
                }
            }
            nodes[off_k + idx] = nd;
        }
    }
}

// acc[0] += sum of min(decoded area / true area, 100) over child boxes, acc[1] += number of child boxes
__global__ void __launch_bounds__(BLK) k_quantise(const Node4<float> *__restrict__ nodes, int n, QGrid g, QNode4 *out, double *acc) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    double ratio = 0, slots = 0;
    if (i < n) {
        const Node4<float> nd = nodes[i];
        QNode4 q;
        for (int c = 0; c < 4; c++) {
            q.c[c].child = nd.c[c].child;
            q.c[c].q[0] = q.c[c].q[1] = q.c[c].q[2] = (uint32_t)Q_MAX;  // empty slot: inverted box (tk_bvh.h: quantise_nodes)
            if (nd.c[c].child == CHILD_EMPTY) continue;
            double et[3], eq[3];
            for (int a = 0; a < 3; a++) {
                long long ql, qh;
                qgrid_snap(g, a, (double)nd.c[c].bmin[a] - g.delta[a], (double)nd.c[c].bmax[a] + g.delta[a], ql, qh);
                q.c[c].q[a] = (uint32_t)ql | ((uint32_t)qh << 16);
                et[a] = (double)nd.c[c].bmax[a] - (double)nd.c[c].bmin[a];
                eq[a] = (double)(qh - ql) * (double)g.step[a];
            }
            const double at = et[0] * et[1] + et[1] * et[2] + et[2] * et[0], aq = eq[0] * eq[1] + eq[1] * eq[2] + eq[2] * eq[0];
            ratio += at > 0 ? fmin(aq / at, 100.0) : (aq > 0 ? 100.0 : 1.0);
            slots += 1;
        }
        out[i] = q;
    }
    for (int off = 32; off > 0; off >>= 1) ratio += __shfl_xor(ratio, off), slots += __shfl_xor(slots, off);
    if ((threadIdx.x & 63) == 0 && slots > 0) {
        atomicAdd(&acc[0], ratio);
        atomicAdd(&acc[1], slots);
    }
}

template <class T>
__global__ void __launch_bounds__(BLK) k_permute(const T *__restrict__ in, const uint32_t *__restrict__ order, int n, T *out) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i < n) out[i] = in[order[i]];
}
__global__ void k_fill_int(int *p, int n, int v) {
    const int i = blockIdx.x * BLK + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace lbvh
}  // namespace tk
