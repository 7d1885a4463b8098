// tk_bvh.h — host-side acceleration-structure build: binned-SAH BVH2 -> 4-wide BVH in breadth-first order.
//
// Replaces the reference's construct_bvh (src/bvh.cpp:8-45: recursive median split on the largest axis, one
// primitive per leaf, a full copy of the box vector per level, O(N log^2 N)).  Closest-hit results do not depend
// on the tree (tk_traverse.h keeps the box test conservative), so the tree is chosen for the GPU:
//   * surface-area heuristic over 16 centroid bins on all three axes, leaves of up to MAX_LEAF primitives;
//   * collapsed to 4-wide nodes by repeatedly opening the child with the largest surface area;
//   * nodes emitted breadth-first (top levels = array prefix), primitives re-ordered into leaf order;
//   * large subtrees are built by std::async tasks over disjoint index ranges.
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <future>
#include <limits>
#include <thread>
#include <vector>

#include "tk_scene.h"

namespace tk {

struct BuildPrim {
    double bmin[3], bmax[3];
    int32_t id;  // >= 0: index of the primitive record; < 0: instance -(1 + id) of a two-level scene (always a leaf of its own)
};
struct Bvh2Node {
    double bmin[3], bmax[3];
    int32_t left, right;   // interior
    int32_t first, count;  // leaf when count > 0 (range in the permuted primitive order)
};

struct Bounds {
    double lo[3], hi[3];
    Bounds() {
        for (int a = 0; a < 3; a++) lo[a] = std::numeric_limits<double>::infinity(), hi[a] = -lo[a];
    }
    void grow(const double *mn, const double *mx) {
        for (int a = 0; a < 3; a++) lo[a] = std::min(lo[a], mn[a]), hi[a] = std::max(hi[a], mx[a]);
    }
    void grow_pt(const double *p) { grow(p, p); }
    double half_area() const {
        double e0 = hi[0] - lo[0], e1 = hi[1] - lo[1], e2 = hi[2] - lo[2];
        if (e0 < 0) return 0.0;
        return e0 * e1 + e1 * e2 + e2 * e0;
    }
};

class Bvh2Builder {
  public:
    Bvh2Builder(std::vector<BuildPrim> &prims, int max_leaf, int threads)
        : prims_(prims), max_leaf_(std::max(1, std::min(max_leaf, (int)MAX_LEAF))), threads_(std::max(1, threads)) {
        nodes_.resize(std::max<size_t>(1, 2 * prims.size()));
        next_.store(0);
    }
    // returns root index; nodes() valid afterwards
    int build() {
        if (prims_.empty()) return -1;
        int root = alloc();
        int spawn_depth = 0;
        while ((1 << spawn_depth) < threads_ * 4) spawn_depth++;
        build_range(root, 0, (int)prims_.size(), threads_ > 1 ? spawn_depth : 0);
        nodes_.resize(next_.load());
        return root;
    }
    std::vector<Bvh2Node> &nodes() { return nodes_; }

  private:
    static constexpr int BINS = 16;
    std::vector<BuildPrim> &prims_;
    std::vector<Bvh2Node> nodes_;
    std::atomic<int> next_;
    int max_leaf_, threads_;

    int alloc() { return next_.fetch_add(1); }

    void build_range(int node, int lo, int hi, int spawn) {
        Bounds b, cb;
        for (int i = lo; i < hi; i++) {
            b.grow(prims_[i].bmin, prims_[i].bmax);
            double c[3];
            for (int a = 0; a < 3; a++) c[a] = 0.5 * (prims_[i].bmin[a] + prims_[i].bmax[a]);
            cb.grow_pt(c);
        }
        Bvh2Node &n = nodes_[node];
        for (int a = 0; a < 3; a++) n.bmin[a] = b.lo[a], n.bmax[a] = b.hi[a];
        n.left = n.right = -1;
        n.first = lo;
        n.count = 0;
        const int cnt = hi - lo;
        if (cnt == 1) {
            n.count = 1;
            return;
        }
        // binned SAH over the three axes
        double best_cost = std::numeric_limits<double>::infinity();
        int best_axis = -1, best_split = -1;
        for (int a = 0; a < 3; a++) {
            const double ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0)) continue;
            Bounds bb[BINS];
            int bc[BINS] = {0};
            const double scale = BINS / ext;
            for (int i = lo; i < hi; i++) {
                double c = 0.5 * (prims_[i].bmin[a] + prims_[i].bmax[a]);
                int k = std::min(BINS - 1, std::max(0, (int)((c - cb.lo[a]) * scale)));
                bb[k].grow(prims_[i].bmin, prims_[i].bmax);
                bc[k]++;
            }
            double right_area[BINS];
            int right_cnt[BINS];
            Bounds acc;
            int c = 0;
            for (int k = BINS - 1; k >= 1; k--) {
                if (bc[k]) acc.grow(bb[k].lo, bb[k].hi);
                c += bc[k];
                right_area[k] = acc.half_area();
                right_cnt[k] = c;
            }
            Bounds accl;
            int cl = 0;
            for (int k = 0; k < BINS - 1; k++) {
                if (bc[k]) accl.grow(bb[k].lo, bb[k].hi);
                cl += bc[k];
                if (cl == 0 || right_cnt[k + 1] == 0) continue;
                // leaves hold up to max_leaf primitives: cost in units of leaf visits keeps leaves full
                double cost = accl.half_area() * std::ceil(cl / (double)max_leaf_) +
                              right_area[k + 1] * std::ceil(right_cnt[k + 1] / (double)max_leaf_);
                if (cost < best_cost) best_cost = cost, best_axis = a, best_split = k;
            }
        }
        bool has_instance = false;
        if (cnt <= max_leaf_)
            for (int i = lo; i < hi; i++) has_instance = has_instance || prims_[i].id < 0;
        if (cnt <= max_leaf_ && !has_instance) {
            // make a leaf unless splitting is clearly cheaper (one node test + two smaller leaves)
            const double leaf_cost = b.half_area() * 1.0;
            const double split_cost = best_axis < 0 ? std::numeric_limits<double>::infinity()
                                                    : 0.5 * b.half_area() + best_cost;
            if (split_cost >= leaf_cost) {
                n.count = cnt;
                return;
            }
        }
        int mid;
        if (best_axis < 0) {
            mid = lo + cnt / 2;  // all centroids coincide: split by index
        } else {
            const double ext = cb.hi[best_axis] - cb.lo[best_axis];
            const double scale = BINS / ext;
            const double clo = cb.lo[best_axis];
            const int a = best_axis, ks = best_split;
            auto it = std::partition(prims_.begin() + lo, prims_.begin() + hi, [&](const BuildPrim &p) {
                double c = 0.5 * (p.bmin[a] + p.bmax[a]);
                int k = std::min(BINS - 1, std::max(0, (int)((c - clo) * scale)));
                return k <= ks;
            });
            mid = (int)(it - prims_.begin());
            if (mid == lo || mid == hi) mid = lo + cnt / 2;
        }
        const int l = alloc(), r = alloc();
        nodes_[node].left = l;
        nodes_[node].right = r;
        if (spawn > 0 && cnt > 8192) {
            auto fut = std::async(std::launch::async, [=]() { build_range(l, lo, mid, spawn - 1); });
            build_range(r, mid, hi, spawn - 1);
            fut.get();
        } else {
            build_range(l, lo, mid, 0);
            build_range(r, mid, hi, 0);
        }
    }
};

// outward rounding of a double box coordinate to R
inline float round_down(double x, float) {
    float f = (float)x;
    if ((double)f > x) f = std::nextafter(f, -std::numeric_limits<float>::infinity());
    return f;
}
inline float round_up(double x, float) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafter(f, std::numeric_limits<float>::infinity());
    return f;
}
inline double round_down(double x, double) { return x; }
inline double round_up(double x, double) { return x; }

struct WideBvhStats {
    int64_t n_nodes = 0, n_prims = 0;
    int depth = 0;
    double sah = 0;  // expected (node fetches, primitive tests) per random ray, for DESIGN figures
};

// Collapse a BVH2 into W-wide records (breadth-first) and the leaf-ordered primitive permutation.
// prim_order[i] = index into the BuildPrim array (post-build order) of the i-th primitive in leaf order.
// `bprims` (the builder's primitive array, post-build order): needed only for two-level scenes, to turn the
// single-primitive leaf of an instance into an instance word.
// Slot order.  W = 4: largest box first = the visiting order of the 4-wide shadow-ray kernel, which does not rank
// its children (any occluder ends the ray): a shadow ray dives into the child it is most likely to be stopped in
// (1M soup, host model of the kernel's order: 34.2 -> 17.3 node steps per shadow ray); closest-hit traversal ranks
// by entry distance and does not care.  W = 8: slot s = the octant of the node the child lies in (bit a of s: high
// side of axis a), by a greedy auction on sum_a (+-)(child centre - node centre)_a, so that a ray visits the
// slots in the order s ^ (octant of its direction) roughly front to back (tk_scene.h: QNode8).
template <class R, int W>
int32_t collapse_to_wide(const std::vector<Bvh2Node> &n2, int root, std::vector<NodeW<R, W>> &out,
                         std::vector<int32_t> &prim_order, WideBvhStats &stats, const std::vector<BuildPrim> *bprims = nullptr) {
    out.clear();
    prim_order.clear();
    stats = WideBvhStats{};
    if (root < 0) return CHILD_EMPTY;
    auto emit_leaf = [&](const Bvh2Node &n) {
        if (bprims && n.count == 1 && (*bprims)[n.first].id < 0) return make_instance_word(-(*bprims)[n.first].id - 1);
        int32_t first = (int32_t)prim_order.size();
        for (int i = 0; i < n.count; i++) prim_order.push_back(n.first + i);
        return make_leaf(first, n.count);
    };
    if (n2[root].count > 0) {
        stats.n_prims = n2[root].count;
        stats.depth = 0;
        return emit_leaf(n2[root]);
    }
    struct Item {
        int n2;
        int depth;
    };
    std::vector<Item> queue;
    queue.push_back({root, 1});
    out.emplace_back();
    size_t head = 0;
    while (head < queue.size()) {
        const Item it = queue[head];
        const size_t self = head++;
        int kids[W];
        int nk = 0;
        kids[nk++] = n2[it.n2].left;
        kids[nk++] = n2[it.n2].right;
        while (nk < W) {
            int best = -1;
            double best_area = -1;
            for (int i = 0; i < nk; i++) {
                const Bvh2Node &c = n2[kids[i]];
                if (c.count > 0) continue;
                Bounds b;
                b.grow(c.bmin, c.bmax);
                double a = b.half_area();
                if (a > best_area) best_area = a, best = i;
            }
            if (best < 0) break;
            const int open = kids[best];
            kids[best] = n2[open].left;
            kids[nk++] = n2[open].right;
        }
        int slot_kid[W];  // the child of each slot, -1 = empty
        for (int i = 0; i < W; i++) slot_kid[i] = -1;
        if (W == 4) {
            double ar[W];
            for (int i = 0; i < nk; i++) {
                Bounds b;
                b.grow(n2[kids[i]].bmin, n2[kids[i]].bmax);
                ar[i] = b.half_area();
            }
            for (int i = 1; i < nk; i++)
                for (int j = i; j > 0 && ar[j] > ar[j - 1]; j--) std::swap(ar[j], ar[j - 1]), std::swap(kids[j], kids[j - 1]);
            for (int i = 0; i < nk; i++) slot_kid[i] = kids[i];
        } else {
            const Bvh2Node &pn = n2[it.n2];
            double gain[W][W];
            for (int i = 0; i < nk; i++) {
                const Bvh2Node &c = n2[kids[i]];
                for (int s = 0; s < W; s++) {
                    double v = 0;
                    for (int a = 0; a < 3; a++) {
                        const double d = 0.5 * (c.bmin[a] + c.bmax[a]) - 0.5 * (pn.bmin[a] + pn.bmax[a]);
                        v += ((s >> a) & 1) ? d : -d;
                    }
                    gain[i][s] = v;
                }
            }
            bool kid_used[W] = {false}, slot_used[W] = {false};
            for (int r = 0; r < nk; r++) {
                int bi = -1, bs = -1;
                double bv = -std::numeric_limits<double>::infinity();
                for (int i = 0; i < nk; i++)
                    if (!kid_used[i])
                        for (int s = 0; s < W; s++)
                            if (!slot_used[s] && gain[i][s] > bv) bv = gain[i][s], bi = i, bs = s;
                if (bi < 0) {  // (NaN boxes: any free pair)
                    for (int i = 0; i < nk && bi < 0; i++)
                        if (!kid_used[i]) bi = i;
                    for (int s = 0; s < W && bs < 0; s++)
                        if (!slot_used[s]) bs = s;
                }
                kid_used[bi] = true, slot_used[bs] = true, slot_kid[bs] = kids[bi];
            }
        }
        NodeW<R, W> node;
        for (int i = 0; i < W; i++) {
            node.c[i].pad = 0;
            if (slot_kid[i] >= 0) {
                const Bvh2Node &c = n2[slot_kid[i]];
                for (int a = 0; a < 3; a++) {
                    node.c[i].bmin[a] = round_down(c.bmin[a], R());
                    node.c[i].bmax[a] = round_up(c.bmax[a], R());
                }
                if (c.count > 0) {
                    node.c[i].child = emit_leaf(c);
                } else {
                    node.c[i].child = (int32_t)queue.size();
                    queue.push_back({slot_kid[i], it.depth + 1});
                    out.emplace_back();
                }
            } else {
                for (int a = 0; a < 3; a++) node.c[i].bmin[a] = Const<R>::inf(), node.c[i].bmax[a] = -Const<R>::inf();
                node.c[i].child = CHILD_EMPTY;
            }
        }
        out[self] = node;
        stats.depth = std::max(stats.depth, it.depth);
    }
    stats.n_nodes = (int64_t)out.size();
    stats.n_prims = (int64_t)prim_order.size();
    return 0;  // root node index
}

// Compressed copy of the nodes (QNode4, tk_scene.h) on one 15-bit grid over the union of all child boxes.  (f64
// scenes traverse the same compressed nodes: a conservative box test may be done in any precision — only the
// primitive tests decide a hit, and those stay in double.)
// Per axis: step = extent / Q_MAX rounded up to a float with slack, every child plane is moved outwards by
// delta = Q_MAX * step * 2^-20 and then snapped outwards to the grid; plane(q) = grid_lo + (Q_BIAS + q) * step, i.e.
// grid_lo is one grid extent below the lowest plane (the bias makes the device's float image of q exact, tk_scene.h).
// delta pays for the rounding of the grid-space slab test (tk_traverse.h: qray_make: <= 0.75 * 2^-21 extent for
// origins within two extents of the scene); the checks below are on exact values (a float plus a 17-bit multiple of
// a float is exact in double to ~1e-16 relative, nothing next to delta).
// Returns the mean over all child boxes of (decoded half-area / true half-area), each ratio capped at 100: how much
// more often a box is entered by the rays that reach its parent.  (Not weighted by absolute area: a cluster of
// small primitives inside a huge scene is exactly where the grid is too coarse, and where the camera usually
// looks.)  The caller keeps the full-width nodes when it is large.
// the grid over [lo_in, hi_in] (shared by the host quantiser and the device one, tk_build_gpu.h)
inline QGrid make_qgrid(const double lo_in[3], const double hi_in[3]) {
    QGrid g;
    for (int a = 0; a < 3; a++) {
        double lo = lo_in[a], hi = hi_in[a];
        if (!(lo <= hi)) lo = hi = 0.0;
        double ext = hi - lo;
        if (!(ext > 0)) ext = std::max(std::fabs(lo), 1.0) * 1e-6;  // flat scene on this axis: any small grid will do
        float step = (float)(ext * (1.0 + 1e-5) / (double)Q_MAX);
        float p = 0;
        for (;; step = std::nextafterf(step * 1.0001f, std::numeric_limits<float>::infinity())) {
            g.delta[a] = (double)Q_MAX * (double)step * 0x1p-20;
            const double x = lo - g.delta[a] - (double)Q_BIAS * (double)step;
            p = (float)x;
            if ((double)p > x) p = std::nextafterf(p, -std::numeric_limits<float>::infinity());
            if ((double)p + (double)(Q_BIAS + Q_MAX) * (double)step >= hi + g.delta[a]) break;
        }
        g.lo[a] = p, g.step[a] = step;
    }
    return g;
}
template <class R, int W>
inline double quantise_nodes(const std::vector<NodeW<R, W>> &in, std::vector<QNodeW<W>> &out, float grid_lo[3],
                             float grid_step[3]) {
    out.assign(in.size(), QNodeW<W>{});
    const double inf = std::numeric_limits<double>::infinity();
    double lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    for (const NodeW<R, W> &nd : in)
        for (int i = 0; i < W; i++)
            if (nd.c[i].child != CHILD_EMPTY)
                for (int a = 0; a < 3; a++) lo[a] = std::min(lo[a], (double)nd.c[i].bmin[a]), hi[a] = std::max(hi[a], (double)nd.c[i].bmax[a]);
    const QGrid g = make_qgrid(lo, hi);
    for (int a = 0; a < 3; a++) grid_lo[a] = g.lo[a], grid_step[a] = g.step[a];
    double ratio_sum = 0;
    int64_t n_slots = 0;
    for (size_t n = 0; n < in.size(); n++) {
        const NodeW<R, W> &nd = in[n];
        QNodeW<W> q{};
        for (int i = 0; i < W; i++) {
            q.c[i].child = nd.c[i].child;
            if (nd.c[i].child == CHILD_EMPTY) {
                // an empty slot holds an inverted box (lo = Q_MAX, hi = 0 on every axis): it fails the slab test by a
                // whole grid extent (the kernels still look at the child word: tk_traverse.h)
                for (int a = 0; a < 3; a++) q.c[i].q[a] = (uint32_t)Q_MAX;
                continue;
            }
            double et[3], eq[3];
            for (int a = 0; a < 3; a++) {
                long long ql, qh;
                qgrid_snap(g, a, (double)nd.c[i].bmin[a] - g.delta[a], (double)nd.c[i].bmax[a] + g.delta[a], ql, qh);
                q.c[i].q[a] = (uint32_t)ql | ((uint32_t)qh << 16);
                et[a] = (double)nd.c[i].bmax[a] - (double)nd.c[i].bmin[a];
                eq[a] = (double)(qh - ql) * (double)g.step[a];
            }
            const double at = et[0] * et[1] + et[1] * et[2] + et[2] * et[0], aq = eq[0] * eq[1] + eq[1] * eq[2] + eq[2] * eq[0];
            ratio_sum += at > 0 ? std::min(aq / at, 100.0) : (aq > 0 ? 100.0 : 1.0);
            n_slots++;
        }
        out[n] = q;
    }
    return n_slots ? ratio_sum / (double)n_slots : 1.0;
}

}  // namespace tk
