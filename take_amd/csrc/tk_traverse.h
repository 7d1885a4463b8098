// tk_traverse.h — ray/primitive tests and the per-lane wide-BVH traversal.
//
// Functional counterpart of the reference's bvh_intersect (src/bvh.cpp:86-109), intersect(BBox,Ray)
// (src/bbox.h:18-32) and intersect_op (src/shape.cpp:13-80) — redesigned, not transliterated:
//   * iterative, front-to-back over a 4-wide BVH with a (child, entry distance) stack, culling deferred
//     subtrees against the current closest hit when they are popped;
//   * the primitive tests keep the reference's exact decision sequence and IEEE operation order (epsilon
//     determinant cull, inclusive [tmin, tmax], u/v/u+v bounds) so that t,u,v of a hit are bit-identical
//     to the f32/f64 restatement in oracle/;
//   * the box test is conservative (both planes widened by a few ulp), so a different tree shape cannot
//     lose a hit the reference's tree finds; closest hit is then tree-independent up to exact ties;
//   * shadow rays stop at the first hit (scene_occluded only uses the boolean, src/scene.cpp:49-53).
#pragma once

#include "tk_scene.h"

namespace tk {

template <class R> struct RayT {
    Vec3<R> o, d;
    R tmin, tmax;
};
template <class R> struct HitT {
    int32_t shape;  // -1 = miss
    int32_t prim;   // index of the hit primitive in leaf order (-1 = miss)
    int32_t meta;   // PrimRec::meta of the hit primitive
    int32_t inst;   // instance the primitive was reached through (two-level scenes), -1 = none
    R t, u, v;
};

// Möller–Trumbore, decision sequence of src/shape.cpp:52-78.  tmax is the current closest distance.
template <class R> TK_HD bool tri_test(const R *a, const RayT<R> &r, R tmax, R &t, R &u, R &v) {
    Vec3<R> v0{a[0], a[1], a[2]}, e1{a[3], a[4], a[5]}, e2{a[6], a[7], a[8]};
    Vec3<R> h = cross(r.d, e2);
    R det = dot(e1, h);
    if (det > -Const<R>::DET_EPS && det < Const<R>::DET_EPS) return false;
    R f = R(1.0) / det;
    Vec3<R> s = r.o - v0;
    u = f * dot(s, h);
    if (u < R(0.0) || u > R(1.0)) return false;
    Vec3<R> q = cross(s, e1);
    v = f * dot(r.d, q);
    if (v < R(0.0) || u + v > R(1.0)) return false;
    t = f * dot(e2, q);
    if (t < r.tmin || tmax < t) return false;
    return true;
}
// src/shape.cpp:13-29
template <class R> TK_HD bool sphere_test(const R *a, const RayT<R> &r, R tmax, R &t) {
    Vec3<R> c{a[0], a[1], a[2]};
    R radius = a[3];
    Vec3<R> oc = r.o - c;
    R qa = dot(r.d, r.d);
    R half_b = dot(oc, r.d);
    R qc = dot(oc, oc) - radius * radius;
    R disc = half_b * half_b - qa * qc;
    if (disc < R(0)) return false;
    R sqrtd = tk_sqrt(disc);
    R root = (-half_b - sqrtd) / qa;
    if (root < r.tmin || tmax < root) {
        root = (-half_b + sqrtd) / qa;
        if (root < r.tmin || tmax < root) return false;
    }
    t = root;
    return true;
}

template <class R> TK_HD R safe_inv(R d) {
    const R tiny = R(1e-30);
    R ad = tk_fabs(d);
    R dd = ad > tiny ? d : __builtin_copysign(tiny, d);
    return R(1) / dd;
}
TK_HD float stack_key(float t) { return t; }
TK_HD float stack_key(double t) {  // largest float <= t: keeps pop-time culling conservative in f64 mode
    float f = (float)t;
    if ((double)f > t) {  // step one float towards -inf (entry distances are finite and >= tmin > 0)
        union {
            float f;
            uint32_t u;
        } b;
        b.f = f;
        b.u = f > 0.0f ? b.u - 1u : (f < 0.0f ? b.u + 1u : 0x80000001u);
        f = b.f;
    }
    return f;
}

// smallest float >= t (t >= 0 or +inf): the f32 image of a double far limit that keeps a box test conservative
TK_HD float float_above(float t) { return t; }
TK_HD float float_above(double t) {
    float f = (float)t;
    if ((double)f < t) {
        union {
            float f;
            uint32_t u;
        } b;
        b.f = f;
        b.u = f > 0.0f ? b.u + 1u : (f < 0.0f ? b.u - 1u : 0x00000001u);
        f = b.f;
    }
    return f;
}

// Conservative slab test of one child slot: true when [tmin, tbest] can overlap the box.  Both planes are widened
// by a few ulp (t >= tmin > 0 on this path, so the scaling is monotone), which keeps closest-hit results
// independent of the tree shape.  tn = entry distance (the traversal order key).
template <class R>
TK_HD bool box_test(const NodeChild<R> &c, Vec3<R> o, R idx, R idy, R idz, R tmin, R tbest, R &tn) {
    R t0x = (c.bmin[0] - o.x) * idx, t1x = (c.bmax[0] - o.x) * idx;
    R t0y = (c.bmin[1] - o.y) * idy, t1y = (c.bmax[1] - o.y) * idy;
    R t0z = (c.bmin[2] - o.z) * idz, t1z = (c.bmax[2] - o.z) * idz;
    tn = tk_fmax(tk_fmax(tk_fmin(t0x, t1x), tk_fmin(t0y, t1y)), tk_fmax(tk_fmin(t0z, t1z), tmin));
    R tf = tk_fmin(tk_fmin(tk_fmax(t0x, t1x), tk_fmax(t0y, t1y)), tk_fmin(tk_fmax(t0z, t1z), tbest));
    return (tn * Const<R>::BOX_SHRINK <= tf * Const<R>::BOX_GROW) && (c.child != CHILD_EMPTY);
}

// ---- compressed nodes (QNode4).  The ray is moved into grid space once (QRay) instead of decoding planes:
//   t(q) = (grid_lo + (Q_BIAS + q) step - o) / d = A + F(q) B,   F(q) = 4 (Q_BIAS + q),   A = (grid_lo - o) * inv_d,
//   B = inv_d * step / 4.
// F(q) is the float with the bits 0x48000000 | q << 8 (q < 2^15: exponent 2^17, q in mantissa bits 8..22) — one byte
// permute of the slot word, no integer-to-float conversion — and the fma rounds F B + A once.
// Rounding (X = (Q_BIAS + q) step in [1, 2] grid extents, Y = grid_lo - o): the computed numerator is
// X (1 + e4) + Y (1 + e2 + e3), |e| <= 2^-24, i.e. the plane looks displaced by <= 2^-24 (X + 2 |Y|).  With the origin
// inside the scene box (|Y| <= 2 extents) that is <= 6 2^-24 = 0.75 2^-21 extents, which the builder adds to every
// child box before snapping it outwards to the grid (tk_bvh.h: make_qgrid, delta = 2^-20 extents); farther away it is
// 3 2^-24 X (inside delta) plus a relative error of 2 2^-24 of the distance, which together with inv_d and the final
// fma rounding (2 more) stays inside BOX_SHRINK / BOX_GROW (6.7 2^-24 each) like the full-width test.
#ifndef TQ_PRESCALE
#define TQ_PRESCALE 0  // 1: the conservative margins are folded into the ray's grid-space constants (experiment)
#endif
struct QRay {
    float ax, ay, az, bx, by, bz;
#if TQ_PRESCALE
    // the far planes' constants: (a, b) * BOX_GROW, while ax.. bz hold (a, b) * BOX_SHRINK — the margins of the slab
    // test cost no instruction.  Scaling the two terms of F b + a separately adds 2^-24 (X + |Y|) to the plane's
    // displacement (<= 0.5 2^-21 extents inside the scene: still within delta) and 2^-24 to the relative error.
    float axf, ayf, azf, bxf, byf, bzf;
#endif
    // per axis the v_perm_b32 selector that turns a slot word (lo | hi << 16) into F of the plane the ray meets FIRST
    // (lo for a positive slope, hi for a negative one); the other plane's selector is this one ^ QSEL_FLIP.  The entry
    // and exit distances come out of the fma directly, without a min / max per axis and box.
    uint32_t sx, sy, sz;
    uint32_t oct16;  // 8-wide nodes: (octant of the direction: bit a set = negative along axis a) << 4 = the byte offset
                     // that turns a visiting rank into a slot (slot = rank ^ octant, tk_scene.h: QNode8)
};
constexpr uint32_t QSEL_LO = 0x0005040Cu;    // bytes (0x00, w.b0, w.b1, K.b0): the low half of the slot word
constexpr uint32_t QSEL_HI = 0x0007060Cu;    // bytes (0x00, w.b2, w.b3, K.b0): the high half
constexpr uint32_t QSEL_FLIP = QSEL_LO ^ QSEL_HI;
constexpr uint32_t QPERM_K = 0x48484848u;    // exponent byte of 2^17
TK_HD void qray_selectors(QRay &f) {
    f.sx = f.bx < 0.0f ? QSEL_HI : QSEL_LO, f.sy = f.by < 0.0f ? QSEL_HI : QSEL_LO, f.sz = f.bz < 0.0f ? QSEL_HI : QSEL_LO;
    f.oct16 = (f.bx < 0.0f ? 16u : 0u) | (f.by < 0.0f ? 32u : 0u) | (f.bz < 0.0f ? 64u : 0u);
}
TK_HD void qray_margins(QRay &f) {
#if TQ_PRESCALE
    const float G = Const<float>::BOX_GROW, S = Const<float>::BOX_SHRINK;
    f.axf = f.ax * G, f.ayf = f.ay * G, f.azf = f.az * G, f.bxf = f.bx * G, f.byf = f.by * G, f.bzf = f.bz * G;
    f.ax *= S, f.ay *= S, f.az *= S, f.bx *= S, f.by *= S, f.bz *= S;
#endif
}
TK_HD QRay qray_make(const float *grid_lo, const float *grid_step, Vec3<float> o, float idx, float idy, float idz) {
    QRay f;
    f.ax = (grid_lo[0] - o.x) * idx, f.ay = (grid_lo[1] - o.y) * idy, f.az = (grid_lo[2] - o.z) * idz;
    f.bx = idx * grid_step[0] * 0.25f, f.by = idy * grid_step[1] * 0.25f, f.bz = idz * grid_step[2] * 0.25f;
    qray_selectors(f);
    qray_margins(f);
    return f;
}
// f64 rays traverse the same compressed nodes with the same f32 slab test: A and B are formed in double and rounded
// to float once (half the rounding of the f32 route above, which the builder's slack and BOX_SHRINK / BOX_GROW
// already cover); the limits tmin / tbest enter the test rounded outwards (stack_key, float_above).
TK_HD QRay qray_make(const float *grid_lo, const float *grid_step, Vec3<double> o, double idx, double idy, double idz) {
    QRay f;
    f.ax = (float)(((double)grid_lo[0] - o.x) * idx), f.ay = (float)(((double)grid_lo[1] - o.y) * idy),
    f.az = (float)(((double)grid_lo[2] - o.z) * idz);
    f.bx = (float)(idx * (double)grid_step[0] * 0.25), f.by = (float)(idy * (double)grid_step[1] * 0.25), f.bz = (float)(idz * (double)grid_step[2] * 0.25);
    qray_selectors(f);
    qray_margins(f);
    return f;
}
#if defined(__HIP_DEVICE_COMPILE__)
typedef float tk_f2 __attribute__((ext_vector_type(2)));
// both planes of one axis: two byte permutes and one packed fma (v_pk_fma_f32); same value per element as the scalar
// form below
__device__ __forceinline__ void qplanes(uint32_t w, uint32_t sel, float b0, float a0, float b1, float a1, float &t0, float &t1) {
    const tk_f2 q = {__uint_as_float(__builtin_amdgcn_perm(w, QPERM_K, sel)), __uint_as_float(__builtin_amdgcn_perm(w, QPERM_K, sel ^ QSEL_FLIP))};
    // (one register pair {b, a} feeding both operands through op_sel — v_pk_fma_f32 d, q, ba, ba op_sel:[0,0,1]
    // op_sel_hi:[1,0,1] — would save six registers of per-ray state, but measured 9 % SLOWER on the closest-hit kernel
    // than the {b, b}, {a, a} pairs the compiler keeps: not used)
    const tk_f2 t = __builtin_elementwise_fma(q, (tk_f2){b0, b1}, (tk_f2){a0, a1});
    t0 = t.x, t1 = t.y;
}
#else
inline float qplane_float(uint32_t q) {
    union {
        uint32_t u;
        float f;
    } c;
    c.u = 0x48000000u | (q << 8);
    return c.f;
}
inline void qplanes(uint32_t w, uint32_t sel, float b0, float a0, float b1, float a1, float &t0, float &t1) {
    const uint32_t lo = w & 0xffffu, hi = w >> 16;
    t0 = __builtin_fmaf(qplane_float(sel == QSEL_LO ? lo : hi), b0, a0);
    t1 = __builtin_fmaf(qplane_float(sel == QSEL_LO ? hi : lo), b1, a1);
}
#endif
// conservative slab test of one compressed child slot
TK_HD bool qbox_test(const QRay &f, uint32_t qx, uint32_t qy, uint32_t qz, int32_t child, float tmin, float tbest, float &tn) {
    // near / far plane per axis by selector (QRay): the same two values min / max would pick — the fma is monotone in
    // the plane coordinate, increasing for a positive slope and decreasing for a negative one
    float nx, fx, ny, fy, nz, fz;
#if TQ_PRESCALE
    qplanes(qx, f.sx, f.bx, f.ax, f.bxf, f.axf, nx, fx);
    qplanes(qy, f.sy, f.by, f.ay, f.byf, f.ayf, ny, fy);
    qplanes(qz, f.sz, f.bz, f.az, f.bzf, f.azf, nz, fz);
    tn = tk_fmax(tk_fmax(nx, ny), tk_fmax(nz, tmin));  // (already shrunk: tn is the culling key as it stands)
    const float tf = tk_fmin(tk_fmin(fx, fy), tk_fmin(fz, tbest));
    return (tn <= tf) && (child != CHILD_EMPTY);
#else
    qplanes(qx, f.sx, f.bx, f.ax, f.bx, f.ax, nx, fx);
    qplanes(qy, f.sy, f.by, f.ay, f.by, f.ay, ny, fy);
    qplanes(qz, f.sz, f.bz, f.az, f.bz, f.az, nz, fz);
    tn = tk_fmax(tk_fmax(nx, ny), tk_fmax(nz, tmin));
    const float tf = tk_fmin(tk_fmin(fx, fy), tk_fmin(fz, tbest));
    return (tn * Const<float>::BOX_SHRINK <= tf * Const<float>::BOX_GROW) && (child != CHILD_EMPTY);
#endif
}

struct TravCount {
    uint32_t nodes = 0, prims = 0, leaves = 0;
};

// Stack concept: void push(int level, int32_t child, float key); void pop(int level, int32_t&, float&)
template <class R, bool ANY_HIT, bool COUNT, class Stack>
TK_HD void traverse(const DeviceScene<R> &sc, const RayT<R> &ray, Stack &stack, HitT<R> &hit, TravCount &tc) {
    hit.shape = -1;
    hit.prim = -1;
    hit.meta = 0;
    hit.inst = -1;
    hit.t = ray.tmax;
    hit.u = hit.v = R(0);
    const R idx = safe_inv(ray.d.x), idy = safe_inv(ray.d.y), idz = safe_inv(ray.d.z);
    R tbest = ray.tmax;
    QRay qr{};
    const bool qn8 = sc.qnodes8 != nullptr;
    const bool qn = sc.qnodes != nullptr || qn8;
    if (qn) qr = qray_make(sc.grid_lo, sc.grid_step, ray.o, idx, idy, idz);
    const float tmin_f = stack_key(ray.tmin);  // largest float <= tmin
    // conservative float image of an entry distance (the stack's culling key): the compressed test computes in f32
    // whatever R is, so its margin is the f32 one
    auto cull_key = [&](R k) -> float { return qn ? (TQ_PRESCALE ? (float)k : (float)k * Const<float>::BOX_SHRINK) : stack_key(k * Const<R>::BOX_SHRINK); };
    // 8-wide nodes: the slots are visited in the order slot ^ octant (tk_scene.h: QNode8)
    const int oct = (int)(qr.oct16 >> 4);
    int sp = 0;
    int32_t cur = sc.root_child;
    for (;;) {
        if (cur >= 0 && qn8) {
            // the order of the trace kernel's 8-wide step: the nearest hit child is visited next (closest hit) / the
            // first one in octant order (any hit), the others are pushed far to near in octant order
            if (COUNT) tc.nodes++;
            const QNode8 &n = sc.qnodes8[cur];
            const float tbest_f = float_above(tbest);
            R key[8];
            int32_t ch[8];
            int nearest = -1;
            for (int k = 0; k < 8; k++) {
                const QChild &c = n.c[k ^ oct];
                float tn;
                const bool ok = qbox_test(qr, c.q[0], c.q[1], c.q[2], c.child, tmin_f, tbest_f, tn);
                key[k] = ok ? (R)tn : Const<R>::inf();
                ch[k] = c.child;
                if (ok && (nearest < 0 || (!ANY_HIT && key[k] < key[nearest]))) nearest = k;
            }
            for (int k = 7; k >= 0; k--)
                if (k != nearest && key[k] < Const<R>::inf()) stack.push(sp++, ch[k], cull_key(key[k]));
            if (nearest >= 0) {
                cur = ch[nearest];
                continue;
            }
            goto pop_next;
        }
        if (cur >= 0) {
            if (COUNT) tc.nodes++;
            R key[4];
            int32_t ch[4];
            if (qn) {  // compressed nodes
                const QNode4 &n = sc.qnodes[cur];
                const float tbest_f = float_above(tbest);
                for (int i = 0; i < 4; i++) {
                    float tn;
                    const bool ok = qbox_test(qr, n.c[i].q[0], n.c[i].q[1], n.c[i].q[2], n.c[i].child, tmin_f, tbest_f, tn);
                    key[i] = ok ? (R)tn : Const<R>::inf();
                    ch[i] = n.c[i].child;
                }
            } else {
                const Node4<R> &n = sc.nodes[cur];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    R tn;
                    bool ok = box_test(n.c[i], ray.o, idx, idy, idz, ray.tmin, tbest, tn);
                    key[i] = ok ? tn : Const<R>::inf();
                    ch[i] = n.c[i].child;
                }
            }
            if (ANY_HIT) {
                // shadow rays: the order of the trace kernel's any-hit instance (tk_trace_quad.h) — no ranking, the
                // first hit slot is visited next, the others are popped last slot first
                int first = -1;
                for (int i = 0; i < 4; i++)
                    if (key[i] < Const<R>::inf()) {
                        if (first < 0) first = i;
                        else stack.push(sp++, ch[i], cull_key(key[i]));
                    }
                if (first >= 0) {
                    cur = ch[first];
                    continue;
                }
                goto pop_next;
            }
            // sorting network, ascending by entry distance; equal distances (boxes that contain the ray origin all
            // enter at tmin) go to the LATER slot first: slots are stored largest box first (tk_bvh.h), and the
            // smallest box is the one most likely to yield a close hit that prunes the others
#define TK_CSWAP(a, b)                        \
    {                                         \
        bool sw = key[b] <= key[a];           \
        R ka = sw ? key[b] : key[a];          \
        R kb = sw ? key[a] : key[b];          \
        int32_t ca = sw ? ch[b] : ch[a];      \
        int32_t cb = sw ? ch[a] : ch[b];      \
        key[a] = ka, key[b] = kb, ch[a] = ca, ch[b] = cb; \
    }
            TK_CSWAP(0, 1) TK_CSWAP(2, 3) TK_CSWAP(0, 2) TK_CSWAP(1, 3) TK_CSWAP(1, 2)
#undef TK_CSWAP
            // far to near: deferred children go on the stack, the nearest is visited next
            if (key[3] < Const<R>::inf()) stack.push(sp++, ch[3], cull_key(key[3]));
            if (key[2] < Const<R>::inf()) stack.push(sp++, ch[2], cull_key(key[2]));
            if (key[1] < Const<R>::inf()) stack.push(sp++, ch[1], cull_key(key[1]));
            if (key[0] < Const<R>::inf()) {
                cur = ch[0];
                continue;
            }
        } else if (is_instance_word(cur)) {
            // two-level scenes: the prototype's tree in its object space (same t), as a nested traversal (the trace
            // kernel does this on one stack with a return marker, tk_trace_quad.h)
            const int32_t id = instance_of_word(cur);
            const InstTrace<R> &it = sc.inst_trace[id];
            const Vec3<R> o = ray.o, d = ray.d;
            RayT<R> r2 = ray;
            r2.o = {it.inv[0] * o.x + it.inv[1] * o.y + it.inv[2] * o.z + it.inv[3],
                    it.inv[4] * o.x + it.inv[5] * o.y + it.inv[6] * o.z + it.inv[7],
                    it.inv[8] * o.x + it.inv[9] * o.y + it.inv[10] * o.z + it.inv[11]};
            r2.d = {it.inv[0] * d.x + it.inv[1] * d.y + it.inv[2] * d.z, it.inv[4] * d.x + it.inv[5] * d.y + it.inv[6] * d.z,
                    it.inv[8] * d.x + it.inv[9] * d.y + it.inv[10] * d.z};
            r2.tmax = tbest;
            DeviceScene<R> sub = sc;
            sub.root_child = it.root_child;
            for (int a = 0; a < 3; a++) sub.grid_lo[a] = it.grid_lo[a], sub.grid_step[a] = it.grid_step[a];
            HitT<R> h2;
            Stack s2;
            traverse<R, ANY_HIT, COUNT>(sub, r2, s2, h2, tc);
            if (h2.prim >= 0 && (h2.t < tbest || hit.prim < 0 || h2.u > hit.u ||
                                 (h2.u == hit.u && (h2.v > hit.v || (h2.v == hit.v && (id > hit.inst || (id == hit.inst && h2.prim > hit.prim))))))) {
                tbest = h2.t;
                hit = h2;
                hit.inst = id;
                hit.shape = sc.inst_shade[id].shape_base + h2.shape;
                if (ANY_HIT) return;
            }
        } else if (cur != CHILD_EMPTY) {
            const int first = leaf_first(cur), cnt = leaf_count(cur);
            if (COUNT) tc.leaves++;
            for (int k = 0; k < cnt; k++) {
                const PrimRec<R> &p = sc.prims[first + k];
                if (COUNT) tc.prims++;
                R t, u = R(0), v = R(0);
                bool ok = ((p.meta & 0xff) == PRIM_TRIANGLE) ? tri_test(p.a, ray, tbest, t, u, v)
                                                             : sphere_test(p.a, ray, tbest, t);
                // ties in t: larger (u, v), then (coincident primitives) the larger primitive index = the larger shape
                // id (the builders order coincident primitives that way) — the same in any tree
                // (coincident candidates from different trees of a two-level scene: the larger instance id wins, a top-level
                // primitive counting as instance -1 — the kernel's rule, tk_trace_quad.h — then the larger primitive index)
                if (ok && (t < tbest || hit.prim < 0 || u > hit.u ||
                           (u == hit.u && (v > hit.v || (v == hit.v && (hit.inst != -1 ? -1 > hit.inst : first + k > hit.prim)))))) {
                    tbest = t;
                    hit.shape = p.shape_id;
                    hit.prim = first + k;
                    hit.meta = p.meta;
                    hit.inst = -1;
                    hit.t = t;
                    hit.u = u;
                    hit.v = v;
                    if (ANY_HIT) return;
                }
            }
        }
        // pop the next deferred subtree that can still contain a closer hit
    pop_next:
        for (;;) {
            if (sp == 0) return;
            float k;
            stack.pop(--sp, cur, k);
            if ((R)k <= tbest) break;
        }
    }
}

// make a ray record from raw components
template <class R> TK_HD RayT<R> make_ray(R ox, R oy, R oz, R dx, R dy, R dz, R tmin, R tmax) {
    RayT<R> r;
    r.o = {ox, oy, oz};
    r.d = {dx, dy, dz};
    r.tmin = tmin;
    r.tmax = tmax;
    return r;
}

}  // namespace tk
