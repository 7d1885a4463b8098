// hostsim.cpp — TEST INFRASTRUCTURE.  Serial host re-execution of the *device* code paths of take_amd/csrc:
// the same tk_host_scene.h (scene preparation + wide BVH), tk_traverse.h, tk_shade.h and tk_integrate.h that
// the HIP kernels call, driven by plain loops in the same round order as the kernels of tk_kernels.h.
//
// Purpose: debug the product's device logic on a machine without a GPU (the authoring container) by comparing
// it with the oracle at small sizes.  It is compiled only by tests/ (tests/hostsim/Makefile), is not part of
// take_amd/ and is never loaded by the product: libtake_hip.so has no CPU path.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "take_hip.h"
#include "tk_host_scene.h"
#include "tk_integrate.h"

using namespace tk;

namespace {
struct ArrayStack {
    int32_t child[128];
    float key[128];
    int max_level = 0;
    void push(int level, int32_t c, float k) {
        child[level] = c;
        key[level] = k;
        if (level + 1 > max_level) max_level = level + 1;
    }
    void pop(int level, int32_t &c, float &k) {
        c = child[level];
        k = key[level];
    }
};
std::string g_err;

template <class R> void dump_slot(const PathState<R> &st, int64_t slot, const char *tag, int k) {
    std::fprintf(stderr, "[slot %lld] k=%d %s R:", (long long)slot, k, tag);
    for (int c = 0; c < PATH_REC; c++)
        if (c != S_HIT && c != S_CTR && c != S_FLAGS) std::fprintf(stderr, " %.17g", (double)st.R_(c, slot));
    std::fprintf(stderr, " I:");
    for (int c : {(int)S_HIT, (int)S_CTR, (int)S_FLAGS}) std::fprintf(stderr, " %d", st.I_(c, slot));
    std::fprintf(stderr, "\n");
}

template <class R> int render_t(const TakeSceneDesc &desc, const TakeRenderOpts &o, void *out_v, uint64_t *stats) {
    HostScene<R> hs;
    g_err = prepare_scene<R>(desc, std::getenv("HOSTSIM_MAX_LEAF") ? std::atoi(std::getenv("HOSTSIM_MAX_LEAF")) : 0, 1, hs);
    if (!g_err.empty()) return TAKE_E_INVALID;
    DeviceScene<R> sc = hs.view();
    const int W = hs.cam.width, H = hs.cam.height;
    const int stride = o.strip_stride > 0 ? o.strip_stride : 1;
    const int n_strips = (H + TILE_ROWS - 1) / TILE_ROWS;
    int n_rows = 0;
    for (int s = o.strip_first; s < n_strips; s += stride) n_rows += std::min(H, (s + 1) * TILE_ROWS) - s * TILE_ROWS;
    const int64_t npix = (int64_t)n_rows * W;
    RenderParams<R> rp{};
    rp.width = W, rp.height = H, rp.n_local_rows = n_rows, rp.npix = (int32_t)npix;
    rp.inv_npix = 1.0 / (double)npix, rp.inv_width = 1.0 / (double)W;
    rp.strip_first = o.strip_first, rp.strip_stride = stride;
    rp.spp = o.spp, rp.max_depth = o.max_depth, rp.seed = o.seed, rp.integrator = o.integrator;
    rp.ray_eps = o.ray_epsilon > 0 ? R(o.ray_epsilon) : (sizeof(R) == 8 ? R(1e-7) : R(1e-4));
    const int spb = o.samples_per_batch > 0 ? std::min(o.samples_per_batch, o.spp) : o.spp;
    const int64_t slots = (int64_t)spb * npix;
    std::vector<R> sr((size_t)PATH_REC * slots);
    PathState<R> st{sr.data(), slots};
    std::vector<R> accum(3 * npix, R(0));
    std::vector<int32_t> q[2], shadow;
    uint64_t n_closest = 0, n_shadow = 0, n_nodes = 0, n_prims = 0, max_stack = 0;
    const char *dump_env = std::getenv("TAKE_HIP_DUMP_SLOT");
    const int64_t dump = dump_env ? std::atoll(dump_env) : -1;
    for (int s0 = 0; s0 < o.spp; s0 += spb) {
        const int nb = std::min(spb, o.spp - s0);
        const int64_t n = (int64_t)nb * npix;
        rp.s0 = s0;
        rp.spb = nb;
        q[0].clear();
        for (int64_t s = 0; s < n; s++) {
            generate_path(sc, rp, st, s);
            q[0].push_back((int32_t)s);
        }
        for (int k = 0; k < o.max_depth + 2; k++) {
            const int cur = k & 1, next = cur ^ 1;
            q[next].clear();
            shadow.clear();
            for (int32_t slot : q[cur]) {  // k_trace<closest>
                RayT<R> ray = make_ray(st.R_(S_OX, slot), st.R_(S_OY, slot), st.R_(S_OZ, slot), st.R_(S_DX, slot),
                                       st.R_(S_DY, slot), st.R_(S_DZ, slot), rp.ray_eps, Const<R>::inf());
                HitT<R> hit;
                ArrayStack stack;
                TravCount tc;
                traverse<R, false, true>(sc, ray, stack, hit, tc);
                n_nodes += tc.nodes, n_prims += tc.prims, n_closest++;
                max_stack = std::max<uint64_t>(max_stack, stack.max_level);
                st.I_(S_HIT, slot) = hit.prim;
                st.I_(S_INST, slot) = hit.inst;
                st.R_(S_HT, slot) = hit.t;
                st.R_(S_HU, slot) = hit.u;
                st.R_(S_HV, slot) = hit.v;
            }
            if (dump >= 0 && dump < slots) dump_slot(st, dump, "after trace_closest", k);
            for (int32_t slot : q[cur]) {  // k_shade
                uint32_t req = rp.integrator ? shade_path_alt(sc, rp, st, (int64_t)slot, k) : shade_path(sc, rp, st, (int64_t)slot, k);
                if (req & REQ_EXTEND) q[next].push_back(slot);
                if (req & REQ_SHADOW) shadow.push_back(slot);
            }
            if (dump >= 0 && dump < slots) dump_slot(st, dump, "after shade", k);
            for (int32_t slot : shadow) {  // k_trace<shadow>
                RayT<R> ray = make_ray(st.R_(S_OX, slot), st.R_(S_OY, slot), st.R_(S_OZ, slot), st.R_(S_SX, slot),
                                       st.R_(S_SY, slot), st.R_(S_SZ, slot), rp.ray_eps, st.R_(S_ST, slot));
                HitT<R> hit;
                ArrayStack stack;
                TravCount tc;
                traverse<R, true, true>(sc, ray, stack, hit, tc);
                n_nodes += tc.nodes, n_prims += tc.prims, n_shadow++;
                if (hit.prim < 0) {
                    st.R_(S_LX, slot) = st.R_(S_LX, slot) + st.R_(S_CX, slot);
                    st.R_(S_LY, slot) = st.R_(S_LY, slot) + st.R_(S_CY, slot);
                    st.R_(S_LZ, slot) = st.R_(S_LZ, slot) + st.R_(S_CZ, slot);
                }
            }
            if (dump >= 0 && dump < slots) dump_slot(st, dump, "after trace_shadow", k);
            if (q[next].empty()) break;
        }
        for (int64_t p = 0; p < npix; p++)  // k_accumulate
            for (int s = 0; s < nb; s++) {
                const int64_t slot = (int64_t)s * npix + p;
                accum[3 * p] = accum[3 * p] + st.R_(S_LX, slot);
                accum[3 * p + 1] = accum[3 * p + 1] + st.R_(S_LY, slot);
                accum[3 * p + 2] = accum[3 * p + 2] + st.R_(S_LZ, slot);
            }
    }
    R *out = (R *)out_v;  // k_resolve
    const R inv = R(1) / R(o.spp);
    for (int64_t p = 0; p < npix; p++) {
        const int lr = (int)(p / W), x = (int)(p % W);
        const int64_t oidx = 3 * ((int64_t)(n_rows - 1 - lr) * W + x);
        for (int c = 0; c < 3; c++) out[oidx + c] = accum[3 * p + c] * inv;
    }
    if (stats) {
        stats[0] = n_closest, stats[1] = n_shadow, stats[2] = n_nodes, stats[3] = n_prims, stats[4] = max_stack;
        stats[5] = (uint64_t)hs.stats.n_nodes, stats[6] = (uint64_t)hs.stats.depth;
    }
    return TAKE_OK;
}

template <class R> int trace_t(const TakeSceneDesc &desc, const void *rays_v, int64_t n, void *hits_v, int any) {
    HostScene<R> hs;
    g_err = prepare_scene<R>(desc, std::getenv("HOSTSIM_MAX_LEAF") ? std::atoi(std::getenv("HOSTSIM_MAX_LEAF")) : 0, 1, hs);
    if (!g_err.empty()) return TAKE_E_INVALID;
    DeviceScene<R> sc = hs.view();
    const R *rays = (const R *)rays_v;  // org3 tmin dir3 tmax
    R *hits = (R *)hits_v;              // shape t u v  (shape as R)
    for (int64_t i = 0; i < n; i++) {
        const R *q = rays + 8 * i;
        RayT<R> ray = make_ray(q[0], q[1], q[2], q[4], q[5], q[6], q[3], q[7]);
        HitT<R> hit;
        ArrayStack stack;
        TravCount tc;
        if (any)
            traverse<R, true, false>(sc, ray, stack, hit, tc);
        else
            traverse<R, false, false>(sc, ray, stack, hit, tc);
        hits[4 * i] = R(hit.shape);
        hits[4 * i + 1] = hit.prim >= 0 ? hit.t : R(0);
        hits[4 * i + 2] = hit.u;
        hits[4 * i + 3] = hit.v;
    }
    return TAKE_OK;
}
// Compressed nodes of the f32 scene against the full-width ones they were made from, in exact (double) arithmetic.
// out[0] = child slots checked, out[1] = slots whose decoded box does NOT contain the true box widened by the
// builder's slack (must be 0), out[2] = child words that differ (must be 0), out[3] = 1e6 * surface-area inflation,
// out[4] = 1 if the scene uses compressed nodes, out[5] = node width (4 or 8), out[6] = 8-wide only: children whose
// centre lies on the wrong side of the node's centre on some axis for their slot (diagnostic, not an error)
template <int W>
static void check_qnodes_w(const HostScene<float> &hs, const std::vector<NodeW<float, W>> &nodes, const std::vector<QNodeW<W>> &qnodes, int64_t *out) {
    for (size_t n = 0; n < nodes.size(); n++)
        for (int i = 0; i < W; i++) {
            const NodeChild<float> &c = nodes[n].c[i];
            const QChild &q = qnodes[n].c[i];
            if (c.child != q.child) out[2]++;
            if (c.child == CHILD_EMPTY) continue;
            out[0]++;
            bool ok = true;
            for (int a = 0; a < 3; a++) {
                const double lo = (double)hs.grid_lo[a] + (double)(Q_BIAS + (q.q[a] & 0xffffu)) * (double)hs.grid_step[a];
                const double hi = (double)hs.grid_lo[a] + (double)(Q_BIAS + (q.q[a] >> 16)) * (double)hs.grid_step[a];
                const double slack = (double)Q_MAX * (double)hs.grid_step[a] * 0x1p-20;
                if (!(lo <= (double)c.bmin[a] - slack && hi >= (double)c.bmax[a] + slack)) ok = false;
                if ((q.q[a] & 0xffffu) > (q.q[a] >> 16) || (q.q[a] >> 16) > (uint32_t)Q_MAX) ok = false;
            }
            if (!ok) out[1]++;
        }
}
}  // namespace

extern "C" {
const char *hostsim_last_error(void) { return g_err.c_str(); }
// out: rows*W*3 Real (float for f32, double for f64); stats: 7 words (may be null)
int hostsim_render(const TakeSceneDesc *desc, int precision, const TakeRenderOpts *opts, void *out, uint64_t *stats) {
    return precision == TAKE_PRECISION_F64 ? render_t<double>(*desc, *opts, out, stats)
                                           : render_t<float>(*desc, *opts, out, stats);
}
// divmod_u31 (tk_integrate.h: the slot -> sample / pixel divisions of every shade round, by reciprocal) against the
// integer division, on divisors and dividends around every power of two, the extremes and `n_random` random pairs;
// returns the number of disagreements
int64_t hostsim_check_divmod(int64_t n_random, uint64_t seed) {
    int64_t bad = 0;
    auto check = [&](uint32_t n, uint32_t d) {
        if (d == 0 || n >= (1u << 31) || d >= (1u << 31)) return;
        uint32_t q, r;
        tk::divmod_u31(n, d, 1.0 / (double)d, q, r);
        bad += (q != n / d) || (r != n % d);
    };
    std::vector<uint32_t> edge;
    for (int b = 0; b < 31; b++)
        for (int64_t k = -2; k <= 2; k++) {
            const int64_t v = ((int64_t)1 << b) + k;
            if (v > 0 && v < ((int64_t)1 << 31)) edge.push_back((uint32_t)v);
        }
    edge.push_back((1u << 31) - 1), edge.push_back(1920 * 1080), edge.push_back(1920), edge.push_back(4096 * 4096), edge.push_back(3);
    for (uint32_t d : edge)
        for (uint32_t n : edge) {
            check(n, d);
            check((uint32_t)std::min<uint64_t>((uint64_t)n * d, (1ull << 31) - 1), d);      // exact multiples
            check((uint32_t)std::min<uint64_t>((uint64_t)n * d + d - 1, (1ull << 31) - 1), d);  // just below the next one
        }
    uint64_t z = seed;
    for (int64_t i = 0; i < n_random; i++) {
        z = tk::rng_mix(z + 0x9E3779B97F4A7C15ull);
        const uint32_t n = (uint32_t)(z >> 33), d = (uint32_t)(tk::rng_mix(z) >> (33 + (z & 31) % 30));
        check(n, d);
    }
    return bad;
}

int hostsim_check_qnodes(const TakeSceneDesc *desc, int64_t *out) {
    HostScene<float> hs;
    std::string err = prepare_scene<float>(*desc, 0, 2, hs);
    if (!err.empty()) {
        g_err = err;
        return -1;
    }
    out[0] = out[1] = out[2] = out[6] = 0;
    out[3] = (int64_t)(hs.q_inflation * 1e6);
    out[4] = hs.qnodes.empty() && hs.qnodes8.empty() ? 0 : 1;
    out[5] = hs.node_width;
    if (!hs.qnodes8.empty()) check_qnodes_w<8>(hs, hs.nodes8, hs.qnodes8, out);
    else if (!hs.qnodes.empty()) check_qnodes_w<4>(hs, hs.nodes, hs.qnodes, out);
    return 0;
}
// rays: n x 8 Real laid out as TakeRayF/TakeRayD; hits: n x 4 Real (shape id as Real, t, u, v)
int hostsim_trace(const TakeSceneDesc *desc, int precision, const void *rays, int64_t n, void *hits, int any) {
    return precision == TAKE_PRECISION_F64 ? trace_t<double>(*desc, rays, n, hits, any)
                                           : trace_t<float>(*desc, rays, n, hits, any);
}
}
