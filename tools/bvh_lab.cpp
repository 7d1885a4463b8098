// bvh_lab.cpp — DEVELOPMENT TOOL (not product, not test): builds the wide BVH of a procedural triangle soup with the
// product's own builder headers and measures traversal work (node visits, leaf visits, primitive tests per ray) for
// incoherent rays on the host, so that builder changes can be ranked without a GPU.
//   g++ -std=c++17 -O2 -I include -I take_amd/csrc tools/bvh_lab.cpp -pthread -o /tmp/bvh_lab && /tmp/bvh_lab 1000000
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "take_hip.h"
#include "tk_bvh.h"
#include "tk_traverse.h"

using namespace tk;

struct Stack {
    int32_t child[128];
    float key[128];
    void push(int l, int32_t c, float k) { child[l] = c, key[l] = k; }
    void pop(int l, int32_t &c, float &k) { c = child[l], k = key[l]; }
};

int main(int argc, char **argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 1000000;
    const int max_leaf = argc > 2 ? std::atoi(argv[2]) : 2;
    const int n_rays = argc > 3 ? std::atoi(argv[3]) : 400000;
    const float jitter = n <= 200000 ? 0.02f : 0.008f;
    std::mt19937_64 rng(1234);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<float> tri;  // 9 floats per triangle
    auto add_tri = [&](const float *a, const float *b, const float *c) {
        for (int k = 0; k < 3; k++) tri.push_back(a[k]);
        for (int k = 0; k < 3; k++) tri.push_back(b[k]);
        for (int k = 0; k < 3; k++) tri.push_back(c[k]);
    };
    auto quad = [&](float cx, float cy, float cz, float ux, float uy, float uz, float vx, float vy, float vz) {
        float p[4][3] = {{cx - ux - vx, cy - uy - vy, cz - uz - vz}, {cx + ux - vx, cy + uy - vy, cz + uz - vz},
                         {cx + ux + vx, cy + uy + vy, cz + uz + vz}, {cx - ux + vx, cy - uy + vy, cz - uz + vz}};
        add_tri(p[0], p[1], p[2]);
        add_tri(p[0], p[2], p[3]);
    };
    quad(0, 0, -1, 1, 0, 0, 0, 1, 0), quad(0, -1, 0, 1, 0, 0, 0, 0, -1), quad(0, 1, 0, 1, 0, 0, 0, 0, 1);
    quad(-1, 0, 0, 0, 0, -1, 0, 1, 0), quad(1, 0, 0, 0, 0, 1, 0, 1, 0), quad(0, .99f, 0, .3f, 0, 0, 0, 0, .3f);
    for (int i = 0; i < n; i++) {
        float c[3] = {0.9f * U(rng), 0.9f * U(rng), 0.9f * U(rng)}, v[3][3];
        for (int k = 0; k < 3; k++)
            for (int a = 0; a < 3; a++) v[k][a] = c[a] + jitter * U(rng);
        add_tri(v[0], v[1], v[2]);
    }
    const int nt = (int)tri.size() / 9;
    std::vector<BuildPrim> bp(nt);
    std::vector<PrimRec<float>> recs(nt);
    const double split_s = std::getenv("LAB_SPLIT") ? std::atof(std::getenv("LAB_SPLIT")) : 0.0;
    std::vector<BuildPrim> extra;
    for (int i = 0; i < nt; i++) {
        const float *t = &tri[9 * i];
        for (int a = 0; a < 3; a++) {
            bp[i].bmin[a] = std::min(t[a], std::min(t[3 + a], t[6 + a]));
            bp[i].bmax[a] = std::max(t[a], std::max(t[3 + a], t[6 + a]));
            recs[i].a[a] = t[a], recs[i].a[3 + a] = t[3 + a] - t[a], recs[i].a[6 + a] = t[6 + a] - t[a];
        }
        bp[i].id = i;
        recs[i].shape_id = i, recs[i].meta = PRIM_TRIANGLE;
        if (split_s > 0 && i >= 12) {  // early split clipping: clip the triangle to boxes no larger than split_s
            struct Poly { double v[10][3]; int n; };
            std::vector<Poly> work, done;
            Poly p0; p0.n = 3;
            for (int k = 0; k < 3; k++) for (int a = 0; a < 3; a++) p0.v[k][a] = t[3 * k + a];
            work.push_back(p0);
            while (!work.empty()) {
                Poly p = work.back(); work.pop_back();
                double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30};
                for (int k = 0; k < p.n; k++) for (int a = 0; a < 3; a++) lo[a] = std::min(lo[a], p.v[k][a]), hi[a] = std::max(hi[a], p.v[k][a]);
                int ax = 0;
                for (int a = 1; a < 3; a++) if (hi[a] - lo[a] > hi[ax] - lo[ax]) ax = a;
                if (hi[ax] - lo[ax] <= split_s) { done.push_back(p); continue; }
                const double mid = 0.5 * (lo[ax] + hi[ax]);
                Poly A, B; A.n = B.n = 0;
                for (int k = 0; k < p.n; k++) {
                    const double *u = p.v[k], *w = p.v[(k + 1) % p.n];
                    const bool ui = u[ax] <= mid, wi = w[ax] <= mid;
                    if (ui) { for (int a = 0; a < 3; a++) A.v[A.n][a] = u[a]; A.n++; }
                    if (!ui || u[ax] == mid) { for (int a = 0; a < 3; a++) B.v[B.n][a] = u[a]; B.n++; }
                    if (ui != wi && u[ax] != mid && w[ax] != mid) {
                        const double f = (mid - u[ax]) / (w[ax] - u[ax]);
                        double x[3];
                        for (int a = 0; a < 3; a++) x[a] = u[a] + f * (w[a] - u[a]);
                        x[ax] = mid;
                        for (int a = 0; a < 3; a++) A.v[A.n][a] = x[a], B.v[B.n][a] = x[a];
                        A.n++, B.n++;
                    }
                }
                if (A.n >= 3) work.push_back(A);
                if (B.n >= 3) work.push_back(B);
            }
            bool firstp = true;
            for (auto &p : done) {
                BuildPrim b;
                b.id = i;
                for (int a = 0; a < 3; a++) b.bmin[a] = 1e30, b.bmax[a] = -1e30;
                for (int k = 0; k < p.n; k++) for (int a = 0; a < 3; a++) b.bmin[a] = std::min(b.bmin[a], p.v[k][a] - 1e-7), b.bmax[a] = std::max(b.bmax[a], p.v[k][a] + 1e-7);
                if (firstp) bp[i] = b, firstp = false; else extra.push_back(b);
            }
        }
    }
    bp.insert(bp.end(), extra.begin(), extra.end());
    std::printf("build prims (refs): %zu\n", bp.size());
    auto t0 = std::chrono::steady_clock::now();
    const int builder_leaf = std::getenv("LAB_BUILDER_LEAF") ? std::atoi(std::getenv("LAB_BUILDER_LEAF")) : max_leaf;
    Bvh2Builder builder(bp, builder_leaf, 8);
    const int root = builder.build();
    std::vector<Node4<float>> nodes;
    std::vector<int32_t> order;
    WideBvhStats stats;
    const int32_t root_child = collapse_to_wide<float>(builder.nodes(), root, nodes, order, stats);
    const double build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<PrimRec<float>> prims(order.size());
    for (size_t k = 0; k < order.size(); k++) prims[k] = recs[bp[order[k]].id];
    std::vector<QNode4> qn;
    float glo[3], gst[3];
    const double infl = quantise_nodes(nodes, qn, glo, gst);
    // node statistics
    int64_t fill[5] = {0}, leaf_sz[5] = {0}, n_leaves = 0;
    for (auto &nd : nodes) {
        int f = 0;
        for (int i = 0; i < 4; i++)
            if (nd.c[i].child != CHILD_EMPTY) {
                f++;
                if (nd.c[i].child < 0) leaf_sz[leaf_count(nd.c[i].child)]++, n_leaves++;
            }
        fill[f]++;
    }
    std::printf("tris %d max_leaf %d: build %.2fs nodes %zu depth %d leaves %lld prim refs %zu q-inflation %.4f\n", nt, max_leaf, build_s,
                nodes.size(), stats.depth, (long long)n_leaves, prims.size(), infl);
    std::printf("  children per node: 1:%lld 2:%lld 3:%lld 4:%lld   leaf sizes: 1:%lld 2:%lld 3:%lld 4:%lld\n", (long long)fill[1],
                (long long)fill[2], (long long)fill[3], (long long)fill[4], (long long)leaf_sz[1], (long long)leaf_sz[2],
                (long long)leaf_sz[3], (long long)leaf_sz[4]);
    DeviceScene<float> sc{};
    sc.nodes = nodes.data(), sc.qnodes = qn.data(), sc.prims = prims.data(), sc.root_child = root_child;
    for (int a = 0; a < 3; a++) sc.grid_lo[a] = glo[a], sc.grid_step[a] = gst[a];
    // rays: bounce-like (origin on a random triangle, random direction) — closest hit; and the same as shadow rays
    const int T = 8;
    std::vector<uint64_t> occl(T, 0), cn(T, 0), cp(T, 0), cl(T, 0), sn(T, 0), sp(T, 0), sl(T, 0), hits(T, 0);
    std::vector<std::thread> pool;
    for (int t = 0; t < T; t++)
        pool.emplace_back([&, t] {
            std::mt19937_64 r(99 + t);
            std::uniform_real_distribution<float> V(-1.f, 1.f);
            for (int i = t; i < n_rays; i += T) {
                const int k = (int)(r() % (uint64_t)nt);
                const float *p = &tri[9 * k];
                float o[3], d[3], l2;
                for (int a = 0; a < 3; a++) o[a] = (p[a] + p[3 + a] + p[6 + a]) / 3.f;
                do {
                    for (int a = 0; a < 3; a++) d[a] = V(r);
                    l2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                } while (l2 > 1.f || l2 < 1e-4f);
                const float inv = 1.f / std::sqrt(l2);
                RayT<float> ray = make_ray(o[0], o[1], o[2], d[0] * inv, d[1] * inv, d[2] * inv, 1e-4f, Const<float>::inf());
                HitT<float> hit;
                Stack st;
                TravCount tc;
                traverse<float, false, true>(sc, ray, st, hit, tc);
                cn[t] += tc.nodes, cp[t] += tc.prims, cl[t] += tc.leaves;
                hits[t] += hit.prim >= 0;
                TravCount ts;
                if (std::getenv("LAB_LIGHT_RAYS")) {  // shadow rays as the integrator makes them: towards a point on the quad light
                    const float lx = 0.3f * V(r), lz = 0.3f * V(r), ly = 0.99f;
                    float dd[3] = {lx - o[0], ly - o[1], lz - o[2]};
                    const float dist = std::sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]);
                    ray = make_ray(o[0], o[1], o[2], dd[0] / dist, dd[1] / dist, dd[2] / dist, 1e-4f, (1.f - 1e-4f) * dist);
                }
                traverse<float, true, true>(sc, ray, st, hit, ts);
                occl[t] += hit.prim >= 0;
                sn[t] += ts.nodes, sp[t] += ts.prims, sl[t] += ts.leaves;
            }
        });
    for (auto &th : pool) th.join();
    if (std::getenv("LAB_LEVELS")) {  // node visits per tree level (closest hit), single thread, 1/8 of the rays
        std::vector<int> depth(nodes.size(), 0);
        for (size_t i = 0; i < nodes.size(); i++)
            for (int k = 0; k < 4; k++)
                if (nodes[i].c[k].child >= 0) depth[nodes[i].c[k].child] = depth[i] + 1;
        std::vector<uint64_t> per_level(64, 0), nodes_at(64, 0);
        for (size_t i = 0; i < nodes.size(); i++) nodes_at[depth[i]]++;
        std::mt19937_64 r(99);
        std::uniform_real_distribution<float> V(-1.f, 1.f);
        int nr = 0;
        for (int i = 0; i < n_rays; i += 8, nr++) {
            const int k = (int)(r() % (uint64_t)nt);
            const float *p = &tri[9 * k];
            float o[3], d[3], l2;
            for (int a = 0; a < 3; a++) o[a] = (p[a] + p[3 + a] + p[6 + a]) / 3.f;
            do {
                for (int a = 0; a < 3; a++) d[a] = V(r);
                l2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            } while (l2 > 1.f || l2 < 1e-4f);
            const float inv = 1.f / std::sqrt(l2);
            RayT<float> ray = make_ray(o[0], o[1], o[2], d[0] * inv, d[1] * inv, d[2] * inv, 1e-4f, Const<float>::inf());
            const float idx = safe_inv(ray.d.x), idy = safe_inv(ray.d.y), idz = safe_inv(ray.d.z);
            float tbest = ray.tmax;
            struct E { int32_t c; float k; } stk[128];
            int sp = 0;
            int32_t cur = root_child;
            for (;;) {
                if (cur >= 0) {
                    per_level[depth[cur]]++;
                    float key[4]; int32_t ch[4];
                    for (int j = 0; j < 4; j++) {
                        float tn;
                        bool ok = box_test(nodes[cur].c[j], ray.o, idx, idy, idz, ray.tmin, tbest, tn);
                        key[j] = ok ? tn : Const<float>::inf(), ch[j] = nodes[cur].c[j].child;
                    }
                    for (int a = 0; a < 4; a++) for (int b = a + 1; b < 4; b++) if (key[b] < key[a]) std::swap(key[a], key[b]), std::swap(ch[a], ch[b]);
                    for (int j = 3; j >= 1; j--) if (key[j] < Const<float>::inf()) stk[sp++] = E{ch[j], key[j]};
                    if (key[0] < Const<float>::inf()) { cur = ch[0]; continue; }
                } else if (cur != CHILD_EMPTY) {
                    const int first = leaf_first(cur), cnt = leaf_count(cur);
                    for (int q = 0; q < cnt; q++) {
                        float t, u, v;
                        if (tri_test(prims[first + q].a, ray, tbest, t, u, v)) tbest = t;
                    }
                }
                bool done = false;
                for (;;) { if (sp == 0) { done = true; break; } --sp; cur = stk[sp].c; if (stk[sp].k <= tbest) break; }
                if (done) break;
            }
        }
        std::printf("  level: nodes, visits/ray:");
        for (int l = 0; l < 64 && nodes_at[l]; l++) std::printf(" [%d] %llu %.2f", l, (unsigned long long)nodes_at[l], (double)per_level[l] / nr);
        std::printf("\n");
    }
    if (std::getenv("LAB_UNORDERED")) {  // the GPU's shadow-ray order: hit slot 0 next, the others popped last-first
        const int mode = std::atoi(std::getenv("LAB_UNORDERED"));  // 0 as built, 1 area descending, 2 prim count descending, 3 area ascending
        std::vector<Node4<float>> nn = nodes;
        if (mode > 0) {
            std::vector<double> cntp(nn.size(), 0);
            for (size_t i = nn.size(); i-- > 0;) {
                double tot = 0;
                for (int k = 0; k < 4; k++) {
                    const int32_t ch = nn[i].c[k].child;
                    if (ch == CHILD_EMPTY) continue;
                    tot += ch < 0 ? leaf_count(ch) : cntp[ch];
                }
                cntp[i] = tot;
            }
            for (auto &nd : nn) {
                double keyv[4];
                for (int k = 0; k < 4; k++) {
                    const auto &c = nd.c[k];
                    if (c.child == CHILD_EMPTY) { keyv[k] = -1e300; continue; }
                    const double ex = c.bmax[0] - c.bmin[0], ey = c.bmax[1] - c.bmin[1], ez = c.bmax[2] - c.bmin[2];
                    const double ar = ex * ey + ey * ez + ez * ex;
                    const double pc = c.child < 0 ? leaf_count(c.child) : cntp[c.child];
                    keyv[k] = mode == 1 ? ar : (mode == 2 ? pc : (mode == 3 ? -ar : pc / (ar + 1e-12)));
                }
                for (int a2 = 0; a2 < 4; a2++) for (int b2 = a2 + 1; b2 < 4; b2++) if (keyv[b2] > keyv[a2]) std::swap(keyv[a2], keyv[b2]), std::swap(nd.c[a2], nd.c[b2]);
                if (std::getenv("LAB_VISIT_DESC")) {  // slots [a0,a3,a2,a1]: the GPU order (first hit, then last-first) visits a0,a1,a2,a3
                    std::swap(nd.c[1], nd.c[3]);
                }
            }
        }
        std::mt19937_64 r(99);
        std::uniform_real_distribution<float> V(-1.f, 1.f);
        uint64_t un = 0, ul = 0; int nr = 0;
        for (int i = 0; i < n_rays; i += 4, nr++) {
            const int k = (int)(r() % (uint64_t)nt);
            const float *p = &tri[9 * k];
            float o[3], d[3], l2;
            for (int a = 0; a < 3; a++) o[a] = (p[a] + p[3 + a] + p[6 + a]) / 3.f;
            do { for (int a = 0; a < 3; a++) d[a] = V(r); l2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]; } while (l2 > 1.f || l2 < 1e-4f);
            const float inv = 1.f / std::sqrt(l2);
            RayT<float> ray = make_ray(o[0], o[1], o[2], d[0] * inv, d[1] * inv, d[2] * inv, 1e-4f, Const<float>::inf());
            const float idx = safe_inv(ray.d.x), idy = safe_inv(ray.d.y), idz = safe_inv(ray.d.z);
            int32_t stk[128]; int sp = 0; int32_t cur = root_child; bool found = false;
            for (;;) {
                if (cur >= 0) {
                    un++;
                    int32_t hc[4]; int nh = 0;
                    for (int j = 0; j < 4; j++) { float tn; if (box_test(nn[cur].c[j], ray.o, idx, idy, idz, ray.tmin, ray.tmax, tn)) hc[nh++] = nn[cur].c[j].child; }
                    if (nh) { for (int j = 1; j < nh; j++) stk[sp++] = hc[j]; cur = hc[0]; continue; }
                } else if (cur != CHILD_EMPTY) {
                    ul++;
                    const int first = leaf_first(cur), cnt = leaf_count(cur);
                    for (int q = 0; q < cnt; q++) { float t, u, v; if (tri_test(prims[first + q].a, ray, ray.tmax, t, u, v)) found = true; }
                    if (found) break;
                }
                if (sp == 0) break;
                cur = stk[--sp];
            }
        }
        std::printf("  GPU-order any-hit (slot mode %d): nodes %.2f leaves %.2f per ray\n", mode, (double)un / nr, (double)ul / nr);
    }
    if (std::getenv("LAB_WIDE8")) {
        // what would 8-wide nodes buy?  Collapse the same BVH2 to W-wide nodes (open the largest child until W slots
        // are used) and count node / leaf visits of ordered closest-hit traversal for the same rays.
        for (int Wd : {4, 8}) {
            struct WNode { float bmin[8][3], bmax[8][3]; int32_t child[8]; int n; };
            std::vector<WNode> wn;
            std::vector<int32_t> worder;
            const auto &n2 = builder.nodes();
            struct Item { int n2; };
            std::vector<Item> queue{{root}};
            wn.emplace_back();
            auto leafw = [&](const Bvh2Node &n) { int32_t f = (int32_t)worder.size(); for (int i = 0; i < n.count; i++) worder.push_back(n.first + i); return make_leaf(f, n.count); };
            for (size_t head = 0; head < queue.size(); head++) {
                int kids[8], nk = 0;
                kids[nk++] = n2[queue[head].n2].left, kids[nk++] = n2[queue[head].n2].right;
                while (nk < Wd) {
                    int best = -1; double ba = -1;
                    for (int i = 0; i < nk; i++) { const Bvh2Node &c = n2[kids[i]]; if (c.count > 0) continue; Bounds b; b.grow(c.bmin, c.bmax); if (b.half_area() > ba) ba = b.half_area(), best = i; }
                    if (best < 0) break;
                    const int open = kids[best]; kids[best] = n2[open].left; kids[nk++] = n2[open].right;
                }
                WNode w{}; w.n = nk;
                for (int i = 0; i < nk; i++) {
                    const Bvh2Node &c = n2[kids[i]];
                    for (int a = 0; a < 3; a++) w.bmin[i][a] = round_down(c.bmin[a], 0.f), w.bmax[i][a] = round_up(c.bmax[a], 0.f);
                    if (c.count > 0) w.child[i] = leafw(c);
                    else { w.child[i] = (int32_t)queue.size(); queue.push_back({kids[i]}); wn.emplace_back(); }
                }
                wn[head] = w;
            }
            std::vector<PrimRec<float>> wp(worder.size());
            for (size_t k = 0; k < worder.size(); k++) wp[k] = recs[bp[worder[k]].id];
            std::mt19937_64 r(99);
            std::uniform_real_distribution<float> V(-1.f, 1.f);
            uint64_t un = 0, ul = 0, ub = 0; int nr = 0;
            for (int i = 0; i < n_rays; i += 4, nr++) {
                const int k = (int)(r() % (uint64_t)nt);
                const float *p = &tri[9 * k];
                float o[3], d[3], l2;
                for (int a = 0; a < 3; a++) o[a] = (p[a] + p[3 + a] + p[6 + a]) / 3.f;
                do { for (int a = 0; a < 3; a++) d[a] = V(r); l2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]; } while (l2 > 1.f || l2 < 1e-4f);
                const float inv = 1.f / std::sqrt(l2);
                RayT<float> ray = make_ray(o[0], o[1], o[2], d[0] * inv, d[1] * inv, d[2] * inv, 1e-4f, Const<float>::inf());
                const float idx = safe_inv(ray.d.x), idy = safe_inv(ray.d.y), idz = safe_inv(ray.d.z);
                float tbest = ray.tmax;
                struct E { int32_t c; float k; } stk[256]; int sp = 0; int32_t cur = 0;
                for (;;) {
                    if (cur >= 0) {
                        un++;
                        const WNode &w = wn[cur];
                        int32_t hc[8]; float hk[8]; int nh = 0;
                        for (int j = 0; j < w.n; j++) {
                            NodeChild<float> c{}; for (int a = 0; a < 3; a++) c.bmin[a] = w.bmin[j][a], c.bmax[a] = w.bmax[j][a]; c.child = w.child[j];
                            float tn; ub++;
                            if (box_test(c, ray.o, idx, idy, idz, ray.tmin, tbest, tn)) hc[nh] = c.child, hk[nh] = tn, nh++;
                        }
                        for (int a2 = 0; a2 < nh; a2++) for (int b2 = a2 + 1; b2 < nh; b2++) if (hk[b2] < hk[a2]) std::swap(hk[a2], hk[b2]), std::swap(hc[a2], hc[b2]);
                        for (int j = nh - 1; j >= 1; j--) stk[sp++] = E{hc[j], hk[j]};
                        if (nh) { cur = hc[0]; continue; }
                    } else if (cur != CHILD_EMPTY) {
                        ul++;
                        const int first = leaf_first(cur), cnt = leaf_count(cur);
                        for (int q = 0; q < cnt; q++) { float t, u, v; if (tri_test(wp[first + q].a, ray, tbest, t, u, v)) tbest = t; }
                    }
                    bool done = false;
                    for (;;) { if (sp == 0) { done = true; break; } --sp; cur = stk[sp].c; if (stk[sp].k <= tbest) break; }
                    if (done) break;
                }
            }
            std::printf("  %d-wide: nodes %zu, closest-hit node visits %.2f, slot tests %.1f, leaf visits %.2f per ray\n", Wd, wn.size(), (double)un / nr, (double)ub / nr, (double)ul / nr);
        }
    }
    if (std::getenv("LAB_ANYORDER")) {
        for (int mode = 0; mode < 5; mode++) {
            std::mt19937_64 r(99);
            std::uniform_real_distribution<float> V(-1.f, 1.f);
            uint64_t un = 0, ul = 0, occ = 0; int nr = 0;
            for (int i = 0; i < n_rays; i += 4, nr++) {
                const int k = (int)(r() % (uint64_t)nt);
                const float *p = &tri[9 * k];
                float o[3];
                for (int a = 0; a < 3; a++) o[a] = (p[a] + p[3 + a] + p[6 + a]) / 3.f;
                const float lx = 0.3f * V(r), lz = 0.3f * V(r), ly = 0.99f;
                float dd[3] = {lx - o[0], ly - o[1], lz - o[2]};
                const float dist = std::sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]);
                RayT<float> ray = make_ray(o[0], o[1], o[2], dd[0] / dist, dd[1] / dist, dd[2] / dist, 1e-4f, (1.f - 1e-4f) * dist);
                const float idx = safe_inv(ray.d.x), idy = safe_inv(ray.d.y), idz = safe_inv(ray.d.z);
                int32_t stk[256]; int sp = 0; int32_t cur = root_child; bool found = false;
                for (;;) {
                    if (cur >= 0) {
                        un++;
                        int32_t hc[4]; float hk[4]; int nh = 0;
                        for (int j = 0; j < 4; j++) {
                            const NodeChild<float> &c = nodes[cur].c[j];
                            float tn;
                            if (!box_test(c, ray.o, idx, idy, idz, ray.tmin, ray.tmax, tn)) continue;
                            // exit distance
                            float t0x = (c.bmin[0] - ray.o.x) * idx, t1x = (c.bmax[0] - ray.o.x) * idx, t0y = (c.bmin[1] - ray.o.y) * idy, t1y = (c.bmax[1] - ray.o.y) * idy;
                            float t0z = (c.bmin[2] - ray.o.z) * idz, t1z = (c.bmax[2] - ray.o.z) * idz;
                            float tf = std::min(std::min(std::max(t0x, t1x), std::max(t0y, t1y)), std::min(std::max(t0z, t1z), ray.tmax));
                            float key = mode == 0 ? (float)j : mode == 1 ? -(tf - tn) : mode == 2 ? tn : mode == 3 ? -tn : -(tf - tn) / (1e-6f + (c.bmax[0]-c.bmin[0]) + (c.bmax[1]-c.bmin[1]) + (c.bmax[2]-c.bmin[2]));
                            hc[nh] = c.child, hk[nh] = key, nh++;
                        }
                        for (int a2 = 0; a2 < nh; a2++) for (int b2 = a2 + 1; b2 < nh; b2++) if (hk[b2] < hk[a2]) std::swap(hk[a2], hk[b2]), std::swap(hc[a2], hc[b2]);
                        if (nh) { for (int j = nh - 1; j >= 1; j--) stk[sp++] = hc[j]; cur = hc[0]; continue; }
                    } else if (cur != CHILD_EMPTY) {
                        ul++;
                        const int first = leaf_first(cur), cnt = leaf_count(cur);
                        for (int q = 0; q < cnt; q++) { float t, u, v; if (tri_test(prims[first + q].a, ray, ray.tmax, t, u, v)) found = true; }
                        if (found) { occ++; break; }
                    }
                    if (sp == 0) break;
                    cur = stk[--sp];
                }
            }
            std::printf("  light rays, any-hit order mode %d: nodes %.2f leaves %.2f per ray (occluded %.1f%%)\n", mode, (double)un / nr, (double)ul / nr, 100.0 * occ / nr);
        }
    }
    uint64_t a = 0, b = 0, c = 0, d = 0, h = 0, e = 0, f = 0;
    for (int t = 0; t < T; t++) a += cn[t], b += cp[t], c += sn[t], d += sp[t], h += hits[t], e += cl[t], f += sl[t];
    const double N = n_rays;
    { uint64_t oc = 0; for (int t = 0; t < T; t++) oc += occl[t]; std::printf("  any-hit rays occluded: %.1f%%\n", 100.0 * oc / N); }
    std::printf("  closest: nodes %.2f leaves %.2f prims %.2f per ray (hit %.1f%%) | any-hit: nodes %.2f leaves %.2f prims %.2f | est. cost (n + 3 l) closest %.1f any %.1f sum %.1f\n",
                a / N, e / N, b / N, 100.0 * h / N, c / N, f / N, d / N, (a + 3.0 * e) / N, (c + 3.0 * f) / N, (a + 3.0 * e + 0.5 * (c + 3.0 * f)) / N);
    return 0;
}
