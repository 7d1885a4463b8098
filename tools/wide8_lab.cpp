// wide8_lab.cpp — DEVELOPMENT TOOL (not product, not test): what would 8-wide nodes cost per ray under the child
// orderings a GPU step can afford?  Builds the product's BVH2 for the bench's soup, collapses it W-wide and counts
// node visits / slot tests / leaf visits / max stack depth per closest-hit and any-hit ray for:
//   sort     full sort by entry distance + pop-time culling (the 4-wide kernel's scheme)
//   near+oct nearest child exact, the others pushed in octant order, pop-time culling
//   oct      octant order only (slot ^ octant mask), pop-time culling
//   oct-nc   octant order, no pop-time culling (one stack entry per node, CWBVH style)
//   g++ -std=c++17 -O2 -I include -I take_amd/csrc tools/wide8_lab.cpp -pthread -o /tmp/wide8_lab && /tmp/wide8_lab 1000000
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include <vector>

#include "take_hip.h"
#include "tk_bvh.h"
#include "tk_traverse.h"

using namespace tk;

struct WNode {
    float bmin[8][3], bmax[8][3];
    int32_t child[8];
};

int main(int argc, char **argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 1000000;
    const int n_rays = argc > 2 ? std::atoi(argv[2]) : 200000;
    const float jitter = n <= 200000 ? 0.02f : 0.008f;
    std::mt19937_64 rng(1234);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<float> tri;
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
    for (int i = 0; i < nt; i++) {
        const float *t = &tri[9 * i];
        for (int a = 0; a < 3; a++) {
            bp[i].bmin[a] = std::min(t[a], std::min(t[3 + a], t[6 + a]));
            bp[i].bmax[a] = std::max(t[a], std::max(t[3 + a], t[6 + a]));
            recs[i].a[a] = t[a], recs[i].a[3 + a] = t[3 + a] - t[a], recs[i].a[6 + a] = t[6 + a] - t[a];
        }
        bp[i].id = i;
        recs[i].shape_id = i, recs[i].meta = PRIM_TRIANGLE;
    }
    Bvh2Builder builder(bp, 1, 8);
    const int root = builder.build();
    const auto &n2 = builder.nodes();
    const int slot_mode = std::getenv("LAB_SLOTS") ? std::atoi(std::getenv("LAB_SLOTS")) : 1;  // 0 as opened, 1 octant auction
    for (int Wd : {2, 3, 4, 5, 6, 8}) {
        std::vector<WNode> wn;
        std::vector<int32_t> worder;
        std::vector<int> queue{root};
        wn.emplace_back();
        int64_t filled = 0;
        for (size_t head = 0; head < queue.size(); head++) {
            int kids[8], nk = 0;
            kids[nk++] = n2[queue[head]].left, kids[nk++] = n2[queue[head]].right;
            while (nk < Wd) {
                int best = -1;
                double ba = -1;
                for (int i = 0; i < nk; i++) {
                    const Bvh2Node &c = n2[kids[i]];
                    if (c.count > 0) continue;
                    Bounds b;
                    b.grow(c.bmin, c.bmax);
                    if (b.half_area() > ba) ba = b.half_area(), best = i;
                }
                if (best < 0) break;
                const int open = kids[best];
                kids[best] = n2[open].left;
                kids[nk++] = n2[open].right;
            }
            filled += nk;
            // slot assignment: greedy auction of children to the 2^3 (8-wide) / first 4 octant slots
            int slot_of[8];
            for (int i = 0; i < 8; i++) slot_of[i] = i;
            if (slot_mode == 1) {
                const Bvh2Node &pn = n2[queue[head]];
                double pc[3];
                for (int a = 0; a < 3; a++) pc[a] = 0.5 * (pn.bmin[a] + pn.bmax[a]);
                double cost[8][8];
                for (int i = 0; i < nk; i++) {
                    const Bvh2Node &c = n2[kids[i]];
                    for (int s = 0; s < Wd; s++) {
                        double v = 0;
                        for (int a = 0; a < 3; a++) {
                            const double d = 0.5 * (c.bmin[a] + c.bmax[a]) - pc[a];
                            // 4-wide: slots use two "virtual" axes = the two largest extents of the parent
                            const int bit = (s >> a) & 1;
                            v += bit ? d : -d;
                        }
                        cost[i][s] = v;
                    }
                }
                bool cu[8] = {false}, su[8] = {false};
                for (int r = 0; r < nk; r++) {
                    int bi = -1, bs = -1;
                    double bv = -1e300;
                    for (int i = 0; i < nk; i++)
                        if (!cu[i])
                            for (int s = 0; s < Wd; s++)
                                if (!su[s] && cost[i][s] > bv) bv = cost[i][s], bi = i, bs = s;
                    cu[bi] = true, su[bs] = true, slot_of[bi] = bs;
                }
            }
            WNode w{};
            for (int s = 0; s < 8; s++) {
                w.child[s] = CHILD_EMPTY;
                for (int a = 0; a < 3; a++) w.bmin[s][a] = Const<float>::inf(), w.bmax[s][a] = -Const<float>::inf();
            }
            for (int i = 0; i < nk; i++) {
                const Bvh2Node &c = n2[kids[i]];
                const int s = slot_of[i];
                for (int a = 0; a < 3; a++) w.bmin[s][a] = round_down(c.bmin[a], 0.f), w.bmax[s][a] = round_up(c.bmax[a], 0.f);
                if (c.count > 0) {
                    int32_t f = (int32_t)worder.size();
                    for (int q = 0; q < c.count; q++) worder.push_back(c.first + q);
                    w.child[s] = make_leaf(f, c.count);
                } else {
                    w.child[s] = (int32_t)queue.size();
                    queue.push_back(kids[i]);
                    wn.emplace_back();
                }
            }
            wn[head] = w;
        }
        std::vector<PrimRec<float>> wp(worder.size());
        for (size_t k = 0; k < worder.size(); k++) wp[k] = recs[bp[worder[k]].id];
        std::printf("%d-wide: %zu nodes, %.2f children per node\n", Wd, wn.size(), (double)filled / wn.size());
        for (int anyhit = 0; anyhit < 2; anyhit++)
            for (int mode = 0; mode < 4; mode++) {
                const int T = 8;
                std::vector<uint64_t> un(T, 0), ul(T, 0), ub(T, 0), md(T, 0), hits(T, 0), dsum(T, 0);
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
                            const float idx = safe_inv(ray.d.x), idy = safe_inv(ray.d.y), idz = safe_inv(ray.d.z);
                            const int oct = (ray.d.x < 0 ? 1 : 0) | (ray.d.y < 0 ? 2 : 0) | (ray.d.z < 0 ? 4 : 0);
                            float tbest = ray.tmax;
                            struct E {
                                int32_t c;
                                float k;
                            } stk[512];
                            int sp = 0, maxsp = 0;
                            int32_t cur = 0;
                            bool found = false;
                            for (;;) {
                                if (cur >= 0) {
                                    un[t]++;
                                    const WNode &w = wn[cur];
                                    int32_t hc[8];
                                    float hk[8];
                                    int hp[8], nh = 0;
                                    for (int j = 0; j < Wd; j++) {
                                        if (w.child[j] == CHILD_EMPTY) continue;
                                        NodeChild<float> c{};
                                        for (int a = 0; a < 3; a++) c.bmin[a] = w.bmin[j][a], c.bmax[a] = w.bmax[j][a];
                                        c.child = w.child[j];
                                        float tn;
                                        ub[t]++;
                                        if (box_test(c, ray.o, idx, idy, idz, ray.tmin, tbest, tn)) hc[nh] = c.child, hk[nh] = tn, hp[nh] = (Wd == 8 ? j ^ oct : j), nh++;
                                    }
                                    // order the hits: index 0 = visited next
                                    auto swp = [&](int a, int b) { std::swap(hk[a], hk[b]), std::swap(hc[a], hc[b]), std::swap(hp[a], hp[b]); };
                                    if (mode == 0) {
                                        for (int a2 = 0; a2 < nh; a2++)
                                            for (int b2 = a2 + 1; b2 < nh; b2++)
                                                if (hk[b2] < hk[a2]) swp(a2, b2);
                                    } else {
                                        for (int a2 = 0; a2 < nh; a2++)
                                            for (int b2 = a2 + 1; b2 < nh; b2++)
                                                if (hp[b2] < hp[a2]) swp(a2, b2);
                                        if (mode == 1 && nh > 1) {  // nearest exact first, the rest stay in octant order
                                            int m = 0;
                                            for (int a2 = 1; a2 < nh; a2++)
                                                if (hk[a2] < hk[m]) m = a2;
                                            for (int a2 = m; a2 > 0; a2--) swp(a2, a2 - 1);
                                        }
                                    }
                                    for (int j = nh - 1; j >= 1; j--) stk[sp++] = E{hc[j], hk[j]};
                                    maxsp = std::max(maxsp, sp);
                                    if (nh) {
                                        cur = hc[0];
                                        continue;
                                    }
                                } else if (cur != CHILD_EMPTY) {
                                    ul[t]++;
                                    const int first = leaf_first(cur), cnt = leaf_count(cur);
                                    for (int q = 0; q < cnt; q++) {
                                        float tt, u, v;
                                        if (tri_test(wp[first + q].a, ray, tbest, tt, u, v)) {
                                            found = true;
                                            if (!anyhit) tbest = tt;
                                        }
                                    }
                                    if (anyhit && found) break;
                                }
                                bool done = false;
                                for (;;) {
                                    if (sp == 0) {
                                        done = true;
                                        break;
                                    }
                                    --sp;
                                    cur = stk[sp].c;
                                    if (mode == 3 || stk[sp].k <= tbest) break;
                                }
                                if (done) break;
                            }
                            hits[t] += found;
                            md[t] = std::max<uint64_t>(md[t], maxsp);
                            dsum[t] += maxsp;
                        }
                    });
                for (auto &th : pool) th.join();
                uint64_t a = 0, b = 0, c = 0, m = 0, h = 0, ds = 0;
                for (int t = 0; t < T; t++) a += un[t], b += ul[t], c += ub[t], m = std::max(m, md[t]), h += hits[t], ds += dsum[t];
                static const char *names[4] = {"sort    ", "near+oct", "oct     ", "oct-nc  "};
                std::printf("  %s %s: node visits %.2f, slot tests %.1f, leaf visits %.2f per ray; stack max %llu mean-max %.1f; hit %.1f%%\n",
                            anyhit ? "any-hit" : "closest", names[mode], (double)a / n_rays, (double)c / n_rays, (double)b / n_rays,
                            (unsigned long long)m, (double)ds / n_rays, 100.0 * h / n_rays);
            }
    }
    return 0;
}
