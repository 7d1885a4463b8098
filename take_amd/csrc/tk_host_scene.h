// tk_host_scene.h — host-side preparation of a scene: validates a TakeSceneDesc, converts it to the
// R-typed arrays of tk_scene.h, builds the wide BVH.  The result is a set of plain host vectors; the C-ABI
// layer (tk_api.hip) uploads them to HBM.  Counterpart of the part of the reference's render() between
// parse_scene and the tile loop (src/render.cpp:37-50) plus build_bvh (src/scene.cpp:4-23).
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "take_hip.h"
#include "tk_bvh.h"
#include "tk_scene.h"

namespace tk {

template <class R> struct HostScene {
    std::vector<Node4<R>> nodes;
    std::vector<QNode4> qnodes;  // compressed copy of nodes (empty = not in use)
    std::vector<Node8<R>> nodes8; // the 8-wide tree at full width: the source of qnodes8 (checks only; never uploaded)
    std::vector<QNode8> qnodes8; // the 8-wide compressed tree (then nodes and qnodes are empty)
    int node_width = 4;          // 4 or 8
    float grid_lo[3] = {0, 0, 0}, grid_step[3] = {1, 1, 1};
    double q_inflation = 1.0;    // mean surface-area inflation of the compressed child boxes (1 = none)
    std::vector<PrimRec<R>> prims;
    int32_t root_child = CHILD_EMPTY;
    std::vector<ShapeInfo> shapes;
    std::vector<MeshInfo> meshes;
    std::vector<int32_t> face_idx;
    std::vector<R> normals, uvs, texels;
    std::vector<MaterialRec<R>> materials;
    std::vector<ImageInfo> images;
    std::vector<LightRec<R>> lights;
    std::vector<R> light_pmf, light_cdf;  // power-based light picking (integrator 3), see prepare_scene
    std::vector<InstTrace<R>> inst_trace;  // two-level scenes (TakeInstance): one record per placement
    std::vector<InstShade<R>> inst_shade;
    int64_t n_blas = 0, blas_nodes = 0, blas_prims = 0;  // stats: prototype trees and their total size
    EnvMap<R> env{-1, 0, 0, 1, 1, 0, 0, {R(0), R(0), R(0)}, nullptr, nullptr, nullptr, nullptr};  // pointers: view() / the uploader
    std::vector<R> env_marginal, env_conditional;
    std::vector<int32_t> env_guide_m, env_guide_c;
    R background[3];
    CameraRec<R> cam;
    WideBvhStats stats;
    int n_material_tags = 0;  // distinct material tags in use (1 => the material sort is skipped)
    uint32_t tag_mask = 0;    // bit t set: some material has tag t
    int single_tag = 0;       // the tag when n_material_tags == 1

    // pointers into the vectors above (a host "device scene" for tests/hostsim; tk_api.hip builds the real one)
    DeviceScene<R> view() const {
        DeviceScene<R> d{};
        d.nodes = nodes.data();
        d.qnodes = qnodes.empty() ? nullptr : qnodes.data();
        d.qnodes8 = qnodes8.empty() ? nullptr : qnodes8.data();
        for (int a = 0; a < 3; a++) d.grid_lo[a] = grid_lo[a], d.grid_step[a] = grid_step[a];
        d.prims = prims.data();
        d.root_child = root_child;
        d.n_nodes = (int32_t)stats.n_nodes;
        d.shapes = shapes.data();
        d.meshes = meshes.data();
        d.face_idx = face_idx.data();
        d.normals = normals.data();
        d.uvs = uvs.data();
        d.materials = materials.data();
        d.images = images.data();
        d.texels = texels.data();
        d.inst_trace = inst_trace.empty() ? nullptr : inst_trace.data();
        d.inst_shade = inst_shade.empty() ? nullptr : inst_shade.data();
        d.n_instances = (int32_t)inst_trace.size();
        d.lights = lights.data();
        d.light_pmf = light_pmf.data();
        d.light_cdf = light_cdf.data();
        d.env = env;
        d.env.marginal = env_marginal.data();
        d.env.conditional = env_conditional.data();
        d.env.guide_m = env_guide_m.data();
        d.env.guide_c = env_guide_c.data();
        d.n_lights = (int32_t)lights.size();
        d.n_shapes = (int32_t)shapes.size();
        for (int a = 0; a < 3; a++) d.background[a] = background[a];
        d.cam = cam;
        return d;
    }
};

// Exactly coincident primitives (identical geometry words: e.g. a duplicated face) tie in t AND in (u, v); the trace
// kernel then lets the larger primitive index win.  For that to mean the same thing in every tree, each group of
// coincident records inside [begin, end) gets its shape ids (with the shading side of the record) in ascending order of
// position — the geometry of the group's records is identical, so no box and no leaf changes.  The device builder needs
// no such pass: its Morton sort is stable, equal codes keep the shape order.
template <class R> inline void order_coincident(std::vector<PrimRec<R>> &prims, size_t begin, size_t end) {
    if (end - begin < 2) return;
    auto geom_hash = [](const PrimRec<R> &p) {
        uint64_t h = 0xcbf29ce484222325ull ^ (uint64_t)(p.meta & 0xff);
        const unsigned char *b = reinterpret_cast<const unsigned char *>(p.a);
        for (size_t i = 0; i < sizeof(p.a); i++) h = (h ^ b[i]) * 0x100000001b3ull;
        return h;
    };
    auto same_geom = [](const PrimRec<R> &x, const PrimRec<R> &y) {
        return (x.meta & 0xff) == (y.meta & 0xff) && std::memcmp(x.a, y.a, sizeof(x.a)) == 0;
    };
    std::vector<std::pair<uint64_t, uint32_t>> keys(end - begin);
    for (size_t i = begin; i < end; i++) keys[i - begin] = {geom_hash(prims[i]), (uint32_t)i};
    std::sort(keys.begin(), keys.end());
    std::vector<PrimRec<R>> group;
    for (size_t i = 0; i < keys.size();) {
        size_t j = i + 1;
        while (j < keys.size() && keys[j].first == keys[i].first) j++;
        if (j - i > 1) {
            // positions keys[i..j) ascend (sorted by (hash, index)); split the run into true geometry groups
            std::vector<char> done(j - i, 0);
            for (size_t a = i; a < j; a++) {
                if (done[a - i]) continue;
                std::vector<uint32_t> pos{keys[a].second};
                for (size_t b = a + 1; b < j; b++)
                    if (!done[b - i] && same_geom(prims[keys[a].second], prims[keys[b].second])) pos.push_back(keys[b].second), done[b - i] = 1;
                if (pos.size() > 1) {
                    group.clear();
                    for (uint32_t q : pos) group.push_back(prims[q]);
                    std::sort(group.begin(), group.end(), [](const PrimRec<R> &x, const PrimRec<R> &y) { return x.shape_id < y.shape_id; });
                    for (size_t q = 0; q < pos.size(); q++) prims[pos[q]] = group[q];
                }
            }
        }
        i = j;
    }
}

// Camera basis of src/render.cpp:37-44, in R arithmetic.
template <class R> inline void make_camera(const TakeCamera &c, CameraRec<R> &out) {
    const R vfov = R(c.vfov);
    const R theta = vfov / R(180) * Const<R>::PI;
    const R h = tk_tan(theta / R(2));
    out.viewport_height = R(2) * h;
    out.viewport_width = out.viewport_height / R(c.height) * R(c.width);
    Vec3<R> from{R(c.lookfrom[0]), R(c.lookfrom[1]), R(c.lookfrom[2])};
    Vec3<R> at{R(c.lookat[0]), R(c.lookat[1]), R(c.lookat[2])};
    Vec3<R> up{R(c.up[0]), R(c.up[1]), R(c.up[2])};
    Vec3<R> w = normalize(from - at);
    Vec3<R> u = normalize(cross(up, w));
    Vec3<R> v = cross(w, u);
    out.u[0] = u.x, out.u[1] = u.y, out.u[2] = u.z;
    out.v[0] = v.x, out.v[1] = v.y, out.v[2] = v.z;
    out.w[0] = w.x, out.w[1] = w.y, out.w[2] = w.z;
    out.lookfrom[0] = from.x, out.lookfrom[1] = from.y, out.lookfrom[2] = from.z;
    out.width = c.width;
    out.height = c.height;
}

// returns "" on success, else an error message (-> TAKE_E_INVALID)
// Run fn(begin, end) -> error string on `threads` contiguous chunks of [0, n); returns the error of the lowest chunk
// that failed ("" if none).  The per-shape loops below are independent per index.
template <class F> inline std::string for_chunks(int64_t n, int threads, F fn) {
    threads = (int)std::max<int64_t>(1, std::min<int64_t>(threads, n / 65536 + 1));
    if (threads == 1) return fn((int64_t)0, n);
    std::vector<std::string> err(threads);
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&, t] { err[t] = fn(n * t / threads, n * (t + 1) / threads); });
    for (auto &th : pool) th.join();
    for (auto &e : err)
        if (!e.empty()) return e;
    return "";
}

// Sampling tables of an environment map (EnvMap, tk_scene.h), in double: per texel f = luminance * sin(theta of the
// row centre) with luminance = 0.2126 r + 0.7152 g + 0.0722 b (negatives count as 0); cond[y][x] = sum of the row's
// f left of x / row sum (x / width for an all-black row), marg[y] = sum of the row sums above y / total.  Both
// start at 0 and end at exactly 1.  (The CPU checker under tests restates this recipe independently.)
inline bool env_tables(const double *rgb, int w, int h, std::vector<double> &marg, std::vector<double> &cond) {
    const double PI_D = 3.14159265358979323846;
    marg.assign((size_t)h + 1, 0.0);
    cond.assign((size_t)h * (w + 1), 0.0);
    std::vector<double> row_sum(h, 0.0);
    for (int y = 0; y < h; y++) {
        const double wy = std::sin(PI_D * (y + 0.5) / h);
        double *c = &cond[(size_t)y * (w + 1)];
        double run = 0;
        for (int x = 0; x < w; x++) {
            const double *t = rgb + 3 * ((size_t)y * w + x);
            const double lum = 0.2126 * t[0] + 0.7152 * t[1] + 0.0722 * t[2];
            c[x] = run;
            run += (lum > 0 ? lum : 0.0) * wy;
        }
        row_sum[y] = run;
        for (int x = 0; x < w; x++) c[x] = run > 0 ? c[x] / run : (double)x / w;
        c[w] = 1.0;
    }
    double total = 0;
    for (int y = 0; y < h; y++) {
        marg[y] = total;
        total += row_sum[y];
    }
    if (!(total > 0)) return false;
    for (int y = 0; y < h; y++) marg[y] /= total;
    marg[h] = 1.0;
    return true;
}

// The host-side trees of a scene, W-wide: the top-level tree over `bp` (shapes, plus one box per placement of a
// two-level scene), the prototype trees, their compressed form.  Out: `nodes` (full width), `qnodes` (compressed; empty
// when the grid is too coarse or on request), hs.root_child / stats / grid / inst_* / the prototypes' records in
// hs.prims, `order` (leaf order of the top-level tree's primitives as indices into bp).
template <class R, int W>
std::string build_host_trees(const TakeSceneDesc &d, HostScene<R> &hs, std::vector<BuildPrim> &bp, int leaf_size, int threads,
                             const std::string &fmt, int64_t ns, std::vector<int32_t> &order, std::vector<NodeW<R, W>> &nodes,
                             std::vector<QNodeW<W>> &qnodes) {
    // ---- two-level scenes (EXTENSION, TakeInstance): one tree per prototype mesh in object space ("BLAS"), their
    // nodes and primitive records appended behind the top-level tree's; an instance enters the top-level build
    // as one box and leaves it as an instance word.
    struct Blas {
        std::vector<NodeW<R, W>> nodes;
        std::vector<PrimRec<R>> prims;
        int32_t root_child = CHILD_EMPTY;
        double lo[3], hi[3];
        int depth = 0;
        // "re-braiding" (Benthin et al. 2017): the entries a placement contributes to the top-level build — subtrees
        // of the prototype's tree (child word local to this tree + object-space box), the root opened largest box
        // first until `braid` entries exist.  Built and MEASURED in round 3 on configs[4] (1000 placements x 10k
        // triangles, boxes of 0.16 overlapping in a 1.7 box): 1 / 4 / 8 / 16 / 32 / 64 entries per placement = 45.0 /
        // 40.7 / 39.0 / 36.8 / 35.1 / 34.1 Msamples/s — every entry a ray enters costs a 96-byte record, a transform
        // and a return marker, and the entries of one placement overlap (their boxes are the corners' boxes of
        // rotated object boxes); that outweighs the shorter descents.  Default 1 (TAKE_HIP_BRAID overrides).
        struct Entry {
            int32_t word;
            double lo[3], hi[3];
        };
        std::vector<Entry> entries;
    };
    const char *braid_env = std::getenv("TAKE_HIP_BRAID");
    const int braid = std::max(1, std::min(braid_env ? std::atoi(braid_env) : 1, 64));
    std::vector<Blas> blas;
    std::vector<int> blas_of_mesh(d.n_meshes, -1);
    std::vector<int> inst_blas;  // per (virtual) instance: its prototype tree
    int max_blas_depth = 0;
    int64_t shape_next = ns;
    // one InstTrace / InstShade per ENTRY of a placement ("virtual instances", placement-major: the tie rule on the
    // instance id keeps ordering placements as the caller numbered them)
    hs.inst_trace.clear(), hs.inst_shade.clear();
    for (int64_t i = 0; i < d.n_instances; i++) {
        const TakeInstance &in = d.instances[i];
        if (in.mesh_id < 0 || in.mesh_id >= d.n_meshes) return "instance " + std::to_string(i) + ": bad mesh id";
        if (in.material_id < -1 || in.material_id >= d.n_materials) return "instance " + std::to_string(i) + ": bad material id";
        const TakeMesh &m = d.meshes[in.mesh_id];
        if (m.n_faces <= 0) return "instance " + std::to_string(i) + ": empty prototype mesh";
        if (blas_of_mesh[in.mesh_id] < 0) {
            blas_of_mesh[in.mesh_id] = (int)blas.size();
            blas.emplace_back();
            Blas &b = blas.back();
            const MeshInfo &mi = hs.meshes[in.mesh_id];
            std::vector<PrimRec<R>> brecs(m.n_faces);
            std::vector<BuildPrim> bbp(m.n_faces);
            for (int a = 0; a < 3; a++) b.lo[a] = std::numeric_limits<double>::infinity(), b.hi[a] = -b.lo[a];
            for (int64_t f = 0; f < m.n_faces; f++) {
                PrimRec<R> &p = brecs[f];
                p = PrimRec<R>{};
                const int32_t *idx = m.indices + 3 * f;
                Vec3<R> v[3];
                for (int k = 0; k < 3; k++)
                    v[k] = {R(m.positions[3 * (int64_t)idx[k]]), R(m.positions[3 * (int64_t)idx[k] + 1]),
                            R(m.positions[3 * (int64_t)idx[k] + 2])};
                const Vec3<R> e1 = v[1] - v[0], e2 = v[2] - v[0];
                p.a[0] = v[0].x, p.a[1] = v[0].y, p.a[2] = v[0].z;
                p.a[3] = e1.x, p.a[4] = e1.y, p.a[5] = e1.z;
                p.a[6] = e2.x, p.a[7] = e2.y, p.a[8] = e2.z;
                p.shape_id = (int32_t)f;  // local: the shape id of a hit is InstShade::shape_base + this
                p.meta = PRIM_TRIANGLE | (hs.materials[m.material_id].tag << 8);
                p.material = m.material_id, p.area_light = -1, p.nidx = -1, p.mesh = in.mesh_id;
                if (mi.nbase >= 0 || mi.uvbase >= 0) p.nidx = mi.fbase + (int32_t)f, p.meta |= META_HAS_ATTR;
                for (int a = 0; a < 3; a++) {
                    const double x0 = (double)(&v[0].x)[a], x1 = (double)(&v[1].x)[a], x2 = (double)(&v[2].x)[a];
                    bbp[f].bmin[a] = std::min(x0, std::min(x1, x2));
                    bbp[f].bmax[a] = std::max(x0, std::max(x1, x2));
                    b.lo[a] = std::min(b.lo[a], bbp[f].bmin[a]), b.hi[a] = std::max(b.hi[a], bbp[f].bmax[a]);
                }
                bbp[f].id = (int32_t)f;
            }
            Bvh2Builder bb(bbp, leaf_size, threads);
            const int broot = bb.build();
            std::vector<int32_t> border;
            WideBvhStats bst;
            b.root_child = collapse_to_wide<R, W>(bb.nodes(), broot, b.nodes, border, bst);
            b.depth = bst.depth;
            b.prims.resize(border.size());
            for (size_t k = 0; k < border.size(); k++) b.prims[k] = brecs[bbp[border[k]].id];
            order_coincident(b.prims, 0, b.prims.size());
            max_blas_depth = std::max(max_blas_depth, b.depth);
            typename Blas::Entry root_e;
            root_e.word = b.root_child;
            for (int a = 0; a < 3; a++) root_e.lo[a] = b.lo[a], root_e.hi[a] = b.hi[a];
            b.entries.assign(1, root_e);
            while ((int)b.entries.size() < braid) {
                int best = -1;
                double best_area = -1;
                for (size_t e = 0; e < b.entries.size(); e++) {
                    if (b.entries[e].word < 0) continue;  // a leaf
                    Bounds bb2;
                    bb2.grow(b.entries[e].lo, b.entries[e].hi);
                    if (bb2.half_area() > best_area) best_area = bb2.half_area(), best = (int)e;
                }
                if (best < 0) break;
                const NodeW<R, W> &nd = b.nodes[b.entries[best].word];
                int nkids = 0;
                for (int j = 0; j < W; j++) nkids += nd.c[j].child != CHILD_EMPTY;
                if ((int)b.entries.size() - 1 + nkids > braid) break;
                b.entries.erase(b.entries.begin() + best);
                for (int j = 0; j < W; j++) {
                    if (nd.c[j].child == CHILD_EMPTY) continue;
                    typename Blas::Entry e;
                    e.word = nd.c[j].child;
                    for (int a = 0; a < 3; a++) e.lo[a] = (double)nd.c[j].bmin[a], e.hi[a] = (double)nd.c[j].bmax[a];
                    b.entries.push_back(e);
                }
            }
        }
        const int this_blas = blas_of_mesh[in.mesh_id];
        const Blas &b = blas[this_blas];
        // transforms: forward linear part for shading, inverse (in double) for the ray
        const double *M = in.xform;
        const double a00 = M[0], a01 = M[1], a02 = M[2], a10 = M[4], a11 = M[5], a12 = M[6], a20 = M[8], a21 = M[9], a22 = M[10];
        const double det = a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20);
        if (!(std::fabs(det) > 1e-300)) return "instance " + std::to_string(i) + ": singular transform";
        const double inv[9] = {(a11 * a22 - a12 * a21) / det, (a02 * a21 - a01 * a22) / det, (a01 * a12 - a02 * a11) / det,
                               (a12 * a20 - a10 * a22) / det, (a00 * a22 - a02 * a20) / det, (a02 * a10 - a00 * a12) / det,
                               (a10 * a21 - a11 * a20) / det, (a01 * a20 - a00 * a21) / det, (a00 * a11 - a01 * a10) / det};
        InstTrace<R> it{};
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) it.inv[4 * r + c] = R(inv[3 * r + c]);
            it.inv[4 * r + 3] = R(-(inv[3 * r] * M[3] + inv[3 * r + 1] * M[7] + inv[3 * r + 2] * M[11]));
        }
        InstShade<R> is{};
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) is.fwd[3 * r + c] = R(M[4 * r + c]);
        is.material = in.material_id >= 0 ? in.material_id : m.material_id;
        is.tag = hs.materials[is.material].tag;
        is.shape_base = (int32_t)shape_next;
        shape_next += m.n_faces;
        if (shape_next >= (int64_t)1 << 31) return "too many instanced faces for 32-bit shape ids";
        // World box of the placement, padded for the rounding of the transformed ray (the specification is the
        // flattened geometry to fp rounding, see take_hip.h).  The TIGHT box: the prototype's vertices under the
        // transform — the object box's eight corners under a rotation span up to sqrt(3) times the extent per axis (5x
        // the volume for a round cloud), and every ray that enters a placement's box pays a descent from the prototype's
        // root (round 2: instanced 42 vs flattened 66 Msamples/s on 1000 x 10k triangles).  Beyond 4e8 vertex
        // transforms in total: the corners' box intersected with the box of the bounding sphere's image.
        BuildPrim ib;
        ib.id = -(int32_t)(i + 1);
        for (int a = 0; a < 3; a++) ib.bmin[a] = std::numeric_limits<double>::infinity(), ib.bmax[a] = -ib.bmin[a];
        double mag = 0;
        if ((double)m.n_vertices * (double)d.n_instances <= 4e8) {
            for (int64_t vtx = 0; vtx < m.n_vertices; vtx++) {
                const double px = m.positions[3 * vtx], py = m.positions[3 * vtx + 1], pz = m.positions[3 * vtx + 2];
                for (int a = 0; a < 3; a++) {
                    const double w = M[4 * a] * px + M[4 * a + 1] * py + M[4 * a + 2] * pz + M[4 * a + 3];
                    ib.bmin[a] = std::min(ib.bmin[a], w), ib.bmax[a] = std::max(ib.bmax[a], w);
                }
            }
            for (int a = 0; a < 3; a++) mag = std::max(mag, std::max(std::fabs(ib.bmin[a]), std::fabs(ib.bmax[a])));
        } else {
            for (int c8 = 0; c8 < 8; c8++) {
                const double px = (c8 & 1) ? b.hi[0] : b.lo[0], py = (c8 & 2) ? b.hi[1] : b.lo[1], pz = (c8 & 4) ? b.hi[2] : b.lo[2];
                for (int a = 0; a < 3; a++) {
                    const double w = M[4 * a] * px + M[4 * a + 1] * py + M[4 * a + 2] * pz + M[4 * a + 3];
                    ib.bmin[a] = std::min(ib.bmin[a], w), ib.bmax[a] = std::max(ib.bmax[a], w);
                }
            }
            // image of the object box's bounding sphere: centre M c, radius r * ||L||_F per axis row
            double c[3], r2 = 0;
            for (int a = 0; a < 3; a++) c[a] = 0.5 * (b.lo[a] + b.hi[a]), r2 += 0.25 * (b.hi[a] - b.lo[a]) * (b.hi[a] - b.lo[a]);
            const double r = std::sqrt(r2);
            for (int a = 0; a < 3; a++) {
                const double wc = M[4 * a] * c[0] + M[4 * a + 1] * c[1] + M[4 * a + 2] * c[2] + M[4 * a + 3];
                const double wr = r * std::sqrt(M[4 * a] * M[4 * a] + M[4 * a + 1] * M[4 * a + 1] + M[4 * a + 2] * M[4 * a + 2]) * (1.0 + 1e-12);
                ib.bmin[a] = std::max(ib.bmin[a], wc - wr), ib.bmax[a] = std::min(ib.bmax[a], wc + wr);
            }
            for (int a = 0; a < 3; a++) mag = std::max(mag, std::max(std::fabs(ib.bmin[a]), std::fabs(ib.bmax[a])));
        }
        const double pad = mag * (sizeof(R) == 4 ? 4e-6 : 1e-13);
        // one top-level entry per braid entry of the prototype: the entry's object box under the transform (its eight
        // corners), clipped to the placement's box, padded
        for (const typename Blas::Entry &e : b.entries) {
            BuildPrim eb;
            const int64_t vid = (int64_t)hs.inst_trace.size();
            if (vid >= ((int64_t)1 << 28)) return "too many instance entries";
            eb.id = -(int32_t)(vid + 1);
            for (int a = 0; a < 3; a++) eb.bmin[a] = std::numeric_limits<double>::infinity(), eb.bmax[a] = -eb.bmin[a];
            for (int c8 = 0; c8 < 8; c8++) {
                const double px = (c8 & 1) ? e.hi[0] : e.lo[0], py = (c8 & 2) ? e.hi[1] : e.lo[1], pz = (c8 & 4) ? e.hi[2] : e.lo[2];
                for (int a = 0; a < 3; a++) {
                    const double w = M[4 * a] * px + M[4 * a + 1] * py + M[4 * a + 2] * pz + M[4 * a + 3];
                    eb.bmin[a] = std::min(eb.bmin[a], w), eb.bmax[a] = std::max(eb.bmax[a], w);
                }
            }
            for (int a = 0; a < 3; a++) {
                eb.bmin[a] = std::max(eb.bmin[a], ib.bmin[a]) - pad, eb.bmax[a] = std::min(eb.bmax[a], ib.bmax[a]) + pad;
                if (eb.bmin[a] > eb.bmax[a]) eb.bmin[a] = eb.bmax[a] = 0.5 * (eb.bmin[a] + eb.bmax[a]);  // (rounding of a flat entry)
            }
            bp.push_back(eb);
            InstTrace<R> ie = it;
            ie.root_child = e.word;  // local to the prototype's tree for now: made global below
            hs.inst_trace.push_back(ie);
            hs.inst_shade.push_back(is);
            inst_blas.push_back(this_blas);
        }
    }
    const int64_t n_virtual = (int64_t)hs.inst_trace.size();

    Bvh2Builder builder(bp, leaf_size, threads);
    const int root = builder.build();
    hs.root_child = collapse_to_wide<R, W>(builder.nodes(), root, nodes, order, hs.stats, d.n_instances > 0 ? &bp : nullptr);
    int32_t top_prims = (int32_t)order.size();
    // append the prototype trees: node indices and leaf ranges become global
    const size_t top_nodes = nodes.size();
    std::vector<size_t> blas_node_base(blas.size()), blas_prim_base(blas.size());
    {
        size_t nb = top_nodes, pb = (size_t)top_prims;
        for (size_t k = 0; k < blas.size(); k++) {
            blas_node_base[k] = nb, blas_prim_base[k] = pb;
            nb += blas[k].nodes.size(), pb += blas[k].prims.size();
        }
        if (pb >= ((size_t)1 << 28)) return "too many primitive records for the 4-wide leaf encoding (2^28)";
        nodes.reserve(nb);
        for (size_t k = 0; k < blas.size(); k++) {
            auto fix = [&](int32_t c) -> int32_t {
                if (c == CHILD_EMPTY) return c;
                if (c >= 0) return c + (int32_t)blas_node_base[k];
                return make_leaf(leaf_first(c) + (int32_t)blas_prim_base[k], leaf_count(c));
            };
            for (NodeW<R, W> nd : blas[k].nodes) {
                for (int j = 0; j < W; j++) nd.c[j].child = fix(nd.c[j].child);
                nodes.push_back(nd);
            }
            blas[k].root_child = fix(blas[k].root_child);
        }
        hs.n_blas = (int64_t)blas.size(), hs.blas_nodes = (int64_t)(nb - top_nodes), hs.blas_prims = (int64_t)(pb - top_prims);
    }
    hs.stats.n_nodes = (int64_t)nodes.size();
    hs.stats.depth += max_blas_depth;  // the traversal stack holds both levels (+ one return marker)
    for (int64_t i = 0; i < n_virtual; i++) {  // entry words: local to the prototype's tree -> global
        const int k = inst_blas[i];
        const int32_t c = hs.inst_trace[i].root_child;
        hs.inst_trace[i].root_child = c == CHILD_EMPTY ? c : (c >= 0 ? c + (int32_t)blas_node_base[k]
                                                                     : make_leaf(leaf_first(c) + (int32_t)blas_prim_base[k], leaf_count(c)));
    }

    qnodes.clear();
    {
        // Both precisions traverse the 64-byte compressed nodes unless the 15-bit grid is too coarse for the
        // geometry (child boxes growing by more than 10 % in area on average: a scene mixing scales by >1e4), or on request
        // (TAKE_HIP_NODES=wide / =q16: A/B runs).  In f64 scenes only the box tests use them (conservative, so
        // exactness is not at stake); hits are decided by the double-precision primitive tests.  Every tree of a
        // two-level scene has its own grid (the top-level one is the scene's, a prototype's is in its InstTrace).
        if (fmt != "wide" && !nodes.empty()) {
            std::vector<NodeW<R, W>> part(nodes.begin(), nodes.begin() + top_nodes);
            std::vector<QNodeW<W>> q;
            hs.q_inflation = top_nodes ? quantise_nodes<R, W>(part, q, hs.grid_lo, hs.grid_step) : 1.0;
            qnodes = q;
            std::vector<std::array<float, 6>> grids(blas.size());
            for (size_t k = 0; k < blas.size(); k++) {
                part.assign(nodes.begin() + blas_node_base[k], nodes.begin() + blas_node_base[k] + blas[k].nodes.size());
                float glo[3], gst[3];
                const double infl = part.empty() ? 1.0 : quantise_nodes<R, W>(part, q, glo, gst);
                if (part.empty()) q.clear(), glo[0] = glo[1] = glo[2] = 0, gst[0] = gst[1] = gst[2] = 1;
                hs.q_inflation = std::max(hs.q_inflation, infl);
                qnodes.insert(qnodes.end(), q.begin(), q.end());
                grids[k] = {glo[0], glo[1], glo[2], gst[0], gst[1], gst[2]};
            }
            for (int64_t i = 0; i < n_virtual; i++)
                for (int a = 0; a < 3; a++)
                    hs.inst_trace[i].grid_lo[a] = grids[inst_blas[i]][a], hs.inst_trace[i].grid_step[a] = grids[inst_blas[i]][3 + a];
            if (hs.q_inflation > 1.10 && fmt != "q16") qnodes.clear();
        }
    }
    // primitive records: the top-level tree's in leaf order, then each prototype's
    hs.prims.resize((size_t)top_prims + (size_t)hs.blas_prims);
    for (size_t k = 0; k < blas.size(); k++)
        std::copy(blas[k].prims.begin(), blas[k].prims.end(), hs.prims.begin() + blas_prim_base[k]);
    return "";
}

// what prepare_scene leaves to the device (TAKE_BUILDER_DEVICE_LBVH): PREP_ALL = nothing (records, host SAH tree);
// PREP_RECORDS = the tree (records in shape order); PREP_TABLES = the tree AND the primitive records (made by
// tk_build_gpu.h::k_make_prims straight from the caller's mesh arrays: at 10M triangles the host loop that writes
// 640 MB of records was 470 of the 570 ms of scene_create) — only validation and the small tables happen here.
enum PrepMode { PREP_ALL = 0, PREP_RECORDS = 1, PREP_TABLES = 2 };
template <class R>
std::string prepare_scene(const TakeSceneDesc &d, int max_leaf, int threads, HostScene<R> &hs, int mode = PREP_ALL,
                          bool burley_lobes = false) {
    const bool build_bvh = mode == PREP_ALL;
    const bool host_records = mode != PREP_TABLES;
    if (d.camera.width <= 0 || d.camera.height <= 0) return "camera width/height must be positive";
    if (d.n_shapes < 0 || d.n_meshes < 0 || d.n_spheres < 0 || d.n_lights < 0 || d.n_materials < 0 || d.n_images < 0)
        return "negative count in scene description";
    if (d.n_shapes > 0 && (!d.shape_kind || !d.shape_ref || !d.shape_face || !d.shape_area_light))
        return "shape arrays missing";
    if (d.n_shapes >= (int64_t)1 << 28) return "too many shapes for the 4-wide leaf encoding (2^28)";
    if (d.n_instances < 0 || (d.n_instances > 0 && !d.instances)) return "instance array missing";
    if (d.n_instances >= (int64_t)1 << 28) return "too many instances";
    if (d.n_instances > 0 && !build_bvh) return "instanced scenes are built by the host builder";
    make_camera<R>(d.camera, hs.cam);
    for (int a = 0; a < 3; a++) hs.background[a] = R(d.background[a]);

    // meshes: concatenate face indices; normals / uvs only for meshes that carry them
    hs.meshes.resize(d.n_meshes);
    int64_t nf = 0, nn = 0, nuv = 0;
    for (int i = 0; i < d.n_meshes; i++) {
        const TakeMesh &m = d.meshes[i];
        if (m.n_vertices < 0 || m.n_faces < 0 || (m.n_faces > 0 && (!m.positions || !m.indices)))
            return "mesh " + std::to_string(i) + ": missing arrays";
        if (m.material_id < 0 || m.material_id >= d.n_materials) return "mesh " + std::to_string(i) + ": bad material id";
        hs.meshes[i] = MeshInfo{(int32_t)nf, m.normals ? (int32_t)nn : -1, m.uvs ? (int32_t)nuv : -1, m.material_id};
        nf += m.n_faces;
        if (m.normals) nn += m.n_vertices;
        if (m.uvs) nuv += m.n_vertices;
    }
    if (nf >= (int64_t)1 << 30 || nn >= (int64_t)1 << 30 || nuv >= (int64_t)1 << 30) return "mesh arrays too large";
    hs.face_idx.resize(3 * (size_t)nf);
    hs.normals.resize(3 * (size_t)nn);
    hs.uvs.resize(2 * (size_t)nuv);
    for (int i = 0; i < d.n_meshes; i++) {
        const TakeMesh &m = d.meshes[i];
        const MeshInfo &mi = hs.meshes[i];
        const std::string ierr = for_chunks(3 * m.n_faces, threads, [&](int64_t k0, int64_t k1) -> std::string {
            for (int64_t k = k0; k < k1; k++) {
                const int32_t vi = m.indices[k];
                if (vi < 0 || vi >= m.n_vertices) return "mesh " + std::to_string(i) + ": vertex index out of range";
                hs.face_idx[3 * (size_t)mi.fbase + k] = vi;
            }
            return "";
        });
        if (!ierr.empty()) return ierr;
        if (m.normals)
            for (int64_t k = 0; k < 3 * m.n_vertices; k++) hs.normals[3 * (size_t)mi.nbase + k] = R(m.normals[k]);
        if (m.uvs)
            for (int64_t k = 0; k < 2 * m.n_vertices; k++) hs.uvs[2 * (size_t)mi.uvbase + k] = R(m.uvs[k]);
    }

    // materials, textures
    hs.materials.resize(d.n_materials);
    bool tag_used[TAKE_MAT_COUNT] = {false};
    for (int i = 0; i < d.n_materials; i++) {
        const TakeMaterial &m = d.materials[i];
        if (m.tag < 0 || m.tag >= TAKE_MAT_COUNT) return "material " + std::to_string(i) + ": unknown tag";
        const TakeTexture &t = m.reflectance;
        if (t.kind == 1 && (t.image_id < 0 || t.image_id >= d.n_images))
            return "material " + std::to_string(i) + ": bad texture image id";
        MaterialRec<R> &o = hs.materials[i];
        o.tag = m.tag;
        // TakeBuildOpts.burley_lobes: the reference's Disney alternatives (Lambert clones upstream) get the real lobes
        if (burley_lobes && m.tag >= TAKE_MAT_DISNEY_METAL && m.tag <= TAKE_MAT_DISNEY_BSDF) o.tag = m.tag + 5;
        o.tex_kind = t.kind;
        o.tex_image = t.image_id;
        o.pad = 0;
        for (int a = 0; a < 3; a++) o.color[a] = R(t.value[a]);
        o.uscale = R(t.uscale), o.vscale = R(t.vscale), o.uoffset = R(t.uoffset), o.voffset = R(t.voffset);
        o.p0 = R(m.param[0]);
        o.p1 = R(m.param[1]);
        for (int k = 0; k < TAKE_MATERIAL_PARAMS; k++) o.p[k] = R(m.param[k]);
        if (o.tag >= TAKE_MAT_BURLEY_METAL && o.tag <= TAKE_MAT_BURLEY_BSDF) {
            // the Burley lobes take square roots and logarithms of their parameters: a value outside the model's range
            // (every parameter in [0, 1], an index of refraction > 0) would render NaN pixels — refuse it here
            auto unit = [&](int k) { return m.param[k] >= 0.0 && m.param[k] <= 1.0; };  // (false for NaN)
            bool ok = true;
            int eta_at = -1;
            switch (o.tag) {
                case TAKE_MAT_BURLEY_METAL: ok = unit(0) && unit(1); break;
                case TAKE_MAT_BURLEY_GLASS: ok = unit(0) && unit(1), eta_at = 2; break;
                case TAKE_MAT_BURLEY_CLEARCOAT:
                case TAKE_MAT_BURLEY_SHEEN: ok = unit(0); break;
                default:
                    for (int k = 0; k < 11; k++) ok = ok && unit(k);
                    eta_at = 11;
            }
            if (eta_at >= 0) ok = ok && m.param[eta_at] > 0.0 && std::isfinite(m.param[eta_at]);
            if (!ok) return "material " + std::to_string(i) + ": Burley parameter outside [0, 1] (or eta <= 0)";
        }
        tag_used[o.tag] = true;
    }
    hs.n_material_tags = 0;
    hs.tag_mask = 0;
    for (int t = 0; t < TAKE_MAT_COUNT; t++)
        if (tag_used[t]) {
            hs.n_material_tags++;
            hs.tag_mask |= 1u << t;
            hs.single_tag = t;
        }
    hs.images.resize(d.n_images);
    int64_t ntex = 0;
    for (int i = 0; i < d.n_images; i++) {
        if (d.images[i].width <= 0 || d.images[i].height <= 0 || !d.images[i].data) return "image: bad dimensions";
        hs.images[i] = ImageInfo{d.images[i].width, d.images[i].height, ntex};
        ntex += (int64_t)d.images[i].width * d.images[i].height;
    }
    hs.texels.resize(3 * (size_t)ntex);
    for (int i = 0; i < d.n_images; i++) {
        const int64_t n = (int64_t)d.images[i].width * d.images[i].height * 3;
        for (int64_t k = 0; k < n; k++) hs.texels[3 * (size_t)hs.images[i].offset + k] = R(d.images[i].data[k]);
    }

    // shapes -> primitive records (shape order for now) + build boxes
    const int64_t ns = d.n_shapes;
    hs.shapes.clear();
    if (host_records) hs.shapes.resize(ns);
    std::vector<PrimRec<R>> recs(host_records ? ns : 0);
    std::vector<BuildPrim> bp(build_bvh ? ns : 0);
    std::string shape_err = for_chunks(ns, threads, [&](int64_t i_begin, int64_t i_end) -> std::string {
    for (int64_t i = i_begin; i < i_end; i++) {
        if (!host_records) {  // validation only: the records are made on the device
            const int32_t al = d.shape_area_light[i];
            if (al < -1 || al >= d.n_lights) return "shape " + std::to_string(i) + ": bad area_light id";
            if (d.shape_kind[i] == 0) {
                const int32_t si = d.shape_ref[i];
                if (si < 0 || si >= d.n_spheres) return "shape " + std::to_string(i) + ": bad sphere index";
                if (d.spheres[si].material_id < 0 || d.spheres[si].material_id >= d.n_materials) return "sphere: bad material id";
            } else if (d.shape_kind[i] == 1) {
                const int32_t mi = d.shape_ref[i], fi = d.shape_face[i];
                if (mi < 0 || mi >= d.n_meshes) return "shape " + std::to_string(i) + ": bad mesh index";
                if (fi < 0 || fi >= d.meshes[mi].n_faces) return "shape " + std::to_string(i) + ": bad face index";
            } else {
                return "shape " + std::to_string(i) + ": unknown kind";
            }
            continue;
        }
        PrimRec<R> &p = recs[i];
        p = PrimRec<R>{};
        p.shape_id = (int32_t)i;
        const int32_t al = d.shape_area_light[i];
        if (al < -1 || al >= d.n_lights) return "shape " + std::to_string(i) + ": bad area_light id";
        int material;
        if (d.shape_kind[i] == 0) {
            const int32_t si = d.shape_ref[i];
            if (si < 0 || si >= d.n_spheres) return "shape " + std::to_string(i) + ": bad sphere index";
            const TakeSphere &s = d.spheres[si];
            if (s.material_id < 0 || s.material_id >= d.n_materials) return "sphere: bad material id";
            material = s.material_id;
            for (int a = 0; a < 3; a++) p.a[a] = R(s.center[a]);
            p.a[3] = R(s.radius);
            hs.shapes[i] = ShapeInfo{-(1 + si), 0, material, al};
            if (build_bvh)
            for (int a = 0; a < 3; a++) {  // bounds of src/scene.cpp:8-10, from the R-typed values
                bp[i].bmin[a] = (double)(p.a[a] - p.a[3]);
                bp[i].bmax[a] = (double)(p.a[a] + p.a[3]);
                // an R-rounded centre-radius can round inwards by an ulp: widen in double
                bp[i].bmin[a] = std::min(bp[i].bmin[a], (double)p.a[a] - (double)p.a[3]);
                bp[i].bmax[a] = std::max(bp[i].bmax[a], (double)p.a[a] + (double)p.a[3]);
            }
            p.meta = PRIM_SPHERE;
        } else if (d.shape_kind[i] == 1) {
            const int32_t mi = d.shape_ref[i], fi = d.shape_face[i];
            if (mi < 0 || mi >= d.n_meshes) return "shape " + std::to_string(i) + ": bad mesh index";
            const TakeMesh &m = d.meshes[mi];
            if (fi < 0 || fi >= m.n_faces) return "shape " + std::to_string(i) + ": bad face index";
            material = m.material_id;
            const int32_t *idx = m.indices + 3 * (int64_t)fi;
            Vec3<R> v[3];
            for (int k = 0; k < 3; k++)
                v[k] = {R(m.positions[3 * (int64_t)idx[k]]), R(m.positions[3 * (int64_t)idx[k] + 1]),
                        R(m.positions[3 * (int64_t)idx[k] + 2])};
            const Vec3<R> e1 = v[1] - v[0], e2 = v[2] - v[0];
            p.a[0] = v[0].x, p.a[1] = v[0].y, p.a[2] = v[0].z;
            p.a[3] = e1.x, p.a[4] = e1.y, p.a[5] = e1.z;
            p.a[6] = e2.x, p.a[7] = e2.y, p.a[8] = e2.z;
            hs.shapes[i] = ShapeInfo{mi, fi, material, al};
            if (build_bvh)
            for (int a = 0; a < 3; a++) {
                const double x0 = (double)(&v[0].x)[a], x1 = (double)(&v[1].x)[a], x2 = (double)(&v[2].x)[a];
                bp[i].bmin[a] = std::min(x0, std::min(x1, x2));
                bp[i].bmax[a] = std::max(x0, std::max(x1, x2));
            }
            p.meta = PRIM_TRIANGLE;
        } else {
            return "shape " + std::to_string(i) + ": unknown kind";
        }
        p.meta |= hs.materials[material].tag << 8;
        if (build_bvh) bp[i].id = (int32_t)i;
    }
    return "";
    });
    if (!shape_err.empty()) return shape_err;

    // lights
    hs.env = EnvMap<R>{-1, 0, 0, 1, 1, 0, 0, {R(0), R(0), R(0)}, nullptr, nullptr, nullptr, nullptr};
    hs.env_marginal.clear(), hs.env_conditional.clear(), hs.env_guide_m.clear(), hs.env_guide_c.clear();
    hs.lights.resize(d.n_lights);
    for (int i = 0; i < d.n_lights; i++) {
        const TakeLight &l = d.lights[i];
        LightRec<R> &o = hs.lights[i];
        o = LightRec<R>{};
        o.kind = l.kind;
        o.shape_id = -1;
        for (int a = 0; a < 3; a++) o.intensity[a] = R(l.intensity[a]);
        if (l.kind == 0) continue;
        if (l.kind == 2) {  // environment map (extension): shape_id = image index, intensity = scale
            if (hs.env.light >= 0) return "more than one environment-map light";
            if (l.shape_id < 0 || l.shape_id >= d.n_images) return "environment map: bad image index";
            const TakeImage3 &im = d.images[l.shape_id];
            std::vector<double> marg, cond;
            if (!env_tables(im.data, im.width, im.height, marg, cond)) return "environment map: no positive luminance";
            hs.env.light = i;
            hs.env.width = im.width, hs.env.height = im.height;
            hs.env.texel0 = hs.images[l.shape_id].offset;
            for (int a = 0; a < 3; a++) hs.env.scale[a] = R(l.intensity[a]);
            hs.env_marginal.assign(marg.begin(), marg.end());
            hs.env_conditional.assign(cond.begin(), cond.end());
            // guide tables over the R-typed CDFs (what the device searches): interval of k / G for k = 0..G
            auto guide = [](const R *cdf, int n, int g, int32_t *out) {
                for (int k = 0; k <= g; k++) {
                    const R xi = R(k) / R(g);
                    int lo = 0, hi = n;
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (cdf[mid] <= xi) lo = mid;
                        else hi = mid;
                    }
                    out[k] = lo;
                }
            };
            // one guide entry per ~quarter row / per column: the bisection that remains is 0-1 steps (round 3: with
            // 256 / 64 entries it was 2-3 and ~5 dependent loads; shade kernel -4 % on the bench workload, same samples)
            auto pow2_at_least = [](int v, int cap) {
                int p = 1;
                while (p < v && p < cap) p <<= 1;
                return p;
            };
            int gm = pow2_at_least(4 * im.height, ENV_GUIDE_M_MAX), gc = pow2_at_least(im.width, ENV_GUIDE_C_MAX);
            if (const char *e = std::getenv("TAKE_HIP_ENV_GUIDE")) {  // "<m>,<c>", powers of two
                int a = 0, b = 0;
                if (std::sscanf(e, "%d,%d", &a, &b) == 2 && a > 0 && b > 0 && !(a & (a - 1)) && !(b & (b - 1)) && a <= (1 << 16) && b <= (1 << 16)) gm = a, gc = b;
            }
            hs.env.n_guide_m = gm, hs.env.n_guide_c = gc;
            hs.env_guide_m.resize(gm + 1);
            guide(hs.env_marginal.data(), im.height, gm, hs.env_guide_m.data());
            hs.env_guide_c.resize((size_t)im.height * (gc + 1));
            for (int y = 0; y < im.height; y++)
                guide(hs.env_conditional.data() + (size_t)y * (im.width + 1), im.width, gc,
                      hs.env_guide_c.data() + (size_t)y * (gc + 1));
            continue;
        }
        if (l.kind != 1) return "light " + std::to_string(i) + ": unknown kind";
        if (l.shape_id < 0 || l.shape_id >= ns) return "light " + std::to_string(i) + ": bad shape id";
        o.shape_id = l.shape_id;
        // (read from the description, not from the records: in PREP_TABLES mode there are none on the host)
        const ShapeInfo si = d.shape_kind[l.shape_id] == 0 ? ShapeInfo{-(1 + d.shape_ref[l.shape_id]), 0, 0, 0}
                                                            : ShapeInfo{d.shape_ref[l.shape_id], d.shape_face[l.shape_id], 0, 0};
        if (si.mesh < 0) {
            o.is_sphere = 1;
            const TakeSphere &sp = d.spheres[-si.mesh - 1];
            for (int a = 0; a < 3; a++) o.v[a] = R(sp.center[a]);
            o.v[3] = R(sp.radius);
        } else {
            const TakeMesh &m = d.meshes[si.mesh];
            // the reference reads mesh.normals.at() when sampling a triangle light and throws on an emissive
            // mesh without vertex normals (src/shape.cpp:163-165): reject it up front
            if (!m.normals) return "light " + std::to_string(i) + ": emissive mesh has no vertex normals";
            const int32_t *idx = m.indices + 3 * (int64_t)si.face;
            for (int k = 0; k < 3; k++)
                for (int a = 0; a < 3; a++) {
                    o.v[3 * k + a] = R(m.positions[3 * (int64_t)idx[k] + a]);
                    o.n[3 * k + a] = R(m.normals[3 * (int64_t)idx[k] + a]);
                }
        }
    }

    // Power-based light picking (src/light.cpp:9-30).  The reference reads Scene::lights_power_pmf / _cdf but never
    // fills them; filled here from its light_power(): luminance(intensity) * area * pi for an area light, 0 otherwise;
    // pmf = power / total, cdf = running sum from 0 (n + 1 entries) — in R arithmetic, in light order (the recipe
    // the golden `ptpow` tables were made with: the test harness applies it to the reference's own Scene).
    {
        std::vector<R> power;
        R total = R(0);
        for (const LightRec<R> &l : hs.lights) {
            R p = R(0);
            if (l.kind == 1) {
                R area;
                if (l.is_sphere) {
                    area = R(4) * Const<R>::PI * l.v[3] * l.v[3];
                } else {
                    const Vec3<R> v0 = {l.v[0], l.v[1], l.v[2]}, v1 = {l.v[3], l.v[4], l.v[5]}, v2 = {l.v[6], l.v[7], l.v[8]};
                    area = length(cross(v1 - v0, v2 - v0)) / R(2);
                }
                p = (l.intensity[0] * R(0.212671) + l.intensity[1] * R(0.715160) + l.intensity[2] * R(0.072169)) * area * Const<R>::PI;
            }
            power.push_back(p);
            total += p;
        }
        hs.light_pmf.clear();
        hs.light_cdf.assign(1, R(0));
        for (R p : power) {
            hs.light_pmf.push_back(p / total);
            hs.light_cdf.push_back(hs.light_cdf.back() + p / total);
        }
    }

    // acceleration structure (build_bvh == false: the caller builds it on the device from the primitives in shape
    // order — tk_build_gpu.h — and only the records are prepared here)
    std::vector<int32_t> order;
    if (!build_bvh) {
        order.resize((size_t)ns);
        for (size_t k = 0; k < order.size(); k++) order[k] = (int32_t)k;
        hs.nodes.clear();
        hs.qnodes.clear(), hs.qnodes8.clear(), hs.nodes8.clear();
        hs.node_width = 4;
        hs.inst_trace.clear(), hs.inst_shade.clear();
        hs.n_blas = hs.blas_nodes = hs.blas_prims = 0;
        hs.root_child = CHILD_EMPTY;
        hs.stats = WideBvhStats{};
        hs.stats.n_prims = (int64_t)order.size();
    } else {
        // default leaf size 2: on triangle soups the tighter leaf boxes save more primitive tests than the extra
        // interior nodes cost (1M soup: 48.5 node + 10.8 primitive tests per ray vs 42.6 + 42.6 with 4 per leaf)
        // one primitive per leaf: with one ray per lane a leaf's primitives are tested one after the other, so a second
        // one doubles the leaf step of the whole wave (measured 1 / 2 / 3 / 4 per leaf: 75.6 / 71.0 / 61.0 / 50.8 Msamples/s)
        const int leaf_size = max_leaf > 0 ? max_leaf : 1;
        const char *fmt_env = std::getenv("TAKE_HIP_NODES");
        const std::string fmt = fmt_env ? fmt_env : "";

        // Tree width.  The 4-wide compressed tree is the default.  TAKE_HIP_NODES=q8 selects the 8-wide one (128-byte
        // nodes, octant-ordered slots; built and measured in round 3: a third fewer node visits per ray — 32.9
        // instead of 49.2 on the 1M soup — but eight 16-byte loads per lane and visit instead of four, and the vector
        // L1 charges per load instruction and distinct line: closest hit +16 %, shadow rays +25 % slower);
        // =wide / =q16 select full-width / forced-compressed 4-wide nodes (A/B runs), and a scene the 15-bit grid is
        // too coarse for falls back to full-width 4-wide nodes.
        const bool want8 = fmt == "q8";
        hs.node_width = 4;
        std::string terr;
        if (want8) {
            std::vector<Node8<R>> &nodes8 = hs.nodes8;
            terr = build_host_trees<R, 8>(d, hs, bp, leaf_size, threads, fmt, ns, order, nodes8, hs.qnodes8);
            if (!terr.empty()) return terr;
            if (!hs.qnodes8.empty() || nodes8.empty()) {
                hs.node_width = 8;
                hs.nodes.clear(), hs.qnodes.clear();
            } else {
                // (grid too coarse: the full-width fall-back is 4-wide; the placements' boxes are appended again)
                bp.erase(std::remove_if(bp.begin(), bp.end(), [](const BuildPrim &b) { return b.id < 0; }), bp.end());
            }
        }
        if (hs.node_width == 4) {
            hs.qnodes8.clear(), hs.nodes8.clear();
            terr = build_host_trees<R, 4>(d, hs, bp, leaf_size, threads, fmt, ns, order, hs.nodes, hs.qnodes);
            if (!terr.empty()) return terr;
        }
    }
    if (!host_records) {
        hs.prims.clear();
        return "";
    }
    if (hs.prims.size() < order.size()) hs.prims.resize(order.size());  // (two-level scenes: the prototypes' records follow)
    for_chunks((int64_t)order.size(), threads, [&](int64_t k_begin, int64_t k_end) -> std::string {
    for (int64_t k = k_begin; k < k_end; k++) {
        hs.prims[k] = recs[build_bvh ? bp[order[k]].id : order[k]];
        const ShapeInfo &si = hs.shapes[hs.prims[k].shape_id];
        PrimRec<R> &pr = hs.prims[k];
        pr.material = si.material, pr.area_light = si.area_light, pr.nidx = -1, pr.mesh = si.mesh;
        if (si.mesh >= 0) {
            const MeshInfo &mi = hs.meshes[si.mesh];
            if (mi.nbase >= 0 || mi.uvbase >= 0) {
                pr.nidx = mi.fbase + si.face;
                pr.meta |= META_HAS_ATTR;
            }
        }
    }
    return "";
    });
    if (build_bvh) order_coincident(hs.prims, 0, order.size());  // (device build: the stable Morton sort does it)
    if ((hs.node_width - 1) * hs.stats.depth + 2 > (hs.node_width == 8 ? MAX_STACK_ENTRIES_W8 : MAX_STACK_ENTRIES)) return "BVH too deep for the traversal stack";
    return "";
}

}  // namespace tk
