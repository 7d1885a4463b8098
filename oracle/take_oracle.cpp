// take_oracle.cpp — TEST INFRASTRUCTURE.  C entry points (ctypes) over take_oracle.hpp.
// Used by tests/ (checker), __graft_entry__.smoke() (checker) and bench.py's cpu_baseline leg only.
// Table entry points mirror the commands of oracle/ref_harness.cpp column for column, so that
// tests/test_oracle_golden.py can compare this restatement with the compiled reference's outputs.
#include "take_oracle.hpp"

#include <chrono>

using namespace oracle;

namespace {
struct Handle {
    int precision;
    Scene<double> d;
    Scene<float> f;
    PathCounters counters;
};
inline V3<double> P3(const double *p) { return {p[0], p[1], p[2]}; }
inline void put3(double *o, V3<double> v) {
    o[0] = v.x;
    o[1] = v.y;
    o[2] = v.z;
}
template <class R> void put_isect(double *o, const std::optional<Intersection<R>> &h) {
    for (int i = 0; i < 15; i++) o[i] = 0.0;
    if (!h) return;
    o[0] = 1.0;
    o[1] = h->t;
    o[2] = h->pos.x, o[3] = h->pos.y, o[4] = h->pos.z;
    o[5] = h->geo_normal.x, o[6] = h->geo_normal.y, o[7] = h->geo_normal.z;
    o[8] = h->shading_normal.x, o[9] = h->shading_normal.y, o[10] = h->shading_normal.z;
    o[11] = h->uv.x, o[12] = h->uv.y;
    o[13] = h->material_id;
    o[14] = h->area_light_id;
}
Image3<double> table_image() {  // the fixed 5x4 image of ref_harness.cpp "material"/"texture"
    Image3<double> img;
    img.width = 5;
    img.height = 4;
    img.data.resize(20);
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 5; x++)
            img.data[y * 5 + x] = {0.1 + 0.15 * x + 0.01 * y, 0.9 - 0.2 * y + 0.02 * x, 0.3 + 0.05 * ((x * 3 + y * 7) % 5)};
    return img;
}
}  // namespace

extern "C" {

int oracle_abi_version(void) { return 1; }
void oracle_pt_mt2(void *p, int max_depth, const double *in, int64_t n, double *out, int integrator);
double oracle_render2(void *p, int spp, int max_depth, int rng_mode, uint64_t seed, int threads, double *out,
                      int with_counters, int integrator);

void oracle_tab_random_real(const double *in, int64_t n, double *out) {
    for (int64_t r = 0; r < n; r++) {
        MtRng rng((unsigned)in[r]);
        for (int i = 0; i < 64; i++) out[r * 64 + i] = rng.real();
    }
}
void oracle_tab_slab(const double *in, int64_t n, double *out) {
    for (int64_t r = 0; r < n; r++) {
        const double *p = in + 14 * r;
        BBox<double> b{P3(p), P3(p + 3)};
        Ray<double> ray{P3(p + 6), P3(p + 9), p[12], p[13]};
        out[r] = intersect_box(b, ray) ? 1.0 : 0.0;
    }
}
void oracle_tab_tri(const double *in, int64_t n, double *out) {
    for (int64_t r = 0; r < n; r++) {
        const double *p = in + 34 * r;
        std::vector<Mesh<double>> meshes(1);
        Mesh<double> &m = meshes[0];
        m.material_id = 3;
        m.positions = {P3(p), P3(p + 3), P3(p + 6)};
        m.indices = {0, 1, 2};
        if (p[17] != 0) m.normals = {P3(p + 19), P3(p + 22), P3(p + 25)};
        if (p[18] != 0) m.uvs = {{p[28], p[29]}, {p[30], p[31]}, {p[32], p[33]}};
        Shape<double> tri{};
        tri.kind = 1;
        tri.material_id = 3;
        tri.area_light_id = 7;
        Ray<double> ray{P3(p + 9), P3(p + 12), p[15], p[16]};
        put_isect(out + 15 * r, intersect_triangle(meshes, tri, 0, ray));
    }
}
void oracle_tab_sphere(const double *in, int64_t n, double *out) {
    for (int64_t r = 0; r < n; r++) {
        const double *p = in + 12 * r;
        Shape<double> s{};
        s.kind = 0;
        s.material_id = 5;
        s.area_light_id = -1;
        s.center = P3(p);
        s.radius = p[3];
        Ray<double> ray{P3(p + 4), P3(p + 7), p[10], p[11]};
        put_isect(out + 15 * r, intersect_sphere(s, 0, ray));
    }
}
void oracle_tab_to_world(const double *in, int64_t n, double *out) {
    for (int64_t r = 0; r < n; r++) put3(out + 3 * r, to_world(P3(in + 6 * r), P3(in + 6 * r + 3)));
}
void oracle_tab_hemicos(const double *in, int64_t n, double *out) {
    for (int64_t r = 0; r < n; r++) {
        MtRng rng((unsigned)in[r]);
        put3(out + 4 * r, sample_hemisphere_cos<double>(rng));
        out[4 * r + 3] = rng.real();
    }
}
void oracle_tab_material(const double *in, int64_t n, double *out) {
    Scene<double> sc;
    sc.images.push_back(table_image());
    for (int64_t r = 0; r < n; r++) {
        const double *p = in + 27 * r;
        double *o = out + 14 * r;
        Material<double> m{};
        m.tag = (int)p[0];
        if (p[22] != 0)
            m.reflectance = {1, 0, {0, 0, 0}, p[23], p[24], p[25], p[26]};
        else
            m.reflectance = {0, 0, P3(p + 1), 1, 1, 0, 0};
        m.p0 = p[4];
        m.p1 = p[5];
        Intersection<double> v{};
        v.geo_normal = P3(p + 6);
        v.shading_normal = P3(p + 9);
        v.uv = {p[12], p[13]};
        v.t = 1;
        v.area_light_id = -1;
        V3<double> dir_in = P3(p + 14), dir_out = P3(p + 17);
        MtRng rng((unsigned)p[21]);
        auto rec = sample_bsdf(m, dir_in, v, sc, rng);
        double next = rng.real();
        for (int i = 0; i < 14; i++) o[i] = 0.0;
        if (rec) {
            o[0] = 1.0;
            put3(o + 1, rec->dir_out);
            o[4] = rec->pdf;
            put3(o + 6, eval_bsdf(m, dir_in, *rec, v, sc));
        }
        o[5] = next;
        o[9] = get_bsdf_pdf(m, dir_in, dir_out, v, sc);
        SampleRecord<double> given{dir_out, p[20]};
        put3(o + 10, eval_bsdf(m, dir_in, given, v, sc));
        o[13] = rec ? get_bsdf_pdf(m, dir_in, rec->dir_out, v, sc) : 0.0;
    }
}
// Burley lobes (tags 12..16; no upstream table: parity unpinned).  Row: tag, colour[3], param[12], geo_normal[3],
// shading_normal[3], dir_in[3], dir_out[3], mt19937 seed, back_face -> the 14 output columns of the material table.
void oracle_tab_burley(const double *in, int64_t n, double *out) {
    Scene<double> sc;
    for (int64_t r = 0; r < n; r++) {
        const double *p = in + 30 * r;
        double *o = out + 14 * r;
        Material<double> m{};
        m.tag = (int)p[0];
        m.reflectance = {0, 0, P3(p + 1), 1, 1, 0, 0};
        for (int k = 0; k < 12; k++) m.q[k] = p[4 + k];
        m.p0 = m.q[0], m.p1 = m.q[1];
        Intersection<double> v{};
        v.geo_normal = P3(p + 16);
        v.shading_normal = P3(p + 19);
        v.t = 1;
        v.area_light_id = -1;
        v.back_face = p[29] != 0;
        V3<double> dir_in = P3(p + 22), dir_out = P3(p + 25);
        MtRng rng((unsigned)p[28]);
        auto rec = sample_bsdf(m, dir_in, v, sc, rng);
        double next = rng.real();
        for (int i = 0; i < 14; i++) o[i] = 0.0;
        if (rec) {
            o[0] = 1.0;
            put3(o + 1, rec->dir_out);
            o[4] = rec->pdf;
            put3(o + 6, eval_bsdf(m, dir_in, *rec, v, sc));
            o[13] = get_bsdf_pdf(m, dir_in, rec->dir_out, v, sc);
        }
        o[5] = next;
        o[9] = get_bsdf_pdf(m, dir_in, dir_out, v, sc);
        SampleRecord<double> given{dir_out, 0.0};
        put3(o + 10, eval_bsdf(m, dir_in, given, v, sc));
    }
}
void oracle_tab_texture(const double *in, int64_t n, double *out) {
    Scene<double> sc;
    sc.images.push_back(table_image());
    for (int64_t r = 0; r < n; r++) {
        const double *p = in + 6 * r;
        Texture<double> t{1, 0, {0, 0, 0}, p[2], p[3], p[4], p[5]};
        put3(out + 3 * r, eval_texture(t, V2<double>{p[0], p[1]}, sc));
    }
}
void oracle_tab_light(const double *in, int64_t n, double *out) {
    for (int64_t r = 0; r < n; r++) {
        const double *p = in + 30 * r;
        double *o = out + 9 * r;
        Scene<double> sc;
        Shape<double> s{};
        if (p[0] == 0) {
            s.kind = 0;
            s.center = P3(p + 1);
            s.radius = p[4];
        } else {
            Mesh<double> m;
            m.material_id = 0;
            m.positions = {P3(p + 1), P3(p + 4), P3(p + 7)};
            m.indices = {0, 1, 2};
            m.normals = {P3(p + 10), P3(p + 13), P3(p + 16)};
            sc.meshes.push_back(m);
            s.kind = 1;
        }
        sc.shapes.push_back(s);
        sc.lights.push_back({1, 0, {1, 2, 3}, {0, 0, 0}});
        V3<double> ref = P3(p + 19);
        MtRng rng((unsigned)p[22]);
        PointAndNormal<double> pn = sample_on_shape(sc, sc.shapes[0], ref, rng);
        put3(o, pn.position);
        put3(o + 3, pn.normal);
        o[6] = rng.real();
        o[7] = get_light_pdf(sc, 0, pn, ref);
        o[8] = get_light_pdf(sc, 0, PointAndNormal<double>{P3(p + 24), P3(p + 27)}, ref);
    }
}
// out must hold 2 + 9 * (2n - 1) doubles
void oracle_tab_bvh(const double *in, int64_t n, double *out) {
    std::vector<BBoxWithID<double>> boxes;
    for (int64_t r = 0; r < n; r++) boxes.push_back({BBox<double>{P3(in + 6 * r), P3(in + 6 * r + 3)}, (int)r});
    std::vector<BVHNode<double>> nodes;
    int root = construct_bvh(boxes, nodes);
    out[0] = root;
    out[1] = (double)nodes.size();
    double *o = out + 2;
    for (auto &nd : nodes) {
        put3(o, nd.box.p_min);
        put3(o + 3, nd.box.p_max);
        o[6] = nd.left;
        o[7] = nd.right;
        o[8] = nd.prim;
        o += 9;
    }
}

// ---------------------------------------------------------------- scene-level entry points
void *oracle_scene_create(const TakeSceneDesc *desc, int precision, double ray_eps) {
    Handle *h = new Handle();
    h->precision = precision;
    if (precision == TAKE_PRECISION_F64) {
        scene_from_desc(*desc, h->d);
        h->d.ray_eps = ray_eps > 0 ? ray_eps : 1e-7;
        build_bvh(h->d);
        fill_light_power(h->d);
    } else {
        scene_from_desc(*desc, h->f);
        h->f.ray_eps = ray_eps > 0 ? (float)ray_eps : 1e-4f;
        build_bvh(h->f);
        fill_light_power(h->f);
    }
    return h;
}
void oracle_scene_destroy(void *p) { delete (Handle *)p; }

// rays: n x 8 (org3 dir3 tmin tmax); out: n x 19 = isect(15) occluded shape_id bu bv
void oracle_isect(void *p, const double *rays, int64_t n, double *out) {
    Handle *h = (Handle *)p;
    for (int64_t r = 0; r < n; r++) {
        const double *q = rays + 8 * r;
        double *o = out + 19 * r;
        if (h->precision == TAKE_PRECISION_F64) {
            Ray<double> ray{P3(q), P3(q + 3), q[6], q[7]};
            auto hit = scene_intersect(h->d, ray);
            put_isect(o, hit);
            o[15] = scene_occluded(h->d, ray) ? 1.0 : 0.0;
            o[16] = hit ? hit->shape_id : -1;
            o[17] = hit ? hit->bu : 0;
            o[18] = hit ? hit->bv : 0;
        } else {
            Ray<float> ray{{(float)q[0], (float)q[1], (float)q[2]}, {(float)q[3], (float)q[4], (float)q[5]}, (float)q[6], (float)q[7]};
            auto hit = scene_intersect(h->f, ray);
            put_isect(o, hit);
            o[15] = hit ? 1.0 : 0.0;
            o[16] = hit ? hit->shape_id : -1;
            o[17] = hit ? hit->bu : 0;
            o[18] = hit ? hit->bv : 0;
        }
    }
}
// brute-force closest hit over all shapes (no BVH): the ground truth of the trace-hook tests.
// out: n x 4 = shape_id t bu bv
void oracle_isect_brute(void *p, const double *rays, int64_t n, double *out) {
    Handle *h = (Handle *)p;
    for (int64_t r = 0; r < n; r++) {
        const double *q = rays + 8 * r;
        double *o = out + 4 * r;
        o[0] = -1, o[1] = o[2] = o[3] = 0;
        if (h->precision == TAKE_PRECISION_F64) {
            Ray<double> ray{P3(q), P3(q + 3), q[6], q[7]};
            double best = K<double>::inf();
            for (int i = 0; i < (int)h->d.shapes.size(); i++) {
                auto v = intersect_shape(h->d, i, ray);
                if (v && v->t < best) best = v->t, o[0] = i, o[1] = v->t, o[2] = v->bu, o[3] = v->bv;
            }
        } else {
            Ray<float> ray{{(float)q[0], (float)q[1], (float)q[2]}, {(float)q[3], (float)q[4], (float)q[5]}, (float)q[6], (float)q[7]};
            float best = K<float>::inf();
            for (int i = 0; i < (int)h->f.shapes.size(); i++) {
                auto v = intersect_shape(h->f, i, ray);
                if (v && v->t < best) best = v->t, o[0] = i, o[1] = v->t, o[2] = v->bu, o[3] = v->bv;
            }
        }
    }
}
// in: n x 7 (org3 dir3 seed) -> out: n x 4 (radiance3, next random_real); double + mt19937 only
void oracle_pt_mt(void *p, int max_depth, const double *in, int64_t n, double *out) { oracle_pt_mt2(p, max_depth, in, n, out, 0); }
// the same with the integrator selectable (0 path_tracing, 1 raw, 2 one-sample MIS, 3 one-sample MIS by power)
void oracle_pt_mt2(void *p, int max_depth, const double *in, int64_t n, double *out, int integrator) {
    Handle *h = (Handle *)p;
    for (int64_t r = 0; r < n; r++) {
        const double *q = in + 7 * r;
        Ray<double> ray{P3(q), P3(q + 3), K<double>::EPS, K<double>::inf()};
        MtRng rng((unsigned)q[6]);
        put3(out + 4 * r, integrate(integrator, h->d, ray, rng, max_depth));
        out[4 * r + 3] = rng.real();
    }
}
// out: H*W*3 doubles, image order.  Returns seconds spent in the tile loop.
double oracle_render(void *p, int spp, int max_depth, int rng_mode, uint64_t seed, int threads, double *out,
                     int with_counters) {
    return oracle_render2(p, spp, max_depth, rng_mode, seed, threads, out, with_counters, 0);
}
double oracle_render2(void *p, int spp, int max_depth, int rng_mode, uint64_t seed, int threads, double *out,
                      int with_counters, int integrator) {
    Handle *h = (Handle *)p;
    h->counters = PathCounters{};
    auto t0 = std::chrono::steady_clock::now();
    if (h->precision == TAKE_PRECISION_F64) {
        render(h->d, spp, max_depth, rng_mode, seed, threads, out, with_counters ? &h->counters : nullptr, integrator);
    } else {
        size_t n = (size_t)h->f.camera.width * h->f.camera.height * 3;
        std::vector<float> tmp(n);
        render(h->f, spp, max_depth, RNG_COUNTER, seed, threads, tmp.data(), with_counters ? &h->counters : nullptr, integrator);
        for (size_t i = 0; i < n; i++) out[i] = tmp[i];
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
// out[9]: closest rays, node visits, box tests, prim tests, shadow rays, node visits, box tests, prim tests, bounces
void oracle_get_counters(void *p, uint64_t *out) {
    Handle *h = (Handle *)p;
    const PathCounters &c = h->counters;
    out[0] = c.closest.rays, out[1] = c.closest.node_visits, out[2] = c.closest.box_tests, out[3] = c.closest.prim_tests;
    out[4] = c.shadow.rays, out[5] = c.shadow.node_visits, out[6] = c.shadow.box_tests, out[7] = c.shadow.prim_tests;
    out[8] = c.bounces;
}
// first words of the counter stream: the RNG specification shared with the HIP kernels (tests pin both to it)
void oracle_counter_words(uint64_t seed, uint64_t pixel, uint64_t sample, int n, uint64_t *out) {
    CounterRng r(seed, pixel, sample);
    for (int i = 0; i < n; i++) out[i] = r.word();
}

}  // extern "C"
