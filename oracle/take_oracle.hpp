// take_oracle.hpp — TEST INFRASTRUCTURE.  CPU restatement of TaKe's render hot path.
//
// This is the oracle of tests/ and of bench.py's `cpu_baseline` leg; nothing in the product
// (take_amd/, libtake_hip.so) includes, links or calls it.  It restates, function by function
// and expression tree by expression tree, the algorithm of the reference (citations are
// file:line in /root/reference), templated on the scalar type `R` and on the random source:
//
//   <double, MtRng>       the reference as built by g++/libstdc++ with the per-tile seed patch
//                          — pinned bit-for-bit against tests/golden/
//   <double, CounterRng>  same arithmetic, counter-based random stream keyed by
//                         (seed, pixel, sample): the stream the GPU kernels implement
//   <float,  CounterRng>  the reference's "Switching to floating point computation is easy —
//                         just set Real = float" (src/take.h:20-28) variant: the f32 twin of the
//                         production GPU path, with the ray offset epsilon as a parameter
//                         (src/take.h:30-31 offers 1e-7 and 1e-4)
//
// Parity status: PINNED.  tests/test_oracle_golden.py checks this file against every table and
// seeded render under tests/golden/, all produced by the compiled reference (oracle/Makefile `ref`,
// oracle/gen_golden.py).  Unpinned by construction: nothing.  (Env-map IBL and instancing do not
// exist upstream and are not restated here.)
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <optional>
#include <random>
#include <thread>
#include <vector>

#include "take_hip.h"

namespace oracle {

// ------------------------------------------------------------------ src/take.h, src/vector.h
template <class R> struct K {
    static constexpr R EPS = R(1e-7);                       // c_EPSILON            take.h:30
    static constexpr R PI = R(3.14159265358979323846);      // c_PI                 take.h:34
    static constexpr R INVPI = R(1.0) / PI;                 //                      take.h:35
    static constexpr R TWOPI = R(2.0) * PI;                 //                      take.h:36
    static constexpr R INVTWOPI = R(1.0) / TWOPI;           //                      take.h:37
    static R inf() { return std::numeric_limits<R>::infinity(); }
};

template <class R> struct V3 {
    R x, y, z;
    R operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
template <class R> struct V2 {
    R x, y;
};
// vector.h:120-240 — note operator/(v, s) multiplies by the reciprocal (vector.h:194-197)
template <class R> inline V3<R> operator+(V3<R> a, V3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class R> inline V3<R> operator-(V3<R> a, V3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class R> inline V3<R> operator-(V3<R> a) { return {-a.x, -a.y, -a.z}; }
template <class R> inline V3<R> operator*(V3<R> a, V3<R> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <class R> inline V3<R> operator*(R s, V3<R> v) { return {s * v.x, s * v.y, s * v.z}; }
template <class R> inline V3<R> operator*(V3<R> v, R s) { return {v.x * s, v.y * s, v.z * s}; }
template <class R> inline V3<R> operator/(V3<R> v, R s) {
    R inv = R(1) / s;
    return {v.x * inv, v.y * inv, v.z * inv};
}
template <class R> inline V3<R> sub_s(V3<R> v, R s) { return {v.x - s, v.y - s, v.z - s}; }  // v - s   vector.h:152
template <class R> inline V3<R> add_s(V3<R> v, R s) { return {v.x + s, v.y + s, v.z + s}; }  // v + s   vector.h:125
template <class R> inline V3<R> s_sub(R s, V3<R> v) { return {s - v.x, s - v.y, s - v.z}; }  // s - v   vector.h:147
template <class R> inline R dot(V3<R> a, V3<R> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class R> inline V3<R> cross(V3<R> a, V3<R> b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class R> inline R length(V3<R> v) { return std::sqrt(dot(v, v)); }
template <class R> inline V3<R> normalize(V3<R> v) {  // vector.h:249-257
    R l = length(v);
    if (l <= 0) return {R(0), R(0), R(0)};
    return v / l;
}
template <class R> inline V3<R> vmin(V3<R> a, V3<R> b) {
    return {a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y, a.z < b.z ? a.z : b.z};
}
template <class R> inline V3<R> vmax(V3<R> a, V3<R> b) {
    return {a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z};
}
template <class R> inline V2<R> operator+(V2<R> a, V2<R> b) { return {a.x + b.x, a.y + b.y}; }
template <class R> inline V2<R> operator*(R s, V2<R> v) { return {s * v.x, s * v.y}; }

// Frisvad ONB, vector.h:314-326
template <class R> inline V3<R> to_world(V3<R> n, V3<R> v) {
    V3<R> x, y;
    if (n.z < R(-1 + 1e-6)) {
        x = {R(0), R(-1), R(0)};
        y = {R(-1), R(0), R(0)};
    } else {
        R a = 1 / (1 + n.z);
        R b = -n.x * n.y * a;
        x = {1 - n.x * n.x * a, b, -n.x};
        y = {b, 1 - n.y * n.y * a, -n.y};
    }
    return x * v.x + y * v.y + n * v.z;
}

inline double modulo(double a, double b) {  // take.h:65-68
    double r = std::fmod(a, b);
    return (r < 0.0) ? r + b : r;
}
inline float modulo(float a, float b) {  // take.h:60-63
    float r = std::fmod(a, b);
    return (r < 0.0f) ? r + b : r;
}
template <class R> inline R clampR(R v, R lo, R hi) { return std::clamp(v, lo, hi); }

// ------------------------------------------------------------------ random sources
// random_real (take.h:89-91) as libstdc++ evaluates it for double: generate_canonical<double,53>
// draws two 32-bit words, (x0 + x1 * 2^32) / 2^64, clamped below 1.
struct MtRng {
    std::mt19937 eng;
    explicit MtRng(unsigned seed) : eng(seed) {}
    double real() {
        double x0 = (double)eng();
        double x1 = (double)eng();
        double r = (x0 + x1 * 4294967296.0) / 18446744073709551616.0;
        if (r >= 1.0) r = std::nextafter(1.0, 0.0);
        return r;
    }
};

// Counter-based stream — the SPECIFICATION the HIP kernels implement:
//   key   = mix(mix(seed + G1*(pixel+1)) + G2*(sample+1)),  mix = splitmix64 finaliser
//   word  = mix(key + G1 * counter++)                       one 64-bit word per random_real
//   f64   = (word >> 11) * 2^-53        f32 = (word >> 40) * 2^-24   (f32 = truncation of f64)
struct CounterRng {
    uint64_t key;
    uint32_t ctr;
    static constexpr uint64_t G1 = 0x9E3779B97F4A7C15ull, G2 = 0xD1B54A32D192ED03ull;
    static uint64_t mix(uint64_t z) {
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    CounterRng(uint64_t seed, uint64_t pixel, uint64_t sample) : ctr(0) {
        key = mix(mix(seed + G1 * (pixel + 1)) + G2 * (sample + 1));
    }
    uint64_t word() { return mix(key + G1 * (uint64_t)(ctr++)); }
};
template <class R, class Rng> struct Draw;
template <> struct Draw<double, MtRng> {
    static double real(MtRng &r) { return r.real(); }
};
template <> struct Draw<double, CounterRng> {
    static double real(CounterRng &r) { return (double)(r.word() >> 11) * (1.0 / 9007199254740992.0); }
};
template <> struct Draw<float, CounterRng> {
    static float real(CounterRng &r) { return (float)(r.word() >> 40) * (1.0f / 16777216.0f); }
};

// ------------------------------------------------------------------ scene (src/scene.h:13-33)
template <class R> struct Ray {  // ray.h:4-9
    V3<R> origin, dir;
    R tmin, tmax;
};
template <class R> struct Intersection {  // intersection.h:4-12 (+ shape_id, for hit-table parity)
    V3<R> pos, geo_normal, shading_normal;
    V2<R> uv;
    R t;
    int material_id, area_light_id;
    int shape_id;
    R bu, bv;  // Möller–Trumbore barycentrics (not in the reference struct)
    bool back_face;  // geo_normal was flipped to face the ray (not in the reference struct; take_burley.hpp)
};
template <class R> struct PointAndNormal {
    V3<R> position, normal;
};
template <class R> struct BBox {  // bbox.h:4-11
    V3<R> p_min{K<R>::inf(), K<R>::inf(), K<R>::inf()};
    V3<R> p_max{-K<R>::inf(), -K<R>::inf(), -K<R>::inf()};
};
template <class R> struct BBoxWithID {
    BBox<R> box;
    int id;
};
template <class R> struct BVHNode {  // bvh.h:5-10
    BBox<R> box;
    int left, right, prim;
};
template <class R> struct Mesh {  // shape.h:13-18
    int material_id;
    std::vector<V3<R>> positions, normals;
    std::vector<V2<R>> uvs;
    std::vector<int32_t> indices;  // 3 per face
};
template <class R> struct Shape {  // shape.h:20-32, flattened variant
    int kind;                       // 0 sphere, 1 triangle
    int material_id, area_light_id;
    V3<R> center;
    R radius;
    int face_id, mesh_id;
};
template <class R> struct Light {  // light.h:9-19
    int kind;                       // 0 point, 1 diffuse area
    int shape_id;
    V3<R> intensity, position;
};
template <class R> struct Texture {  // texture.h:16-27
    int kind, image_id;
    V3<R> value;
    R uscale, vscale, uoffset, voffset;
};
template <class R> struct Material {  // material.h:7-93
    int tag;
    Texture<R> reflectance;
    R p0, p1;
    R q[12];  // every TakeMaterial::param (tags 12..16, oracle/take_burley.hpp); p0 = q[0], p1 = q[1]
};
template <class R> struct Image3 {  // image.h:13-39
    int width, height;
    std::vector<V3<R>> data;
    const V3<R> &at(int x, int y) const { return data[(size_t)y * width + x]; }
};
// EXTENSION (not in the reference, which has only a constant background — SURVEY.md §0): an environment-map light,
// TakeLight kind 2.  Equirectangular image, y up, row 0 = zenith; radiance = nearest texel * scale; sampled with the
// piecewise-constant density luminance * sin(theta_row).  This is the specification the device code
// (take_amd/csrc/tk_shade.h: env_*) is held to; parity is against THIS restatement plus analytic tests
// ("parity unpinned" with respect to the reference).
template <class R> struct EnvLight {
    int light = -1;  // index in `lights`, -1 = none
    int image = -1;
    V3<R> scale{R(0), R(0), R(0)};
    std::vector<R> marginal;     // height + 1 row CDF values
    std::vector<R> conditional;  // height rows of width + 1 column CDF values
};
template <class R> struct Scene {
    TakeCamera camera;
    std::vector<Shape<R>> shapes;
    std::vector<Mesh<R>> meshes;
    std::vector<Light<R>> lights;
    std::vector<Material<R>> materials;
    std::vector<Image3<R>> images;
    V3<R> background;
    EnvLight<R> env;
    std::vector<BVHNode<R>> bvh_nodes;
    int bvh_root = -1;
    // Scene::lights_power_pmf / _cdf (scene.h:28-29): read by sample_light_power / get_light_pmf (light.cpp:9-23) but
    // never filled by the reference.  Filled by fill_light_power() below from light_power() (light.cpp:25-30), the
    // same recipe oracle/ref_harness.cpp applies to the reference's Scene for the `ptpow` golden tables.
    std::vector<R> lights_power_pmf, lights_power_cdf;
    R ray_eps = K<R>::EPS;  // c_EPSILON in its ray-offset role (render.cpp:75, path_tracing.h:53,79)
};

template <class R> inline V3<R> cv3(const double *p) { return {R(p[0]), R(p[1]), R(p[2])}; }

template <class R> void scene_from_desc(const TakeSceneDesc &d, Scene<R> &s) {
    s.camera = d.camera;
    s.background = cv3<R>(d.background);
    s.meshes.resize(d.n_meshes);
    for (int i = 0; i < d.n_meshes; i++) {
        const TakeMesh &m = d.meshes[i];
        Mesh<R> &o = s.meshes[i];
        o.material_id = m.material_id;
        o.positions.resize(m.n_vertices);
        for (int64_t k = 0; k < m.n_vertices; k++) o.positions[k] = cv3<R>(m.positions + 3 * k);
        o.indices.assign(m.indices, m.indices + 3 * m.n_faces);
        if (m.normals) {
            o.normals.resize(m.n_vertices);
            for (int64_t k = 0; k < m.n_vertices; k++) o.normals[k] = cv3<R>(m.normals + 3 * k);
        }
        if (m.uvs) {
            o.uvs.resize(m.n_vertices);
            for (int64_t k = 0; k < m.n_vertices; k++) o.uvs[k] = {R(m.uvs[2 * k]), R(m.uvs[2 * k + 1])};
        }
    }
    s.shapes.resize(d.n_shapes);
    for (int64_t i = 0; i < d.n_shapes; i++) {
        Shape<R> &o = s.shapes[i];
        o = Shape<R>{};
        o.kind = d.shape_kind[i];
        o.area_light_id = d.shape_area_light[i];
        if (o.kind == 0) {
            const TakeSphere &sp = d.spheres[d.shape_ref[i]];
            o.center = cv3<R>(sp.center);
            o.radius = R(sp.radius);
            o.material_id = sp.material_id;
        } else {
            o.mesh_id = d.shape_ref[i];
            o.face_id = d.shape_face[i];
            o.material_id = d.meshes[o.mesh_id].material_id;
        }
    }
    s.lights.resize(d.n_lights);
    for (int i = 0; i < d.n_lights; i++)
        s.lights[i] = {d.lights[i].kind, d.lights[i].shape_id, cv3<R>(d.lights[i].intensity), cv3<R>(d.lights[i].position)};
    s.materials.resize(d.n_materials);
    for (int i = 0; i < d.n_materials; i++) {
        const TakeMaterial &m = d.materials[i];
        const TakeTexture &t = m.reflectance;
        s.materials[i] = {m.tag,
                          {t.kind, t.image_id, cv3<R>(t.value), R(t.uscale), R(t.vscale), R(t.uoffset), R(t.voffset)},
                          R(m.param[0]),
                          R(m.param[1]),
                          {}};
        for (int k = 0; k < TAKE_MATERIAL_PARAMS; k++) s.materials[i].q[k] = R(m.param[k]);
    }
    s.images.resize(d.n_images);
    for (int i = 0; i < d.n_images; i++) {
        s.images[i].width = d.images[i].width;
        s.images[i].height = d.images[i].height;
        size_t n = (size_t)d.images[i].width * d.images[i].height;
        s.images[i].data.resize(n);
        for (size_t k = 0; k < n; k++) s.images[i].data[k] = cv3<R>(d.images[i].data + 3 * k);
    }
    // environment map: sampling tables from the image as given (double), stored in R
    s.env = EnvLight<R>{};
    for (int i = 0; i < d.n_lights; i++) {
        if (d.lights[i].kind != 2) continue;
        const TakeImage3 &im = d.images[d.lights[i].shape_id];
        const int w = im.width, h = im.height;
        std::vector<double> cond((size_t)h * (w + 1)), marg((size_t)h + 1), rows(h);
        for (int y = 0; y < h; y++) {
            const double weight = std::sin(3.14159265358979323846 * (y + 0.5) / h);
            double acc = 0;
            for (int x = 0; x < w; x++) {
                const double *t = im.data + 3 * ((size_t)y * w + x);
                const double lum = 0.2126 * t[0] + 0.7152 * t[1] + 0.0722 * t[2];
                cond[(size_t)y * (w + 1) + x] = acc;
                acc += (lum > 0 ? lum : 0.0) * weight;
            }
            rows[y] = acc;
            for (int x = 0; x < w; x++) {
                double &c = cond[(size_t)y * (w + 1) + x];
                c = acc > 0 ? c / acc : (double)x / w;
            }
            cond[(size_t)y * (w + 1) + w] = 1.0;
        }
        double total = 0;
        for (int y = 0; y < h; y++) marg[y] = total, total += rows[y];
        for (int y = 0; y < h; y++) marg[y] /= total;
        marg[h] = 1.0;
        s.env.light = i;
        s.env.image = d.lights[i].shape_id;
        s.env.scale = cv3<R>(d.lights[i].intensity);
        s.env.marginal.assign(marg.begin(), marg.end());
        s.env.conditional.assign(cond.begin(), cond.end());
    }
}

// ------------------------------------------------------------------ environment map (extension, see EnvLight)
template <class R> struct EnvSample {
    V3<R> dir, radiance;
    R pdf;
};
template <class R> inline V3<R> env_radiance(const Scene<R> &sc, int x, int y) {
    return sc.images[sc.env.image].at(x, y) * sc.env.scale;
}
template <class R> inline R env_density(const Scene<R> &sc, int x, int y, R sin_theta) {
    if (!(sin_theta > R(0))) return R(0);
    const int w = sc.images[sc.env.image].width, h = sc.images[sc.env.image].height;
    const R *row = sc.env.conditional.data() + (size_t)y * (w + 1);
    const R p = (sc.env.marginal[y + 1] - sc.env.marginal[y]) * (row[x + 1] - row[x]) * R(w) * R(h);
    return p / (R(2) * K<R>::PI * K<R>::PI * sin_theta);
}
template <class R> inline V3<R> env_eval(const Scene<R> &sc, const V3<R> &d, R &pdf) {
    const int w = sc.images[sc.env.image].width, h = sc.images[sc.env.image].height;
    const R cy = std::clamp(d.y, R(-1), R(1));
    const R theta = std::acos(cy);
    const R u = std::atan2(d.z, d.x) * K<R>::INVTWOPI + R(0.5);
    const int x = std::clamp((int)std::floor(u * R(w)), 0, w - 1);
    const int y = std::clamp((int)std::floor(theta * K<R>::INVPI * R(h)), 0, h - 1);
    pdf = env_density(sc, x, y, std::sqrt(std::fmax(R(0), R(1) - cy * cy)));
    return env_radiance(sc, x, y);
}
template <class R> inline int cdf_interval(const R *cdf, int n, R xi) {  // largest i < n with cdf[i] <= xi
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] <= xi) lo = mid;
        else hi = mid;
    }
    return lo;
}
template <class R, class Rng> inline EnvSample<R> env_sample(const Scene<R> &sc, Rng &rng) {
    const int w = sc.images[sc.env.image].width, h = sc.images[sc.env.image].height;
    const R u1 = Draw<R, Rng>::real(rng);
    const R u2 = Draw<R, Rng>::real(rng);
    const int y = cdf_interval(sc.env.marginal.data(), h, u1);
    const R *row = sc.env.conditional.data() + (size_t)y * (w + 1);
    const int x = cdf_interval(row, w, u2);
    const R m0 = sc.env.marginal[y], m1 = sc.env.marginal[y + 1], c0 = row[x], c1 = row[x + 1];
    const R dv = m1 > m0 ? (u1 - m0) / (m1 - m0) : R(0.5), du = c1 > c0 ? (u2 - c0) / (c1 - c0) : R(0.5);
    const R theta = (R(y) + dv) / R(h) * K<R>::PI;
    const R phi = ((R(x) + du) / R(w) - R(0.5)) * K<R>::TWOPI;
    const R st = std::sin(theta);
    return {V3<R>{st * std::cos(phi), std::cos(theta), st * std::sin(phi)}, env_radiance(sc, x, y), env_density(sc, x, y, st)};
}

// ------------------------------------------------------------------ src/bbox.h:18-55
template <class R> inline bool intersect_box(const BBox<R> &b, const Ray<R> &r) {
    R t_min = r.tmin, t_max = r.tmax;
    for (int a = 0; a < 3; a++) {
        R t0 = std::fmin((b.p_min[a] - r.origin[a]) / r.dir[a], (b.p_max[a] - r.origin[a]) / r.dir[a]);
        R t1 = std::fmax((b.p_min[a] - r.origin[a]) / r.dir[a], (b.p_max[a] - r.origin[a]) / r.dir[a]);
        t_min = std::fmax(t0, t_min);
        t_max = std::fmin(t1, t_max);
        if (t_max < t_min) return false;
    }
    return true;
}
template <class R> inline int largest_axis(const BBox<R> &b) {
    V3<R> e = b.p_max - b.p_min;
    if (e.x > e.y && e.x > e.z) return 0;
    if (e.y > e.x && e.y > e.z) return 1;
    return 2;
}
template <class R> inline BBox<R> merge(const BBox<R> &a, const BBox<R> &b) {
    return {{std::min(a.p_min.x, b.p_min.x), std::min(a.p_min.y, b.p_min.y), std::min(a.p_min.z, b.p_min.z)},
            {std::max(a.p_max.x, b.p_max.x), std::max(a.p_max.y, b.p_max.y), std::max(a.p_max.z, b.p_max.z)}};
}

// ------------------------------------------------------------------ src/bvh.cpp:8-45
template <class R> int construct_bvh(const std::vector<BBoxWithID<R>> &boxes, std::vector<BVHNode<R>> &pool) {
    if (boxes.size() == 1) {
        BVHNode<R> n;
        n.left = n.right = -1;
        n.prim = boxes[0].id;
        n.box = boxes[0].box;
        pool.push_back(n);
        return (int)pool.size() - 1;
    }
    BBox<R> big;
    for (const auto &b : boxes) big = merge(big, b.box);
    int axis = largest_axis(big);
    std::vector<BBoxWithID<R>> local = boxes;
    std::sort(local.begin(), local.end(), [&](const BBoxWithID<R> &b1, const BBoxWithID<R> &b2) {
        V3<R> c1 = (b1.box.p_max + b1.box.p_min) / R(2);
        V3<R> c2 = (b2.box.p_max + b2.box.p_min) / R(2);
        return c1[axis] < c2[axis];
    });
    std::vector<BBoxWithID<R>> lb(local.begin(), local.begin() + local.size() / 2);
    std::vector<BBoxWithID<R>> rb(local.begin() + local.size() / 2, local.end());
    BVHNode<R> n;
    n.box = big;
    n.left = construct_bvh(lb, pool);
    n.right = construct_bvh(rb, pool);
    n.prim = -1;
    pool.push_back(n);
    return (int)pool.size() - 1;
}

// ------------------------------------------------------------------ src/shape.cpp:3-110
template <class R> inline V2<R> get_sphere_uv(V3<R> p) {
    R theta = std::acos(-p.y);
    R phi = std::atan2(-p.z, p.x) + K<R>::PI;
    return {phi / (2 * K<R>::PI), -theta / K<R>::PI};
}
template <class R>
std::optional<Intersection<R>> intersect_sphere(const Shape<R> &s, int shape_id, const Ray<R> &r) {
    V3<R> oc = r.origin - s.center;
    R a = dot(r.dir, r.dir);
    R half_b = dot(oc, r.dir);
    R c = dot(oc, oc) - s.radius * s.radius;
    R disc = half_b * half_b - a * c;
    if (disc < 0) return {};
    R sqrtd = std::sqrt(disc);
    R root = (-half_b - sqrtd) / a;
    if (root < r.tmin || r.tmax < root) {
        root = (-half_b + sqrtd) / a;
        if (root < r.tmin || r.tmax < root) return {};
    }
    Intersection<R> v{};
    v.t = root;
    v.pos = r.origin + r.dir * v.t;
    v.geo_normal = normalize(v.pos - s.center);
    v.back_face = !(dot(r.dir, v.geo_normal) < 0);
    v.geo_normal = dot(r.dir, v.geo_normal) < 0 ? v.geo_normal : -v.geo_normal;
    v.shading_normal = v.geo_normal;
    v.material_id = s.material_id;
    v.uv = get_sphere_uv(v.geo_normal);
    v.area_light_id = s.area_light_id;
    v.shape_id = shape_id;
    v.bu = v.bv = 0;
    return v;
}
template <class R>
std::optional<Intersection<R>> intersect_triangle(const std::vector<Mesh<R>> &meshes, const Shape<R> &tri, int shape_id,
                                                  const Ray<R> &r) {
    const Mesh<R> &mesh = meshes[tri.mesh_id];
    const int32_t *idx = &mesh.indices[3 * (size_t)tri.face_id];
    V3<R> v0 = mesh.positions[idx[0]], v1 = mesh.positions[idx[1]], v2 = mesh.positions[idx[2]];
    V3<R> e1 = v1 - v0, e2 = v2 - v0;
    V3<R> h = cross(r.dir, e2);
    R a = dot(e1, h);
    if (a > -K<R>::EPS && a < K<R>::EPS) return {};
    R f = R(1.0) / a;
    V3<R> s = r.origin - v0;
    R u = f * dot(s, h);
    if (u < 0.0 || u > 1.0) return {};
    V3<R> q = cross(s, e1);
    R v = f * dot(r.dir, q);
    if (v < 0.0 || u + v > 1.0) return {};
    R t = f * dot(e2, q);
    if (t < r.tmin || r.tmax < t) return {};
    Intersection<R> in{};
    in.t = t;
    in.pos = r.origin + r.dir * t;
    in.geo_normal = normalize(cross(e1, e2));
    in.back_face = !(dot(r.dir, in.geo_normal) < 0);
    in.geo_normal = dot(r.dir, in.geo_normal) < 0 ? in.geo_normal : -in.geo_normal;
    in.material_id = mesh.material_id;
    in.area_light_id = tri.area_light_id;
    if (mesh.uvs.empty()) {
        in.uv = {u, v};
    } else {
        V2<R> uv0 = mesh.uvs[idx[0]], uv1 = mesh.uvs[idx[1]], uv2 = mesh.uvs[idx[2]];
        in.uv = (1 - u - v) * uv0 + u * uv1 + v * uv2;
    }
    if (mesh.normals.empty()) {
        in.shading_normal = in.geo_normal;
    } else {
        V3<R> n0 = mesh.normals[idx[0]], n1 = mesh.normals[idx[1]], n2 = mesh.normals[idx[2]];
        in.shading_normal = normalize((1 - u - v) * n0 + u * n1 + v * n2);
    }
    in.shape_id = shape_id;
    in.bu = u;
    in.bv = v;
    return in;
}
template <class R>
inline std::optional<Intersection<R>> intersect_shape(const Scene<R> &sc, int shape_id, const Ray<R> &r) {  // shape.h:42
    const Shape<R> &s = sc.shapes[shape_id];
    return s.kind == 0 ? intersect_sphere(s, shape_id, r) : intersect_triangle(sc.meshes, s, shape_id, r);
}

// ------------------------------------------------------------------ src/bvh.cpp:86-109, src/scene.cpp:4-64
struct TraversalCounters {
    uint64_t rays = 0, node_visits = 0, box_tests = 0, prim_tests = 0;
};
template <class R>
std::optional<Intersection<R>> bvh_intersect(const Scene<R> &sc, int node_id, Ray<R> ray, TraversalCounters *tc) {
    const BVHNode<R> &node = sc.bvh_nodes[node_id];
    if (tc) tc->node_visits++;
    if (node.prim != -1) {
        if (tc) tc->prim_tests++;
        return intersect_shape(sc, node.prim, ray);
    }
    const BVHNode<R> &left = sc.bvh_nodes[node.left];
    const BVHNode<R> &right = sc.bvh_nodes[node.right];
    std::optional<Intersection<R>> isect_left;
    if (tc) tc->box_tests += 2;
    if (intersect_box(left.box, ray)) {
        isect_left = bvh_intersect(sc, node.left, ray, tc);
        if (isect_left) ray.tmax = isect_left->t;
    }
    if (intersect_box(right.box, ray)) {
        if (auto isect_right = bvh_intersect(sc, node.right, ray, tc)) return isect_right;
    }
    return isect_left;
}
template <class R> void build_bvh(Scene<R> &sc) {
    std::vector<BBoxWithID<R>> boxes(sc.shapes.size());
    for (int i = 0; i < (int)boxes.size(); i++) {
        const Shape<R> &s = sc.shapes[i];
        if (s.kind == 0) {
            boxes[i] = {BBox<R>{sub_s(s.center, s.radius), add_s(s.center, s.radius)}, i};
        } else {
            const Mesh<R> &m = sc.meshes[s.mesh_id];
            const int32_t *idx = &m.indices[3 * (size_t)s.face_id];
            V3<R> p0 = m.positions[idx[0]], p1 = m.positions[idx[1]], p2 = m.positions[idx[2]];
            boxes[i] = {BBox<R>{vmin(vmin(p0, p1), p2), vmax(vmax(p0, p1), p2)}, i};
        }
    }
    sc.bvh_nodes.clear();
    sc.bvh_root = boxes.empty() ? -1 : construct_bvh(boxes, sc.bvh_nodes);
}
template <class R>
std::optional<Intersection<R>> scene_intersect(const Scene<R> &sc, const Ray<R> &r, TraversalCounters *tc = nullptr) {
    if (tc) tc->rays++;
    if (!sc.bvh_nodes.empty()) return bvh_intersect(sc, sc.bvh_root, r, tc);
    R t = K<R>::inf();
    Intersection<R> v{};
    for (int i = 0; i < (int)sc.shapes.size(); i++) {
        auto v_ = intersect_shape(sc, i, r);
        if (v_ && v_->t < t) {
            t = v_->t;
            v = *v_;
        }
    }
    if (t < K<R>::inf()) return v;
    return {};
}
template <class R> bool scene_occluded(const Scene<R> &sc, const Ray<R> &r, TraversalCounters *tc = nullptr) {
    return scene_intersect(sc, r, tc) ? true : false;  // scene.cpp:49-53: a full closest-hit query
}

// ------------------------------------------------------------------ src/texture.cpp:3-25
template <class R> V3<R> eval_texture(const Texture<R> &t, V2<R> uv, const Scene<R> &sc) {
    if (t.kind == 0) return t.value;
    const Image3<R> &img = sc.images[t.image_id];
    R x = img.width * modulo(t.uscale * uv.x + t.uoffset, R(1));
    R y = img.height * modulo(t.vscale * uv.y + t.voffset, R(1));
    int x1 = static_cast<int>(std::floor(x));
    int x2 = (x1 + 1) == img.width ? 0 : (x1 + 1);
    int y1 = static_cast<int>(std::floor(y));
    int y2 = (y1 + 1) == img.height ? 0 : (y1 + 1);
    V3<R> q11 = img.at(x1, y1), q12 = img.at(x1, y2), q21 = img.at(x2, y1), q22 = img.at(x2, y2);
    if (x1 == x2) x2 += 1;
    if (y1 == y2) y2 += 1;
    return (q11 * (x2 - x) * (y2 - y) + q21 * (x - x1) * (y2 - y) + q12 * (x2 - x) * (y - y1) +
            q22 * (x - x1) * (y - y1)) /
           R((x2 - x1) * (y2 - y1));
}

// ------------------------------------------------------------------ src/material.h:95-140, src/materials/*.inl
template <class R> struct SampleRecord {
    V3<R> dir_out;
    R pdf;
};
template <class R, class Rng> inline V3<R> sample_hemisphere_cos(Rng &rng) {  // material.h:121-132
    R u1 = Draw<R, Rng>::real(rng);
    R u2 = Draw<R, Rng>::real(rng);
    R phi = K<R>::TWOPI * u2;
    R sqrt_u1 = std::sqrt(clampR(u1, R(0), R(1)));
    return {std::cos(phi) * sqrt_u1, std::sin(phi) * sqrt_u1, std::sqrt(clampR(1 - u1, R(0), R(1)))};
}
template <class R> inline R blinn_phong_G_hat(V3<R> omega, V3<R> n, R alpha) {  // material.h:134-140
    R odn = dot(omega, n);
    R a = std::sqrt(R(0.5) * alpha + 1) / std::sqrt(1 / (odn * odn) - 1);
    R a2 = a * a;
    return a < R(1.6) ? (R(3.535) * a + R(2.181) * a2) / (1 + R(2.276) * a + R(2.577) * a2) : R(1);
}
template <class R> inline V3<R> facing_normal(V3<R> dir_in, const Intersection<R> &v) {
    return dot(dir_in, v.shading_normal) < 0 ? -v.shading_normal : v.shading_normal;
}
template <class R> inline SampleRecord<R> cosine_record(V3<R> n, const Intersection<R> &v, V3<R> local) {
    SampleRecord<R> rec;
    rec.dir_out = to_world(n, local);
    if (dot(v.geo_normal, rec.dir_out) < 0)
        rec.pdf = R(0);
    else
        rec.pdf = std::fmax(dot(n, rec.dir_out), R(0)) / K<R>::PI;
    return rec;
}
// half-vector lobe shared by Phong / BlinnPhong / BlinnPhongMicrofacet (phong.inl:7-18, blinn_phong.inl:7-18)
template <class R, class Rng> inline V3<R> sample_power_lobe(R exponent, Rng &rng) {
    R u1 = Draw<R, Rng>::real(rng);
    R u2 = Draw<R, Rng>::real(rng);
    R ra1 = 1 / (exponent + 1);
    R phi = K<R>::TWOPI * u2;
    R sqrt_u1 = std::sqrt(clampR(1 - std::pow(u1, 2 * ra1), R(0), R(1)));
    return normalize(V3<R>{std::cos(phi) * sqrt_u1, std::sin(phi) * sqrt_u1, clampR(std::pow(u1, ra1), R(0), R(1))});
}

}  // namespace oracle
#include "take_burley.hpp"  // tags 12..16 (extension; parity unpinned, see its header)
namespace oracle {

template <class R, class Rng>
std::optional<SampleRecord<R>> sample_bsdf(const Material<R> &m, V3<R> dir_in, const Intersection<R> &v,
                                           const Scene<R> &sc, Rng &rng) {
    if (is_burley<R>(m.tag)) return burley_sample(m, dir_in, v, rng);
    if (dot(v.geo_normal, dir_in) < 0) return {};  // first statement of every sample_bsdf_op
    V3<R> n = facing_normal(dir_in, v);
    SampleRecord<R> rec;
    switch (m.tag) {
        case TAKE_MAT_MIRROR: {  // mirror.inl:1-10
            rec.dir_out = -dir_in + 2 * dot(dir_in, n) * n;
            rec.pdf = 1;
            return rec;
        }
        case TAKE_MAT_PLASTIC: {  // plastic.inl:1-27
            V3<R> reflect_dir = -dir_in + 2 * dot(dir_in, n) * n;
            R eta = m.p0;
            R F0 = std::pow((eta - 1) / (eta + 1), R(2));
            R F = F0 + (1 - F0) * std::pow(1 - dot(n, reflect_dir), R(5));
            R u = Draw<R, Rng>::real(rng);
            if (u <= F) {
                rec.dir_out = reflect_dir;
                rec.pdf = R(1);
                return rec;
            }
            return cosine_record(n, v, sample_hemisphere_cos<R>(rng));
        }
        case TAKE_MAT_PHONG: {  // phong.inl:1-28
            V3<R> local_out = sample_power_lobe<R>(m.p0, rng);
            V3<R> reflect_dir = normalize(-dir_in + 2 * dot(dir_in, n) * n);
            rec.dir_out = normalize(to_world(reflect_dir, local_out));
            if (dot(v.geo_normal, rec.dir_out) < 0)
                rec.pdf = R(0);
            else
                rec.pdf = std::fmax(R(0), (m.p0 + 1) / K<R>::TWOPI * std::pow(dot(reflect_dir, rec.dir_out), m.p0));
            return rec;
        }
        case TAKE_MAT_BLINN_PHONG:               // blinn_phong.inl:1-28
        case TAKE_MAT_BLINN_PHONG_MICROFACET: {  // blinn_phong_microfacet.inl:1-28
            V3<R> local_h = sample_power_lobe<R>(m.p0, rng);
            V3<R> h = normalize(to_world(n, local_h));
            rec.dir_out = normalize(-dir_in + 2 * dot(dir_in, h) * h);
            if (dot(v.geo_normal, rec.dir_out) <= 0 || dot(h, n) <= 0 || dot(rec.dir_out, h) <= 0) {
                rec.pdf = R(0);
            } else if (m.tag == TAKE_MAT_BLINN_PHONG) {
                rec.pdf = (m.p0 + 1) * R(0.25) * K<R>::INVTWOPI * std::pow(dot(n, h), m.p0) / dot(rec.dir_out, h);
            } else {
                rec.pdf = (m.p0 + 1) * R(0.25) * K<R>::INVTWOPI * std::pow(clampR(dot(n, h), R(0), R(1)), m.p0) /
                          dot(rec.dir_out, h);
            }
            return rec;
        }
        default:  // Diffuse (diffuse.inl:1-13) and every Disney* (disney_*.inl:1-13): cosine hemisphere
            return cosine_record(n, v, sample_hemisphere_cos<R>(rng));
    }
}

template <class R>
R get_bsdf_pdf(const Material<R> &m, V3<R> dir_in, V3<R> dir_out, const Intersection<R> &v, const Scene<R> &) {
    if (is_burley<R>(m.tag)) return burley_pdf(m, dir_in, dir_out, v);
    if (m.tag == TAKE_MAT_MIRROR) return R(0);  // mirror.inl:12-14
    if (dot(v.geo_normal, dir_out) < 0) return R(0);
    V3<R> n = facing_normal(dir_in, v);
    switch (m.tag) {
        case TAKE_MAT_PLASTIC: {  // plastic.inl:29-38
            R eta = m.p0;
            R F0 = std::pow((eta - 1) / (eta + 1), R(2));
            R F = F0 + (1 - F0) * std::pow(1 - dot(n, dir_out), R(5));
            return (1 - F) * std::fmax(dot(n, dir_out), R(0)) / K<R>::PI;
        }
        case TAKE_MAT_PHONG: {  // phong.inl:30-40
            V3<R> reflect_dir = normalize(-dir_in + 2 * dot(dir_in, n) * n);
            return std::fmax(R(0), (m.p0 + 1) / K<R>::TWOPI * std::pow(dot(reflect_dir, dir_out), m.p0));
        }
        case TAKE_MAT_BLINN_PHONG:               // blinn_phong.inl:30-40
        case TAKE_MAT_BLINN_PHONG_MICROFACET: {  // blinn_phong_microfacet.inl:30-40
            V3<R> h = normalize(dir_out + dir_in);
            if (dot(v.geo_normal, dir_out) <= 0 || dot(h, n) <= 0 || dot(dir_out, h) <= 0) return R(0);
            if (m.tag == TAKE_MAT_BLINN_PHONG)
                return (m.p0 + 1) * R(0.25) * K<R>::INVTWOPI * std::pow(dot(n, h), m.p0) / dot(dir_out, h);
            return (m.p0 + 1) * R(0.25) * K<R>::INVTWOPI * std::pow(clampR(dot(n, h), R(0), R(1)), m.p0) / dot(dir_out, h);
        }
        default:  // diffuse.inl:15-20 and the Disney clones
            return std::fmax(dot(n, dir_out), R(0)) / K<R>::PI;
    }
}

template <class R>
V3<R> eval_bsdf(const Material<R> &m, V3<R> dir_in, const SampleRecord<R> &rec, const Intersection<R> &v,
                const Scene<R> &sc) {
    const V3<R> zero{R(0), R(0), R(0)};
    if (is_burley<R>(m.tag)) return burley_eval(m, dir_in, rec.dir_out, v, sc);
    if (dot(v.geo_normal, dir_in) < 0 || dot(v.geo_normal, rec.dir_out) < 0) return zero;
    V3<R> n = facing_normal(dir_in, v);
    const V3<R> &dir_out = rec.dir_out;
    switch (m.tag) {
        case TAKE_MAT_MIRROR: {  // mirror.inl:16-22 — no cosine
            V3<R> F0 = eval_texture(m.reflectance, v.uv, sc);
            return F0 + s_sub(R(1), F0) * std::pow(R(1 - dot(n, dir_out)), R(5));
        }
        case TAKE_MAT_PLASTIC: {  // plastic.inl:40-50 — specular branch detected by pdf == 1
            if (rec.pdf == R(1)) return {R(1), R(1), R(1)};
            V3<R> Kd = eval_texture(m.reflectance, v.uv, sc);
            return Kd * std::fmax(dot(n, dir_out), R(0)) / K<R>::PI;
        }
        case TAKE_MAT_PHONG: {  // phong.inl:42-53
            V3<R> reflect_dir = normalize(-dir_in + 2 * dot(dir_in, n) * n);
            V3<R> Ks = eval_texture(m.reflectance, v.uv, sc);
            if (dot(n, dir_out) <= 0) return zero;
            return Ks * (m.p0 + 1) / K<R>::TWOPI * std::pow(std::fmax(dot(dir_out, reflect_dir), R(0)), m.p0);
        }
        case TAKE_MAT_BLINN_PHONG: {  // blinn_phong.inl:42-54
            if (dot(n, dir_out) <= 0) return zero;
            V3<R> h = normalize(dir_out + dir_in);
            V3<R> Ks = eval_texture(m.reflectance, v.uv, sc);
            V3<R> Fh = Ks + s_sub(R(1), Ks) * std::pow(R(1 - dot(h, dir_out)), R(5));
            return (m.p0 + 2) * R(0.25) * K<R>::INVPI / (2 - std::pow(R(2), -m.p0 / 2)) * Fh *
                   std::pow(std::fmax(R(0), dot(n, h)), m.p0);
        }
        case TAKE_MAT_BLINN_PHONG_MICROFACET: {  // blinn_phong_microfacet.inl:42-59
            V3<R> h = normalize(dir_out + dir_in);
            if (dot(n, dir_out) <= 0 || dot(dir_out, h) <= 0 || dot(dir_in, h) <= 0) return zero;
            V3<R> Ks = eval_texture(m.reflectance, v.uv, sc);
            V3<R> Fh = Ks + s_sub(R(1), Ks) * std::pow(R(1 - dot(h, dir_out)), R(5));
            R Dh = (m.p0 + 2) * K<R>::INVTWOPI * std::pow(clampR(dot(n, h), R(0), R(1)), m.p0);
            R G = blinn_phong_G_hat(dir_out, n, m.p0) * blinn_phong_G_hat(dir_in, n, m.p0);
            return Fh * Dh * G * R(0.25) / dot(n, dir_in);
        }
        case TAKE_MAT_DISNEY_DIFFUSE:  // disney_diffuse.inl:22-46 (body in take_burley.hpp: shared with tag 16)
            return disney_diffuse_f(eval_texture(m.reflectance, v.uv, sc), m.p0, m.p1, n, dir_in, dir_out);
        case TAKE_MAT_DISNEY_CLEARCOAT:
            // disney_clearcoat.inl:26 returns an uninitialised Vector3{} (vector.h:30); defined here as zero
            // and excluded from the golden tables.
            return zero;
        default: {  // diffuse.inl:22-28, disney_metal/glass/sheen/bsdf.inl:22-27 (Lambert clones)
            V3<R> Kd = eval_texture(m.reflectance, v.uv, sc);
            return Kd * std::fmax(dot(n, dir_out), R(0)) / K<R>::PI;
        }
    }
}

// ------------------------------------------------------------------ src/light.cpp, src/shape.cpp:125-184
template <class R> inline R get_area(const Scene<R> &sc, const Shape<R> &s) {  // shape.cpp:171-184
    if (s.kind == 0) return 4 * K<R>::PI * s.radius * s.radius;
    const Mesh<R> &m = sc.meshes[s.mesh_id];
    const int32_t *idx = &m.indices[3 * (size_t)s.face_id];
    V3<R> v0 = m.positions[idx[0]], v1 = m.positions[idx[1]], v2 = m.positions[idx[2]];
    return length(cross(v1 - v0, v2 - v0)) / 2;
}
template <class R, class Rng>
PointAndNormal<R> sample_on_shape(const Scene<R> &sc, const Shape<R> &s, V3<R> ref_pos, Rng &rng) {
    if (s.kind == 0) {  // shape.cpp:125-144 — cone sampling
        R u1 = Draw<R, Rng>::real(rng);
        R u2 = Draw<R, Rng>::real(rng);
        R r = s.radius;
        R d = length(s.center - ref_pos);
        R z = 1 + u1 * (r / d - 1);
        R z2 = z * z;
        R sin_theta = std::sqrt(clampR(1 - z2, R(0), R(1)));
        V3<R> local_p = normalize(
            V3<R>{std::cos(2 * K<R>::PI * u2) * sin_theta, std::sin(2 * K<R>::PI * u2) * sin_theta, z});
        V3<R> normal = normalize(to_world(normalize(ref_pos - s.center), local_p));
        V3<R> point = s.center + s.radius * normal;
        return {point, normal};
    }
    const Mesh<R> &m = sc.meshes[s.mesh_id];  // shape.cpp:146-169
    const int32_t *idx = &m.indices[3 * (size_t)s.face_id];
    V3<R> v0 = m.positions[idx[0]], v1 = m.positions[idx[1]], v2 = m.positions[idx[2]];
    R u1 = Draw<R, Rng>::real(rng);
    R u2 = Draw<R, Rng>::real(rng);
    R b1 = 1 - std::sqrt(u1);
    R b2 = std::sqrt(u1) * u2;
    V3<R> point = (1 - b1 - b2) * v0 + b1 * v1 + b2 * v2;
    V3<R> normal = normalize(cross(v1 - v0, v2 - v0));
    // the reference reads mesh.normals.at() and throws on meshes without normals;
    // a scene that reaches this without normals is rejected at scene creation
    V3<R> n0 = m.normals[idx[0]], n1 = m.normals[idx[1]], n2 = m.normals[idx[2]];
    V3<R> sn = (1 - b1 - b2) * n0 + b1 * n1 + b2 * n2;
    return {point, dot(sn, normal) > 0 ? normal : -normal};
}
template <class R>
R get_light_pdf(const Scene<R> &sc, int light_id, const PointAndNormal<R> &lp, V3<R> ref_pos) {  // light.cpp:32-48
    const Light<R> &l = sc.lights[light_id];
    if (l.kind == 1) {
        const Shape<R> &s = sc.shapes[l.shape_id];
        if (s.kind == 1) return 1 / get_area(sc, s);
        R r = s.radius;
        R d = length(lp.position - ref_pos);
        return 1 / (K<R>::TWOPI * r * r * (1 - r / d));
    }
    return 0;
}

// ------------------------------------------------------------------ src/integrator/path_tracing.h:5-111
struct PathCounters {
    TraversalCounters closest, shadow;
    uint64_t bounces = 0;
};
template <class R, class Rng>
V3<R> path_tracing(const Scene<R> &sc, const Ray<R> &ray, Rng &rng, int max_depth, PathCounters *pc = nullptr) {
    Ray<R> r = ray;
    auto v_ = scene_intersect(sc, r, pc ? &pc->closest : nullptr);
    if (!v_) {
        if (sc.env.light >= 0) {  // extension: the camera ray sees the environment map
            R unused;
            return env_eval(sc, r.dir, unused);
        }
        return sc.background;
    }
    Intersection<R> v = *v_;
    V3<R> radiance{R(0), R(0), R(0)};
    V3<R> throughput{R(1), R(1), R(1)};
    const R nlights = R(sc.lights.size());
    if (v.area_light_id != -1) {
        const Light<R> &light = sc.lights.at(v.area_light_id);
        if (light.kind == 1) radiance = radiance + throughput * light.intensity;
    }
    for (int i = 0; i <= max_depth; ++i) {
        if (pc) pc->bounces++;
        V3<R> dir_in = -r.dir;
        const Material<R> &m = sc.materials[v.material_id];
        bool is_specular = (m.tag == TAKE_MAT_PLASTIC || m.tag == TAKE_MAT_MIRROR);
        V3<R> C1{R(0), R(0), R(0)};
        if (sc.lights.size() > 0 && !is_specular) {
            int light_id = static_cast<int>(std::floor(Draw<R, Rng>::real(rng) * nlights));  // light.cpp:5-7
            const Light<R> &light = sc.lights[light_id];
            if (light.kind == 2) {
                // extension: next-event estimation towards the environment map, in the form of the area-light
                // term below with the map's solid-angle density (over the number of lights) as light_pdf
                EnvSample<R> es = env_sample(sc, rng);
                R light_pdf = es.pdf / nlights;
                if (light_pdf <= 0) break;
                R bsdf_pdf = get_bsdf_pdf(m, dir_in, es.dir, v, sc);
                if (bsdf_pdf > 0 && !std::isinf(light_pdf)) {
                    SampleRecord<R> rec{};
                    rec.dir_out = es.dir;
                    V3<R> FG = eval_bsdf(m, dir_in, rec, v, sc);
                    Ray<R> shadow_r{v.pos, es.dir, sc.ray_eps, K<R>::inf()};
                    if (!scene_occluded(sc, shadow_r, pc ? &pc->shadow : nullptr)) {
                        C1 = FG * es.radiance * light_pdf / (light_pdf * light_pdf + bsdf_pdf * bsdf_pdf);
                    }
                }
            } else if (light.kind == 1) {
                PointAndNormal<R> lp = sample_on_shape(sc, sc.shapes.at(light.shape_id), v.pos, rng);
                R d = length(lp.position - v.pos);
                V3<R> light_dir = normalize(lp.position - v.pos);
                R light_pdf = get_light_pdf(sc, light_id, lp, v.pos) * (d * d) /
                              (std::fmax(dot(-lp.normal, light_dir), R(0)) * nlights);
                if (light_pdf <= 0) break;
                R bsdf_pdf = get_bsdf_pdf(m, dir_in, light_dir, v, sc);
                if (bsdf_pdf > 0 && !std::isinf(light_pdf)) {
                    SampleRecord<R> rec{};
                    rec.dir_out = light_dir;
                    V3<R> FG = eval_bsdf(m, dir_in, rec, v, sc);
                    Ray<R> shadow_r{v.pos, light_dir, sc.ray_eps, (1 - sc.ray_eps) * d};
                    if (!scene_occluded(sc, shadow_r, pc ? &pc->shadow : nullptr)) {
                        C1 = FG * light.intensity * light_pdf / (light_pdf * light_pdf + bsdf_pdf * bsdf_pdf);
                    }
                }
            }
        }
        radiance = radiance + throughput * C1;

        V3<R> C2{R(0), R(0), R(0)};
        auto rec_ = sample_bsdf(m, dir_in, v, sc, rng);
        if (!rec_) break;
        SampleRecord<R> &rec = *rec_;
        V3<R> FG = eval_bsdf(m, dir_in, rec, v, sc);
        V3<R> dir_out = normalize(rec.dir_out);
        R bsdf_pdf = rec.pdf;
        R light_pdf = R(0);
        if (bsdf_pdf <= R(0)) break;
        r = Ray<R>{v.pos, dir_out, sc.ray_eps, K<R>::inf()};
        auto new_v = scene_intersect(sc, r, pc ? &pc->closest : nullptr);
        if (!new_v) {
            if (sc.env.light >= 0) {
                // extension: the sampled ray sees the environment map — the emitter-hit term C2 below with the
                // map's density in the role of light_pdf
                R env_pdf;
                V3<R> L = env_eval(sc, dir_out, env_pdf);
                R lp = env_pdf / nlights;
                radiance = radiance +
                           throughput * (FG * L * (is_specular ? (1 / bsdf_pdf) : (bsdf_pdf / (lp * lp + bsdf_pdf * bsdf_pdf))));
                break;
            }
            throughput = throughput * (FG / bsdf_pdf);
            radiance = radiance + throughput * sc.background;
            break;
        }
        if (new_v->area_light_id != -1) {
            V3<R> light_pos = new_v->pos;
            R d = length(light_pos - v.pos);
            V3<R> light_dir = normalize(light_pos - v.pos);
            light_pdf = get_light_pdf(sc, new_v->area_light_id, PointAndNormal<R>{new_v->pos, new_v->geo_normal}, v.pos) *
                        (d * d) / (std::fmax(dot(-new_v->geo_normal, light_dir), R(0)) * nlights);
            if (light_pdf <= 0) break;
            const Light<R> &light = sc.lights[new_v->area_light_id];
            if (light.kind == 1) {
                C2 = FG * light.intensity *
                     (is_specular ? (1 / bsdf_pdf) : (bsdf_pdf / (light_pdf * light_pdf + bsdf_pdf * bsdf_pdf)));
            }
        }
        radiance = radiance + throughput * C2;
        throughput = throughput * (FG / bsdf_pdf);
        v = *new_v;
    }
    return radiance;
}

// ------------------------------------------------------------------ src/light.cpp:9-30 (power-based light picking)
template <class R> inline R luminance(V3<R> s) {  // vector.h:309-311
    return s.x * R(0.212671) + s.y * R(0.715160) + s.z * R(0.072169);
}
template <class R> inline R light_power(const Scene<R> &sc, const Light<R> &l) {  // light.cpp:25-30
    if (l.kind == 1) return luminance(l.intensity) * get_area(sc, sc.shapes[l.shape_id]) * K<R>::PI;
    return 0;
}
template <class R> void fill_light_power(Scene<R> &sc) {
    R total = 0;
    std::vector<R> power;
    for (auto &l : sc.lights) power.push_back(light_power(sc, l)), total += power.back();
    sc.lights_power_pmf.clear();
    sc.lights_power_cdf.assign(1, R(0));
    for (R p : power) {
        sc.lights_power_pmf.push_back(p / total);
        sc.lights_power_cdf.push_back(sc.lights_power_cdf.back() + p / total);
    }
}
template <class R, class Rng> int sample_light_power(const Scene<R> &sc, Rng &rng) {  // light.cpp:9-17
    const std::vector<R> &cdf = sc.lights_power_cdf;
    R u = Draw<R, Rng>::real(rng);
    int size = (int)cdf.size() - 1;
    const R *ptr = std::upper_bound(cdf.data(), cdf.data() + size + 1, u);
    return std::clamp(int(ptr - cdf.data() - 1), 0, size - 1);
}

// ------------------------------------------------------------------ src/integrator/path_tracing.h:114-158
// Path tracing without MIS (defined by the reference, called by nothing there).
template <class R, class Rng>
V3<R> path_tracing_raw(const Scene<R> &sc, const Ray<R> &ray, Rng &rng, int max_depth, PathCounters *pc = nullptr) {
    Ray<R> r = ray;
    auto v_ = scene_intersect(sc, r, pc ? &pc->closest : nullptr);
    if (!v_) return sc.background;
    Intersection<R> v = *v_;
    V3<R> radiance{R(0), R(0), R(0)};
    V3<R> throughput{R(1), R(1), R(1)};
    for (int i = 0; i <= max_depth; ++i) {
        if (pc) pc->bounces++;
        if (v.area_light_id != -1) {
            const Light<R> &light = sc.lights.at(v.area_light_id);
            if (light.kind == 1) {
                radiance = radiance + throughput * light.intensity;
                break;
            }
        } else {
            V3<R> dir_in = -r.dir;
            const Material<R> &m = sc.materials[v.material_id];
            auto rec_ = sample_bsdf(m, dir_in, v, sc, rng);
            if (!rec_) break;
            SampleRecord<R> &rec = *rec_;
            V3<R> FG = eval_bsdf(m, dir_in, rec, v, sc);
            V3<R> dir_out = normalize(rec.dir_out);
            R pdf = rec.pdf;
            if (pdf <= R(0)) break;
            throughput = throughput * (FG / pdf);
            r = Ray<R>{v.pos, dir_out, sc.ray_eps, K<R>::inf()};
            auto nv = scene_intersect(sc, r, pc ? &pc->closest : nullptr);
            if (!nv) {
                radiance = radiance + throughput * sc.background;
                break;
            }
            v = *nv;
        }
    }
    return radiance;
}

// ------------------------------------------------------------------ src/integrator/path_tracing.h:161-271, :274-380
// One-sample MIS: per vertex EITHER a light sample OR a BSDF sample (coin flip), one closest-hit ray either way, the
// sampled direction weighted by the mixture density.  POWER = false: uniform light pick (path_tracing_one_sample_MIS);
// POWER = true: pick by power (path_tracing_one_sample_MIS_power, "seems to have bugs" upstream — restated as written).
// One undefined spot upstream: in the uniform variant the ray towards the sampled light point is dereferenced without
// a check (`v = *v_`, :222; "we will always hit the light or an obstacle"); a miss ends the path here.
template <class R, class Rng, bool POWER>
V3<R> path_tracing_one_sample(const Scene<R> &sc, const Ray<R> &ray, Rng &rng, int max_depth, PathCounters *pc = nullptr) {
    Ray<R> r = ray;
    auto v_ = scene_intersect(sc, r, pc ? &pc->closest : nullptr);
    if (!v_) return sc.background;
    Intersection<R> v = *v_;
    V3<R> radiance{R(0), R(0), R(0)};
    V3<R> throughput{R(1), R(1), R(1)};
    const R nlights = R(sc.lights.size());
    for (int i = 0; i <= max_depth; ++i) {
        if (pc) pc->bounces++;
        if (v.area_light_id != -1) {
            const Light<R> &light = sc.lights.at(v.area_light_id);
            if (light.kind == 1) {
                radiance = radiance + throughput * light.intensity;
                break;
            }
        }
        V3<R> dir_in = -r.dir;
        const Material<R> &m = sc.materials[v.material_id];
        bool is_specular = (m.tag == TAKE_MAT_PLASTIC || m.tag == TAKE_MAT_MIRROR);
        if (sc.lights.size() > 0 && !is_specular && Draw<R, Rng>::real(rng) <= R(0.5)) {
            // sampling a light
            int light_id = POWER ? sample_light_power(sc, rng) : static_cast<int>(std::floor(Draw<R, Rng>::real(rng) * nlights));
            const Light<R> &light = sc.lights[light_id];
            if (light.kind == 1) {
                PointAndNormal<R> lp = sample_on_shape(sc, sc.shapes.at(light.shape_id), v.pos, rng);
                R d = length(lp.position - v.pos);
                V3<R> light_dir = normalize(lp.position - v.pos);
                R light_pdf = POWER ? get_light_pdf(sc, light_id, lp, v.pos) * (d * d) * sc.lights_power_pmf[light_id] /
                                          (std::fmax(dot(-lp.normal, light_dir), R(0)))
                                    : get_light_pdf(sc, light_id, lp, v.pos) * (d * d) /
                                          (std::fmax(dot(-lp.normal, light_dir), R(0)) * nlights);
                if (light_pdf <= 0) break;
                R bsdf_pdf = get_bsdf_pdf(m, dir_in, light_dir, v, sc);
                if (bsdf_pdf <= 0) break;
                SampleRecord<R> rec{};
                rec.dir_out = light_dir;
                V3<R> FG = eval_bsdf(m, dir_in, rec, v, sc);
                r = Ray<R>{v.pos, light_dir, sc.ray_eps, K<R>::inf()};
                auto nv = scene_intersect(sc, r, pc ? &pc->closest : nullptr);
                if (!nv) {
                    if (POWER) radiance = radiance + throughput * sc.background;  // :327-331
                    break;                                                         // uniform variant: undefined upstream
                }
                v = *nv;
                if (POWER && v.area_light_id == -1) break;  // :333-335
                throughput = throughput * (FG / (R(0.5) * light_pdf + R(0.5) * bsdf_pdf));
            }
        } else {
            // sampling the BSDF
            auto rec_ = sample_bsdf(m, dir_in, v, sc, rng);
            if (!rec_) break;
            SampleRecord<R> &rec = *rec_;
            V3<R> FG = eval_bsdf(m, dir_in, rec, v, sc);
            V3<R> dir_out = normalize(rec.dir_out);
            R bsdf_pdf = rec.pdf;
            if (bsdf_pdf <= R(0)) break;
            r = Ray<R>{v.pos, dir_out, sc.ray_eps, K<R>::inf()};
            auto nv = scene_intersect(sc, r, pc ? &pc->closest : nullptr);
            R pdf = (sc.lights.empty() || is_specular) ? bsdf_pdf : R(0.5) * bsdf_pdf;
            if (!nv) {
                throughput = throughput * (FG / pdf);
                radiance = radiance + throughput * sc.background;
                break;
            }
            if (!is_specular && nv->area_light_id != -1) {
                V3<R> light_pos = nv->pos;
                R d = length(light_pos - v.pos);
                V3<R> light_dir = normalize(light_pos - v.pos);
                R lpdf = get_light_pdf(sc, nv->area_light_id, PointAndNormal<R>{nv->pos, nv->geo_normal}, v.pos) * (d * d);
                R light_pdf = POWER ? lpdf * sc.lights_power_pmf[nv->area_light_id] / std::fmax(dot(-nv->geo_normal, light_dir), R(0))
                                    : lpdf / (std::fmax(dot(-nv->geo_normal, light_dir), R(0)) * nlights);
                if (light_pdf <= 0) break;
                pdf += R(0.5) * light_pdf;
            }
            throughput = throughput * (FG / pdf);
            v = *nv;
        }
    }
    return radiance;
}

// TakeRenderOpts.integrator -> the reference function it selects
enum Integrator { INT_PATH_MIS = 0, INT_RAW = 1, INT_ONE_SAMPLE = 2, INT_ONE_SAMPLE_POWER = 3 };
template <class R, class Rng>
V3<R> integrate(int integrator, const Scene<R> &sc, const Ray<R> &ray, Rng &rng, int max_depth, PathCounters *pc = nullptr) {
    switch (integrator) {
        case INT_RAW: return path_tracing_raw(sc, ray, rng, max_depth, pc);
        case INT_ONE_SAMPLE: return path_tracing_one_sample<R, Rng, false>(sc, ray, rng, max_depth, pc);
        case INT_ONE_SAMPLE_POWER: return path_tracing_one_sample<R, Rng, true>(sc, ray, rng, max_depth, pc);
        default: return path_tracing(sc, ray, rng, max_depth, pc);
    }
}

// ------------------------------------------------------------------ src/render.cpp:37-82
template <class R> struct CameraBasis {
    V3<R> u, v, w, lookfrom;
    R viewport_width, viewport_height;
    int width, height;
};
template <class R> CameraBasis<R> camera_basis(const TakeCamera &cam) {
    CameraBasis<R> b;
    b.width = cam.width;
    b.height = cam.height;
    R vfov = R(cam.vfov);
    R theta = vfov / 180 * K<R>::PI;
    R h = std::tan(theta / 2);
    b.viewport_height = 2 * h;
    b.viewport_width = b.viewport_height / cam.height * cam.width;
    b.lookfrom = cv3<R>(cam.lookfrom);
    V3<R> lookat = cv3<R>(cam.lookat), up = cv3<R>(cam.up);
    b.w = normalize(b.lookfrom - lookat);
    b.u = normalize(cross(up, b.w));
    b.v = cross(b.w, b.u);
    return b;
}
// ry is drawn BEFORE rx under g++; both sources keep that order
template <class R> inline Ray<R> camera_ray(const CameraBasis<R> &b, int x, int y, R rx, R ry, R eps) {
    V3<R> d = normalize(b.u * ((x + rx) / b.width - R(0.5)) * b.viewport_width +
                        b.v * ((y + ry) / b.height - R(0.5)) * b.viewport_height - b.w);
    return {b.lookfrom, d, eps, K<R>::inf()};
}

enum RngMode { RNG_MT_PER_TILE = 0, RNG_COUNTER = 1 };

// The tile loop of render() with the pbrt-style self-scheduling of src/parallel.cpp:183-237 reduced to an
// atomic tile counter (same x-fastest tile order, one tile per grab).  out: H*W*3, image order (row 0 = top).
template <class R>
void render(const Scene<R> &sc, int spp, int max_depth, int rng_mode, uint64_t seed, int threads, R *out,
            PathCounters *total = nullptr, int integrator = 0) {
    const CameraBasis<R> cb = camera_basis<R>(sc.camera);
    const int W = cb.width, H = cb.height;
    constexpr int tile_size = 16;
    const int ntx = (W + tile_size - 1) / tile_size, nty = (H + tile_size - 1) / tile_size;
    std::atomic<int> next{0};
    std::vector<PathCounters> counters(std::max(threads, 1));
    auto worker = [&](int tid) {
        PathCounters *pc = total ? &counters[tid] : nullptr;
        for (;;) {
            int idx = next.fetch_add(1);
            if (idx >= ntx * nty) break;
            int tx = idx % ntx, ty = idx / ntx;
            MtRng mt((unsigned)(ty * ntx + tx));  // the oracle seed patch at render.cpp:60
            int x0 = tx * tile_size, x1 = std::min(x0 + tile_size, W);
            int y0 = ty * tile_size, y1 = std::min(y0 + tile_size, H);
            for (int y = y0; y < y1; y++) {
                for (int x = x0; x < x1; x++) {
                    V3<R> color{R(0), R(0), R(0)};
                    for (int i = 0; i < spp; i++) {
                        if constexpr (std::is_same<R, double>::value) {
                            if (rng_mode == RNG_MT_PER_TILE) {
                                R ry = Draw<R, MtRng>::real(mt);
                                R rx = Draw<R, MtRng>::real(mt);
                                Ray<R> r = camera_ray(cb, x, y, rx, ry, sc.ray_eps);
                                color = color + integrate(integrator, sc, r, mt, max_depth, pc);
                                continue;
                            }
                        }
                        CounterRng cr(seed, (uint64_t)y * W + x, (uint64_t)i);
                        R ry = Draw<R, CounterRng>::real(cr);
                        R rx = Draw<R, CounterRng>::real(cr);
                        Ray<R> r = camera_ray(cb, x, y, rx, ry, sc.ray_eps);
                        color = color + integrate(integrator, sc, r, cr, max_depth, pc);
                    }
                    V3<R> px = color / R(spp);
                    R *o = out + ((size_t)(H - y - 1) * W + x) * 3;  // img(x, height - y - 1)
                    o[0] = px.x;
                    o[1] = px.y;
                    o[2] = px.z;
                }
            }
        }
    };
    if (threads <= 1) {
        worker(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(worker, t);
        for (auto &t : pool) t.join();
    }
    if (total) {
        for (auto &c : counters) {
            total->bounces += c.bounces;
            for (int k = 0; k < 2; k++) {
                TraversalCounters &dst = k ? total->shadow : total->closest;
                const TraversalCounters &src = k ? c.shadow : c.closest;
                dst.rays += src.rays;
                dst.node_visits += src.node_visits;
                dst.box_tests += src.box_tests;
                dst.prim_tests += src.prim_tests;
            }
        }
    }
}

}  // namespace oracle
