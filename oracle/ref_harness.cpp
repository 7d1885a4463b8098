// ref_harness.cpp — TEST INFRASTRUCTURE.  Drives the *compiled reference* (TaKe, built from
// /root/reference by oracle/Makefile target `ref`, outputs in oracle/_ref/) to produce the
// golden vectors under tests/golden/.  It exists only in the authoring container: the GPU box
// never sees /root/reference, so nothing here runs there.
//
// It contains no reference code: it includes the reference's headers from where they lie
// (-I/root/reference/src) and calls the reference's own functions:
//   render()            src/render.cpp:9      (seed-patched at :60, see oracle/Makefile)
//   path_tracing()      src/integrator/path_tracing.h:5, and the three integrators the reference defines but never
//                       calls: path_tracing_raw :114, path_tracing_one_sample_MIS :161, .._power :274 (+ light_power()
//                       src/light.cpp:25, with which the power tables the parser leaves empty are filled here)
//   scene_intersect()   src/scene.cpp:25,  scene_occluded() src/scene.cpp:49
//   intersect(BBox,Ray) src/bbox.h:18,     intersect_shape() src/shape.h:42
//   construct_bvh()     src/bvh.cpp:8
//   sample_bsdf / get_bsdf_pdf / eval      src/material.cpp:76-98
//   sample_on_shape / get_light_pdf        src/shape.h:55, src/light.cpp:32
//   eval(Texture)       src/texture.h:48,  to_world() src/vector.h:314, random_real() src/take.h:89
//
// Tables are raw little-endian float64, row-major; the column layouts are defined by
// oracle/gen_golden.py, which generates the inputs and documents the outputs.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "image.h"
#include "parallel.h"
#include "parse/parse_ply.h"
#include "parse/parse_scene.h"
#include "parse/parse_serialized.h"
#include "render.h"
#include "scene.h"

#include "take_flatten.hpp"
#include "take_sceneio.hpp"

// defined (non-inline) in the render.cpp translation unit via integrator/path_tracing.h (:5, :114, :161, :274)
Vector3 path_tracing(const Scene &scene, const Ray &ray, std::mt19937 &rng);
Vector3 path_tracing_raw(const Scene &scene, const Ray &ray, std::mt19937 &rng);
Vector3 path_tracing_one_sample_MIS(const Scene &scene, const Ray &ray, std::mt19937 &rng);
Vector3 path_tracing_one_sample_MIS_power(const Scene &scene, const Ray &ray, std::mt19937 &rng);

static std::vector<double> read_f64(const std::string &path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) {
        fprintf(stderr, "cannot open %s\n", path.c_str());
        exit(2);
    }
    size_t n = (size_t)f.tellg();
    f.seekg(0);
    std::vector<double> v(n / 8);
    f.read((char *)v.data(), (std::streamsize)(v.size() * 8));
    return v;
}
static void write_f64(const std::string &path, const std::vector<double> &v) {
    std::ofstream f(path, std::ios::binary);
    f.write((const char *)v.data(), (std::streamsize)(v.size() * 8));
}
static Vector3 V3(const double *p) { return Vector3{p[0], p[1], p[2]}; }
static void push3(std::vector<double> &o, const Vector3 &v) {
    o.push_back(v.x);
    o.push_back(v.y);
    o.push_back(v.z);
}
static void push_isect(std::vector<double> &o, const std::optional<Intersection> &h) {
    // 1 + 1 + 3 + 3 + 3 + 2 + 2 = 15 columns
    if (!h) {
        for (int i = 0; i < 15; i++) o.push_back(i == 0 ? 0.0 : 0.0);
        return;
    }
    o.push_back(1.0);
    o.push_back(h->t);
    push3(o, h->pos);
    push3(o, h->geo_normal);
    push3(o, h->shading_normal);
    o.push_back(h->uv.x);
    o.push_back(h->uv.y);
    o.push_back((double)h->material_id);
    o.push_back((double)h->area_light_id);
}

static Material make_material(int tag, const Texture &tex, double p0, double p1) {
    switch (tag) {
        case 0: return Diffuse{tex};
        case 1: return Mirror{tex, p0};
        case 2: return Plastic{tex, p0};
        case 3: return Phong{tex, p0};
        case 4: return BlinnPhong{tex, p0};
        case 5: return BlinnPhongMicrofacet{tex, p0};
        case 6: return DisneyDiffuse{tex, p0, p1};
        case 7: return DisneyMetal{tex, p0, p1};
        case 8: return DisneyGlass{tex, p0, p1, 1.5};
        case 9: return DisneyClearcoat{p0};
        case 10: return DisneySheen{tex, p0};
        default: return DisneyBSDF{tex, 0, 0, 0, 0.5, 0.5, 0, 0, 0, 0.5, 0, 1, 1.5};
    }
}

static Scene load_scene(const std::string &xml, int max_depth) {
    Scene scene = parse_scene(xml);
    scene.options.max_depth = max_depth;
    build_bvh(scene);
    return scene;
}

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: ref_harness <cmd> ...\n");
        return 2;
    }
    std::string cmd = argv[1];

    if (cmd == "render") {
        // render <scene.xml> <max_depth> <threads> <out.f64>   (out: i32 w,h as two doubles, then w*h*3 doubles)
        int threads = atoi(argv[4]);
        parallel_init(threads);
        Image3 img = render({argv[2], "-max_depth", argv[3]});
        parallel_cleanup();
        std::vector<double> o;
        o.push_back(img.width);
        o.push_back(img.height);
        for (auto &p : img.data) push3(o, p);
        write_f64(argv[5], o);
        return 0;
    }
    if (cmd == "imwrite") {
        // imwrite <in.f64: w h then w*h*3 doubles, Image3 order> <out.exr|.pfm>   the reference's own writer
        auto in = read_f64(argv[2]);
        Image3 img((int)in[0], (int)in[1]);
        for (size_t i = 0; i < img.data.size(); i++) img.data[i] = Vector3{in[2 + 3 * i], in[3 + 3 * i], in[4 + 3 * i]};
        imwrite(argv[3], img);
        return 0;
    }
    if (cmd == "flatten") {
        // flatten <scene.xml> <out.tkscene>
        Scene scene = parse_scene(std::string(argv[2]));
        take_hip::FlatScene flat;
        take_hip::flatten_scene(scene, flat);
        take_hip::write_tkscene(argv[3], flat.desc, scene.options.spp, scene.options.max_depth);
        return 0;
    }
    if (cmd == "ply") {
        // ply <mesh.ply> <in.f64: to_world, 16 doubles row-major> <out.f64>   the reference's own parse_ply
        // (src/parse/parse_ply.cpp:9-123).  out: nv nf has_normals has_uvs, the reference's inverse(to_world) (16, what
        // parse_ply pushes the normals through), then positions (3 nv), indices (3 nf, as doubles), normals (3 nv if
        // any), uvs (2 nv if any)
        auto in = read_f64(argv[3]);
        Matrix4x4 m;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) m(i, j) = in[4 * i + j];
        TriangleMesh mesh = parse_ply(argv[2], m);
        std::vector<double> o;
        o.push_back((double)mesh.positions.size());
        o.push_back((double)mesh.indices.size());
        o.push_back(mesh.normals.empty() ? 0.0 : 1.0);
        o.push_back(mesh.uvs.empty() ? 0.0 : 1.0);
        const Matrix4x4 inv = inverse(m);
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) o.push_back(inv(i, j));
        for (auto &v : mesh.positions) push3(o, v);
        for (auto &f : mesh.indices) o.push_back(f[0]), o.push_back(f[1]), o.push_back(f[2]);
        for (auto &v : mesh.normals) push3(o, v);
        for (auto &v : mesh.uvs) o.push_back(v.x), o.push_back(v.y);
        write_f64(argv[4], o);
        return 0;
    }
    if (cmd == "serialized") {
        // serialized <mesh.serialized> <shape_index> <in.f64: to_world> <out.f64>   the reference's own parse_serialized
        // (src/parse/parse_serialized.cpp:174-256); out as in `ply`
        auto in = read_f64(argv[4]);
        Matrix4x4 m;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) m(i, j) = in[4 * i + j];
        TriangleMesh mesh = parse_serialized(argv[2], atoi(argv[3]), m);
        std::vector<double> o;
        o.push_back((double)mesh.positions.size());
        o.push_back((double)mesh.indices.size());
        o.push_back(mesh.normals.empty() ? 0.0 : 1.0);
        o.push_back(mesh.uvs.empty() ? 0.0 : 1.0);
        const Matrix4x4 inv = inverse(m);
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) o.push_back(inv(i, j));
        for (auto &v : mesh.positions) push3(o, v);
        for (auto &f : mesh.indices) o.push_back(f[0]), o.push_back(f[1]), o.push_back(f[2]);
        for (auto &v : mesh.normals) push3(o, v);
        for (auto &v : mesh.uvs) o.push_back(v.x), o.push_back(v.y);
        write_f64(argv[5], o);
        return 0;
    }
    if (cmd == "serialized_time") {
        // serialized_time <mesh.serialized>   seconds the reference's parse_serialized takes on this host
        const auto t0 = std::chrono::steady_clock::now();
        TriangleMesh mesh = parse_serialized(argv[2], 0, Matrix4x4::identity());
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%.6f %zu %zu\n", s, mesh.positions.size(), mesh.indices.size());
        return 0;
    }
    if (cmd == "ply_time") {
        // ply_time <mesh.ply>   seconds the reference's parse_ply takes on this host (tools/diag_ply.py)
        const auto t0 = std::chrono::steady_clock::now();
        TriangleMesh mesh = parse_ply(argv[2], Matrix4x4::identity());
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%.6f %zu %zu\n", s, mesh.positions.size(), mesh.indices.size());
        return 0;
    }
    if (cmd == "random_real") {
        // random_real <in: seed> <out: 64 values per seed>
        auto in = read_f64(argv[2]);
        std::vector<double> o;
        for (double s : in) {
            std::mt19937 rng{(unsigned)s};
            for (int i = 0; i < 64; i++) o.push_back(random_real(rng));
        }
        write_f64(argv[3], o);
        return 0;
    }
    if (cmd == "slab") {
        // in: bmin3 bmax3 org3 dir3 tmin tmax (14) -> out: hit (1)
        auto in = read_f64(argv[2]);
        std::vector<double> o;
        for (size_t r = 0; r + 14 <= in.size(); r += 14) {
            const double *p = &in[r];
            BBox b{V3(p), V3(p + 3)};
            Ray ray{V3(p + 6), V3(p + 9), p[12], p[13]};
            o.push_back(intersect(b, ray) ? 1.0 : 0.0);
        }
        write_f64(argv[3], o);
        return 0;
    }
    if (cmd == "tri") {
        // in: v0 v1 v2 (9) org dir (6) tmin tmax (2) has_n has_uv (2) n0 n1 n2 (9) uv0 uv1 uv2 (6) = 34
        // out: isect (15)
        auto in = read_f64(argv[2]);
        std::vector<double> o;
        for (size_t r = 0; r + 34 <= in.size(); r += 34) {
            const double *p = &in[r];
            std::vector<TriangleMesh> meshes(1);
            TriangleMesh &m = meshes[0];
            m.material_id = 3;
            m.positions = {V3(p), V3(p + 3), V3(p + 6)};
            m.indices = {Vector3i{0, 1, 2}};
            if (p[17] != 0) m.normals = {V3(p + 19), V3(p + 22), V3(p + 25)};
            if (p[18] != 0) m.uvs = {Vector2{p[28], p[29]}, Vector2{p[30], p[31]}, Vector2{p[32], p[33]}};
            Triangle tri{{3, 7}, 0, 0};
            Ray ray{V3(p + 9), V3(p + 12), p[15], p[16]};
            push_isect(o, intersect_shape(Shape{tri}, meshes, ray));
        }
        write_f64(argv[3], o);
        return 0;
    }
    if (cmd == "sphere") {
        // in: center3 radius org3 dir3 tmin tmax (12) -> out: isect (15)
        auto in = read_f64(argv[2]);
        std::vector<double> o;
        std::vector<TriangleMesh> meshes;
        for (size_t r = 0; r + 12 <= in.size(); r += 12) {
            const double *p = &in[r];
            Sphere s{{5, -1}, V3(p), p[3]};
            Ray ray{V3(p + 4), V3(p + 7), p[10], p[11]};
            push_isect(o, intersect_shape(Shape{s}, meshes, ray));
        }
        write_f64(argv[3], o);
        return 0;
    }
    if (cmd == "to_world") {
        // in: n3 v3 (6) -> out: 3
        auto in = read_f64(argv[2]);
        std::vector<double> o;
        for (size_t r = 0; r + 6 <= in.size(); r += 6) push3(o, to_world(V3(&in[r]), V3(&in[r + 3])));
        write_f64(argv[3], o);
        return 0;
    }
    if (cmd == "hemicos") {
        // in: seed -> out: dir3, next random_real (4)
        auto in = read_f64(argv[2]);
        std::vector<double> o;
        for (double s : in) {
            std::mt19937 rng{(unsigned)s};
            push3(o, sample_hemisphere_cos(rng));
            o.push_back(random_real(rng));
        }
        write_f64(argv[3], o);
        return 0;
    }
    if (cmd == "material") {
        // in (27): tag, color3, p0, p1, geo_n3, sh_n3, uv2, dir_in3, dir_out3, rec_pdf, seed, tex_kind, uscale,vscale,uoff,voff
        //   -> cols 0 | 1..3 | 4 5 | 6..8 | 9..11 | 12 13 | 14..16 | 17..19 | 20 | 21 | 22 | 23..26
        // out (14): has_rec, rec_dir3, rec_pdf, next_random, eval_sampled3, pdf(dir_in,dir_out), eval_given3 (record{dir_out,rec_pdf}), pdf_sampled_dir
        // The texture pool holds one fixed procedural 5x4 image (same formula in gen_golden.py).
        auto in = read_f64(argv[2]);
        TexturePool pool;
        Image3 img(5, 4);
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 5; x++)
                img(x, y) = Vector3{0.1 + 0.15 * x + 0.01 * y, 0.9 - 0.2 * y + 0.02 * x, 0.3 + 0.05 * ((x * 3 + y * 7) % 5)};
        pool.image3s.push_back(img);
        std::vector<double> o;
        for (size_t r = 0; r + 27 <= in.size(); r += 27) {
            const double *p = &in[r];
            Texture tex = (p[22] != 0) ? Texture{ImageTexture{0, p[23], p[24], p[25], p[26]}}
                                       : Texture{ConstTexture{V3(p + 1)}};
            Material m = make_material((int)p[0], tex, p[4], p[5]);
            Intersection v{};
            v.pos = Vector3{0, 0, 0};
            v.geo_normal = V3(p + 6);
            v.shading_normal = V3(p + 9);
            v.uv = Vector2{p[12], p[13]};
            v.t = 1;
            v.material_id = 0;
            v.area_light_id = -1;
            Vector3 dir_in = V3(p + 14), dir_out = V3(p + 17);
            std::mt19937 rng{(unsigned)p[21]};
            auto rec = sample_bsdf(m, dir_in, v, pool, rng);
            double next = random_real(rng);
            if (rec) {
                o.push_back(1.0);
                push3(o, rec->dir_out);
                o.push_back(rec->pdf);
                o.push_back(next);
                push3(o, eval(m, dir_in, *rec, v, pool));
            } else {
                o.push_back(0.0);
                for (int i = 0; i < 4; i++) o.push_back(0.0);
                o.push_back(next);
                for (int i = 0; i < 3; i++) o.push_back(0.0);
            }
            o.push_back(get_bsdf_pdf(m, dir_in, dir_out, v, pool));
            SampleRecord given{dir_out, p[20]};
            push3(o, eval(m, dir_in, given, v, pool));
            o.push_back(rec ? get_bsdf_pdf(m, dir_in, rec->dir_out, v, pool) : 0.0);
        }
        write_f64(argv[3], o);
        return 0;
    }
    if (cmd == "texture") {
        // in: u v uscale vscale uoff voff (6) -> out rgb (3); same 5x4 image as "material"
        auto in = read_f64(argv[2]);
        TexturePool pool;
        Image3 img(5, 4);
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 5; x++)
                img(x, y) = Vector3{0.1 + 0.15 * x + 0.01 * y, 0.9 - 0.2 * y + 0.02 * x, 0.3 + 0.05 * ((x * 3 + y * 7) % 5)};
        pool.image3s.push_back(img);
        std::vector<double> o;
        for (size_t r = 0; r + 6 <= in.size(); r += 6) {
            const double *p = &in[r];
            Texture tex{ImageTexture{0, p[2], p[3], p[4], p[5]}};
            push3(o, eval(tex, Vector2{p[0], p[1]}, pool));
        }
        write_f64(argv[3], o);
        return 0;
    }
    if (cmd == "light") {
        // in (30): kind(0 sphere,1 tri), center3/radius or v0 v1 v2 (9, cols 1..9), n0 n1 n2 (9, cols 10..18),
        //          ref_pos3 (19..21), seed (22), n_lights (23), query point3 (24..26), query normal3 (27..29)
        // out (9): sampled pos3, normal3, next_random, get_light_pdf(sampled point), get_light_pdf(query point)
        auto in = read_f64(argv[2]);
        std::vector<double> o;
        for (size_t r = 0; r + 30 <= in.size(); r += 30) {
            const double *p = &in[r];
            Scene scene;
            if (p[0] == 0) {
                scene.shapes.push_back(Sphere{{0, 0}, V3(p + 1), p[4]});
            } else {
                TriangleMesh m;
                m.material_id = 0;
                m.positions = {V3(p + 1), V3(p + 4), V3(p + 7)};
                m.indices = {Vector3i{0, 1, 2}};
                m.normals = {V3(p + 10), V3(p + 13), V3(p + 16)};
                scene.meshes.push_back(m);
                scene.shapes.push_back(Triangle{{0, 0}, 0, 0});
            }
            scene.lights.push_back(DiffuseAreaLight{0, Vector3{1, 2, 3}});
            Vector3 ref = V3(p + 19);
            std::mt19937 rng{(unsigned)p[22]};
            PointAndNormal pn = sample_on_light(scene, scene.lights[0], ref, rng);
            push3(o, pn.position);
            push3(o, pn.normal);
            o.push_back(random_real(rng));
            o.push_back(get_light_pdf(scene, 0, pn, ref));
            o.push_back(get_light_pdf(scene, 0, PointAndNormal{V3(p + 24), V3(p + 27)}, ref));
        }
        write_f64(argv[3], o);
        return 0;
    }
    if (cmd == "bvh") {
        // in: n boxes (bmin3 bmax3) -> out: root_id, n_nodes, then per node: bmin3 bmax3 left right prim (9)
        auto in = read_f64(argv[2]);
        std::vector<BBoxWithID> boxes;
        for (size_t r = 0; r + 6 <= in.size(); r += 6) boxes.push_back({BBox{V3(&in[r]), V3(&in[r + 3])}, (int)(r / 6)});
        std::vector<BVHNode> nodes;
        int root = construct_bvh(boxes, nodes);
        std::vector<double> o{(double)root, (double)nodes.size()};
        for (auto &n : nodes) {
            push3(o, n.box.p_min);
            push3(o, n.box.p_max);
            o.push_back(n.left_node_id);
            o.push_back(n.right_node_id);
            o.push_back(n.primitive_id);
        }
        write_f64(argv[3], o);
        return 0;
    }
    if (cmd == "isect") {
        // isect <scene.xml> <in: org3 dir3 tmin tmax (8)> <out: isect(15) + occluded(1)>
        Scene scene = load_scene(argv[2], 50);
        auto in = read_f64(argv[3]);
        std::vector<double> o;
        for (size_t r = 0; r + 8 <= in.size(); r += 8) {
            const double *p = &in[r];
            Ray ray{V3(p), V3(p + 3), p[6], p[7]};
            push_isect(o, scene_intersect(scene, ray));
            o.push_back(scene_occluded(scene, ray) ? 1.0 : 0.0);
        }
        write_f64(argv[4], o);
        return 0;
    }
    if (cmd == "pt") {
        // pt <scene.xml> <max_depth> <in: org3 dir3 seed (7)> <out: radiance3 next_random (4)> [integrator]
        // integrator: 0 path_tracing (default), 1 path_tracing_raw, 2 path_tracing_one_sample_MIS,
        //             3 path_tracing_one_sample_MIS_power
        Scene scene = load_scene(argv[2], atoi(argv[3]));
        const int integrator = argc > 6 ? atoi(argv[6]) : 0;
        if (integrator == 3) {
            // Scene::lights_power_pmf / _cdf (src/scene.h:28-29) are read by sample_light_power / get_light_pmf
            // (src/light.cpp:9-23) but no code of the reference fills them.  Filled here from the reference's own
            // light_power() (src/light.cpp:25-30): pmf = power / sum, cdf = running sum starting at 0 (n + 1 entries,
            // the layout sample_light_power's upper_bound expects).
            Real total = 0;
            std::vector<Real> power;
            for (auto &l : scene.lights) power.push_back(light_power(scene, l)), total += power.back();
            scene.lights_power_pmf.clear();
            scene.lights_power_cdf.assign(1, Real(0));
            for (Real p : power) {
                scene.lights_power_pmf.push_back(p / total);
                scene.lights_power_cdf.push_back(scene.lights_power_cdf.back() + p / total);
            }
        }
        auto in = read_f64(argv[4]);
        std::vector<double> o;
        for (size_t r = 0; r + 7 <= in.size(); r += 7) {
            const double *p = &in[r];
            Ray ray{V3(p), V3(p + 3), c_EPSILON, infinity<Real>()};
            std::mt19937 rng{(unsigned)p[6]};
            Vector3 L = integrator == 1   ? path_tracing_raw(scene, ray, rng)
                        : integrator == 2 ? path_tracing_one_sample_MIS(scene, ray, rng)
                        : integrator == 3 ? path_tracing_one_sample_MIS_power(scene, ray, rng)
                                          : path_tracing(scene, ray, rng);
            push3(o, L);
            o.push_back(random_real(rng));
        }
        write_f64(argv[5], o);
        return 0;
    }
    if (cmd == "time") {
        // time <scene.xml> <max_depth> <threads>  -> prints the reference's own timers (render.cpp:58,83)
        int threads = atoi(argv[4]);
        parallel_init(threads);
        Image3 img = render({argv[2], "-max_depth", argv[3]});
        parallel_cleanup();
        double s = 0;
        for (auto &p : img.data) s += p.x + p.y + p.z;
        printf("checksum %.17g\n", s);
        return 0;
    }
    fprintf(stderr, "unknown command %s\n", cmd.c_str());
    return 2;
}
