// take_sceneio.hpp — write a TakeSceneDesc as a `.tkscene` file (little-endian).
//
// The format exists so that a scene parsed once on the host (by the reference's own
// XML parser through take_flatten.hpp, or by take_amd/scenes.py) can be replayed on a
// GPU box that has neither the reference nor its scene files.  The reader lives in
// take_amd/sceneio.py; layout:
//
//   char[8]  "TKSCENE1"
//   i32 w,h; f64 lookfrom[3], lookat[3], up[3], vfov; f64 background[3]; i32 spp, max_depth
//   i32 n_meshes;    per mesh: i64 nv, nf; i32 material_id, has_normals, has_uvs, 0;
//                              f64 pos[nv*3]; i32 idx[nf*3]; [f64 nrm[nv*3]]; [f64 uv[nv*2]]
//   i32 n_spheres;   per sphere: f64 center[3], radius; i32 material_id, 0
//   i64 n_shapes;    i32 kind[n], ref[n], face[n], area_light[n]
//   i32 n_lights;    per light: i32 kind, shape_id; f64 intensity[3], position[3]
//   i32 n_materials; per material: i32 tag, tex_kind, tex_image, n_more (0 or 8);
//                              f64 value[3], uscale, vscale, uoffset, voffset, param[4], param[4 .. 4 + n_more)
//   i32 n_images;    per image: i32 w, h; f64 data[w*h*3]
#pragma once

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "take_hip.h"

namespace take_hip {

namespace detail {
struct FileWriter {
    FILE *f;
    explicit FileWriter(const std::string &path) : f(std::fopen(path.c_str(), "wb")) {
        if (!f) throw std::runtime_error("cannot open " + path);
    }
    ~FileWriter() {
        if (f) std::fclose(f);
    }
    void raw(const void *p, size_t n) {
        if (n && std::fwrite(p, 1, n, f) != n) throw std::runtime_error("short write");
    }
    void i32(int32_t v) { raw(&v, 4); }
    void i64(int64_t v) { raw(&v, 8); }
    void f64(double v) { raw(&v, 8); }
    void f64n(const double *p, size_t n) { raw(p, n * 8); }
    void i32n(const int32_t *p, size_t n) { raw(p, n * 4); }
};
}  // namespace detail

inline void write_tkscene(const std::string &path, const TakeSceneDesc &d, int spp, int max_depth) {
    detail::FileWriter w(path);
    w.raw("TKSCENE1", 8);
    w.i32(d.camera.width);
    w.i32(d.camera.height);
    w.f64n(d.camera.lookfrom, 3);
    w.f64n(d.camera.lookat, 3);
    w.f64n(d.camera.up, 3);
    w.f64(d.camera.vfov);
    w.f64n(d.background, 3);
    w.i32(spp);
    w.i32(max_depth);
    w.i32(d.n_meshes);
    for (int i = 0; i < d.n_meshes; i++) {
        const TakeMesh &m = d.meshes[i];
        w.i64(m.n_vertices);
        w.i64(m.n_faces);
        w.i32(m.material_id);
        w.i32(m.normals ? 1 : 0);
        w.i32(m.uvs ? 1 : 0);
        w.i32(0);
        w.f64n(m.positions, (size_t)m.n_vertices * 3);
        w.i32n(m.indices, (size_t)m.n_faces * 3);
        if (m.normals) w.f64n(m.normals, (size_t)m.n_vertices * 3);
        if (m.uvs) w.f64n(m.uvs, (size_t)m.n_vertices * 2);
    }
    w.i32(d.n_spheres);
    for (int i = 0; i < d.n_spheres; i++) {
        w.f64n(d.spheres[i].center, 3);
        w.f64(d.spheres[i].radius);
        w.i32(d.spheres[i].material_id);
        w.i32(0);
    }
    w.i64(d.n_shapes);
    w.i32n(d.shape_kind, (size_t)d.n_shapes);
    w.i32n(d.shape_ref, (size_t)d.n_shapes);
    w.i32n(d.shape_face, (size_t)d.n_shapes);
    w.i32n(d.shape_area_light, (size_t)d.n_shapes);
    w.i32(d.n_lights);
    for (int i = 0; i < d.n_lights; i++) {
        w.i32(d.lights[i].kind);
        w.i32(d.lights[i].shape_id);
        w.f64n(d.lights[i].intensity, 3);
        w.f64n(d.lights[i].position, 3);
    }
    w.i32(d.n_materials);
    for (int i = 0; i < d.n_materials; i++) {
        const TakeMaterial &m = d.materials[i];
        w.i32(m.tag);
        w.i32(m.reflectance.kind);
        w.i32(m.reflectance.image_id);
        bool more = false;  // files of scenes without Disney parameters stay byte-identical to the first format
        for (int k = 4; k < TAKE_MATERIAL_PARAMS; k++) more = more || m.param[k] != 0.0;
        w.i32(more ? TAKE_MATERIAL_PARAMS - 4 : 0);
        w.f64n(m.reflectance.value, 3);
        w.f64(m.reflectance.uscale);
        w.f64(m.reflectance.vscale);
        w.f64(m.reflectance.uoffset);
        w.f64(m.reflectance.voffset);
        w.f64n(m.param, more ? TAKE_MATERIAL_PARAMS : 4);
    }
    w.i32(d.n_images);
    for (int i = 0; i < d.n_images; i++) {
        w.i32(d.images[i].width);
        w.i32(d.images[i].height);
        w.f64n(d.images[i].data, (size_t)d.images[i].width * d.images[i].height * 3);
    }
}

}  // namespace take_hip
