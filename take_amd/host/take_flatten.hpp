// take_flatten.hpp — host-side adapter: a TaKe `Scene` -> `TakeSceneDesc`.
//
// This is the reference-side half of the drop-in boundary (include/take_hip.h).  It is
// a header-only template so that it compiles against the reference's own headers
// without this repository containing any of them: `SceneT` is the reference `Scene`
// (src/scene.h:13-33) and the alternative types are found through std::variant
// introspection, by position:
//   Shape    = variant<Sphere, Triangle>                         (src/shape.h:32)
//   Light    = variant<PointLight, DiffuseAreaLight>             (src/light.h:19)
//   Texture  = variant<ConstTexture, ImageTexture>               (src/texture.h:27)
//   Material = variant<Diffuse, Mirror, ... DisneyBSDF>          (src/material.h:82-93)
// INTEGRATION.md shows the call site a maintainer adds to src/render.cpp.
//
// The adapter copies nothing it does not have to: mesh arrays are referenced in place
// (TVector3<double> is 3 packed doubles, src/vector.h:30-48), only the per-shape SoA
// arrays and the small material/light tables are materialised here.
#pragma once

#include <cstdint>
#include <cstring>
#include <type_traits>
#include <variant>
#include <vector>

#include "take_hip.h"

namespace take_hip {

// Owns the small tables a TakeSceneDesc points at.  Keep it alive until
// take_hip_scene_create() has returned.
struct FlatScene {
    TakeSceneDesc desc{};
    std::vector<TakeMesh> meshes;
    std::vector<TakeSphere> spheres;
    std::vector<int32_t> shape_kind, shape_ref, shape_face, shape_area_light;
    std::vector<TakeLight> lights;
    std::vector<TakeMaterial> materials;
    std::vector<TakeImage3> images;
};

namespace detail {

template <class V3> inline void put3(double *dst, const V3 &v) {
    dst[0] = double(v.x);
    dst[1] = double(v.y);
    dst[2] = double(v.z);
}

template <class TextureT> inline TakeTexture flatten_texture(const TextureT &t) {
    TakeTexture out{};
    if (t.index() == 0) {
        const auto &c = std::get<0>(t);
        out.kind = 0;
        put3(out.value, c.value);
    } else {
        const auto &im = std::get<1>(t);
        out.kind = 1;
        out.image_id = im.texture_id;
        out.uscale = im.uscale;
        out.vscale = im.vscale;
        out.uoffset = im.uoffset;
        out.voffset = im.voffset;
    }
    return out;
}

// members are probed by name so one visitor serves all 12 alternatives: member_<name>(alt) is the member's value,
// or 0 where the alternative has no such member
template <class M, class = void> struct has_reflectance : std::false_type {};
template <class M> struct has_reflectance<M, std::void_t<decltype(std::declval<M>().reflectance)>> : std::true_type {};
#define TAKE_HIP_MEMBER_PROBE(name)                                                                          \
    template <class M, class = void> struct has_##name : std::false_type {};                                   \
    template <class M> struct has_##name<M, std::void_t<decltype(std::declval<M>().name)>> : std::true_type {}; \
    template <class M> inline double member_##name(const M &m) {                                               \
        if constexpr (has_##name<M>::value)                                                                    \
            return double(m.name);                                                                             \
        else                                                                                                   \
            return 0.0;                                                                                        \
    }
TAKE_HIP_MEMBER_PROBE(eta)
TAKE_HIP_MEMBER_PROBE(exponent)
TAKE_HIP_MEMBER_PROBE(roughness)
TAKE_HIP_MEMBER_PROBE(subsurface)
TAKE_HIP_MEMBER_PROBE(anisotropic)
TAKE_HIP_MEMBER_PROBE(clearcoat_gloss)
TAKE_HIP_MEMBER_PROBE(sheen_tint)
TAKE_HIP_MEMBER_PROBE(specular_transmission)
TAKE_HIP_MEMBER_PROBE(metallic)
TAKE_HIP_MEMBER_PROBE(specular)
TAKE_HIP_MEMBER_PROBE(specular_tint)
TAKE_HIP_MEMBER_PROBE(sheen)
TAKE_HIP_MEMBER_PROBE(clearcoat)
#undef TAKE_HIP_MEMBER_PROBE

}  // namespace detail

template <class SceneT> inline void flatten_scene(const SceneT &scene, FlatScene &out) {
    using namespace detail;
    TakeSceneDesc &d = out.desc;
    std::memset(&d, 0, sizeof d);

    d.camera.width = scene.camera.width;
    d.camera.height = scene.camera.height;
    put3(d.camera.lookfrom, scene.camera.lookfrom);
    put3(d.camera.lookat, scene.camera.lookat);
    put3(d.camera.up, scene.camera.up);
    d.camera.vfov = scene.camera.vfov;
    put3(d.background, scene.background_color);

    out.meshes.clear();
    for (const auto &m : scene.meshes) {
        static_assert(sizeof(m.positions[0]) == 3 * sizeof(double), "Real must be double");
        TakeMesh tm{};
        tm.n_vertices = (int64_t)m.positions.size();
        tm.n_faces = (int64_t)m.indices.size();
        tm.positions = reinterpret_cast<const double *>(m.positions.data());
        tm.indices = reinterpret_cast<const int32_t *>(m.indices.data());
        tm.normals = m.normals.empty() ? nullptr : reinterpret_cast<const double *>(m.normals.data());
        tm.uvs = m.uvs.empty() ? nullptr : reinterpret_cast<const double *>(m.uvs.data());
        tm.material_id = m.material_id;
        out.meshes.push_back(tm);
    }

    const size_t ns = scene.shapes.size();
    out.spheres.clear();
    out.shape_kind.assign(ns, 0);
    out.shape_ref.assign(ns, 0);
    out.shape_face.assign(ns, 0);
    out.shape_area_light.assign(ns, -1);
    for (size_t i = 0; i < ns; i++) {
        const auto &s = scene.shapes[i];
        if (s.index() == 0) {  // Sphere
            const auto &sp = std::get<0>(s);
            TakeSphere ts{};
            put3(ts.center, sp.center);
            ts.radius = sp.radius;
            ts.material_id = sp.material_id;
            out.shape_kind[i] = 0;
            out.shape_ref[i] = (int32_t)out.spheres.size();
            out.shape_area_light[i] = sp.area_light_id;
            out.spheres.push_back(ts);
        } else {  // Triangle
            const auto &tr = std::get<1>(s);
            out.shape_kind[i] = 1;
            out.shape_ref[i] = tr.mesh_id;
            out.shape_face[i] = tr.face_id;
            out.shape_area_light[i] = tr.area_light_id;
        }
    }

    out.lights.clear();
    for (const auto &l : scene.lights) {
        TakeLight tl{};
        if (l.index() == 0) {
            const auto &p = std::get<0>(l);
            tl.kind = 0;
            tl.shape_id = -1;
            put3(tl.intensity, p.intensity);
            put3(tl.position, p.position);
        } else {
            const auto &a = std::get<1>(l);
            tl.kind = 1;
            tl.shape_id = a.shape_id;
            put3(tl.intensity, a.intensity);
        }
        out.lights.push_back(tl);
    }

    out.materials.clear();
    for (const auto &m : scene.materials) {
        TakeMaterial tm{};
        tm.tag = (int32_t)m.index();
        tm.reflectance.kind = 0;
        std::visit(
            [&](const auto &alt) {
                using A = std::decay_t<decltype(alt)>;
                if constexpr (has_reflectance<A>::value) tm.reflectance = flatten_texture(alt.reflectance);
                // TakeMaterial::param = the alternative's scalar members in declaration order (src/material.h:7-80)
                double *p = tm.param;
                switch (tm.tag) {
                    case TAKE_MAT_MIRROR:
                    case TAKE_MAT_PLASTIC: p[0] = member_eta(alt); break;
                    case TAKE_MAT_PHONG:
                    case TAKE_MAT_BLINN_PHONG:
                    case TAKE_MAT_BLINN_PHONG_MICROFACET: p[0] = member_exponent(alt); break;
                    case TAKE_MAT_DISNEY_DIFFUSE: p[0] = member_roughness(alt), p[1] = member_subsurface(alt); break;
                    case TAKE_MAT_DISNEY_METAL: p[0] = member_roughness(alt), p[1] = member_anisotropic(alt); break;
                    case TAKE_MAT_DISNEY_GLASS:
                        p[0] = member_roughness(alt), p[1] = member_anisotropic(alt), p[2] = member_eta(alt);
                        break;
                    case TAKE_MAT_DISNEY_CLEARCOAT: p[0] = member_clearcoat_gloss(alt); break;
                    case TAKE_MAT_DISNEY_SHEEN: p[0] = member_sheen_tint(alt); break;
                    case TAKE_MAT_DISNEY_BSDF:
                        p[0] = member_specular_transmission(alt), p[1] = member_metallic(alt);
                        p[2] = member_subsurface(alt), p[3] = member_specular(alt), p[4] = member_roughness(alt);
                        p[5] = member_specular_tint(alt), p[6] = member_anisotropic(alt), p[7] = member_sheen(alt);
                        p[8] = member_sheen_tint(alt), p[9] = member_clearcoat(alt);
                        p[10] = member_clearcoat_gloss(alt), p[11] = member_eta(alt);
                        break;
                    default: break;
                }
            },
            m);
        // tags 7..11 carry their parameters although the reference's lobes (and tags 7..11 here) ignore them:
        // TakeBuildOpts.burley_lobes turns them into tags 12..16, which read them
        out.materials.push_back(tm);
    }

    out.images.clear();
    for (const auto &img : scene.textures.image3s) {
        TakeImage3 ti{};
        ti.width = img.width;
        ti.height = img.height;
        ti.data = reinterpret_cast<const double *>(img.data.data());
        out.images.push_back(ti);
    }

    d.n_meshes = (int32_t)out.meshes.size();
    d.meshes = out.meshes.data();
    d.n_spheres = (int32_t)out.spheres.size();
    d.spheres = out.spheres.data();
    d.n_shapes = (int64_t)ns;
    d.shape_kind = out.shape_kind.data();
    d.shape_ref = out.shape_ref.data();
    d.shape_face = out.shape_face.data();
    d.shape_area_light = out.shape_area_light.data();
    d.n_lights = (int32_t)out.lights.size();
    d.lights = out.lights.data();
    d.n_materials = (int32_t)out.materials.size();
    d.materials = out.materials.data();
    d.n_images = (int32_t)out.images.size();
    d.images = out.images.data();
}

}  // namespace take_hip
