// tk_shade.h — per-vertex work of the integrator as device functions: hit attributes, textures,
// the 12-way material dispatch (a switch on the tag the queue was sorted by), area-light sampling.
//
// Functional counterparts in the reference (the formulas and branch conditions are kept exactly, because
// they define the estimator; the dispatch, data layout and control flow are not):
//   hit attributes    src/shape.cpp:30-40, 80-108      textures   src/texture.cpp:3-25
//   materials         src/material.cpp:76-98 + src/materials/*.inl, src/material.h:121-140
//   lights            src/light.cpp:5-7,32-56, src/shape.cpp:125-184
#pragma once

#include "tk_scene.h"
#include "tk_traverse.h"

namespace tk {

template <class R> struct Isect {  // reference `Intersection` (src/intersection.h:4-12)
    Vec3<R> pos, gn, sn;
    Vec2<R> uv;
    int32_t material, area_light;
    bool back;  // gn was flipped to face the ray (not in the reference struct; which side of a dielectric: tk_burley.h)
};

template <class R> TK_HD Vec3<R> ld3(const R *p) { return {p[0], p[1], p[2]}; }

// Rebuild the full intersection record from (primitive, t, u, v) and the ray: src/shape.cpp:30-40 (sphere),
// :80-108 (triangle).  `prim` indexes the leaf-ordered primitive records (the closest-hit kernel reports it);
// vertex attributes come from the shading-side arrays through the shape id stored in the record.
// `inst` >= 0 (two-level scenes, EXTENSION): the primitive record is in the object space of its prototype; the
// placement's transform takes the triangle's edges (geometric normal) and the interpolated vertex normal to world
// space, exactly what flattening the instance would have produced up to rounding.  pos = ro + rd * t needs nothing:
// t is the same number in both spaces.
template <class R>
TK_HD void make_isect(const DeviceScene<R> &sc, Vec3<R> ro, Vec3<R> rd, int32_t prim, R t, R u, R v, Isect<R> &out,
                      int32_t inst = -1) {
#if defined(__HIP_DEVICE_COMPILE__)
    // the whole record with 16-byte loads (four in f32, six in f64), all in flight at once (tk_common.h: load_record;
    // measured on the f32 shade kernel: -8 %)
    PrimRec<R> whole;
    load_record(sc.prims + prim, whole);
    const PrimRec<R> &p = whole;
#else
    const PrimRec<R> &p = sc.prims[prim];
#endif
    out.material = p.material;
    out.area_light = p.area_light;
    out.pos = ro + rd * t;
    if ((p.meta & 0xff) == PRIM_SPHERE) {
        Vec3<R> n = normalize(out.pos - Vec3<R>{p.a[0], p.a[1], p.a[2]});
        out.back = !(dot(rd, n) < R(0));
        n = dot(rd, n) < R(0) ? n : -n;
        out.gn = n;
        out.sn = n;
        R theta = tk_acos(-n.y);  // get_sphere_uv, src/shape.cpp:3-11
        R phi = tk_atan2(-n.z, n.x) + Const<R>::PI;
        out.uv = {phi / (R(2) * Const<R>::PI), -theta / Const<R>::PI};
        return;
    }
    Vec3<R> e1{p.a[3], p.a[4], p.a[5]}, e2{p.a[6], p.a[7], p.a[8]};
    if (inst >= 0) {
        const InstShade<R> &is = sc.inst_shade[inst];
        const R *f = is.fwd;
        e1 = {f[0] * e1.x + f[1] * e1.y + f[2] * e1.z, f[3] * e1.x + f[4] * e1.y + f[5] * e1.z, f[6] * e1.x + f[7] * e1.y + f[8] * e1.z};
        e2 = {f[0] * e2.x + f[1] * e2.y + f[2] * e2.z, f[3] * e2.x + f[4] * e2.y + f[5] * e2.z, f[6] * e2.x + f[7] * e2.y + f[8] * e2.z};
        out.material = is.material;
    }
    Vec3<R> gn = normalize(cross(e1, e2));
    out.back = !(dot(rd, gn) < R(0));
    gn = dot(rd, gn) < R(0) ? gn : -gn;
    out.gn = gn;
    out.uv = {u, v};
    out.sn = gn;
    if (p.nidx < 0) return;  // mesh without vertex normals and uvs (src/shape.cpp:90,101)
    const MeshInfo mi = sc.meshes[p.mesh];
    const int32_t *idx = sc.face_idx + 3 * (int64_t)p.nidx;
    const int32_t i0 = idx[0], i1 = idx[1], i2 = idx[2];
    if (mi.uvbase >= 0) {
        const R *uv = sc.uvs + 2 * (int64_t)mi.uvbase;
        Vec2<R> uv0{uv[2 * i0], uv[2 * i0 + 1]}, uv1{uv[2 * i1], uv[2 * i1 + 1]}, uv2{uv[2 * i2], uv[2 * i2 + 1]};
        out.uv = (R(1) - u - v) * uv0 + u * uv1 + v * uv2;
    }
    if (mi.nbase >= 0) {
        const R *nn = sc.normals + 3 * (int64_t)mi.nbase;
        Vec3<R> n0 = ld3(nn + 3 * i0), n1 = ld3(nn + 3 * i1), n2 = ld3(nn + 3 * i2);
        Vec3<R> sn = (R(1) - u - v) * n0 + u * n1 + v * n2;
        if (inst >= 0) {  // normals transform with the transposed inverse: n_w = (L^-1)^T n
            const R *b = sc.inst_trace[inst].inv;
            sn = {b[0] * sn.x + b[4] * sn.y + b[8] * sn.z, b[1] * sn.x + b[5] * sn.y + b[9] * sn.z, b[2] * sn.x + b[6] * sn.y + b[10] * sn.z};
        }
        out.sn = normalize(sn);
    }
}

// ---- textures (src/texture.cpp:3-25; the wrap-seam arithmetic of :13-24 is kept as is)
template <class R> TK_HD R modulo1(R a) {
    R r = tk_fmod(a, R(1));
    return (r < R(0)) ? r + R(1) : r;
}
template <class R> TK_HD Vec3<R> eval_texture(const DeviceScene<R> &sc, const MaterialRec<R> &m, Vec2<R> uv) {
    if (m.tex_kind == 0) return ld3(m.color);
    const ImageInfo im = sc.images[m.tex_image];
    R x = R(im.width) * modulo1(m.uscale * uv.x + m.uoffset);
    R y = R(im.height) * modulo1(m.vscale * uv.y + m.voffset);
    int x1 = (int)tk_floor(x);
    int x2 = (x1 + 1) == im.width ? 0 : (x1 + 1);
    int y1 = (int)tk_floor(y);
    int y2 = (y1 + 1) == im.height ? 0 : (y1 + 1);
    const R *tx = sc.texels + 3 * im.offset;
    Vec3<R> q11 = ld3(tx + 3 * ((int64_t)y1 * im.width + x1)), q12 = ld3(tx + 3 * ((int64_t)y2 * im.width + x1));
    Vec3<R> q21 = ld3(tx + 3 * ((int64_t)y1 * im.width + x2)), q22 = ld3(tx + 3 * ((int64_t)y2 * im.width + x2));
    if (x1 == x2) x2 += 1;
    if (y1 == y2) y2 += 1;
    R fx2 = R(x2) - x, fx1 = x - R(x1), fy2 = R(y2) - y, fy1 = y - R(y1);
    return (q11 * fx2 * fy2 + q21 * fx1 * fy2 + q12 * fx2 * fy1 + q22 * fx1 * fy1) / R((x2 - x1) * (y2 - y1));
}

// ---- environment map (extension, see tk_scene.h: EnvMap).  Everything is piecewise constant per texel.
template <class R> struct EnvSample {
    Vec3<R> dir;
    Vec3<R> radiance;
    R pdf;  // solid-angle density of `dir`
};
template <class R> TK_HD Vec3<R> env_texel(const DeviceScene<R> &sc, int x, int y) {
    const R *t = sc.texels + 3 * (sc.env.texel0 + (int64_t)y * sc.env.width + x);
    return Vec3<R>{t[0] * sc.env.scale[0], t[1] * sc.env.scale[1], t[2] * sc.env.scale[2]};
}
// density of the texel (x, y) over the unit square times the Jacobian of the equirectangular map at polar angle theta
template <class R> TK_HD R env_texel_pdf(const DeviceScene<R> &sc, int x, int y, R sin_theta) {
    if (!(sin_theta > R(0))) return R(0);
    const R *row = sc.env.conditional + (int64_t)y * (sc.env.width + 1);
    const R p = (sc.env.marginal[y + 1] - sc.env.marginal[y]) * (row[x + 1] - row[x]) * R(sc.env.width) * R(sc.env.height);
    return p / (R(2) * Const<R>::PI * Const<R>::PI * sin_theta);
}
template <class R> TK_HD void env_lookup(const DeviceScene<R> &sc, Vec3<R> d, int &x, int &y, R &sin_theta) {
    const R cy = tk_clamp(d.y, R(-1), R(1));
    const R theta = tk_acos(cy);
    R u = tk_atan2(d.z, d.x) * Const<R>::INVTWOPI + R(0.5);
    x = (int)tk_floor(u * R(sc.env.width));
    y = (int)tk_floor(theta * Const<R>::INVPI * R(sc.env.height));
    x = x < 0 ? 0 : (x >= sc.env.width ? sc.env.width - 1 : x);
    y = y < 0 ? 0 : (y >= sc.env.height ? sc.env.height - 1 : y);
    sin_theta = tk_sqrt(tk_fmax(R(0), R(1) - cy * cy));
}
// radiance arriving from direction d, and the density with which env_sample would have produced d
template <class R> TK_HD Vec3<R> env_eval(const DeviceScene<R> &sc, Vec3<R> d, R &pdf) {
    int x, y;
    R st;
    env_lookup(sc, d, x, y, st);
    pdf = env_texel_pdf(sc, x, y, st);
    return env_texel(sc, x, y);
}
// largest i in [0, n) with cdf[i] <= xi (cdf[0] = 0, cdf[n] = 1, non-decreasing)
template <class R> TK_HD int cdf_find(const R *cdf, int n, R xi) {
    int lo = 0, hi = n;  // invariant: cdf[lo] <= xi, and (hi == n or cdf[hi] > xi)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] <= xi) lo = mid;
        else hi = mid;
    }
    return lo;
}
// the same interval, found from a guide table (EnvMap::guide_*): bisection between two neighbouring guide entries
template <class R> TK_HD int cdf_find_guided(const R *cdf, int n, const int32_t *guide, int n_guide, R xi) {
    int k = (int)(xi * R(n_guide));
    k = k < 0 ? 0 : (k >= n_guide ? n_guide - 1 : k);
    int lo = guide[k], hi = guide[k + 1] + 1;  // cdf[lo] <= k / n_guide <= xi < (k + 1) / n_guide < cdf[hi] (or hi == n)
    hi = hi > n ? n : hi;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] <= xi) lo = mid;
        else hi = mid;
    }
    return lo;
}
template <class R, class G> TK_HD EnvSample<R> env_sample(const DeviceScene<R> &sc, G &rng) {
    const R u1 = random_real<R>(rng);
    const R u2 = random_real<R>(rng);
    const int y = cdf_find_guided(sc.env.marginal, sc.env.height, sc.env.guide_m, sc.env.n_guide_m, u1);
    const R *row = sc.env.conditional + (int64_t)y * (sc.env.width + 1);
    const int x = cdf_find_guided(row, sc.env.width, sc.env.guide_c + (int64_t)y * (sc.env.n_guide_c + 1), sc.env.n_guide_c, u2);
    const R m0 = sc.env.marginal[y], m1 = sc.env.marginal[y + 1], c0 = row[x], c1 = row[x + 1];
    const R dv = m1 > m0 ? (u1 - m0) / (m1 - m0) : R(0.5), du = c1 > c0 ? (u2 - c0) / (c1 - c0) : R(0.5);
    const R theta = (R(y) + dv) / R(sc.env.height) * Const<R>::PI;
    const R phi = ((R(x) + du) / R(sc.env.width) - R(0.5)) * Const<R>::TWOPI;
    const R st = tk_sin(theta);
    EnvSample<R> s;
    s.dir = Vec3<R>{st * tk_cos(phi), tk_cos(theta), st * tk_sin(phi)};
    s.radiance = env_texel(sc, x, y);
    s.pdf = env_texel_pdf(sc, x, y, st);
    return s;
}

// ---- materials
template <class R> struct BsdfSample {
    Vec3<R> dir_out;
    R pdf;
};
template <class R> TK_HD Vec3<R> facing(Vec3<R> dir_in, const Isect<R> &v) {
    return dot(dir_in, v.sn) < R(0) ? -v.sn : v.sn;
}
template <class R, class G> TK_HD Vec3<R> hemisphere_cos(G &rng) {  // src/material.h:121-132
    R u1 = random_real<R>(rng);
    R u2 = random_real<R>(rng);
    R phi = Const<R>::TWOPI * u2;
    R s = tk_sqrt(tk_clamp(u1, R(0), R(1)));
    return {tk_cos(phi) * s, tk_sin(phi) * s, tk_sqrt(tk_clamp(R(1) - u1, R(0), R(1)))};
}
template <class R, class G> TK_HD Vec3<R> power_lobe(R exponent, G &rng) {  // src/materials/phong.inl:7-18
    R u1 = random_real<R>(rng);
    R u2 = random_real<R>(rng);
    R ra1 = R(1) / (exponent + R(1));
    R phi = Const<R>::TWOPI * u2;
    R s = tk_sqrt(tk_clamp(R(1) - tk_pow(u1, R(2) * ra1), R(0), R(1)));
    return normalize(Vec3<R>{tk_cos(phi) * s, tk_sin(phi) * s, tk_clamp(tk_pow(u1, ra1), R(0), R(1))});
}
template <class R> TK_HD R g_hat(Vec3<R> w, Vec3<R> n, R alpha) {  // src/material.h:134-140
    R odn = dot(w, n);
    R a = tk_sqrt(R(0.5) * alpha + R(1)) / tk_sqrt(R(1) / (odn * odn) - R(1));
    R a2 = a * a;
    return a < R(1.6) ? (R(3.535) * a + R(2.181) * a2) / (R(1) + R(2.276) * a + R(2.577) * a2) : R(1);
}
template <class R> TK_HD BsdfSample<R> cosine_sample(Vec3<R> n, const Isect<R> &v, Vec3<R> local) {
    BsdfSample<R> s;
    s.dir_out = to_world(n, local);
    s.pdf = dot(v.gn, s.dir_out) < R(0) ? R(0) : tk_fmax(dot(n, s.dir_out), R(0)) / Const<R>::PI;
    return s;
}

}  // namespace tk
#include "tk_burley.h"  // tags 12..16
namespace tk {

// TAG: the material tag as a compile-time constant (the shade kernels are instantiated per tag and launched over
// the tag's segment of the material-sorted queue, so the 12-way dispatch folds away); TAG = -1 reads m.tag.
// returns false when the reference returns an empty optional (dir_in below the geometric surface)
template <class R, int TAG = -1, class G>
TK_HD bool sample_bsdf(const MaterialRec<R> &m, Vec3<R> dir_in, const Isect<R> &v, G &rng, BsdfSample<R> &out) {
    if constexpr (TAG < 0 || tag_is_burley(TAG)) {
        if (tag_is_burley(TAG >= 0 ? TAG : m.tag)) return burley_sample<R, TAG>(m, dir_in, v, rng, out);
    }
    if (dot(v.gn, dir_in) < R(0)) return false;
    const Vec3<R> n = facing(dir_in, v);
    const int tag = TAG >= 0 ? TAG : m.tag;
    switch (tag) {
        case 1: {  // Mirror
            out.dir_out = -dir_in + R(2) * dot(dir_in, n) * n;
            out.pdf = R(1);
            return true;
        }
        case 2: {  // Plastic
            Vec3<R> refl = -dir_in + R(2) * dot(dir_in, n) * n;
            R F0 = tk_pow((m.p0 - R(1)) / (m.p0 + R(1)), R(2));
            R F = F0 + (R(1) - F0) * tk_pow(R(1) - dot(n, refl), R(5));
            R u = random_real<R>(rng);
            if (u <= F) {
                out.dir_out = refl;
                out.pdf = R(1);
            } else {
                out = cosine_sample(n, v, hemisphere_cos<R>(rng));
            }
            return true;
        }
        case 3: {  // Phong
            Vec3<R> local = power_lobe<R>(m.p0, rng);
            Vec3<R> refl = normalize(-dir_in + R(2) * dot(dir_in, n) * n);
            out.dir_out = normalize(to_world(refl, local));
            out.pdf = dot(v.gn, out.dir_out) < R(0)
                          ? R(0)
                          : tk_fmax(R(0), (m.p0 + R(1)) / Const<R>::TWOPI * tk_pow(dot(refl, out.dir_out), m.p0));
            return true;
        }
        case 4:    // BlinnPhong
        case 5: {  // BlinnPhongMicrofacet
            Vec3<R> h = normalize(to_world(n, power_lobe<R>(m.p0, rng)));
            out.dir_out = normalize(-dir_in + R(2) * dot(dir_in, h) * h);
            if (dot(v.gn, out.dir_out) <= R(0) || dot(h, n) <= R(0) || dot(out.dir_out, h) <= R(0)) {
                out.pdf = R(0);
            } else {
                R ndh = tag == 4 ? dot(n, h) : tk_clamp(dot(n, h), R(0), R(1));
                out.pdf = (m.p0 + R(1)) * R(0.25) * Const<R>::INVTWOPI * tk_pow(ndh, m.p0) / dot(out.dir_out, h);
            }
            return true;
        }
        default:  // Diffuse and the Disney family: cosine hemisphere
            out = cosine_sample(n, v, hemisphere_cos<R>(rng));
            return true;
    }
}

template <class R, int TAG = -1>
TK_HD R bsdf_pdf(const MaterialRec<R> &m, Vec3<R> dir_in, Vec3<R> dir_out, const Isect<R> &v) {
    const int tag = TAG >= 0 ? TAG : m.tag;
    if constexpr (TAG < 0 || tag_is_burley(TAG)) {
        if (tag_is_burley(tag)) return burley_pdf<R, TAG>(m, dir_in, dir_out, v);
    }
    if (tag == 1) return R(0);
    if (dot(v.gn, dir_out) < R(0)) return R(0);
    const Vec3<R> n = facing(dir_in, v);
    switch (tag) {
        case 2: {
            R F0 = tk_pow((m.p0 - R(1)) / (m.p0 + R(1)), R(2));
            R F = F0 + (R(1) - F0) * tk_pow(R(1) - dot(n, dir_out), R(5));
            return (R(1) - F) * tk_fmax(dot(n, dir_out), R(0)) / Const<R>::PI;
        }
        case 3: {
            Vec3<R> refl = normalize(-dir_in + R(2) * dot(dir_in, n) * n);
            return tk_fmax(R(0), (m.p0 + R(1)) / Const<R>::TWOPI * tk_pow(dot(refl, dir_out), m.p0));
        }
        case 4:
        case 5: {
            Vec3<R> h = normalize(dir_out + dir_in);
            if (dot(v.gn, dir_out) <= R(0) || dot(h, n) <= R(0) || dot(dir_out, h) <= R(0)) return R(0);
            R ndh = m.tag == 4 ? dot(n, h) : tk_clamp(dot(n, h), R(0), R(1));
            return (m.p0 + R(1)) * R(0.25) * Const<R>::INVTWOPI * tk_pow(ndh, m.p0) / dot(dir_out, h);
        }
        default:
            return tk_fmax(dot(n, dir_out), R(0)) / Const<R>::PI;
    }
}

// BSDF * cosine ("FG").  rec_pdf is SampleRecord::pdf: Plastic's specular branch is recognised by pdf == 1
// (src/materials/plastic.inl:44).
template <class R, int TAG = -1>
TK_HD Vec3<R> eval_bsdf(const DeviceScene<R> &sc, const MaterialRec<R> &m, Vec3<R> dir_in, Vec3<R> dir_out, R rec_pdf,
                        const Isect<R> &v) {
    const Vec3<R> zero{R(0), R(0), R(0)};
    if constexpr (TAG < 0 || tag_is_burley(TAG)) {
        if (tag_is_burley(TAG >= 0 ? TAG : m.tag)) return burley_eval<R, TAG>(sc, m, dir_in, dir_out, v);
    }
    if (dot(v.gn, dir_in) < R(0) || dot(v.gn, dir_out) < R(0)) return zero;
    const Vec3<R> n = facing(dir_in, v);
    const int tag = TAG >= 0 ? TAG : m.tag;
    switch (tag) {
        case 1: {
            Vec3<R> F0 = eval_texture(sc, m, v.uv);
            return F0 + one_minus(F0) * tk_pow5(R(1) - dot(n, dir_out));
        }
        case 2: {
            if (rec_pdf == R(1)) return {R(1), R(1), R(1)};
            Vec3<R> Kd = eval_texture(sc, m, v.uv);
            return Kd * tk_fmax(dot(n, dir_out), R(0)) / Const<R>::PI;
        }
        case 3: {
            Vec3<R> refl = normalize(-dir_in + R(2) * dot(dir_in, n) * n);
            Vec3<R> Ks = eval_texture(sc, m, v.uv);
            if (dot(n, dir_out) <= R(0)) return zero;
            return Ks * (m.p0 + R(1)) / Const<R>::TWOPI * tk_pow(tk_fmax(dot(dir_out, refl), R(0)), m.p0);
        }
        case 4: {
            if (dot(n, dir_out) <= R(0)) return zero;
            Vec3<R> h = normalize(dir_out + dir_in);
            Vec3<R> Ks = eval_texture(sc, m, v.uv);
            Vec3<R> Fh = Ks + one_minus(Ks) * tk_pow5(R(1) - dot(h, dir_out));
            R norm = (m.p0 + R(2)) * R(0.25) * Const<R>::INVPI / (R(2) - tk_pow(R(2), -m.p0 / R(2)));
            return norm * Fh * tk_pow(tk_fmax(R(0), dot(n, h)), m.p0);
        }
        case 5: {
            Vec3<R> h = normalize(dir_out + dir_in);
            if (dot(n, dir_out) <= R(0) || dot(dir_out, h) <= R(0) || dot(dir_in, h) <= R(0)) return zero;
            Vec3<R> Ks = eval_texture(sc, m, v.uv);
            Vec3<R> Fh = Ks + one_minus(Ks) * tk_pow5(R(1) - dot(h, dir_out));
            R Dh = (m.p0 + R(2)) * Const<R>::INVTWOPI * tk_pow(tk_clamp(dot(n, h), R(0), R(1)), m.p0);
            R G = g_hat(dir_out, n, m.p0) * g_hat(dir_in, n, m.p0);
            return Fh * Dh * G * R(0.25) / dot(n, dir_in);
        }
        case 6:  // DisneyDiffuse, src/materials/disney_diffuse.inl:22-46 (body in tk_burley.h: shared with tag 16)
            return disney_diffuse_value(eval_texture(sc, m, v.uv), m.p0, m.p1, n, dir_in, dir_out);
        case 9:  // DisneyClearcoat: the reference returns an uninitialised value; defined as zero
            return zero;
        default: {
            Vec3<R> Kd = eval_texture(sc, m, v.uv);
            return Kd * tk_fmax(dot(n, dir_out), R(0)) / Const<R>::PI;
        }
    }
}

// ---- lights
template <class R> struct LightSample {
    Vec3<R> pos, n;
};
template <class R, class G> TK_HD LightSample<R> sample_light_point(const LightRec<R> &l, Vec3<R> ref, G &rng) {
    LightSample<R> s;
    R u1 = random_real<R>(rng);
    R u2 = random_real<R>(rng);
    if (l.is_sphere) {  // cone sampling, src/shape.cpp:125-144
        Vec3<R> c = ld3(l.v);
        R r = l.v[3];
        R d = length(c - ref);
        R z = R(1) + u1 * (r / d - R(1));
        R z2 = z * z;
        R st = tk_sqrt(tk_clamp(R(1) - z2, R(0), R(1)));
        Vec3<R> local = normalize(Vec3<R>{tk_cos(R(2) * Const<R>::PI * u2) * st, tk_sin(R(2) * Const<R>::PI * u2) * st, z});
        s.n = normalize(to_world(normalize(ref - c), local));
        s.pos = c + r * s.n;
        return s;
    }
    Vec3<R> v0 = ld3(l.v), v1 = ld3(l.v + 3), v2 = ld3(l.v + 6);  // src/shape.cpp:146-169
    R b1 = R(1) - tk_sqrt(u1);
    R b2 = tk_sqrt(u1) * u2;
    s.pos = (R(1) - b1 - b2) * v0 + b1 * v1 + b2 * v2;
    Vec3<R> gn = normalize(cross(v1 - v0, v2 - v0));
    Vec3<R> sn = (R(1) - b1 - b2) * ld3(l.n) + b1 * ld3(l.n + 3) + b2 * ld3(l.n + 6);
    s.n = dot(sn, gn) > R(0) ? gn : -gn;
    return s;
}
// area-measure pdf of a point on light `l` seen from ref (src/light.cpp:32-48; the sphere branch measures d
// to the light POINT, as the reference does)
template <class R> TK_HD R light_pdf_area(const LightRec<R> &l, Vec3<R> light_pos, Vec3<R> ref) {
    if (l.kind != 1) return R(0);
    if (l.is_sphere) {
        R r = l.v[3];
        R d = length(light_pos - ref);
        return R(1) / (Const<R>::TWOPI * r * r * (R(1) - r / d));
    }
    Vec3<R> v0 = ld3(l.v), v1 = ld3(l.v + 3), v2 = ld3(l.v + 6);
    return R(1) / (length(cross(v1 - v0, v2 - v0)) / R(2));
}

}  // namespace tk
