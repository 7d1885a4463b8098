// tk_burley.h — the Burley ("Disney principled") lobes, material tags 12..16 (include/take_hip.h).
//
// EXTENSION, parity unpinned: the reference declares and parses these materials (src/material.h:44-80,
// src/parse/parse_scene.cpp:578-700) but evaluates Lambert clones (src/materials/disney_{metal,glass,clearcoat,sheen,
// bsdf}.inl); tags 7..11 keep doing exactly that.  Tags 12..16 implement the published model — Burley 2012/2015 in
// the five-lobe form of UCSD CSE 272 homework 1, GGX visible normals after Heitz 2018 — as specified in DESIGN.md §4d;
// the CPU statement they are tested against is oracle/take_burley.hpp.
//
// Everything works in the tangent frame of the shading normal flipped to dir_in's side (the basis to_world builds):
//   wi = dir_in (wi.z >= 0), wo = dir_out; "upper" = dir_out on dir_in's side of the geometric surface.
// eval returns BSDF * |cos(dir_out)|, as every reference material does.  Isect::back (the geometric normal had to be
// flipped to face the ray) tells which side of a dielectric interface dir_in is on.
#pragma once
// (included by tk_shade.h after Isect, BsdfSample, hemisphere_cos and eval_texture are declared)

namespace tk {

TK_HD constexpr bool tag_is_burley(int tag) { return tag >= TAKE_MAT_BURLEY_METAL && tag <= TAKE_MAT_BURLEY_BSDF; }

// parameters of one material, by tag (TakeMaterial::param order)
template <class R> struct BurleyMat {
    R transmission, metallic, subsurface, specular, roughness, specular_tint, anisotropic, sheen, sheen_tint, clearcoat,
        gloss, ior;
};
template <class R> TK_HD BurleyMat<R> burley_unpack(const MaterialRec<R> &m, int tag) {
    BurleyMat<R> b{R(0), R(0), R(0), R(0), R(0), R(0), R(0), R(0), R(0), R(0), R(0), R(1.5)};
    const R *p = m.p;
    if (tag == TAKE_MAT_BURLEY_METAL) {
        b.roughness = p[0], b.anisotropic = p[1];
    } else if (tag == TAKE_MAT_BURLEY_GLASS) {
        b.roughness = p[0], b.anisotropic = p[1], b.ior = p[2];
    } else if (tag == TAKE_MAT_BURLEY_CLEARCOAT) {
        b.gloss = p[0];
    } else if (tag == TAKE_MAT_BURLEY_SHEEN) {
        b.sheen_tint = p[0];
    } else {
        b.transmission = p[0], b.metallic = p[1], b.subsurface = p[2], b.specular = p[3], b.roughness = p[4];
        b.specular_tint = p[5], b.anisotropic = p[6], b.sheen = p[7], b.sheen_tint = p[8], b.clearcoat = p[9];
        b.gloss = p[10], b.ior = p[11];
    }
    return b;
}

// what a vertex and dir_in fix for every lobe
template <class R> struct BurleyFrame {
    Vec3<R> n, wi;
    R ax, ay;   // GGX widths along the two tangents
    R eta;      // index behind the surface / index on dir_in's side
    TK_HD Vec3<R> local(Vec3<R> w) const {
        // the rows of the basis to_world(n, .) builds
        Vec3<R> s, t;
        if (n.z < R(-1 + 1e-6)) {
            s = {R(0), R(-1), R(0)};
            t = {R(-1), R(0), R(0)};
        } else {
            R a = R(1) / (R(1) + n.z);
            R b = -n.x * n.y * a;
            s = {R(1) - n.x * n.x * a, b, -n.x};
            t = {b, R(1) - n.y * n.y * a, -n.y};
        }
        return {dot(w, s), dot(w, t), dot(w, n)};
    }
};
template <class R> TK_HD BurleyFrame<R> burley_frame(const BurleyMat<R> &b, Vec3<R> dir_in, const Isect<R> &v) {
    BurleyFrame<R> f;
    f.n = dot(dir_in, v.sn) < R(0) ? -v.sn : v.sn;
    f.wi = f.local(dir_in);
    const R aspect = tk_sqrt(R(1) - R(0.9) * b.anisotropic);
    const R r2 = b.roughness * b.roughness;
    f.ax = tk_fmax(R(1e-4), r2 / aspect);
    f.ay = tk_fmax(R(1e-4), r2 * aspect);
    f.eta = v.back ? R(1) / b.ior : b.ior;
    return f;
}

template <class R> TK_HD R schlick5(R c) {  // (1 - c)^5 with c clamped to [0, 1]
    const R m = tk_clamp(R(1) - c, R(0), R(1));
    return (m * m) * (m * m) * m;
}
template <class R> TK_HD Vec3<R> tint(Vec3<R> base) {
    const R lum = R(0.212671) * base.x + R(0.715160) * base.y + R(0.072169) * base.z;
    return lum > R(0) ? base / lum : Vec3<R>{R(1), R(1), R(1)};
}
template <class R> TK_HD Vec3<R> mix_white(R t, Vec3<R> c) {  // (1 - t) * white + t * c
    return Vec3<R>{R(1) - t, R(1) - t, R(1) - t} + t * c;
}
template <class R> TK_HD R dielectric_fresnel(R cos_i, R eta) {
    const R ct2 = R(1) - (R(1) - cos_i * cos_i) / (eta * eta);
    if (ct2 < R(0)) return R(1);
    const R ci = tk_fabs(cos_i), ct = tk_sqrt(ct2);
    const R rs = (ci - eta * ct) / (ci + eta * ct);
    const R rp = (eta * ci - ct) / (eta * ci + ct);
    return (rs * rs + rp * rp) / R(2);
}
template <class R> TK_HD R ggx_ndf(Vec3<R> h, R ax, R ay) {
    const R t = h.x * h.x / (ax * ax) + h.y * h.y / (ay * ay) + h.z * h.z;
    return R(1) / (Const<R>::PI * ax * ay * t * t);
}
template <class R> TK_HD R smith_masking(Vec3<R> w, R ax, R ay) {
    const R a = (w.x * ax) * (w.x * ax) + (w.y * ay) * (w.y * ay);
    const R lambda = (tk_sqrt(R(1) + a / (w.z * w.z)) - R(1)) / R(2);
    return R(1) / (R(1) + lambda);
}
template <class R> TK_HD Vec3<R> visible_normal(Vec3<R> wi, R ax, R ay, R u0, R u1) {  // Heitz 2018
    const Vec3<R> vh = normalize(Vec3<R>{ax * wi.x, ay * wi.y, wi.z});
    const R l2 = vh.x * vh.x + vh.y * vh.y;
    const Vec3<R> t1 = l2 > R(0) ? Vec3<R>{-vh.y, vh.x, R(0)} / tk_sqrt(l2) : Vec3<R>{R(1), R(0), R(0)};
    const Vec3<R> t2 = cross(vh, t1);
    const R r = tk_sqrt(u0), phi = Const<R>::TWOPI * u1;
    const R p1 = r * tk_cos(phi);
    R p2 = r * tk_sin(phi);
    const R s = (R(1) + vh.z) / R(2);
    p2 = (R(1) - s) * tk_sqrt(tk_fmax(R(0), R(1) - p1 * p1)) + s * p2;
    const Vec3<R> nh = t1 * p1 + t2 * p2 + vh * tk_sqrt(tk_fmax(R(0), R(1) - p1 * p1 - p2 * p2));
    return normalize(Vec3<R>{ax * nh.x, ay * nh.y, tk_fmax(R(0), nh.z)});
}

// ---- reflection lobes (wi.z > 0 and wo.z > 0 are the caller's business)
// specular reflection off GGX: value / Fresnel colour = D G / (4 cos_i); pdf of a visible-normal sample
template <class R> struct GgxReflection {
    R dg_over_4ci;  // D * G1(wi) * G1(wo) / (4 wi.z)
    R pdf;          // D * G1(wi) / (4 wi.z)
    R h_dot_wo;
};
template <class R> TK_HD GgxReflection<R> ggx_reflection(Vec3<R> wi, Vec3<R> wo, R ax, R ay) {
    const Vec3<R> h = normalize(wi + wo);
    GgxReflection<R> g;
    const R d = ggx_ndf(h, ax, ay), g1i = smith_masking(wi, ax, ay);
    g.dg_over_4ci = d * g1i * smith_masking(wo, ax, ay) / (R(4) * wi.z);
    g.pdf = d * g1i / (R(4) * wi.z);
    g.h_dot_wo = tk_fabs(dot(h, wo));
    return g;
}
template <class R> TK_HD Vec3<R> schlick_colour(Vec3<R> f0, R h_dot_wo) { return f0 + one_minus(f0) * schlick5(h_dot_wo); }

template <class R> struct CoatReflection {
    R value, pdf;
};
template <class R> TK_HD CoatReflection<R> coat_reflection(Vec3<R> wi, Vec3<R> wo, R gloss) {
    const Vec3<R> h = normalize(wi + wo);
    const R ag = (R(1) - gloss) * R(0.1) + gloss * R(0.001), a2 = ag * ag;
    const R d = (a2 - R(1)) / (Const<R>::PI * tk_log(a2) * (R(1) + (a2 - R(1)) * h.z * h.z));
    const R hw = tk_fabs(dot(h, wo));
    const R fr = R(0.04) + (R(1) - R(0.04)) * schlick5(hw);
    const R g = smith_masking(wi, R(0.25), R(0.25)) * smith_masking(wo, R(0.25), R(0.25));
    return {fr * d * g / (R(4) * wi.z), d * h.z / (R(4) * hw)};
}
template <class R> TK_HD Vec3<R> sheen_value(Vec3<R> base, R sheen_tint, Vec3<R> wi, Vec3<R> wo) {
    const Vec3<R> h = normalize(wi + wo);
    return mix_white(sheen_tint, tint(base)) * (schlick5(tk_fabs(dot(h, wo))) * wo.z);
}
// src/materials/disney_diffuse.inl:22-46 (tag 6 evaluates the same expression)
template <class R> TK_HD Vec3<R> disney_diffuse_value(Vec3<R> Kd, R roughness, R subsurface, Vec3<R> n, Vec3<R> dir_in, Vec3<R> dir_out) {
    Vec3<R> h = normalize(dir_in + dir_out);
    R hdout = dot(h, dir_out), ndout = dot(n, dir_out), ndin = dot(n, dir_in);
    R fd90 = R(0.5) + R(2) * roughness * hdout * hdout;
    R fi = R(1) + (fd90 - R(1)) * tk_pow(R(1) - dot(n, dir_in), R(5));
    R fo = R(1) + (fd90 - R(1)) * tk_pow(R(1) - dot(n, dir_out), R(5));
    Vec3<R> base = Kd * Const<R>::INVPI * fi * fo * ndout;
    R fss90 = roughness * hdout * hdout;
    R si = R(1) + (fss90 - R(1)) * tk_pow(R(1) - dot(n, dir_in), R(5));
    R so = R(1) + (fss90 - R(1)) * tk_pow(R(1) - dot(n, dir_out), R(5));
    Vec3<R> ss = R(1.25) * Kd * Const<R>::INVPI * (si * so * (R(1) / (tk_fabs(ndin) + tk_fabs(ndout)) - R(0.5)) + R(0.5)) * ndout;
    return (R(1) - subsurface) * base + subsurface * ss;
}

// ---- rough dielectric: one evaluation gives the value per unit base colour factor and the pdf
template <class R> struct GlassTerm {
    R value;  // reflection: F D G / (4 cos_i); refraction: (1 - F) D G |h.wo h.wi| / (cos_i (h.wi + eta h.wo)^2)
    R pdf;
    bool valid;
};
template <class R> TK_HD GlassTerm<R> glass_term(const BurleyFrame<R> &f, Vec3<R> wo, bool upper) {
    GlassTerm<R> g{R(0), R(0), false};
    const Vec3<R> wi = f.wi;
    if (wi.z <= R(0) || (upper ? wo.z <= R(0) : wo.z >= R(0))) return g;
    const Vec3<R> s = upper ? wi + wo : wi + wo * f.eta;
    const R l2 = dot(s, s);
    if (!(l2 > R(0))) return g;
    Vec3<R> h = s / tk_sqrt(l2);
    if (h.z < R(0)) h = -h;
    const R hi = dot(h, wi), ho = dot(h, wo);
    if (!(hi > R(0)) || (!upper && !(ho < R(0)))) return g;
    const R fr = dielectric_fresnel(hi, f.eta);
    const R d = ggx_ndf(h, f.ax, f.ay), g1i = smith_masking(wi, f.ax, f.ay);
    const R g1o = smith_masking(Vec3<R>{wo.x, wo.y, tk_fabs(wo.z)}, f.ax, f.ay);
    g.valid = true;
    if (upper) {
        g.value = fr * (d * g1i * g1o) / (R(4) * wi.z);
        g.pdf = fr * (d * g1i) / (R(4) * wi.z);
    } else {
        const R denom = hi + f.eta * ho;
        g.value = (R(1) - fr) * (d * g1i * g1o) * tk_fabs(ho * hi) / (wi.z * denom * denom);
        const R dh_dout = f.eta * f.eta * ho / (denom * denom);
        g.pdf = (R(1) - fr) * (d * g1i) * tk_fabs(dh_dout * hi / wi.z);
    }
    return g;
}
template <class R> TK_HD Vec3<R> glass_colour(Vec3<R> base, bool upper) {
    // max(0, .): a textured base colour can be slightly negative (the reference's filter extrapolates at the wrap seam)
    return upper ? base : Vec3<R>{tk_sqrt(tk_fmax(R(0), base.x)), tk_sqrt(tk_fmax(R(0), base.y)), tk_sqrt(tk_fmax(R(0), base.z))};
}

// sampling weights of the principled material: diffuse | metal | glass | clearcoat
template <class R> struct BurleyMix {
    R diffuse, metal, glass, coat;
};
template <class R> TK_HD BurleyMix<R> burley_mix(const BurleyMat<R> &b, bool back) {
    if (back) return {R(0), R(0), R(1), R(0)};
    const R d = (R(1) - b.metallic) * (R(1) - b.transmission);
    const R m = R(1) - b.transmission * (R(1) - b.metallic);
    const R g = (R(1) - b.metallic) * b.transmission;
    const R c = R(0.25) * b.clearcoat;
    const R sum = d + m + g + c;
    return {d / sum, m / sum, g / sum, c / sum};
}

// ---- the three entry points tk_shade.h dispatches tags 12..16 to
template <class R, int TAG>
TK_HD Vec3<R> burley_eval(const DeviceScene<R> &sc, const MaterialRec<R> &m, Vec3<R> dir_in, Vec3<R> dir_out, const Isect<R> &v) {
    const Vec3<R> zero{R(0), R(0), R(0)};
    if (dot(v.gn, dir_in) < R(0)) return zero;
    const int tag = TAG >= 0 ? TAG : m.tag;
    const bool upper = !(dot(v.gn, dir_out) < R(0));
    const BurleyMat<R> b = burley_unpack(m, tag);
    const BurleyFrame<R> f = burley_frame(b, dir_in, v);
    const Vec3<R> wo = f.local(dir_out);
    const bool mirror_side = upper && f.wi.z > R(0) && wo.z > R(0);  // where the reflection lobes live
    if (tag == TAKE_MAT_BURLEY_CLEARCOAT) {
        if (!mirror_side) return zero;
        const R c = coat_reflection(f.wi, wo, b.gloss).value;
        return {c, c, c};
    }
    const Vec3<R> base = eval_texture(sc, m, v.uv);
    if (tag == TAKE_MAT_BURLEY_METAL) {
        if (!mirror_side) return zero;
        const GgxReflection<R> g = ggx_reflection(f.wi, wo, f.ax, f.ay);
        return schlick_colour(base, g.h_dot_wo) * g.dg_over_4ci;
    }
    if (tag == TAKE_MAT_BURLEY_SHEEN) return mirror_side ? sheen_value(base, b.sheen_tint, f.wi, wo) : zero;
    const Vec3<R> glass = glass_colour(base, upper) * glass_term(f, wo, upper).value;
    if (tag == TAKE_MAT_BURLEY_GLASS) return glass;
    // principled
    Vec3<R> sum = ((R(1) - b.metallic) * b.transmission) * glass;
    if (v.back || !mirror_side) return sum;
    sum = sum + (R(1) - b.transmission) * (R(1) - b.metallic) *
                    disney_diffuse_value(base, b.roughness, b.subsurface, f.n, dir_in, dir_out);
    sum = sum + (R(1) - b.metallic) * b.sheen * sheen_value(base, b.sheen_tint, f.wi, wo);
    const R r0 = (b.ior - R(1)) / (b.ior + R(1));
    const Vec3<R> c0 = (b.specular * r0 * r0 * (R(1) - b.metallic)) * mix_white(b.specular_tint, tint(base)) + b.metallic * base;
    const GgxReflection<R> g = ggx_reflection(f.wi, wo, f.ax, f.ay);
    sum = sum + (R(1) - b.transmission * (R(1) - b.metallic)) * (schlick_colour(c0, g.h_dot_wo) * g.dg_over_4ci);
    const R c = coat_reflection(f.wi, wo, b.gloss).value;
    return sum + R(0.25) * b.clearcoat * Vec3<R>{c, c, c};
}

template <class R, int TAG> TK_HD R burley_pdf(const MaterialRec<R> &m, Vec3<R> dir_in, Vec3<R> dir_out, const Isect<R> &v) {
    if (dot(v.gn, dir_in) < R(0)) return R(0);
    const int tag = TAG >= 0 ? TAG : m.tag;
    const bool upper = !(dot(v.gn, dir_out) < R(0));
    const BurleyMat<R> b = burley_unpack(m, tag);
    const BurleyFrame<R> f = burley_frame(b, dir_in, v);
    const Vec3<R> wo = f.local(dir_out);
    const bool mirror_side = upper && f.wi.z > R(0) && wo.z > R(0);
    if (tag == TAKE_MAT_BURLEY_METAL) return mirror_side ? ggx_reflection(f.wi, wo, f.ax, f.ay).pdf : R(0);
    if (tag == TAKE_MAT_BURLEY_CLEARCOAT) return mirror_side ? coat_reflection(f.wi, wo, b.gloss).pdf : R(0);
    if (tag == TAKE_MAT_BURLEY_SHEEN) return mirror_side ? wo.z / Const<R>::PI : R(0);
    const R glass = glass_term(f, wo, upper).pdf;
    if (tag == TAKE_MAT_BURLEY_GLASS) return glass;
    const BurleyMix<R> w = burley_mix(b, v.back);
    R pdf = w.glass * glass;
    if (upper) {
        const R cosine = mirror_side ? wo.z / Const<R>::PI : R(0);
        const R metal = mirror_side ? ggx_reflection(f.wi, wo, f.ax, f.ay).pdf : R(0);
        const R coat = mirror_side ? coat_reflection(f.wi, wo, b.gloss).pdf : R(0);
        pdf += w.diffuse * cosine + w.metal * metal + w.coat * coat;
    }
    return pdf;
}

// Draw order (the specification): principled — one number picks the lobe; then cosine lobe: u1 u2 of
// hemisphere_cos; metal, clearcoat: u0 u1; glass: u0 u1 and a third for reflect-or-refract.
// A microfacet reflection that leaves through the macro-surface, or a refraction that stays above it, gets pdf 0
// (the path ends), so that pdf() never prices a sample as the event it was not.
template <class R, int TAG, class G>
TK_HD bool burley_sample(const MaterialRec<R> &m, Vec3<R> dir_in, const Isect<R> &v, G &rng, BsdfSample<R> &out) {
    if (dot(v.gn, dir_in) < R(0)) return false;
    const int tag = TAG >= 0 ? TAG : m.tag;
    const BurleyMat<R> b = burley_unpack(m, tag);
    const BurleyFrame<R> f = burley_frame(b, dir_in, v);
    enum { COSINE, METAL, GLASS, COAT } lobe;
    if (tag == TAKE_MAT_BURLEY_METAL) {
        lobe = METAL;
    } else if (tag == TAKE_MAT_BURLEY_GLASS) {
        lobe = GLASS;
    } else if (tag == TAKE_MAT_BURLEY_CLEARCOAT) {
        lobe = COAT;
    } else if (tag == TAKE_MAT_BURLEY_SHEEN) {
        lobe = COSINE;
    } else {
        const BurleyMix<R> w = burley_mix(b, v.back);
        const R u = random_real<R>(rng);
        lobe = u < w.diffuse ? COSINE : (u < w.diffuse + w.metal ? METAL : (u < w.diffuse + w.metal + w.glass ? GLASS : COAT));
        if (v.back) lobe = GLASS;
    }
    Vec3<R> wo;
    bool want_upper = true;
    if (lobe == COSINE) {
        wo = hemisphere_cos<R>(rng);
    } else {
        const R u0 = random_real<R>(rng);
        const R u1 = random_real<R>(rng);
        Vec3<R> h;
        if (lobe == COAT) {
            const R ag = (R(1) - b.gloss) * R(0.1) + b.gloss * R(0.001), a2 = ag * ag;
            const R ch = tk_sqrt(tk_clamp((R(1) - tk_pow(a2, R(1) - u0)) / (R(1) - a2), R(0), R(1)));
            const R sh = tk_sqrt(tk_fmax(R(0), R(1) - ch * ch)), phi = Const<R>::TWOPI * u1;
            h = {sh * tk_cos(phi), sh * tk_sin(phi), ch};
        } else {
            h = visible_normal(f.wi, f.ax, f.ay, u0, u1);
        }
        const R hi = dot(f.wi, h);
        if (lobe == GLASS) {
            const R fr = dielectric_fresnel(hi, f.eta);
            want_upper = random_real<R>(rng) <= fr;
            if (!want_upper) {
                const R ho = tk_sqrt(tk_fmax(R(0), R(1) - (R(1) - hi * hi) / (f.eta * f.eta)));
                wo = -f.wi / f.eta + (tk_fabs(hi) / f.eta - ho) * h;
            }
        }
        if (want_upper) wo = -f.wi + R(2) * hi * h;
    }
    out.dir_out = to_world(f.n, wo);
    const bool below = dot(v.gn, out.dir_out) < R(0);
    const bool as_sampled = want_upper ? (!below && wo.z > R(0)) : (below && wo.z < R(0));
    out.pdf = as_sampled ? burley_pdf<R, TAG>(m, dir_in, out.dir_out, v) : R(0);
    return true;
}

}  // namespace tk
