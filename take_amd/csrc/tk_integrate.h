// tk_integrate.h — the integrator as per-path step functions of the wavefront pipeline.
//
// The reference's path_tracing() (src/integrator/path_tracing.h:5-111) is one loop per camera ray:
//     primary hit; for i in 0..max_depth { NEE (shadow ray) ; BSDF sample ; extend ray ; MIS bookkeeping }
// Here every path of a batch advances one loop iteration per launch round:
//     generate -> [ trace_closest -> shade(k) -> trace_shadow ]  for k = 0 .. max_depth+1 -> accumulate
// shade(k) finishes iteration k-1 with the hit the extend ray found (emitter MIS term C2, throughput update)
// and starts iteration k (NEE sample -> shadow-ray request, BSDF sample -> extend-ray request).  The shadow
// kernel adds throughput*C1 to the path's radiance when unoccluded; stream order puts that add before the next
// shade's C2 add, which is the reference's order of additions.  All per-sample arithmetic, the order of random
// draws and every `break` of the reference loop are kept; what changes is who executes them and when.
#pragma once

#include "tk_shade.h"

namespace tk {

template <class R> struct RenderParams {
    int32_t width, height;
    int32_t n_local_rows;   // rows this rank renders
    int32_t npix;           // n_local_rows * width
    int32_t strip_first, strip_stride;
    int32_t spp;            // total samples per pixel
    int32_t s0;             // first sample index of this batch
    int32_t spb;            // samples per pixel in this batch
    int32_t max_depth;
    int32_t integrator;     // TakeRenderOpts.integrator: 0 path_tracing, 1 raw, 2 one-sample MIS, 3 one-sample MIS by power
    uint64_t seed;
    R ray_eps;
    double inv_npix, inv_width;  // 1.0 / npix, 1.0 / width for slot_pixel (0: not set — plain division)
};

// n / d and n % d for n, d < 2^31 with the reciprocal of the (launch-invariant) divisor: the product is within 2^-20
// of the true quotient, so its floor is off by at most one and one correction makes it exact.  Every path of every
// shade round starts with two such divisions (slot -> sample, pixel -> row, column); as integer divisions they were
// ~130 of the shade kernel's instructions per path.
TK_HD void divmod_u31(uint32_t n, uint32_t d, double inv_d, uint32_t &q, uint32_t &r) {
    if (inv_d == 0.0) {
        q = n / d, r = n % d;
        return;
    }
    q = (uint32_t)((double)n * inv_d);
    r = n - q * d;
    if ((int32_t)r < 0) q -= 1, r += d;
    else if (r >= d) q += 1, r -= d;
}
// path slot -> (sample of the batch, local row, column)
template <class R> TK_HD void slot_pixel(const RenderParams<R> &rp, int64_t slot, int &sl, int &lr, int &x) {
    uint32_t q, p, row, col;
    divmod_u31((uint32_t)slot, (uint32_t)rp.npix, rp.inv_npix, q, p);
    divmod_u31(p, (uint32_t)rp.width, rp.inv_width, row, col);
    sl = (int)q, lr = (int)row, x = (int)col;
}

// Multi-GPU unit: a strip of STRIP_ROWS image rows, dealt round-robin to the ranks.  (The reference's own unit is
// a 16-row tile row, src/render.cpp:52; 4-row strips balance better: 1080 rows over 8 ranks = 136 vs 132 rows per
// rank instead of 144 vs 128.  Pixel keys do not depend on the sharding, so the image does not either.)
constexpr int TILE_ROWS = 4;

// local row -> render-loop y (y = 0 is the bottom image row: the reference stores img(x, height - y - 1))
template <class R> TK_HD int local_row_to_y(const RenderParams<R> &rp, int lr) {
    return (rp.strip_first + (lr / TILE_ROWS) * rp.strip_stride) * TILE_ROWS + (lr % TILE_ROWS);
}

template <class R> TK_HD Rng path_rng(const RenderParams<R> &rp, int64_t slot, uint32_t ctr) {
    int sl, lr, x;
    slot_pixel(rp, slot, sl, lr, x);
    const int y = local_row_to_y(rp, lr);
    Rng r;
    r.key = rng_key(rp.seed, (uint64_t)y * (uint64_t)rp.width + (uint64_t)x, (uint64_t)(rp.s0 + sl));
    r.ctr = ctr;
    return r;
}

// Camera ray of src/render.cpp:69-75.  The two jitters are drawn y first (the g++ evaluation order the golden
// vectors pin).
constexpr uint32_t CAMERA_DRAWS = 2;  // random numbers a camera ray consumes: a path's stream continues at this counter
template <class R> TK_HD Vec3<R> camera_dir(const CameraRec<R> &c, const RenderParams<R> &rp, int64_t slot) {
    int sl, lr, x;
    slot_pixel(rp, slot, sl, lr, x);
    const int y = local_row_to_y(rp, lr);
    Rng rng = path_rng(rp, slot, 0);
    const R ry = random_real<R>(rng);
    const R rx = random_real<R>(rng);
    return normalize(ld3(c.u) * ((R(x) + rx) / R(c.width) - R(0.5)) * c.viewport_width +
                     ld3(c.v) * ((R(y) + ry) / R(c.height) - R(0.5)) * c.viewport_height - ld3(c.w));
}
// The whole initial record of a path.  (The render loop of the default integrator does not run this: its first
// closest-hit launch makes the camera rays itself — CameraIo, tk_kernels.h — and round 0 of shade_path starts from
// throughput 1, radiance 0 and counter CAMERA_DRAWS whatever the record holds; this function states what that equals.)
template <class R>
TK_HD void generate_path(const DeviceScene<R> &sc, const RenderParams<R> &rp, const PathState<R> &st, int64_t slot) {
    const CameraRec<R> &c = sc.cam;
    const Vec3<R> d = camera_dir(c, rp, slot);
    Rng rng = path_rng(rp, slot, 0);
    rng.ctr = CAMERA_DRAWS;
    st.R_(S_OX, slot) = c.lookfrom[0];
    st.R_(S_OY, slot) = c.lookfrom[1];
    st.R_(S_OZ, slot) = c.lookfrom[2];
    st.R_(S_DX, slot) = d.x;
    st.R_(S_DY, slot) = d.y;
    st.R_(S_DZ, slot) = d.z;
    st.R_(S_TX, slot) = R(1);
    st.R_(S_TY, slot) = R(1);
    st.R_(S_TZ, slot) = R(1);
    st.R_(S_LX, slot) = R(0);
    st.R_(S_LY, slot) = R(0);
    st.R_(S_LZ, slot) = R(0);
    st.I_(S_CTR, slot) = (int32_t)rng.ctr;
    st.I_(S_FLAGS, slot) = 0;
    st.I_(S_OCC, slot) = -1;
    st.I_(S_CONV, slot) = 0;
}

constexpr uint32_t REQ_EXTEND = 1, REQ_SHADOW = 2;
constexpr int TAG_ANY = -1;    // material tag read from the material record
constexpr int TAG_MISS = TAKE_MAT_COUNT;   // the segment of the sorted queue holding the paths whose extend ray missed

// One launch round for one path.  `k` is the shade round (uniform over the launch): k = 0 handles the camera
// ray's hit, k >= 1 finishes loop iteration k-1; iteration k is started when k <= max_depth.
// TAG: compile-time material tag of the vertex being shaded (see tk_shade.h); a TAG instance may still meet a
// miss when the queue is not sorted (single-tag scenes).  Returns REQ_* bits: which rays to trace next.
template <class R, int TAG = TAG_ANY, class ST = PathState<R>>
TK_HD uint32_t shade_path(const DeviceScene<R> &sc, const RenderParams<R> &rp, const ST &st, int64_t slot,
                          int k) {
    const int32_t hit_prim = st.I_(S_HIT, slot);
    const Vec3<R> ro{st.R_(S_OX, slot), st.R_(S_OY, slot), st.R_(S_OZ, slot)};
    const Vec3<R> rd{st.R_(S_DX, slot), st.R_(S_DY, slot), st.R_(S_DZ, slot)};
    Vec3<R> thr{st.R_(S_TX, slot), st.R_(S_TY, slot), st.R_(S_TZ, slot)};
    Vec3<R> rad{st.R_(S_LX, slot), st.R_(S_LY, slot), st.R_(S_LZ, slot)};
    if (k == 0) {  // a path starts here: what generate_path writes, without needing it written (see there)
        thr = Vec3<R>{R(1), R(1), R(1)};
        rad = Vec3<R>{R(0), R(0), R(0)};
        if (sizeof(R) == 8) st.I_(S_CONV, slot) = 0;  // (f64 records only: in the f32 shade kernel the store cost two registers and with them the fifth wave)
    }
    const Vec3<R> bg = ld3(sc.background);
    const R nlights = R(sc.n_lights);
    const bool miss = (TAG == TAG_MISS) || hit_prim < 0;
    Isect<R> v{};
    bool alive = !miss;
    if (!miss) make_isect(sc, ro, rd, hit_prim, st.R_(S_HT, slot), st.R_(S_HU, slot), st.R_(S_HV, slot), v,
                            sc.inst_shade ? st.I_(S_INST, slot) : -1);

    if (k == 0) {
        // src/integrator/path_tracing.h:7-18
        if (miss) {
            rad = bg;
            if (sc.env.light >= 0) {  // extension: the camera ray sees the environment map
                R unused;
                rad = env_eval(sc, rd, unused);
            }
        } else if (v.area_light != -1) {
            const LightRec<R> &l = sc.lights[v.area_light];
            if (l.kind == 1) rad = rad + thr * ld3(l.intensity);
        }
    } else {
        // second half of loop iteration k-1: src/integrator/path_tracing.h:82-108
        const Vec3<R> FG{st.R_(S_FX, slot), st.R_(S_FY, slot), st.R_(S_FZ, slot)};
        const R pdf = st.R_(S_PDF, slot);
        const bool was_specular = (st.I_(S_FLAGS, slot) & FLAG_SPECULAR) != 0;
        if (miss && sc.env.light >= 0) {
            // extension: the BSDF-sampled ray left the scene and sees the environment map — the C2 term of an
            // emitter hit (path_tracing.h:97-102) with the map's density in the role of light_pdf
            R env_pdf;
            const Vec3<R> L = env_eval(sc, rd, env_pdf);
            const R light_pdf = env_pdf / nlights;
            rad = rad + thr * (FG * L * (was_specular ? (R(1) / pdf) : (pdf / (light_pdf * light_pdf + pdf * pdf))));
            thr = thr * (FG / pdf);
        } else if (miss) {
            thr = thr * (FG / pdf);
            rad = rad + thr * bg;
        } else {
            Vec3<R> C2{R(0), R(0), R(0)};
            if (v.area_light != -1) {
                const LightRec<R> &l = sc.lights[v.area_light];
                const R d = length(v.pos - ro);
                const Vec3<R> light_dir = normalize(v.pos - ro);
                const R light_pdf =
                    light_pdf_area(l, v.pos, ro) * (d * d) / (tk_fmax(dot(-v.gn, light_dir), R(0)) * nlights);
                if (light_pdf <= R(0)) {
                    alive = false;  // `break` at :93-96 — before the C2 add and the throughput update
                } else if (l.kind == 1) {
                    C2 = FG * ld3(l.intensity) *
                         (was_specular ? (R(1) / pdf) : (pdf / (light_pdf * light_pdf + pdf * pdf)));
                }
            }
            if (alive) {
                rad = rad + thr * C2;
                thr = thr * (FG / pdf);
            }
        }
    }

    uint32_t req = 0;
    if (TAG != TAG_MISS && alive && k <= rp.max_depth) {
        // first half of loop iteration k: src/integrator/path_tracing.h:22-81
        Rng rng = path_rng(rp, slot, k == 0 ? CAMERA_DRAWS : (uint32_t)st.I_(S_CTR, slot));
        const Vec3<R> dir_in = -rd;
        const MaterialRec<R> &m = sc.materials[v.material];
        constexpr int MT = (TAG >= 0 && TAG < TAG_MISS) ? TAG : -1;
        const int tag = MT >= 0 ? MT : m.tag;
        const bool is_specular = (tag == 2 || tag == 1);
        if (sc.n_lights > 0 && !is_specular) {
            const int light_id = (int)tk_floor(random_real<R>(rng) * nlights);
            const LightRec<R> &l = sc.lights[light_id];  // (whole-record load measured: +2.7 % shade time — the env branch needs 4 words of it)
            if (l.kind == 2) {
                // extension: next-event estimation towards the environment map (same C1 form as an area light,
                // path_tracing.h:33-58; the shadow ray has no far end)
                const EnvSample<R> es = env_sample(sc, rng);
                const R light_pdf = es.pdf / nlights;
                if (light_pdf <= R(0)) {
                    alive = false;
                } else {
                    const R bp = bsdf_pdf<R, MT>(m, dir_in, es.dir, v);
                    if (bp > R(0) && !tk_isinf(light_pdf)) {
                        const Vec3<R> FGl = eval_bsdf<R, MT>(sc, m, dir_in, es.dir, R(0), v);
                        const Vec3<R> add = thr * (FGl * es.radiance * light_pdf / (light_pdf * light_pdf + bp * bp));
                        st.R_(S_SX, slot) = es.dir.x;
                        st.R_(S_SY, slot) = es.dir.y;
                        st.R_(S_SZ, slot) = es.dir.z;
                        st.R_(S_ST, slot) = Const<R>::inf();
                        st.R_(S_CX, slot) = add.x;
                        st.R_(S_CY, slot) = add.y;
                        st.R_(S_CZ, slot) = add.z;
                        req |= REQ_SHADOW;
                    }
                }
            } else if (l.kind == 1) {
                const LightSample<R> lp = sample_light_point(l, v.pos, rng);
                const R d = length(lp.pos - v.pos);
                const Vec3<R> light_dir = normalize(lp.pos - v.pos);
                const R light_pdf =
                    light_pdf_area(l, lp.pos, v.pos) * (d * d) / (tk_fmax(dot(-lp.n, light_dir), R(0)) * nlights);
                if (light_pdf <= R(0)) {
                    alive = false;  // `break` at :40-43
                } else {
                    const R bp = bsdf_pdf<R, MT>(m, dir_in, light_dir, v);
                    if (bp > R(0) && !tk_isinf(light_pdf)) {
                        const Vec3<R> FGl = eval_bsdf<R, MT>(sc, m, dir_in, light_dir, R(0), v);
                        const Vec3<R> C1 = FGl * ld3(l.intensity) * light_pdf / (light_pdf * light_pdf + bp * bp);
                        const Vec3<R> add = thr * C1;
                        st.R_(S_SX, slot) = light_dir.x;
                        st.R_(S_SY, slot) = light_dir.y;
                        st.R_(S_SZ, slot) = light_dir.z;
                        st.R_(S_ST, slot) = (R(1) - rp.ray_eps) * d;
                        st.R_(S_CX, slot) = add.x;
                        st.R_(S_CY, slot) = add.y;
                        st.R_(S_CZ, slot) = add.z;
                        req |= REQ_SHADOW;
                    }
                }
            }
        }
        if (alive) {
            BsdfSample<R> rec;
            if (sample_bsdf<R, MT>(m, dir_in, v, rng, rec)) {
                const Vec3<R> FG = eval_bsdf<R, MT>(sc, m, dir_in, rec.dir_out, rec.pdf, v);
                const Vec3<R> dir_out = normalize(rec.dir_out);
                if (rec.pdf > R(0)) {
                    st.R_(S_DX, slot) = dir_out.x;
                    st.R_(S_DY, slot) = dir_out.y;
                    st.R_(S_DZ, slot) = dir_out.z;
                    st.R_(S_FX, slot) = FG.x;
                    st.R_(S_FY, slot) = FG.y;
                    st.R_(S_FZ, slot) = FG.z;
                    st.R_(S_PDF, slot) = rec.pdf;
                    st.I_(S_FLAGS, slot) = is_specular ? FLAG_SPECULAR : 0;
                    req |= REQ_EXTEND;
                }
            }
        }
        // both rays of this iteration start at the vertex
        st.R_(S_OX, slot) = v.pos.x;
        st.R_(S_OY, slot) = v.pos.y;
        st.R_(S_OZ, slot) = v.pos.z;
        st.I_(S_CTR, slot) = (int32_t)rng.ctr;
    }
    st.R_(S_TX, slot) = thr.x;
    st.R_(S_TY, slot) = thr.y;
    st.R_(S_TZ, slot) = thr.z;
    st.R_(S_LX, slot) = rad.x;
    st.R_(S_LY, slot) = rad.y;
    st.R_(S_LZ, slot) = rad.z;
    return req;
}

// ---- the reference's other integrators (src/integrator/path_tracing.h:114-380; defined there, called by nothing):
//   1  path_tracing_raw                    BSDF sampling only, emission on hit, no next-event estimation
//   2  path_tracing_one_sample_MIS         per vertex EITHER a light sample OR a BSDF sample (coin flip), the one ray
//                                          weighted by the mixture density; uniform light pick
//   3  path_tracing_one_sample_MIS_power   the same with the light picked by power (light.cpp:9-30; the tables the
//                                          parser leaves empty are filled from light_power(), tk_host_scene.h)
// All three trace ONE closest-hit ray per vertex and no shadow rays: a round is trace -> shade.  An iteration of the
// reference loop that does nothing (a PointLight was picked: `if (auto *l = get_if<DiffuseAreaLight>)` fails) is
// run again inside the same shade call, so the loop index is carried in the path's flag word.
constexpr int32_t FLAG_LIGHT_BRANCH = 2;  // the pending ray was aimed at a sampled light point
constexpr int FLAG_ITER_SHIFT = 8;        // loop index `i` of the reference in bits 8..

// src/light.cpp:9-17: std::upper_bound over the n + 1 entries of the power CDF, clamped to [0, n - 1]
template <class R, class G> TK_HD int sample_light_by_power(const DeviceScene<R> &sc, G &rng) {
    const R u = random_real<R>(rng);
    const int size = sc.n_lights;
    int lo = 0, hi = size + 1;  // first index in [0, size] whose entry is > u (size + 1 if none)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sc.light_cdf[mid] > u) hi = mid;
        else lo = mid + 1;
    }
    const int off = lo - 1;
    return off < 0 ? 0 : (off > size - 1 ? size - 1 : off);
}

template <class R, int TAG = TAG_ANY, class ST = PathState<R>>
TK_HD uint32_t shade_path_alt(const DeviceScene<R> &sc, const RenderParams<R> &rp, const ST &st, int64_t slot,
                              int k) {
    const bool raw = rp.integrator == 1, power = rp.integrator == 3;
    const int32_t hit_prim = st.I_(S_HIT, slot);
    const Vec3<R> ro{st.R_(S_OX, slot), st.R_(S_OY, slot), st.R_(S_OZ, slot)};
    const Vec3<R> rd{st.R_(S_DX, slot), st.R_(S_DY, slot), st.R_(S_DZ, slot)};
    Vec3<R> thr{st.R_(S_TX, slot), st.R_(S_TY, slot), st.R_(S_TZ, slot)};
    Vec3<R> rad{st.R_(S_LX, slot), st.R_(S_LY, slot), st.R_(S_LZ, slot)};
    const Vec3<R> bg = ld3(sc.background);
    const R nlights = R(sc.n_lights);
    const bool miss = (TAG == TAG_MISS) || hit_prim < 0;
    Isect<R> v{};
    if (!miss) make_isect(sc, ro, rd, hit_prim, st.R_(S_HT, slot), st.R_(S_HU, slot), st.R_(S_HV, slot), v,
                            sc.inst_shade ? st.I_(S_INST, slot) : -1);
    const int32_t flags = st.I_(S_FLAGS, slot);
    int iter = flags >> FLAG_ITER_SHIFT;
    bool alive = true;
    if (k == 0) {
        iter = 0;
        if (miss) rad = bg, alive = false;  // :117, :164, :277
    } else {
        // the ray of loop iteration `iter` has been traced: finish that iteration
        const Vec3<R> FG{st.R_(S_FX, slot), st.R_(S_FY, slot), st.R_(S_FZ, slot)};
        const R pdf = st.R_(S_PDF, slot);
        if (raw) {
            if (miss) rad = rad + thr * bg, alive = false;  // :148-152 (throughput was updated before the ray, :146)
        } else if (flags & FLAG_LIGHT_BRANCH) {
            if (miss) {
                // :327-331 (power variant); the uniform variant dereferences the empty optional (:222) — a miss
                // ends the path here, as in the oracle
                if (power) rad = rad + thr * bg;
                alive = false;
            } else if (power && v.area_light == -1) {
                alive = false;  // :333-335
            } else {
                thr = thr * (FG / pdf);  // pdf = 0.5 light_pdf + 0.5 bsdf_pdf, stored by the first half
            }
        } else {
            const bool was_specular = (flags & FLAG_SPECULAR) != 0;
            R p = (sc.n_lights == 0 || was_specular) ? pdf : R(0.5) * pdf;
            if (miss) {
                thr = thr * (FG / p);
                rad = rad + thr * bg;
                alive = false;
            } else {
                if (!was_specular && v.area_light != -1) {
                    const LightRec<R> &l = sc.lights[v.area_light];
                    const R d = length(v.pos - ro);
                    const Vec3<R> light_dir = normalize(v.pos - ro);
                    const R lpd = light_pdf_area(l, v.pos, ro) * (d * d);
                    const R light_pdf = power ? lpd * sc.light_pmf[v.area_light] / tk_fmax(dot(-v.gn, light_dir), R(0))
                                              : lpd / (tk_fmax(dot(-v.gn, light_dir), R(0)) * nlights);
                    if (light_pdf <= R(0)) alive = false;
                    else p = p + R(0.5) * light_pdf;
                }
                if (alive) thr = thr * (FG / p);
            }
        }
        iter++;
    }

    uint32_t req = 0;
    if (TAG != TAG_MISS && alive) {
        Rng rng = path_rng(rp, slot, (uint32_t)st.I_(S_CTR, slot));
        const Vec3<R> dir_in = -rd;
        const MaterialRec<R> &m = sc.materials[v.material];
        constexpr int MT = (TAG >= 0 && TAG < TAG_MISS) ? TAG : -1;
        const int tag = MT >= 0 ? MT : m.tag;
        const bool is_specular = (tag == 2 || tag == 1);
        int32_t new_flags = 0;
        for (; iter <= rp.max_depth; iter++) {
            if (v.area_light != -1) {
                const LightRec<R> &l = sc.lights[v.area_light];
                if (l.kind == 1) {
                    rad = rad + thr * ld3(l.intensity);
                    break;
                }
                if (raw) continue;  // `else` of :121-127: an emitter id that is not a DiffuseAreaLight does nothing
            }
            if (!raw && sc.n_lights > 0 && !is_specular && random_real<R>(rng) <= R(0.5)) {
                const int light_id = power ? sample_light_by_power(sc, rng) : (int)tk_floor(random_real<R>(rng) * nlights);
                const LightRec<R> &l = sc.lights[light_id];
                if (l.kind != 1) continue;  // a PointLight: the iteration does nothing
                const LightSample<R> lp = sample_light_point(l, v.pos, rng);
                const R d = length(lp.pos - v.pos);
                const Vec3<R> light_dir = normalize(lp.pos - v.pos);
                const R lpd = light_pdf_area(l, lp.pos, v.pos) * (d * d);
                const R light_pdf = power ? lpd * sc.light_pmf[light_id] / tk_fmax(dot(-lp.n, light_dir), R(0))
                                          : lpd / (tk_fmax(dot(-lp.n, light_dir), R(0)) * nlights);
                if (light_pdf <= R(0)) break;
                const R bp = bsdf_pdf<R, MT>(m, dir_in, light_dir, v);
                if (bp <= R(0)) break;
                const Vec3<R> FG = eval_bsdf<R, MT>(sc, m, dir_in, light_dir, R(0), v);
                st.R_(S_DX, slot) = light_dir.x, st.R_(S_DY, slot) = light_dir.y, st.R_(S_DZ, slot) = light_dir.z;
                st.R_(S_FX, slot) = FG.x, st.R_(S_FY, slot) = FG.y, st.R_(S_FZ, slot) = FG.z;
                st.R_(S_PDF, slot) = R(0.5) * light_pdf + R(0.5) * bp;
                new_flags = FLAG_LIGHT_BRANCH;
                req = REQ_EXTEND;
                break;
            }
            BsdfSample<R> rec;
            if (!sample_bsdf<R, MT>(m, dir_in, v, rng, rec)) break;
            const Vec3<R> FG = eval_bsdf<R, MT>(sc, m, dir_in, rec.dir_out, rec.pdf, v);
            const Vec3<R> dir_out = normalize(rec.dir_out);
            if (rec.pdf <= R(0)) break;
            if (raw) thr = thr * (FG / rec.pdf);
            st.R_(S_DX, slot) = dir_out.x, st.R_(S_DY, slot) = dir_out.y, st.R_(S_DZ, slot) = dir_out.z;
            st.R_(S_FX, slot) = FG.x, st.R_(S_FY, slot) = FG.y, st.R_(S_FZ, slot) = FG.z;
            st.R_(S_PDF, slot) = rec.pdf;
            new_flags = is_specular ? FLAG_SPECULAR : 0;
            req = REQ_EXTEND;
            break;
        }
        st.R_(S_OX, slot) = v.pos.x, st.R_(S_OY, slot) = v.pos.y, st.R_(S_OZ, slot) = v.pos.z;
        st.I_(S_CTR, slot) = (int32_t)rng.ctr;
        st.I_(S_FLAGS, slot) = new_flags | (iter << FLAG_ITER_SHIFT);
    }
    st.R_(S_TX, slot) = thr.x, st.R_(S_TY, slot) = thr.y, st.R_(S_TZ, slot) = thr.z;
    st.R_(S_LX, slot) = rad.x, st.R_(S_LY, slot) = rad.y, st.R_(S_LZ, slot) = rad.z;
    return req;
}

}  // namespace tk
