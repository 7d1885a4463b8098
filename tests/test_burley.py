"""The Burley lobes (material tags 12..16, an EXTENSION: upstream's disney_*.inl are Lambert clones, so there is no
reference output to pin these to — "parity unpinned").  What stands in for golden vectors is the set of properties the
published model has; they are checked here on the oracle's statement (oracle/take_burley.hpp), and the device code is
then held to the oracle value for value (tests/test_gpu_burley.py).

  reciprocity        f(a, b) / cos(b) == f(b, a) / cos(a) for every reflection lobe and their sum
  white furnace      the cosine-weighted albedo of each lobe at base colour 1 is <= 1
  pdf                integrates to <= 1 over the sphere (== 1 up to the mass of directions sampling rejects), is what
                     sample() reports, and sampled directions follow it (histogram against the integrated pdf)
  glass              reflection + transmission at base colour 1 keeps the energy the masking term does not remove
  reductions         principled(metallic = 1) == metal; principled(transmission = 1, specular = 0) == glass;
                     principled(all extras 0) == (tag 6) Disney diffuse
"""
import numpy as np
import pytest

import oracle

METAL, GLASS, CLEARCOAT, SHEEN, BSDF = 12, 13, 14, 15, 16
Z = np.array([0.0, 0.0, 1.0])


def rows(tag, params, dir_in, dir_out, color=(1.0, 1.0, 1.0), seeds=None, back=0.0, gn=Z, sn=Z):
    """table rows (oracle_tab_burley): tag, colour[3], param[12], gn[3], sn[3], dir_in[3], dir_out[3], seed, back"""
    dir_out = np.atleast_2d(np.asarray(dir_out, np.float64))
    n = dir_out.shape[0]
    a = np.zeros((n, 30))
    a[:, 0] = tag
    a[:, 1:4] = color
    p = np.zeros(12)
    p[:len(params)] = params
    a[:, 4:16] = p
    a[:, 16:19] = gn
    a[:, 19:22] = sn
    a[:, 22:25] = dir_in
    a[:, 25:28] = dir_out
    a[:, 28] = np.arange(n) if seeds is None else seeds
    a[:, 29] = back
    return a


def run(a):
    return oracle.table("burley", a)


def unit(theta, phi):
    return np.stack([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)], -1)


def sphere_grid(nt=360, nphi=720):
    """midpoint rule in (cos theta, phi): directions and the solid angle of each cell"""
    ct = -1 + (np.arange(nt) + 0.5) * 2 / nt
    ph = (np.arange(nphi) + 0.5) * 2 * np.pi / nphi
    CT, PH = np.meshgrid(ct, ph, indexing="ij")
    st = np.sqrt(1 - CT * CT)
    d = np.stack([st * np.cos(PH), st * np.sin(PH), CT], -1).reshape(-1, 3)
    return d, (2 / nt) * (2 * np.pi / nphi), (nt, nphi)


def bsdf_params(**kw):
    order = ["specular_transmission", "metallic", "subsurface", "specular", "roughness", "specular_tint", "anisotropic",
             "sheen", "sheen_tint", "clearcoat", "clearcoat_gloss", "eta"]
    d = dict(specular_transmission=0.0, metallic=0.0, subsurface=0.0, specular=0.5, roughness=0.5, specular_tint=0.0,
             anisotropic=0.0, sheen=0.0, sheen_tint=0.5, clearcoat=0.0, clearcoat_gloss=1.0, eta=1.5)
    d.update(kw)
    return [d[k] for k in order]


REFLECTION_CASES = [
    (METAL, [0.4, 0.0], (0.9, 0.6, 0.3)),
    (METAL, [0.3, 0.7], (0.9, 0.6, 0.3)),
    (CLEARCOAT, [0.3], (1, 1, 1)),
    (SHEEN, [0.6], (0.2, 0.5, 0.9)),
    (BSDF, bsdf_params(metallic=0.3, subsurface=0.4, sheen=0.7, clearcoat=0.8, clearcoat_gloss=0.4, specular_tint=0.6,
                       anisotropic=0.5, roughness=0.35), (0.8, 0.4, 0.2)),
]


@pytest.mark.parametrize("tag,params,color", REFLECTION_CASES)
def test_reciprocity(tag, params, color):
    rng = np.random.default_rng(1)
    n = 2000
    a = unit(np.arccos(rng.uniform(0.05, 1, n)), rng.uniform(0, 2 * np.pi, n))
    b = unit(np.arccos(rng.uniform(0.05, 1, n)), rng.uniform(0, 2 * np.pi, n))
    fab = run(rows(tag, params, a, b, color))[:, 10:13]
    fba = run(rows(tag, params, b, a, color))[:, 10:13]
    ca, cb = a[:, 2:3], b[:, 2:3]
    assert np.all(fab >= 0)
    assert fab.max() > 0
    np.testing.assert_allclose(fab / cb, fba / ca, rtol=1e-10, atol=1e-14)


@pytest.mark.parametrize("tag,params", [(METAL, [0.5, 0.0]), (METAL, [0.25, 0.8]), (CLEARCOAT, [0.5]), (SHEEN, [0.0]),
                                        (GLASS, [0.5, 0.0, 1.5]), (GLASS, [0.3, 0.5, 1.33]),
                                        (BSDF, bsdf_params(metallic=0.5, clearcoat=1.0, sheen=0.0, roughness=0.5)),
                                        (BSDF, bsdf_params(specular_transmission=0.7, roughness=0.4))])
@pytest.mark.parametrize("theta_in", [0.2, 1.0])
def test_white_furnace_and_pdf_mass(tag, params, theta_in):
    d, dw, _ = sphere_grid()
    din = unit(np.float64(theta_in), np.float64(0.7))
    o = run(rows(tag, params, din, d))
    f, pdf = o[:, 10:13], o[:, 9]
    assert np.all(f >= 0) and np.all(pdf >= 0) and np.all(np.isfinite(f)) and np.all(np.isfinite(pdf))
    albedo = f.sum(0) * dw
    mass = pdf.sum() * dw
    assert np.all(albedo <= 1.0 + 2e-3), albedo
    assert mass <= 1.0 + 2e-3, mass
    if tag == SHEEN:
        assert abs(mass - 1) < 2e-3  # cosine hemisphere
    else:
        # the missing part: half vectors whose reflection leaves the hemisphere (GGX tails are heavy: alpha 0.25 has
        # 8 % of its visible normals tilted beyond 40 degrees; the clearcoat's GTR1 more)
        assert mass > (0.8 if tag == CLEARCOAT else 0.9), mass
    if tag == GLASS:
        # nothing is absorbed at base colour 1: what is missing is what the masking term drops (no multiple
        # scattering).  The lobe is the one camera paths use — radiance crossing into the denser medium is scaled by
        # 1 / eta^2 — so the energy balance counts the transmitted part eta^2 times.
        up = d[:, 2] > 0
        energy = f[up].sum(0) * dw + params[2] ** 2 * f[~up].sum(0) * dw
        assert np.all(energy <= 1 + 2e-3), energy
        assert np.all(energy > 0.85), energy


@pytest.mark.parametrize("tag,params,back", [(METAL, [0.4, 0.6], 0), (CLEARCOAT, [0.2], 0), (SHEEN, [0.5], 0),
                                             (GLASS, [0.4, 0.0, 1.5], 0), (GLASS, [0.4, 0.3, 1.5], 1),
                                             (BSDF, bsdf_params(specular_transmission=0.5, metallic=0.2, clearcoat=0.6,
                                                                clearcoat_gloss=0.3, roughness=0.4), 0),
                                             (BSDF, bsdf_params(specular_transmission=0.5, metallic=0.2, clearcoat=0.6,
                                                                clearcoat_gloss=0.3, roughness=0.4), 1)])
def test_samples_follow_the_pdf(tag, params, back):
    din = unit(np.float64(0.8), np.float64(2.1))
    n = 60000
    s = run(rows(tag, params, np.tile(din, (n, 1)), np.tile(Z, (n, 1)), seeds=np.arange(n) + 17, back=back))
    assert np.all(s[:, 0] == 1)
    # what sample() reports is what pdf() says about the sampled direction — or zero: a reflection that left through
    # the surface / a refraction that stayed above it is dropped, not priced as the other event
    ok = s[:, 4] > 0
    np.testing.assert_allclose(s[ok, 4], s[ok, 13], rtol=1e-12)
    assert ((s[:, 4] == 0) & (s[:, 13] > 0)).mean() < 0.03
    assert ok.mean() > 0.8
    w = s[ok, 1:4]
    np.testing.assert_allclose(np.linalg.norm(w, axis=1), 1, atol=1e-9)
    # histogram of sampled directions against the pdf integrated per bin
    d, dw, (nt, nphi) = sphere_grid(240, 480)
    pdf = run(rows(tag, params, din, d, back=back))[:, 9].reshape(nt, nphi)
    bt, bp = 8, 8
    want = pdf.reshape(bt, nt // bt, bp, nphi // bp).sum((1, 3)) * dw
    it = np.minimum(((w[:, 2] + 1) / 2 * bt).astype(int), bt - 1)
    ip = np.minimum((np.mod(np.arctan2(w[:, 1], w[:, 0]), 2 * np.pi) / (2 * np.pi) * bp).astype(int), bp - 1)
    got = np.zeros((bt, bp))
    np.add.at(got, (it, ip), 1.0 / n)
    sigma = np.sqrt(np.maximum(want, 1e-6) / n)
    # quadrature of a peaked pdf is the looser side: 5 sigma + 2 % of the bin + 1e-3
    assert np.all(np.abs(got - want) <= 5 * sigma + 0.02 * want + 1e-3), np.abs(got - want).max()


def test_sample_estimator_is_bounded_for_vndf_lobes():
    """f / pdf of a visible-normal sample is F * G1(out) <= 1 per channel (metal), and <= 1 for glass at base 1"""
    din = unit(np.float64(1.2), np.float64(0.3))
    n = 5000
    for tag, params in ((METAL, [0.5, 0.4]), (GLASS, [0.5, 0.4, 1.5])):
        s = run(rows(tag, params, np.tile(din, (n, 1)), np.tile(Z, (n, 1)), seeds=np.arange(n)))
        ok = s[:, 4] > 0
        wgt = s[ok, 6:9] / s[ok, 4:5]
        assert wgt.max() <= 1 + 1e-9
        assert wgt.min() >= 0


def test_reductions_of_the_principled_material():
    rng = np.random.default_rng(5)
    n = 300
    a = unit(np.arccos(rng.uniform(0.05, 1, n)), rng.uniform(0, 2 * np.pi, n))
    b = unit(np.arccos(rng.uniform(-1, 1, n)), rng.uniform(0, 2 * np.pi, n))
    col = (0.7, 0.5, 0.2)
    # the geometric normal stays +z: a dir_out below it is a transmission

    def both(tag1, p1, tag2, p2, cols=slice(9, 13)):
        x = run(rows(tag1, p1, a, b, col))[:, cols]
        y = run(rows(tag2, p2, a, b, col))[:, cols]
        assert np.abs(y).max() > 0
        np.testing.assert_allclose(x, y, rtol=1e-12, atol=1e-15)

    # metallic = 1: the metal lobe with F0 = base colour, sampled by (diffuse 0, metal 1, glass 0, clearcoat 0)
    both(BSDF, bsdf_params(metallic=1.0, roughness=0.3, anisotropic=0.4), METAL, [0.3, 0.4])
    # transmission = 1 with no specular boost: eval is the glass lobe (the pdf also holds the metal lobe's share)
    both(BSDF, bsdf_params(specular_transmission=1.0, specular=0.0, roughness=0.3, eta=1.4), GLASS, [0.3, 0.0, 1.4],
         cols=slice(10, 13))


def test_principled_diffuse_corner_is_the_upstream_disney_diffuse():
    """all extras off: what is left is the (tag 6) Disney diffuse lobe — the one real Disney lobe upstream, pinned to
    the reference's golden table — plus the colourless Schlick reflection of the specular lobe at F0 = 0"""
    rng = np.random.default_rng(6)
    n = 200
    a = unit(np.arccos(rng.uniform(0.05, 1, n)), rng.uniform(0, 2 * np.pi, n))
    b = unit(np.arccos(rng.uniform(0.05, 1, n)), rng.uniform(0, 2 * np.pi, n))
    col = (0.7, 0.5, 0.2)
    x = run(rows(BSDF, bsdf_params(specular=0.0, roughness=0.6, subsurface=0.3), a, b, col))[:, 10:13]
    spec = run(rows(METAL, [0.6, 0.0], a, b, (0.0, 0.0, 0.0)))[:, 10:13]
    m = np.zeros((n, 27))
    m[:, 0] = 6
    m[:, 1:4] = col
    m[:, 4:6] = (0.6, 0.3)
    m[:, 6:9] = Z
    m[:, 9:12] = Z
    m[:, 14:17] = a
    m[:, 17:20] = b
    y = oracle.table("material", m)[:, 10:13]
    np.testing.assert_allclose(x, y + spec, rtol=1e-12, atol=1e-15)


def test_back_face_inverts_the_index():
    """Snell's law on a nearly smooth interface: entering (front face) sin(t) = sin(i) / eta, leaving (back face)
    sin(t) = sin(i) * eta; and a sampled refraction, reversed, is a refraction from the other side (pdf > 0)"""
    din = unit(np.float64(0.3), np.float64(1.0))
    n = 2000
    p = [0.02, 0.0, 1.5]
    for back, ratio in ((0.0, 1 / 1.5), (1.0, 1.5)):
        s = run(rows(GLASS, p, np.tile(din, (n, 1)), np.tile(Z, (n, 1)), seeds=np.arange(n), back=back))
        tr = (s[:, 3] < 0) & (s[:, 4] > 0)
        assert tr.mean() > 0.8
        sin_t = np.hypot(s[tr, 1], s[tr, 2])
        assert abs(np.median(sin_t) / np.sin(0.3) - ratio) < 0.01
    p = [0.2, 0.0, 1.5]
    s = run(rows(GLASS, p, np.tile(din, (n, 1)), np.tile(Z, (n, 1)), seeds=np.arange(n)))
    tr = (s[:, 3] < 0) & (s[:, 4] > 0)
    wo = s[tr, 1:4]
    # from inside: the geometric and shading normals face the new dir_in (= -z side), back_face = 1
    back = run(rows(GLASS, p, wo, np.tile(din, (tr.sum(), 1)), gn=-Z, sn=-Z, back=1.0))
    assert np.all(back[:, 9] > 0)
    assert np.all(back[:, 10:13] > 0)


# ---- the parameters' way in: reference parser -> take_flatten.hpp -> .tkscene -> TakeMaterial::param
def test_flatten_carries_the_disney_parameters():
    """tests/golden/scenes/disney.tkscene = the reference's parse of disney.xml through take_amd/host/take_flatten.hpp:
    every Disney alternative arrives with its members in declaration order (parser defaults where the XML is silent:
    src/parse/parse_scene.cpp:562-700)"""
    from helpers import golden_scene
    from take_amd import cdefs as D

    sd = golden_scene("disney")
    got = [(m.tag, tuple(float(x) for x in m.param)) for m in sd.materials[:7]]
    f32 = lambda x: float(np.float32(x))  # noqa: E731  (the parser reads floats)
    z = 0.0
    want = [
        (D.MAT_DISNEY_METAL, (f32(0.35), f32(0.7)) + (z,) * 10),
        (D.MAT_DISNEY_GLASS, (f32(0.15), z, f32(1.5)) + (z,) * 9),
        (D.MAT_DISNEY_GLASS, (f32(0.3), f32(0.4), f32(1.33)) + (z,) * 9),
        (D.MAT_DISNEY_CLEARCOAT, (f32(0.6),) + (z,) * 11),
        (D.MAT_DISNEY_SHEEN, (f32(0.7),) + (z,) * 11),
        (D.MAT_DISNEY_BSDF, (f32(0.6), z, z, 0.5, f32(0.25), z, z, f32(0.4), 0.5, f32(0.8), 0.5, f32(1.45))),
        (D.MAT_DISNEY_BSDF, (z, f32(0.8), f32(0.3), 0.5, f32(0.4), 0.5, f32(0.6), z, 0.5, z, 1.0, 1.5)),
    ]
    for (gt, gp), (wt, wp) in zip(got, want):
        assert gt == wt
        np.testing.assert_allclose(gp, wp, rtol=1e-7, atol=0)


def test_tkscene_keeps_twelve_parameters_and_the_old_bytes(tmp_path):
    import os

    from helpers import GOLD, golden_scene
    from take_amd import scenes
    from take_amd.scene import load_tkscene, save_tkscene

    sd = scenes.burley_scene(16, 16, 1)
    p = str(tmp_path / "b.tkscene")
    save_tkscene(p, sd)
    back = load_tkscene(p)
    assert [m.tag for m in back.materials] == [m.tag for m in sd.materials]
    for a, b in zip(back.materials, sd.materials):
        assert tuple(a.param) == tuple(b.param) and len(a.param) == 12
    # a scene without Disney parameters is written exactly as before the parameter block grew
    for name in ("cbox", "mats"):
        q = str(tmp_path / (name + ".tkscene"))
        save_tkscene(q, golden_scene(name))
        assert open(q, "rb").read() == open(os.path.join(GOLD, "scenes", name + ".tkscene"), "rb").read()


@pytest.mark.parametrize("precision", [1, 0])
def test_device_code_on_the_host_matches_the_oracle_bit_for_bit(precision):
    """tests/hostsim runs take_amd/csrc/tk_burley.h (the headers the HIP kernels call) on the host: with one libm on
    both sides the image of a scene holding all five lobes — glass sphere, glass cube with triangle back faces,
    anisotropic metal, sheen, clearcoat, two principled mixes — equals the oracle's, f64 and the f32 twin"""
    from helpers import hostsim_render
    from take_amd import scenes

    sd = scenes.burley_scene(48, 48, 4)
    osc = oracle.OracleScene(sd, precision=precision)
    want = osc.render(spp=4, max_depth=8, seed=3)
    osc.close()
    got, _ = hostsim_render(sd, precision, 4, 8, seed=3)
    assert np.isfinite(want).all() and want.mean() > 0.1
    assert np.array_equal(got, want.astype(got.dtype))
    # and the lobes are not the Lambert clones
    stub = scenes.burley_scene(48, 48, 4, real=False)
    lam, _ = hostsim_render(stub, precision, 4, 8, seed=3)
    assert np.abs(lam.astype(np.float64) - want).mean() > 0.01


@pytest.mark.parametrize("integrator", [1, 2, 3])
def test_other_integrators_take_the_lobes_too(integrator):
    from helpers import hostsim_render
    from take_amd import scenes

    sd = scenes.burley_scene(32, 32, 2)
    osc = oracle.OracleScene(sd, precision=1)
    want = osc.render(spp=2, max_depth=6, seed=1, integrator=integrator)
    osc.close()
    got, _ = hostsim_render(sd, 1, 2, 6, seed=1, integrator=integrator)
    assert np.array_equal(got, want)


def test_out_of_range_parameters_are_refused():
    """sqrt(1 - 0.9 anisotropic), log(alpha_g^2), 1 / eta: a parameter outside the model's range would render NaN —
    scene preparation (the code the C ABI runs before anything reaches the GPU) rejects it with a message"""
    from helpers import hostsim_render
    from take_amd import scenes
    from take_amd.scene import SceneData
    from take_amd import cdefs as D

    def scene(tag, params):
        sd = SceneData(width=8, height=8, lookfrom=(0.0, 0.0, 3.0), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), vfov=40.0,
                       background=(0.5, 0.5, 0.5), spp=1, max_depth=2)
        sd.add_sphere((0.0, 0.0, 0.0), 0.5, sd.add_material(tag, (0.5, 0.5, 0.5), params))
        return sd

    hostsim_render(scene(D.MAT_BURLEY_METAL, (0.3, 1.0)), 1, 1, 2)  # in range: fine
    hostsim_render(scene(D.MAT_DISNEY_METAL, (0.3, 7.0)), 1, 1, 2)  # the reference's stub ignores its parameters
    for tag, params in ((D.MAT_BURLEY_METAL, (0.3, 1.5)), (D.MAT_BURLEY_GLASS, (0.3, 0.0, 0.0)),
                        (D.MAT_BURLEY_GLASS, (float("nan"), 0.0, 1.5)), (D.MAT_BURLEY_CLEARCOAT, (-0.1,)),
                        (D.MAT_BURLEY_BSDF, scenes.principled(metallic=1.2)), (D.MAT_BURLEY_BSDF, scenes.principled(eta=-1.0))):
        with pytest.raises(RuntimeError, match="Burley parameter"):
            hostsim_render(scene(tag, params), 1, 1, 2)
