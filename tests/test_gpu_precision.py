"""GPU: the PRODUCTION precision (f32, the path bench.py times) against the PINNED reference-precision oracle
(oracle<double>, bit-exact with the compiled reference) at matched counter seeds — and the two waivers of the
reference's quirk list that the f32 path takes, pinned here instead of implicit.

What f32 can and cannot match.  At matched seeds an f32 path follows its f64 twin until a discrete decision flips
(a hit/miss at a triangle edge, `floor(xi * n_lights)`, Plastic's `xi <= F`, a `light_pdf <= 0` break): one flipped
sample of 16 changes its pixel by O(radiance / 16).  Measured on the CPU twin (16 spp, depth 50, three seeds): about
one sample in 10^4..10^5 flips.  That sets a floor of a few 1e-4 on the per-pixel RMSE of a 64x64 x 16 spp image that
no epsilon choice removes (sweeping the ray epsilon from 1e-7 to 1e-3: 1e-4..2e-4 is the flat minimum; 1e-7, the
reference's f64 value, self-intersects in f32 and gives RMSE 0.5).  Two scenes sit above 1e-3 for a stated reason:
  * spherelight: a path standing ON the sphere light samples that sphere from its own surface; the reference's
    `light_pdf <= 0 -> break` (path_tracing.h:40-43) is then decided by the last bit of r/d (also in f64, also
    between glibc and ocml): ~100 of 3072 pixels differ by a few 1e-2;
  * meshlight: 32 emissive triangles of radiance 17; a single flipped sample on the light moves one pixel by 0.6.
Bars below = 2x the worst value measured over seeds 1..3 on the MI355X, and a robust statistic (share of pixels within
1e-3) next to the RMSE so that one outlier cannot hide a systematic error.
"""
import numpy as np
import pytest

import oracle
from helpers import GOLDEN_SCENES, golden_scene, random_rays, rays_to_abi, rmse
from take_amd import capi, scenes
from take_amd import cdefs as D
from take_amd.scene import SceneData

pytestmark = pytest.mark.gpu

# scene -> (RMSE bar, minimum share of pixels within 1e-3 of the f64 oracle)
# measured on the MI355X (worst of seeds 1..3): cbox 3.3e-4 / 0.997, mats 2.2e-3 / 0.993, soup1k 1.5e-4 / 0.998,
# spherelight 2.7e-3 / 0.961, meshlight 2.7e-4 / 0.999 (the CPU twin at seed 3 has one flipped sample ON the light: 5.9e-3)
F32_VS_F64_BARS = {"cbox": (1e-3, 0.995), "mats": (4.5e-3, 0.985), "soup1k": (1e-3, 0.995), "spherelight": (6e-3, 0.93),
                   "meshlight": (1.5e-2, 0.99)}


@pytest.mark.parametrize("name", GOLDEN_SCENES)
def test_f32_gpu_render_vs_pinned_f64_oracle(name, record_property):
    sd = golden_scene(name)
    sc = capi.Scene(sd, precision=D.TAKE_PRECISION_F32)
    osc = oracle.OracleScene(sd, precision=1)  # reference arithmetic: double, ray epsilon 1e-7
    worst, share = 0.0, 1.0
    try:
        for seed in (1, 2, 3):
            want = osc.render(16, 50, rng_mode=oracle.RNG_COUNTER, seed=seed)
            got = sc.render(spp=16, max_depth=50, seed=seed).astype(np.float64)
            d = np.abs(got - want).max(axis=2)
            worst = max(worst, rmse(got, want))
            share = min(share, float((d < 1e-3).mean()))
            assert abs(got.mean() - want.mean()) / want.mean() < 2e-3  # no systematic brightness error
    finally:
        sc.close()
        osc.close()
    record_property("f32_vs_f64_rmse", worst)
    print(f"{name}: f32 GPU vs f64 oracle, worst RMSE over 3 seeds {worst:.3e}, pixels within 1e-3: {share:.4f}")
    bar, min_share = F32_VS_F64_BARS[name]
    assert worst < bar, worst
    assert share >= min_share, share


def test_f32_vs_f64_on_the_bench_workload_shape():
    """100k-triangle soup (configs[1]'s scene), 256x256, 16 spp, depth 50: f32 GPU vs f64 GPU at matched seeds (the f64
    GPU path is itself at rounding level of the pinned oracle, test_gpu_parity.py).  This is the statistic bench.py
    reports for the full configs[2] workload (`parity.f32_vs_f64_rmse`, there at 256 spp).  On a soup a path makes
    ~25 rays x ~11 triangle tests against triangles of 1 % of the scene size: an f32 ray passes on the other side of
    some edge once in ~100 paths, and a flipped sample differs by its whole radiance (the sky's sun is 400): measured
    RMSE 3.6e-3 at 16 spp, 95.9 % of the pixels within 1e-3; the error of a pixel falls with 1/sqrt(spp)."""
    sd = scenes.soup_scene(100_000, 256, 256, spp=16, envmap=(256, 128))
    a = capi.Scene(sd, precision=D.TAKE_PRECISION_F32)
    b = capi.Scene(sd, precision=D.TAKE_PRECISION_F64)
    try:
        f32 = a.render(spp=16, max_depth=50, seed=2).astype(np.float64)
        f64 = b.render(spp=16, max_depth=50, seed=2)
    finally:
        a.close()
        b.close()
    e = rmse(f32, f64)
    d = np.abs(f32 - f64).max(axis=2)
    print(f"100k soup + env-map: f32 vs f64 RMSE {e:.3e}, pixels within 1e-3: {(d < 1e-3).mean():.4f}")
    assert e < 8e-3 and (d < 1e-3).mean() > 0.92
    assert abs(f32.mean() - f64.mean()) / f64.mean() < 1e-3


def _tie_scene():
    """two coplanar overlapping quads (z = 0) of different materials: every ray into the overlap hits a triangle of
    each quad at exactly the same t"""
    sd = SceneData(width=32, height=32, lookfrom=(0.0, 0.0, 3.0), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), vfov=45.0,
                   background=(0.1, 0.1, 0.1), spp=1, max_depth=1)
    red = sd.add_material(D.MAT_DIFFUSE, (0.8, 0.1, 0.1))
    blue = sd.add_material(D.MAT_DIFFUSE, (0.1, 0.1, 0.8))
    for c, m in (((-0.25, 0.0, 0.0), red), ((0.25, 0.125, 0.0), blue)):
        pos, idx, nrm, uv = scenes._quad(c, (0.75, 0, 0), (0, 0.75, 0), (0, 0, 1))
        sd.add_mesh(pos, idx, m, normals=nrm, uvs=uv)
    pos, idx, nrm, uv = scenes._quad((0, 0, -1.5), (2, 0, 0), (0, 2, 0), (0, 0, 1))
    sd.add_mesh(pos, idx, red, normals=nrm, uvs=uv)
    return sd


@pytest.mark.parametrize("precision", [D.TAKE_PRECISION_F32, D.TAKE_PRECISION_F64])
def test_exact_ties_vs_reference_visiting_order(precision):
    """WAIVER of SURVEY.md App. B #9, pinned.  The reference gives an exact tie in t to the LATER subtree of its
    median-split tree (bvh.cpp:100-106: the right child is tested with tmax = the left hit's t and the triangle test
    accepts t == tmax, shape.cpp:77) — a property of that tree.  The GPU resolves a tie on values: the candidate with
    the lexicographically larger (u, v) wins, which every tree computes identically (tk_trace_quad.h: finish()).
    Here: distance and hit/miss always equal the reference's; where the winning shape differs from the reference's,
    both are tied candidates and the GPU's has the larger (u, v)."""
    sd = _tie_scene()
    rng = np.random.default_rng(3)
    n = 4000
    o = np.tile(np.array([0.0, 0.0, 3.0]), (n, 1)) + rng.uniform(-0.5, 0.5, (n, 3)) * np.array([1, 1, 0])
    tgt = np.stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-0.6, 0.7, n), np.zeros(n)], 1)  # inside the overlap region
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.hstack([o, d, np.full((n, 1), 1e-7), np.full((n, 1), np.inf)])
    if precision == D.TAKE_PRECISION_F32:
        rays = rays.astype(np.float32).astype(np.float64)
    osc = oracle.OracleScene(sd, precision=precision)
    ref = osc.isect(rays)  # reference BVH + reference traversal: columns 0 hit, 1 t, 16 shape, 17 u, 18 v
    osc.close()
    sc = capi.Scene(sd, precision=precision)
    got = sc.trace_closest(rays_to_abi(rays, precision))
    sc.close()
    hit = ref[:, 0] > 0
    assert np.array_equal(got["shape_id"] >= 0, hit)
    assert np.array_equal(got["t"][hit].astype(np.float64), ref[hit, 1])  # the distance never depends on the rule
    differ = hit & (got["shape_id"] != ref[:, 16].astype(np.int32))
    same = hit & ~differ
    assert np.array_equal(got["u"][same].astype(np.float64), ref[same, 17])
    assert differ.sum() > 100, "the scene is built so that ties are common"
    gu, gv = got["u"][differ].astype(np.float64), got["v"][differ].astype(np.float64)
    ru, rv = ref[differ, 17], ref[differ, 18]
    assert np.all((gu > ru) | ((gu == ru) & (gv >= rv)))  # the documented rule: larger (u, v) wins


def _coincident_scene():
    """three exactly coincident triangles (same vertices) with DIFFERENT vertex normals and materials, among a few
    others: (t, u, v) of a ray are identical for all three"""
    sd = SceneData(width=48, height=48, lookfrom=(0.0, 0.3, 3.0), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), vfov=40.0,
                   background=(0.2, 0.2, 0.3), spp=4, max_depth=3)
    mats = [sd.add_material(D.MAT_DIFFUSE, c) for c in ((0.8, 0.1, 0.1), (0.1, 0.8, 0.1), (0.1, 0.1, 0.8), (0.6, 0.6, 0.6))]
    tri = np.array([[-0.8, -0.5, 0.0], [0.8, -0.5, 0.1], [0.0, 0.9, -0.1]])
    rng = np.random.default_rng(2)
    for k in range(3):
        n = np.array([[0.1 * k, 0.2, 1.0]] * 3) + rng.normal(0, 0.05, (3, 3))
        sd.add_mesh(tri, [[0, 1, 2]], mats[k], normals=n / np.linalg.norm(n, axis=1, keepdims=True), uvs=rng.uniform(0, 1, (3, 2)))
    pos, idx = scenes.soup_triangles(200, 5, 0.9, 0.1)
    sd.add_mesh(pos + np.array([0, 0, -0.6]), idx, mats[3])
    lp, li, ln, lu = scenes._quad((0, 1.6, 0.5), (0.5, 0, 0), (0, 0, 0.5), (0, -1, 0))
    sd.add_mesh(lp, li, mats[3], normals=ln, uvs=lu, emission=(9.0, 9.0, 9.0))
    return sd


@pytest.mark.parametrize("precision", [D.TAKE_PRECISION_F32, D.TAKE_PRECISION_F64])
def test_coincident_primitives_resolve_by_shape_id_in_every_tree(precision):
    """Exactly coincident primitives tie in t AND (u, v): the larger shape id wins (found by the fuzz tests: before,
    the winner depended on the visiting order, i.e. on the tree).  Host and device builders, compressed and
    full-width nodes: one image; the trace hook reports the largest of the coincident shape ids."""
    import os

    sd = _coincident_scene()
    rays = np.array([[0.0, 0.1, 3.0, 0.0, 0.0, -1.0, 1e-4, np.inf], [0.2, -0.1, 2.0, 0.01, 0.02, -1.0, 1e-4, np.inf]])
    rays[:, 3:6] /= np.linalg.norm(rays[:, 3:6], axis=1, keepdims=True)
    if precision == D.TAKE_PRECISION_F32:
        rays = rays.astype(np.float32).astype(np.float64)
    imgs = []
    for builder, fmt in ((D.TAKE_BUILDER_HOST_SAH, None), (D.TAKE_BUILDER_DEVICE_LBVH, None), (D.TAKE_BUILDER_HOST_SAH, "wide")):
        if fmt:
            os.environ["TAKE_HIP_NODES"] = fmt
        try:
            sc = capi.Scene(sd, precision=precision, builder=builder, max_leaf_size=1 if builder == D.TAKE_BUILDER_HOST_SAH and fmt else 0)
        finally:
            os.environ.pop("TAKE_HIP_NODES", None)
        try:
            hits = sc.trace_closest(rays_to_abi(rays, precision))
            assert list(hits["shape_id"]) == [2, 2]  # shapes 0, 1, 2 coincide: the largest id
            imgs.append(sc.render(spp=4, max_depth=3, seed=6))
        finally:
            sc.close()
    assert np.array_equal(imgs[0], imgs[1]) and np.array_equal(imgs[0], imgs[2])
