"""Instancing (TakeInstance, BASELINE configs[4]: "10M-triangle instanced scene"): an EXTENSION — the reference has no
instancing, so PARITY IS UNPINNED with respect to it.  The specification is SURVEY.md §8(c)'s:
an instanced scene renders as the same geometry flattened to world-space triangles, to fp rounding (the ray is moved
into the prototype's object space instead of the triangles into world space: same t, different rounding).

  * CPU: the device code on the host (two-level traversal as a nested call) — shape ids identical, t within 1e-13
    (f64) / 1e-5 (f32), renders RMSE < 1e-12 / 2e-4;
  * GPU: the trace kernel's one-stack version (return marker) against the flattened scene through the same C ABI:
    same bars; any-hit == closest-hit boolean; determinism, strip and batch invariance; a transform with scale and
    shear; vertex normals on the prototype; a per-placement material; memory: one prototype + N transforms.
"""
import numpy as np
import pytest

from helpers import hostsim_render, hostsim_trace, random_rays, rays_to_abi, rmse
from take_amd import cdefs as D
from take_amd import scenes
from take_amd.dist import strip_rows
from take_amd.scene import SceneData


def small(n_inst=40, tris=200, res=48):
    return scenes.instanced_scene(n_inst, tris, res, res, spp=4, max_depth=8)


def sheared_with_normals():
    """two placements of a quad with vertex normals and uvs under a non-rigid transform (scale + shear), one with its
    own material; a plain quad light above"""
    sd = SceneData(width=40, height=40, lookfrom=(0.0, 1.2, 3.0), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), vfov=40.0,
                   background=(0.2, 0.3, 0.4), spp=4, max_depth=4)
    grey = sd.add_material(D.MAT_DIFFUSE, (0.7, 0.7, 0.7))
    gold = sd.add_material(D.MAT_BLINN_PHONG_MICROFACET, (0.8, 0.7, 0.3), (30.0,))
    pos, idx, nrm, uv = scenes._quad((0, 0, 0), (0.5, 0, 0), (0, 0, -0.5), (0, 1, 0))
    nrm = nrm + np.array([[0.2, 0, 0.1], [-0.2, 0, 0.1], [-0.2, 0, -0.1], [0.2, 0, -0.1]])  # bent vertex normals
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    proto = sd.add_prototype(pos, idx, grey, normals=nrm, uvs=uv)
    sd.add_instance(proto, [[1.5, 0.3, 0.0, -0.4], [0.0, 1.0, 0.2, 0.0], [0.1, 0.0, 0.8, 0.2]])
    sd.add_instance(proto, [[0.6, 0.0, 0.0, 0.6], [0.2, 0.7, 0.0, 0.3], [0.0, 0.0, 1.1, -0.3]], gold)
    lp, li, ln, lu = scenes._quad((0, 1.5, 0), (0.4, 0, 0), (0, 0, 0.4), (0, -1, 0))
    sd.add_mesh(lp, li, grey, normals=ln, uvs=lu, emission=(12.0, 12.0, 12.0))
    return sd


@pytest.mark.parametrize("precision", [1, 0])
def test_device_code_on_host_instanced_equals_flattened(precision):
    sd = small()
    fl = sd.flattened()
    assert len(sd.meshes) == 7 and len(fl.meshes) == 47 and fl.n_shapes == 12 + 40 * 200
    rays = random_rays(8000, 3, tmin=1e-7 if precision else 1e-4)
    if precision == 0:
        rays = rays.astype(np.float32).astype(np.float64)
    a = hostsim_trace(sd, precision, rays).astype(np.float64)
    b = hostsim_trace(fl, precision, rays).astype(np.float64)
    assert np.array_equal(a[:, 0], b[:, 0])  # same shape ids: numbering follows the flattened scene's
    hit = a[:, 0] >= 0
    assert hit.sum() > 4000 and np.abs(a[hit, 1] - b[hit, 1]).max() < (1e-13 if precision else 1e-5)
    assert np.array_equal(hostsim_trace(sd, precision, rays, any_hit=True)[:, 0] >= 0, hit)
    ia, _ = hostsim_render(sd, precision, 4, 8, seed=1)
    ib, _ = hostsim_render(fl, precision, 4, 8, seed=1)
    assert rmse(ia, ib) < (1e-12 if precision else 2e-4), rmse(ia, ib)


def test_host_shear_normals_and_material_override():
    sd = sheared_with_normals()
    ia, _ = hostsim_render(sd, 1, 4, 4, seed=2)
    ib, _ = hostsim_render(sd.flattened(), 1, 4, 4, seed=2)
    assert rmse(ia, ib) < 1e-12 and ia.mean() > 0.05


@pytest.mark.parametrize("seed", [1, 2, 3, 5, 8, 13])
def test_host_random_scenes_with_sheared_and_mirrored_placements(seed):
    from fuzz_scenes import add_random_instances, random_rays, random_scene

    sd, scale = random_scene(seed, res=16)
    add_random_instances(sd, scale, seed)
    fl = sd.flattened()
    for precision in (1, 0):
        rays = random_rays(sd, scale, 800, seed, 1e-7 if precision else 1e-4)
        if precision == 0:
            rays = rays.astype(np.float32).astype(np.float64)
        a = hostsim_trace(sd, precision, rays).astype(np.float64)
        b = hostsim_trace(fl, precision, rays).astype(np.float64)
        same = a[:, 0] == b[:, 0]
        assert same.mean() > 0.995
        hit = same & (a[:, 0] >= 0)
        assert (np.abs(a[hit, 1] - b[hit, 1]) / np.maximum(scale, b[hit, 1])).max() < (1e-12 if precision else 1e-4)
        ia, _ = hostsim_render(sd, precision, 2, 4, seed=seed)
        ib, _ = hostsim_render(fl, precision, 2, 4, seed=seed)
        d = np.abs(ia.astype(np.float64) - ib).max(axis=2)
        assert (d > (1e-9 if precision else 1e-3)).mean() < 0.02  # a path may flip at an edge; nothing systematic


# ------------------------------------------------------------------ GPU, through the C ABI
@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 5, 8, 13])
def test_gpu_random_scenes_with_sheared_and_mirrored_placements(seed):
    from fuzz_scenes import add_random_instances, random_rays, random_scene
    from take_amd import capi

    sd, scale = random_scene(seed, res=24)
    add_random_instances(sd, scale, seed)
    fl = sd.flattened()
    for precision in (D.TAKE_PRECISION_F64, D.TAKE_PRECISION_F32):
        f64 = precision == D.TAKE_PRECISION_F64
        a, b = capi.Scene(sd, precision=precision), capi.Scene(fl, precision=precision)
        try:
            rays = random_rays(sd, scale, 20000, seed, 1e-7 if f64 else 1e-4)
            if not f64:
                rays = rays.astype(np.float32).astype(np.float64)
            ha, hb = a.trace_closest(rays_to_abi(rays, precision)), b.trace_closest(rays_to_abi(rays, precision))
            same = ha["shape_id"] == hb["shape_id"]
            assert same.mean() > 0.995, same.mean()
            hit = same & (ha["shape_id"] >= 0)
            dt = np.abs(ha["t"][hit].astype(np.float64) - hb["t"][hit]) / np.maximum(scale, hb["t"][hit])
            assert dt.max() < (1e-12 if f64 else 1e-4), dt.max()
            assert np.array_equal(a.trace_any(rays_to_abi(rays, precision)).astype(bool), ha["shape_id"] >= 0)
            ia, ib = a.render(spp=4, max_depth=4, seed=seed), b.render(spp=4, max_depth=4, seed=seed)
            d = np.abs(ia.astype(np.float64) - ib).max(axis=2)
            assert (d > (1e-8 if f64 else 1e-3) * max(1.0, float(ib.max()))).mean() < 0.03
        finally:
            a.close()
            b.close()



@pytest.mark.gpu
@pytest.mark.parametrize("precision", [D.TAKE_PRECISION_F64, D.TAKE_PRECISION_F32])
def test_gpu_instanced_equals_flattened(precision):
    from take_amd import capi

    sd = scenes.instanced_scene(60, 500, 96, 96, spp=4, max_depth=50)
    fl = sd.flattened()
    a, b = capi.Scene(sd, precision=precision), capi.Scene(fl, precision=precision)
    try:
        f64 = precision == D.TAKE_PRECISION_F64
        rays = random_rays(50000, 4, tmin=1e-7 if f64 else 1e-4)
        if not f64:
            rays = rays.astype(np.float32).astype(np.float64)
        ha, hb = a.trace_closest(rays_to_abi(rays, precision)), b.trace_closest(rays_to_abi(rays, precision))
        same = ha["shape_id"] == hb["shape_id"]
        assert same.mean() > 0.9999, same.mean()  # a ray through a shared edge may pick the neighbour in the other space
        hit = same & (ha["shape_id"] >= 0)
        dt = np.abs(ha["t"][hit].astype(np.float64) - hb["t"][hit]) / np.maximum(1.0, hb["t"][hit])
        assert dt.max() < (1e-13 if f64 else 1e-5), dt.max()  # relative: t reaches 5 in this scene (measured 2.9e-6)
        assert np.array_equal(a.trace_any(rays_to_abi(rays, precision)).astype(bool), ha["shape_id"] >= 0)
        ia = a.render(spp=4, max_depth=50, seed=5)
        ib = b.render(spp=4, max_depth=50, seed=5)
        d = np.abs(ia.astype(np.float64) - ib).max(axis=2)
        if f64:
            assert np.median(d) < 1e-12 and (d < 1e-9).mean() > 0.99, (d < 1e-9).mean()
        else:
            assert rmse(ia, ib) < 3e-3 and (d < 1e-3).mean() > 0.97, (rmse(ia, ib), (d < 1e-3).mean())
        # invariances of the two-level path
        assert np.array_equal(ia, a.render(spp=4, max_depth=50, seed=5))
        assert np.array_equal(ia, a.render(spp=4, max_depth=50, seed=5, samples_per_batch=1))
        img = np.zeros_like(ia)
        for r in range(3):
            img[strip_rows(96, r, 3)] = a.render(spp=4, max_depth=50, seed=5, strip_first=r, strip_stride=3)
        assert np.array_equal(img, ia)
    finally:
        a.close()
        b.close()


@pytest.mark.gpu
def test_gpu_shear_normals_material_override_and_memory():
    from take_amd import capi

    sd = sheared_with_normals()
    a, b = capi.Scene(sd, precision=D.TAKE_PRECISION_F64), capi.Scene(sd.flattened(), precision=D.TAKE_PRECISION_F64)
    try:
        ia, ib = a.render(spp=8, max_depth=4, seed=2), b.render(spp=8, max_depth=4, seed=2)
        assert rmse(ia, ib) < 1e-9 and ia.mean() > 0.05
    finally:
        a.close()
        b.close()
    # configs[4]'s shape at a tenth of its size: 1000 placements of a 1000-triangle mesh
    big = scenes.instanced_scene(1000, 1000, 320, 180, spp=2, max_depth=50)
    inst, flat = capi.Scene(big), capi.Scene(big.flattened())
    try:
        si, sf = inst.stats(), flat.stats()
        assert sf["device_bytes"] > 50 * si["device_bytes"], (si, sf)  # one prototype + 1000 transforms vs 1M triangles
        ia, ib = inst.render(spp=2, max_depth=50, seed=1), flat.render(spp=2, max_depth=50, seed=1)
        d = np.abs(ia.astype(np.float64) - ib).max(axis=2)
        assert rmse(ia, ib) < 5e-3 and (d < 1e-3).mean() > 0.95, (rmse(ia, ib), (d < 1e-3).mean())
        assert abs(ia.mean() - ib.mean()) / ib.mean() < 2e-3
    finally:
        inst.close()
        flat.close()


def coincident_instance_scene():
    """a quad as ordinary geometry AND, exactly coincident with it, an identity-transform placement of the same quad
    (its own material): every ray through the quad ties in (t, u, v) between a top-level primitive and a primitive
    inside an instance.  Rule (tk_trace_quad.h, tk_traverse.h): the larger (instance, primitive) wins, a top-level
    primitive counting as instance -1 — so the placement's copy is the hit, in every tree and visiting order."""
    sd = SceneData(width=32, height=32, lookfrom=(0.0, 0.0, 3.0), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), vfov=40.0,
                   background=(0.3, 0.3, 0.3), spp=2, max_depth=3)
    red = sd.add_material(D.MAT_DIFFUSE, (0.8, 0.1, 0.1))
    blue = sd.add_material(D.MAT_DIFFUSE, (0.1, 0.1, 0.8))
    pos, idx, nrm, uv = scenes._quad((0, 0, 0), (0.7, 0, 0), (0, 0.7, 0), (0, 0, 1))
    sd.add_mesh(pos, idx, red, normals=nrm, uvs=uv)                       # shapes 0, 1
    far, fi, fn, fu = scenes._quad((0, 0, -1.0), (1.5, 0, 0), (0, 1.5, 0), (0, 0, 1))
    sd.add_mesh(far, fi, red, normals=fn, uvs=fu)                         # shapes 2, 3 (a backdrop: more than one leaf)
    proto = sd.add_prototype(pos, idx, blue, normals=nrm, uvs=uv)
    sd.add_instance(proto, [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]])   # shapes 4, 5
    return sd


def _rays_at_the_quad(n, seed):
    rng = np.random.default_rng(seed)
    rays = np.zeros((n, 8))
    rays[:, 0:2] = rng.uniform(-0.6, 0.6, (n, 2))
    rays[:, 2] = 2.0
    rays[:, 3:6] = (0.0, 0.0, -1.0)
    rays[:, 6], rays[:, 7] = 1e-4, np.inf
    return rays


@pytest.mark.parametrize("precision", [1, 0])
def test_host_instance_coincident_with_plain_geometry_tie_rule(precision):
    sd = coincident_instance_scene()
    rays = _rays_at_the_quad(500, 4)
    hits = hostsim_trace(sd, precision, rays)
    assert np.isin(hits[:, 0], (4, 5)).all()  # the placement's faces, never the coincident top-level ones (0, 1)
    assert np.abs(hits[:, 1].astype(np.float64) - 2.0).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [D.TAKE_PRECISION_F64, D.TAKE_PRECISION_F32])
def test_gpu_instance_coincident_with_plain_geometry_tie_rule(precision):
    from take_amd import capi

    sd = coincident_instance_scene()
    rays = _rays_at_the_quad(5000, 4)
    p = 1 if precision == D.TAKE_PRECISION_F64 else 0
    want = hostsim_trace(sd, p, rays)
    sc = capi.Scene(sd, precision=precision)
    try:
        got = sc.trace_closest(rays_to_abi(rays, p))
        assert np.isin(got["shape_id"], (4, 5)).all()
        assert np.array_equal(got["shape_id"].astype(np.float64), want[:, 0].astype(np.float64))
        for k, col in (("t", 1), ("u", 2), ("v", 3)):
            assert np.array_equal(got[k].astype(np.float64), want[:, col].astype(np.float64)), k
        img = sc.render(spp=2, max_depth=3, seed=1)
        centre = img[12:20, 12:20].mean(axis=(0, 1))
        assert centre[2] > centre[0]  # the quad renders in the placement's blue, not the coincident red
    finally:
        sc.close()


@pytest.mark.parametrize("braid", [4, 16])
def test_host_braided_placements_equal_flattened(braid, monkeypatch):
    """TAKE_HIP_BRAID > 1: a placement enters the top-level tree as several entries (subtrees of the prototype's tree,
    each with its own instance record and root) — same hits, same shape ids"""
    monkeypatch.setenv("TAKE_HIP_BRAID", str(braid))
    sd = small(n_inst=30, tris=300)
    fl = sd.flattened()
    rays = random_rays(6000, 11, tmin=1e-7)
    a = hostsim_trace(sd, 1, rays).astype(np.float64)
    monkeypatch.delenv("TAKE_HIP_BRAID")
    b = hostsim_trace(fl, 1, rays).astype(np.float64)
    assert np.array_equal(a[:, 0], b[:, 0])
    hit = a[:, 0] >= 0
    assert hit.sum() > 2500 and np.abs(a[hit, 1] - b[hit, 1]).max() < 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("precision,builder", [(D.TAKE_PRECISION_F32, D.TAKE_BUILDER_HOST_SAH), (D.TAKE_PRECISION_F32, D.TAKE_BUILDER_DEVICE_LBVH),
                                               (D.TAKE_PRECISION_F64, D.TAKE_BUILDER_AUTO), (D.TAKE_PRECISION_MIXED, D.TAKE_BUILDER_AUTO)])
def test_gpu_library_flattening_is_the_python_flattening(precision, builder):
    """TakeBuildOpts.instances = TAKE_INSTANCES_FLATTEN: scene_create expands the placements itself (world-space
    positions, n^T L^-1 normals, material override, shape ids in placement order).  Same arithmetic as
    SceneData.flattened(), so the scene is the same scene: statistics, images and hit tables bit for bit."""
    from helpers import random_rays, rays_to_abi
    from take_amd import capi

    for sd in (small(60, 300, 64), sheared_with_normals()):
        a = capi.Scene(sd, precision=precision, builder=builder, flatten_instances=True)
        b = capi.Scene(sd.flattened(), precision=precision, builder=builder)
        c = capi.Scene(sd, precision=precision)  # two levels
        try:
            assert a.stats() == b.stats()
            ia, ib = a.render(spp=4, max_depth=6, seed=5), b.render(spp=4, max_depth=6, seed=5)
            assert np.array_equal(ia, ib) and ia.mean() > 0.01
            rays = rays_to_abi(random_rays(4096, 3, tmin=1e-4), 0 if precision == D.TAKE_PRECISION_F32 else 1)
            ha, hb, hc = a.trace_closest(rays), b.trace_closest(rays), c.trace_closest(rays)
            for k in ("shape_id", "t", "u", "v"):
                assert np.array_equal(ha[k], hb[k]), k
            # the two-level scene numbers its shapes the same way (hits agree except where rounding moves an edge)
            assert (ha["shape_id"] == hc["shape_id"]).mean() > 0.995 and (ha["shape_id"] >= 0).mean() > 0.05
        finally:
            a.close(), b.close(), c.close()


@pytest.mark.gpu
def test_gpu_flattening_refuses_what_it_cannot_expand():
    from take_amd import capi

    sd = small(4, 50, 16)
    sd.instance_mesh[2] = 99
    with pytest.raises(capi.TakeError) as e:
        capi.Scene(sd, flatten_instances=True)
    assert e.value.code == D.TAKE_E_INVALID and "bad mesh index" in str(e.value)
    sd = small(4, 50, 16)
    sd.instance_xform[1] = np.zeros((3, 4))
    sd.meshes[sd.instance_mesh[1]].normals = np.tile([0.0, 1.0, 0.0], (len(sd.meshes[sd.instance_mesh[1]].positions), 1))
    with pytest.raises(capi.TakeError) as e:
        capi.Scene(sd, flatten_instances=True)
    assert "singular transform" in str(e.value)
