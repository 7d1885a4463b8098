"""BVH built on the device (TakeBuildOpts.builder = TAKE_BUILDER_DEVICE_LBVH, take_amd/csrc/tk_build_gpu.h) against the
host SAH build and the oracle.  The tree is different (Morton order, no SAH), the RESULTS must not be: the box tests
are conservative, so closest hits, occlusion and whole images are required to be bit-identical."""
import time

import numpy as np
import pytest

import oracle
from helpers import GOLDEN_SCENES, golden_scene, random_rays, rays_to_abi
from take_amd import capi, scenes
from take_amd import cdefs as D

pytestmark = pytest.mark.gpu
DEV = D.TAKE_BUILDER_DEVICE_LBVH


@pytest.mark.parametrize("name", GOLDEN_SCENES)
def test_device_built_tree_gives_exhaustive_search_hits(name):
    sd = golden_scene(name)
    rays = random_rays(20000, 5, tmin=1e-7).astype(np.float32).astype(np.float64)
    osc = oracle.OracleScene(sd, precision=0)
    want = osc.isect_brute(rays)
    osc.close()
    sc = capi.Scene(sd, precision=D.TAKE_PRECISION_F32, builder=DEV)
    hits = sc.trace_closest(rays_to_abi(rays, 0))
    occ = sc.trace_any(rays_to_abi(rays, 0))
    st = sc.stats()
    sc.close()
    assert st["n_prims"] == sd.n_shapes and st["depth"] >= 1
    assert np.array_equal(hits["shape_id"], want[:, 0].astype(np.int32))
    hit = want[:, 0] >= 0
    for k, col in (("t", 1), ("u", 2), ("v", 3)):
        assert np.array_equal(hits[k][hit].astype(np.float64), want[hit, col]), k
    assert np.array_equal(occ.astype(bool), hit)


@pytest.mark.parametrize("name", ["cbox", "mats", "meshlight"])
def test_device_and_host_builds_render_the_same_image(name):
    sd = golden_scene(name)
    a = capi.Scene(sd, builder=D.TAKE_BUILDER_HOST_SAH)
    b = capi.Scene(sd, builder=DEV)
    try:
        assert np.array_equal(a.render(spp=4, max_depth=50, seed=3), b.render(spp=4, max_depth=50, seed=3))
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("leaf", [1, 2, 4])
def test_device_build_leaf_sizes_100k(leaf):
    sd = scenes.soup_scene(100_000, 256, 256, spp=1)
    a = capi.Scene(sd)
    b = capi.Scene(sd, builder=DEV, max_leaf_size=leaf)
    try:
        rays = rays_to_abi(random_rays(100_000, 11, tmin=1e-4).astype(np.float32).astype(np.float64), 0)
        ha, hb = a.trace_closest(rays), b.trace_closest(rays)
        for f in ("shape_id", "t", "u", "v"):
            assert np.array_equal(ha[f], hb[f]), f
        assert np.array_equal(a.trace_any(rays), b.trace_any(rays))
        assert np.array_equal(a.render(spp=2, max_depth=50, seed=1), b.render(spp=2, max_depth=50, seed=1))
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("leaf", [2, 4])
@pytest.mark.parametrize("precision", [D.TAKE_PRECISION_F32, D.TAKE_PRECISION_F64])
def test_host_build_leaf_sizes_give_the_same_results(leaf, precision):
    """the host builder's default is one primitive per leaf (one ray per lane tests a leaf's primitives one after the
    other); larger leaves go through the per-lane primitive loop and must not change a hit or a pixel"""
    sd = scenes.soup_scene(20_000, 128, 128, spp=1)
    a = capi.Scene(sd, precision=precision, builder=D.TAKE_BUILDER_HOST_SAH)
    b = capi.Scene(sd, precision=precision, builder=D.TAKE_BUILDER_HOST_SAH, max_leaf_size=leaf)
    try:
        assert b.stats()["n_nodes"] < a.stats()["n_nodes"]
        r = random_rays(50_000, 12, tmin=1e-4 if precision == D.TAKE_PRECISION_F32 else 1e-7)
        if precision == D.TAKE_PRECISION_F32:
            r = r.astype(np.float32).astype(np.float64)
        rays = rays_to_abi(r, precision)
        ha, hb = a.trace_closest(rays), b.trace_closest(rays)
        for f in ("shape_id", "t", "u", "v"):
            assert np.array_equal(ha[f], hb[f]), f
        assert np.array_equal(a.trace_any(rays), b.trace_any(rays))
        assert np.array_equal(a.render(spp=2, max_depth=20, seed=1), b.render(spp=2, max_depth=20, seed=1))
    finally:
        a.close()
        b.close()


def test_device_build_1m_triangles_same_results_and_build_time():
    sd = scenes.soup_scene(1_000_000, 1920, 1080, spp=1)
    t0 = time.time()
    a = capi.Scene(sd)
    t_host = time.time() - t0
    t0 = time.time()
    b = capi.Scene(sd, builder=DEV)
    t_dev = time.time() - t0
    try:
        sa, sb = a.stats(), b.stats()
        print(f"\n1M triangles: scene_create host SAH {t_host:.2f} s ({sa['n_nodes']} nodes, depth {sa['depth']}), "
              f"device LBVH {t_dev:.2f} s ({sb['n_nodes']} nodes, depth {sb['depth']})")
        assert sb["n_prims"] == sa["n_prims"]
        rays = rays_to_abi(random_rays(200_000, 3, tmin=1e-4).astype(np.float32).astype(np.float64), 0)
        ha, hb = a.trace_closest(rays), b.trace_closest(rays)
        for f in ("shape_id", "t", "u", "v"):
            assert np.array_equal(ha[f], hb[f]), f
        assert np.array_equal(a.render(spp=1, max_depth=50, seed=4), b.render(spp=1, max_depth=50, seed=4))
        for sc, nm in ((a, "host SAH"), (b, "device LBVH")):
            sc.set_instrumentation(timing=True, counting=True)
            sc.render(spp=1, max_depth=50, seed=4)
            c = sc.counters()
            sc.set_instrumentation(False, False)
            rays_n = c["rays_closest"] + c["rays_shadow"]
            print(f"  {nm}: {c['node_visits'] / rays_n:.1f} nodes/ray, {c['prim_tests'] / rays_n:.1f} prims/ray")
    finally:
        a.close()
        b.close()


def test_device_build_with_coincident_primitives():
    """thousands of primitives with the same Morton code (identical triangles): the index tie-break keeps the tree
    balanced; whatever the builder decides (device tree or host fall-back), results equal the host build's"""
    sd = scenes.soup_scene(64, 64, 64, spp=1)
    tri = np.array([[0.1, 0.1, 0.0], [0.3, 0.1, 0.0], [0.2, 0.3, 0.0]])
    pos = np.tile(tri, (5000, 1))
    sd.add_mesh(pos, np.arange(15000, dtype=np.int32).reshape(-1, 3), 0)
    a = capi.Scene(sd)
    b = capi.Scene(sd, builder=DEV)
    try:
        rays = rays_to_abi(random_rays(20000, 9, tmin=1e-4).astype(np.float32).astype(np.float64), 0)
        ha, hb = a.trace_closest(rays), b.trace_closest(rays)
        assert np.array_equal(ha["t"], hb["t"]) and np.array_equal(a.trace_any(rays), b.trace_any(rays))
    finally:
        a.close()
        b.close()


def test_device_builder_request_on_tiny_and_f64_scenes_uses_the_host_builder():
    """fewer than 8 shapes, or a double-precision scene: the request is honoured by the host builder (documented
    fall-back), results as usual"""
    from take_amd.scene import SceneData

    sd = SceneData(width=16, height=16, lookfrom=(0.0, 0.0, 3.0), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), vfov=40.0,
                   background=(0.2, 0.3, 0.4), spp=1, max_depth=3)
    m = sd.add_material(D.MAT_DIFFUSE, (0.5, 0.5, 0.5))
    pos, idx, nrm, uv = scenes._quad((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1))
    sd.add_mesh(pos, idx, m, normals=nrm, uvs=uv)
    a = capi.Scene(sd)
    b = capi.Scene(sd, builder=DEV)
    assert np.array_equal(a.render(spp=2, max_depth=3, seed=1), b.render(spp=2, max_depth=3, seed=1))
    a.close(), b.close()
    big = golden_scene("soup1k")
    a = capi.Scene(big, precision=D.TAKE_PRECISION_F64)
    b = capi.Scene(big, precision=D.TAKE_PRECISION_F64, builder=DEV)
    assert np.array_equal(a.render(spp=1, max_depth=5, seed=1), b.render(spp=1, max_depth=5, seed=1))
    a.close(), b.close()


def test_device_build_wide_node_format():
    """TAKE_HIP_NODES=wide with the device builder: full-width nodes straight from the collapse kernel"""
    import os

    if os.environ.get("TAKE_HIP_NODES"):
        pytest.skip("experiment knobs select the node format")

    sd = scenes.soup_scene(50_000, 128, 128, spp=1)
    a = capi.Scene(sd, builder=DEV)
    os.environ["TAKE_HIP_NODES"] = "wide"
    try:
        b = capi.Scene(sd, builder=DEV)
    finally:
        del os.environ["TAKE_HIP_NODES"]
    try:
        for sc, nb in ((a, 64), (b, 128)):
            sc.set_instrumentation(timing=False, counting=True)
            img = sc.render(spp=1, max_depth=10, seed=2)
            assert sc.counters()["node_bytes"] == nb
            sc.set_instrumentation(False, False)
        assert np.array_equal(a.render(spp=1, max_depth=10, seed=2), b.render(spp=1, max_depth=10, seed=2))
    finally:
        a.close()
        b.close()


def test_exact_ties_resolve_the_same_way_in_every_tree():
    """two coplanar, overlapping quads of different colour (every ray into the overlap has an exact tie in t between
    triangles of the two quads), axis-aligned boxes sharing planes: the (u, v) rule must give one image for the host
    tree, the device tree, compressed and full-width nodes"""
    import os

    from take_amd.scene import SceneData

    sd = SceneData(width=96, height=96, lookfrom=(0.3, 0.4, 3.0), lookat=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), vfov=45.0,
                   background=(0.1, 0.1, 0.1), spp=4, max_depth=6)
    red = sd.add_material(D.MAT_DIFFUSE, (0.8, 0.1, 0.1))
    blue = sd.add_material(D.MAT_DIFFUSE, (0.1, 0.1, 0.8))
    grey = sd.add_material(D.MAT_DIFFUSE, (0.6, 0.6, 0.6))
    for c, m in (((-0.3, 0.0, 0.0), red), ((0.3, 0.1, 0.0), blue)):  # coplanar (z = 0), overlapping in the middle
        pos, idx, nrm, uv = scenes._quad(c, (0.7, 0, 0), (0, 0.7, 0), (0, 0, 1))
        sd.add_mesh(pos, idx, m, normals=nrm, uvs=uv)
    rng = np.random.default_rng(5)
    for k in range(64):  # a grid of boxes standing on the plane y = -1, touching their neighbours
        x, z = (k % 8) * 0.25 - 1.0, (k // 8) * 0.25 - 1.0
        h = rng.uniform(0.1, 0.5)
        lo, hi = np.array([x, -1.0, z]), np.array([x + 0.25, -1.0 + h, z + 0.25])
        corners = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                            [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]])
        faces = np.array([[0, 1, 2], [0, 2, 3], [4, 6, 5], [4, 7, 6], [0, 4, 5], [0, 5, 1], [3, 2, 6], [3, 6, 7],
                          [0, 3, 7], [0, 7, 4], [1, 5, 6], [1, 6, 2]], np.int32)
        sd.add_mesh(corners, faces, grey)
    pos, idx, nrm, uv = scenes._quad((0, 1.5, 0), (0.5, 0, 0), (0, 0, 0.5), (0, -1, 0))
    sd.add_mesh(pos, idx, grey, normals=nrm, uvs=uv, emission=(10.0, 10.0, 10.0))
    imgs = []
    for builder, fmt in ((D.TAKE_BUILDER_HOST_SAH, None), (DEV, None), (D.TAKE_BUILDER_HOST_SAH, "wide"), (DEV, "wide")):
        if fmt:
            os.environ["TAKE_HIP_NODES"] = fmt
        try:
            sc = capi.Scene(sd, builder=builder)
        finally:
            os.environ.pop("TAKE_HIP_NODES", None)
        imgs.append(sc.render(spp=4, max_depth=6, seed=8))
        sc.close()
    for im in imgs[1:]:
        assert np.array_equal(imgs[0], im)
    overlap = imgs[0][40:56, 44:52]  # pixels looking into the overlap: one of the two colours, not a blend of garbage
    assert np.isfinite(imgs[0]).all() and overlap.mean() > 0
