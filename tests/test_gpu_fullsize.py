"""GPU parity at BASELINE.json's full sizes (configs[1]: 100k soup 1024x1024; configs[2]/[3]: 1M soup 1920x1080),
where the oracle cannot run in seconds: size-independent properties, all bit-exact.

  * the image does not depend on the node format: 64-byte compressed nodes (16-bit scene grid, the default) and
    full-width 128-byte nodes give identical renders and identical hit tables — the compressed boxes are
    conservative, so quantisation can change the work, never the result;
  * closest hit == exhaustive search over all 1M triangles for a sample of rays (oracle brute force, same
    triangle test): the BVH, its compression and the traversal lose no hit;
  * any-hit == (closest hit exists); determinism; strip sharding (N = 8) re-assembles to the single-GPU image;
    batch-size invariance.
"""
import os

import numpy as np
import pytest

import oracle
from helpers import random_rays, rays_to_abi
from take_amd import capi, scenes
from take_amd import cdefs as D
from take_amd.dist import strip_rows

pytestmark = pytest.mark.gpu


def _scene(n_tris, w, h, node_format=None):
    old = os.environ.get("TAKE_HIP_NODES")
    try:
        if node_format is None:
            os.environ.pop("TAKE_HIP_NODES", None)
        else:
            os.environ["TAKE_HIP_NODES"] = node_format  # read once, when the scene is built
        return capi.Scene(scenes.soup_scene(n_tris, w, h, spp=1), precision=D.TAKE_PRECISION_F32)
    finally:
        if old is None:
            os.environ.pop("TAKE_HIP_NODES", None)
        else:
            os.environ["TAKE_HIP_NODES"] = old


@pytest.fixture(scope="module")
def soup1m():
    sc = _scene(1_000_000, 1920, 1080)
    yield sc
    sc.close()


@pytest.mark.skipif(bool(os.environ.get("TAKE_HIP_NODES")),
                    reason="experiment knobs select the node format")
def test_1m_compressed_nodes_in_use(soup1m):
    soup1m.set_instrumentation(timing=False, counting=True)
    soup1m.render(spp=1, max_depth=2, seed=0)
    c = soup1m.counters()
    soup1m.set_instrumentation(False, False)
    assert c["node_bytes"] == 64  # the 15-bit grid is fine enough for 1M triangles: no fall-back to 128-byte nodes
    assert c["samples"] == 1920 * 1080


@pytest.mark.parametrize("fmt", ["wide", "q8"])
def test_1m_node_format_does_not_change_results(soup1m, fmt):
    """full-width 4-wide nodes and the 8-wide compressed tree (TAKE_HIP_NODES=q8: another tree shape, octant-ordered
    slots, its own kernel instances) against the default 4-wide compressed tree: bit-identical images and hit tables"""
    wide = _scene(1_000_000, 1920, 1080, fmt)
    try:
        wide.set_instrumentation(timing=False, counting=True)
        wide.render(spp=1, max_depth=2, seed=0)
        assert wide.counters()["node_bytes"] == 128  # (both alternatives have 128-byte nodes)
        wide.set_instrumentation(False, False)
        a = soup1m.render(spp=1, max_depth=50, seed=5)
        b = wide.render(spp=1, max_depth=50, seed=5)
        assert np.array_equal(a, b)
        rays = rays_to_abi(random_rays(200_000, 23, tmin=1e-4).astype(np.float32).astype(np.float64), 0)
        ha, hb = soup1m.trace_closest(rays), wide.trace_closest(rays)
        for f in ("shape_id", "t", "u", "v"):
            assert np.array_equal(ha[f], hb[f]), f
        assert np.array_equal(soup1m.trace_any(rays), wide.trace_any(rays))
    finally:
        wide.close()


def test_1m_closest_hit_equals_exhaustive_search(soup1m):
    rays = random_rays(256, 31, tmin=1e-4).astype(np.float32).astype(np.float64)
    osc = oracle.OracleScene(soup1m.sd, precision=0)
    want = osc.isect_brute(rays)  # every triangle tested, reference triangle test
    osc.close()
    hits = soup1m.trace_closest(rays_to_abi(rays, 0))
    assert np.array_equal(hits["shape_id"], want[:, 0].astype(np.int32))
    hit = want[:, 0] >= 0
    assert hit.sum() > 100
    for k, col in (("t", 1), ("u", 2), ("v", 3)):
        assert np.array_equal(hits[k][hit].astype(np.float64), want[hit, col]), k
    assert np.array_equal(soup1m.trace_any(rays_to_abi(rays, 0)).astype(bool), hit)


def test_1m_f64_through_compressed_nodes_equals_exhaustive_search():
    """f64 scenes traverse the 64-byte compressed nodes with f32 box tests (conservative: limits rounded outwards) and
    decide hits with the double-precision primitive tests: at 1M triangles closest hits must still equal the
    exhaustive search of the reference triangle test in double, bit for bit, and the image must equal the one rendered
    through the full-width double nodes (TAKE_HIP_NODES=wide)."""
    sd = scenes.soup_scene(1_000_000, 480, 270, spp=1)
    sc = capi.Scene(sd, precision=D.TAKE_PRECISION_F64)
    old = os.environ.get("TAKE_HIP_NODES")
    os.environ["TAKE_HIP_NODES"] = "wide"
    try:
        wide = capi.Scene(sd, precision=D.TAKE_PRECISION_F64)
    finally:
        if old is None:
            os.environ.pop("TAKE_HIP_NODES", None)
        else:
            os.environ["TAKE_HIP_NODES"] = old
    try:
        sc.set_instrumentation(timing=False, counting=True)
        img = sc.render(spp=1, max_depth=50, seed=4)
        assert sc.counters()["node_bytes"] == 64
        sc.set_instrumentation(False, False)
        wide.set_instrumentation(timing=False, counting=True)
        img_w = wide.render(spp=1, max_depth=50, seed=4)
        assert wide.counters()["node_bytes"] == 256
        assert np.array_equal(img, img_w)
        rays = random_rays(256, 37, tmin=1e-7)
        osc = oracle.OracleScene(sd, precision=1)
        want = osc.isect_brute(rays)
        osc.close()
        hits = sc.trace_closest(rays_to_abi(rays, 1))
        assert np.array_equal(hits["shape_id"], want[:, 0].astype(np.int32))
        hit = want[:, 0] >= 0
        assert hit.sum() > 100
        for k, col in (("t", 1), ("u", 2), ("v", 3)):
            assert np.array_equal(hits[k][hit], want[hit, col]), k
        many = rays_to_abi(random_rays(200_000, 41, tmin=1e-7), 1)
        ha, hb = sc.trace_closest(many), wide.trace_closest(many)
        for f in ("shape_id", "t", "u", "v"):
            assert np.array_equal(ha[f], hb[f]), f
        assert np.array_equal(sc.trace_any(many), wide.trace_any(many))
    finally:
        sc.close()
        wide.close()


def test_1m_deterministic_sharded_and_batch_invariant(soup1m):
    full = soup1m.render(spp=2, max_depth=50, seed=9)
    assert np.array_equal(full, soup1m.render(spp=2, max_depth=50, seed=9))
    assert np.array_equal(full, soup1m.render(spp=2, max_depth=50, seed=9, samples_per_batch=1))
    assert np.isfinite(full).all() and full.min() >= 0 and 0.01 < full.mean() < 5.0
    img = np.zeros_like(full)
    for r in range(8):
        rows = strip_rows(1080, r, 8)
        img[rows] = soup1m.render(spp=2, max_depth=50, seed=9, strip_first=r, strip_stride=8)
    assert np.array_equal(img, full)


def test_100k_config_full_resolution_properties():
    sc = _scene(100_000, 1024, 1024)
    wide = _scene(100_000, 1024, 1024, "wide")
    try:
        a = sc.render(spp=2, max_depth=50, seed=2)
        assert np.array_equal(a, sc.render(spp=2, max_depth=50, seed=2, samples_per_batch=1))
        assert np.array_equal(a, wide.render(spp=2, max_depth=50, seed=2))
        img = np.zeros_like(a)
        for r in range(4):
            img[strip_rows(1024, r, 4)] = sc.render(spp=2, max_depth=50, seed=2, strip_first=r, strip_stride=4)
        assert np.array_equal(img, a)
    finally:
        sc.close()
        wide.close()
