"""The product's device code, executed on the host (tests/hostsim = the same tk_*.h headers the HIP kernels
call, driven by serial loops in kernel order), against the oracle.

This is what can be verified without a GPU: scene preparation, the SAH/4-wide BVH, the traversal, the
integrator rounds, random-stream consumption, strip sharding, batching.  On the host both sides use the same
libm, so the bar is BIT equality (the GPU tests, which add ocml's libm and the real queues, carry the
toleranced comparisons)."""
import numpy as np
import pytest

import oracle
from helpers import GOLDEN_SCENES, golden_scene, hostsim_render, hostsim_trace, n_local_rows, random_rays
from take_amd import scenes
from take_amd.dist import strip_rows


@pytest.mark.parametrize("name", GOLDEN_SCENES)
@pytest.mark.parametrize("precision", [1, 0])
def test_wide_bvh_closest_hit_equals_exhaustive_search(name, precision):
    sd = golden_scene(name)
    rays = random_rays(3000, 5, tmin=1e-7)
    if precision == 0:
        rays = rays.astype(np.float32).astype(np.float64)
    osc = oracle.OracleScene(sd, precision=precision)
    want = osc.isect_brute(rays)
    osc.close()
    got = hostsim_trace(sd, precision, rays).astype(np.float64)
    assert (got[:, 0] == want[:, 0]).all()
    hit = want[:, 0] >= 0
    assert hit.sum() > 1000
    assert np.array_equal(got[hit, 1:4], want[hit, 1:4])  # t, u, v bit for bit


@pytest.mark.parametrize("precision", [1, 0])
def test_eight_wide_tree_equals_exhaustive_search(precision, monkeypatch):
    """TAKE_HIP_NODES=q8: the 8-wide compressed tree (octant-ordered slots, nearest child first, the others in
    octant order) finds the same closest hits and the same occlusion as the exhaustive search — on a soup deep enough
    for several 8-wide levels and on two golden scenes"""
    monkeypatch.setenv("TAKE_HIP_NODES", "q8")
    for sd in (scenes.soup_scene(20_000, 32, 32, spp=1), golden_scene("mats"), golden_scene("soup1k")):
        rays = random_rays(2000, 9, bounded_fraction=0.5, tmin=1e-7)
        if precision == 0:
            rays = rays.astype(np.float32).astype(np.float64)
        osc = oracle.OracleScene(sd, precision=precision)
        want = osc.isect_brute(rays)
        osc.close()
        got = hostsim_trace(sd, precision, rays).astype(np.float64)
        assert (got[:, 0] == want[:, 0]).all()
        hit = want[:, 0] >= 0
        assert hit.sum() > 500
        assert np.array_equal(got[hit, 1:4], want[hit, 1:4])
        occ = hostsim_trace(sd, precision, rays, any_hit=True)[:, 0] >= 0
        assert np.array_equal(occ, hit)


@pytest.mark.parametrize("name", GOLDEN_SCENES)
def test_any_hit_equals_closest_hit_boolean(name):
    sd = golden_scene(name)
    rays = random_rays(3000, 6, bounded_fraction=0.8, tmin=1e-7)
    osc = oracle.OracleScene(sd, precision=1)
    want = osc.isect(rays)[:, 15]
    osc.close()
    got = hostsim_trace(sd, 1, rays, any_hit=True)[:, 0] >= 0
    assert np.array_equal(got, want.astype(bool))


@pytest.mark.parametrize("name", GOLDEN_SCENES)
@pytest.mark.parametrize("precision", [1, 0])
def test_wavefront_render_equals_oracle(name, precision):
    sd = golden_scene(name)
    osc = oracle.OracleScene(sd, precision=precision)
    for depth in (1, 50):
        want = osc.render(2, depth, rng_mode=oracle.RNG_COUNTER, seed=11, threads=8)
        got, _ = hostsim_render(sd, precision, 2, depth, seed=11)
        assert np.array_equal(got.astype(np.float64), want), f"{name} depth {depth}"
    osc.close()


def test_strip_sharding_reassembles_the_full_image():
    sd = golden_scene("mats")  # 64 x 48: three strips
    full, _ = hostsim_render(sd, 0, 2, 5, seed=3)
    for world in (2, 3, 4):
        img = np.zeros_like(full)
        total = 0
        for r in range(world):
            part, _ = hostsim_render(sd, 0, 2, 5, seed=3, strip_first=r, strip_stride=world)
            rows = strip_rows(sd.height, r, world)
            assert part.shape[0] == len(rows) == n_local_rows(sd.height, r, world)
            img[rows] = part
            total += len(rows)
        assert total == sd.height
        assert np.array_equal(img, full)


def test_batching_does_not_change_the_image():
    sd = golden_scene("cbox")
    a, _ = hostsim_render(sd, 0, 4, 5, seed=9, samples_per_batch=4)
    b, _ = hostsim_render(sd, 0, 4, 5, seed=9, samples_per_batch=1)
    c, _ = hostsim_render(sd, 0, 4, 5, seed=9, samples_per_batch=3)
    assert np.array_equal(a, b) and np.array_equal(a, c)


def test_ragged_image_sizes():
    # width / height not multiples of the tile or the wave
    sd = golden_scene("cbox")
    sd.width, sd.height = 37, 21
    osc = oracle.OracleScene(sd, precision=1)
    want = osc.render(1, 3, rng_mode=oracle.RNG_COUNTER, seed=2, threads=4)
    osc.close()
    got, _ = hostsim_render(sd, 1, 1, 3, seed=2)
    assert np.array_equal(got, want)


def test_max_depth_minus_one_and_zero():
    sd = golden_scene("cbox")
    osc = oracle.OracleScene(sd, precision=1)
    for depth in (-1, 0):
        want = osc.render(1, depth, rng_mode=oracle.RNG_COUNTER, seed=4, threads=4)
        got, _ = hostsim_render(sd, 1, 1, depth, seed=4)
        assert np.array_equal(got, want)
    osc.close()


def test_procedural_soup_scene_matches_oracle():
    sd = scenes.soup_scene(3000, 48, 32, spp=1, seed=7, jitter=0.05)
    osc = oracle.OracleScene(sd, precision=0)
    want = osc.render(1, 50, seed=1, threads=8)
    osc.close()
    got, st = hostsim_render(sd, 0, 1, 50, seed=1)
    assert np.array_equal(got.astype(np.float64), want)
    assert st["max_stack"] <= 3 * st["bvh_depth"] + 1


def test_emissive_mesh_without_normals_is_rejected():
    # the reference throws std::out_of_range when it samples such a light (src/shape.cpp:163-165)
    sd = scenes.soup_scene(10, 16, 16, spp=1)
    pos, idx = scenes.soup_triangles(4, 3)
    sd.add_mesh(pos, idx, 0, emission=(1, 1, 1))
    with pytest.raises(RuntimeError, match="no vertex normals"):
        hostsim_render(sd, 0, 1, 1)


# ------------------------------------------------------------------ compressed BVH nodes (15-bit scene grid)
NODE_FORMATS = [("", 4), ("q8", 8)]  # TAKE_HIP_NODES value -> node width (default: the 4-wide tree)


def _check_qnodes(sd, fmt=""):
    import ctypes as C
    import os

    from helpers import hostsim

    out = (C.c_int64 * 8)()
    desc, keep = sd.to_desc()
    old = os.environ.get("TAKE_HIP_NODES")
    try:
        if fmt:
            os.environ["TAKE_HIP_NODES"] = fmt
        else:
            os.environ.pop("TAKE_HIP_NODES", None)
        rc = hostsim().hostsim_check_qnodes(C.byref(desc), out)
    finally:
        if old is None:
            os.environ.pop("TAKE_HIP_NODES", None)
        else:
            os.environ["TAKE_HIP_NODES"] = old
    assert rc == 0, hostsim().hostsim_last_error()
    return list(out)[:6]


@pytest.mark.parametrize("fmt,width", NODE_FORMATS)
@pytest.mark.parametrize("name", GOLDEN_SCENES)
def test_compressed_nodes_contain_the_true_boxes(name, fmt, width):
    """every compressed child box (exact arithmetic) contains the true box plus the builder's slack, child words
    are unchanged, and the f32 golden scenes all use the compressed format — for the 4-wide tree (default) and the
    8-wide one"""
    slots, bad, diff, infl, in_use, w = _check_qnodes(golden_scene(name), fmt)
    assert bad == 0 and diff == 0
    if slots:
        assert in_use == 1 and 1.0 <= infl / 1e6 < 1.10 and w == width


@pytest.mark.parametrize("fmt,width", NODE_FORMATS)
def test_compressed_nodes_on_a_large_soup_and_scale_mixing_fallback(fmt, width):
    slots, bad, diff, infl, in_use, w = _check_qnodes(scenes.soup_scene(200_000, 64, 64, spp=1), fmt)
    assert slots > 100_000 and bad == 0 and diff == 0 and in_use == 1 and w == width
    assert infl / 1e6 < 1.02  # a 15-bit cell is far below a soup triangle
    # a scene mixing scales by 1e6 (tiny triangles next to a huge one): the 15-bit grid would inflate the small
    # boxes many times over, the builder must keep the full-width (4-wide) nodes
    rng = np.random.default_rng(3)
    sd = scenes.soup_scene(64, 32, 32, spp=1)
    tiny = (rng.uniform(-1, 1, (2000, 1, 3)) * 1e-3 + rng.uniform(-1, 1, (2000, 3, 3)) * 1e-6).astype(np.float64)
    sd.add_mesh(tiny.reshape(-1, 3), np.arange(6000, dtype=np.int32).reshape(-1, 3), 0)
    big = np.array([[-1e3, -1e3, -5.0], [1e3, -1e3, -5.0], [0.0, 1e3, -5.0]])
    sd.add_mesh(big, np.array([[0, 1, 2]], np.int32), 0)
    slots, bad, diff, infl, in_use, w = _check_qnodes(sd, fmt)
    assert in_use == 0 and infl / 1e6 > 1.10 and w == 4


def test_slot_to_pixel_division_by_reciprocal_is_exact():
    """every shade round turns a path slot into (sample, row, column) with two divisions by launch constants; the device
    code multiplies by their reciprocals and corrects by one (tk_integrate.h::divmod_u31) — checked against the integer
    division on ~100k edge pairs around the powers of two and 5 M random pairs below 2^31"""
    import ctypes as C

    from helpers import hostsim

    L = hostsim()
    L.hostsim_check_divmod.argtypes = [C.c_int64, C.c_uint64]
    L.hostsim_check_divmod.restype = C.c_int64
    assert L.hostsim_check_divmod(5_000_000, 12345) == 0
