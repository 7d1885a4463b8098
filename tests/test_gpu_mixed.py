"""Mixed precision (TAKE_PRECISION_MIXED): the first `exact_bounces` rounds of every path in the reference's arithmetic
(double, on the f64 scene), the surviving paths' records converted to float and finished on the f32 scene.

  * with exact_bounces >= the number of rounds the render IS the f64 render: bit-identical images;
  * the error against the f64 image falls with every exact bounce (a flipped hit / miss decision costs what the path
    still carries), and with the default three it is well inside the f32 path's;
  * the usual invariances hold (determinism, batch size, strip sharding, progressive accumulation, scene groups).
The reference has one arithmetic; this mode is specified against the repo's own f64 path (itself at rounding level
of the pinned oracle, tests/test_gpu_parity.py)."""
import numpy as np
import pytest
import torch

from helpers import golden_scene, rmse
from take_amd import capi, scenes
from take_amd import cdefs as D
from take_amd.dist import strip_rows

pytestmark = pytest.mark.gpu


def _render(sd, precision, spp, depth, seed, exact=0, **kw):
    sc = capi.Scene(sd, precision=precision)
    sc.exact_bounces = exact
    try:
        return sc.render(spp=spp, max_depth=depth, seed=seed, **kw)
    finally:
        sc.close()


@pytest.mark.parametrize("name", ["cbox", "mats", "meshlight"])
def test_all_rounds_exact_is_the_f64_render(name):
    sd = golden_scene(name)
    want = _render(sd, D.TAKE_PRECISION_F64, 4, 6, 3)
    got = _render(sd, D.TAKE_PRECISION_MIXED, 4, 6, 3, exact=8)  # max_depth 6 -> 8 rounds
    assert got.dtype == np.float64 and np.array_equal(got, want)


def test_error_falls_with_the_exact_bounces_on_a_soup():
    sd = scenes.soup_scene(100_000, 640, 360, spp=16, envmap=(512, 256))
    ref = _render(sd, D.TAKE_PRECISION_F64, 16, 50, 1)
    e32 = rmse(_render(sd, D.TAKE_PRECISION_F32, 16, 50, 1), ref)
    errs = [rmse(_render(sd, D.TAKE_PRECISION_MIXED, 16, 50, 1, exact=k), ref) for k in (1, 3, 6)]
    print("f32", e32, "mixed", errs)
    assert errs[0] < e32 and errs[1] < 0.6 * e32 and errs[2] < 0.35 * e32, (e32, errs)
    assert errs[0] > errs[1] > errs[2] > 0
    # the default is three exact bounces
    assert np.array_equal(_render(sd, D.TAKE_PRECISION_MIXED, 16, 50, 1), _render(sd, D.TAKE_PRECISION_MIXED, 16, 50, 1, exact=3))


def test_mixed_invariances():
    sd = golden_scene("mats")  # 64 x 48, every reference material tag: the material sort runs in both halves
    sc = capi.Scene(sd, precision=D.TAKE_PRECISION_MIXED)
    try:
        a = sc.render(spp=6, max_depth=10, seed=4)
        assert np.isfinite(a).all() and a.mean() > 0.01
        assert np.array_equal(a, sc.render(spp=6, max_depth=10, seed=4))
        assert np.array_equal(a, sc.render(spp=6, max_depth=10, seed=4, samples_per_batch=1))
        img = np.zeros_like(a)
        for r in range(3):
            img[strip_rows(sd.height, r, 3)] = sc.render(spp=6, max_depth=10, seed=4, strip_first=r, strip_stride=3)
        assert np.array_equal(img, a)
        out = torch.zeros((sd.height, sd.width, 3), dtype=torch.float64, device="cuda")
        sc.render_accumulate(out.data_ptr(), 4, 10, seed=4, restart=True)
        assert sc.render_accumulate(out.data_ptr(), 2, 10, seed=4) == 6
        assert np.array_equal(out.cpu().numpy(), a)
        # integrators 1..3 are refused (the mixed path is the reference's path_tracing)
        with pytest.raises(capi.TakeError):
            sc.render(spp=1, max_depth=3, seed=1, integrator=2)
        # the trace hooks of a mixed scene are the f64 scene's
        rays = np.zeros((4, 8))
        rays[:, 2], rays[:, 5], rays[:, 7] = 3.0, -1.0, np.inf
        assert sc.trace_closest(np.concatenate([rays[:, 0:3], rays[:, 6:7], rays[:, 3:6], rays[:, 7:8]], axis=1))["t"].dtype == np.float64
    finally:
        sc.close()
    f64 = _render(sd, D.TAKE_PRECISION_F64, 6, 10, 4)
    assert rmse(a, f64) < 5e-3  # (bounded radiance, 6 spp: f32 alone is at 4.5e-3 on this scene, tests/test_gpu_precision.py)


def test_mixed_scene_group_equals_single_scene():
    sd = golden_scene("cbox")
    want = _render(sd, D.TAKE_PRECISION_MIXED, 4, 8, 2)
    g = capi.SceneGroup(sd, [0, 0, 0], precision=D.TAKE_PRECISION_MIXED)
    try:
        got = g.render(spp=4, max_depth=8, seed=2)
    finally:
        g.close()
    assert np.array_equal(got, want)


def test_mixed_on_an_instanced_scene():
    """two-level scenes (TakeInstance): the conversion hands over rays, not hits, so the f32 rounds traverse the f32
    twin of the same two-level tree.  All rounds exact = the f64 render; the default = closer to it than the f32 path."""
    sd = scenes.instanced_scene(60, 400, 160, 96, spp=8)
    ref = _render(sd, D.TAKE_PRECISION_F64, 8, 12, 5)
    assert np.array_equal(_render(sd, D.TAKE_PRECISION_MIXED, 8, 12, 5, exact=14), ref)
    e32 = rmse(_render(sd, D.TAKE_PRECISION_F32, 8, 12, 5), ref)
    emx = rmse(_render(sd, D.TAKE_PRECISION_MIXED, 8, 12, 5), ref)
    print("instanced: f32", e32, "mixed", emx)
    assert 0 < emx < 0.7 * e32
