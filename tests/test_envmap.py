"""Environment-map light (TakeLight kind 2): an EXTENSION — the reference has only a constant background
(SURVEY.md §0), BASELINE configs[2] asks for "env-map IBL importance sampling".  PARITY UNPINNED with respect to the
reference: there is nothing upstream to compare with.  What pins it instead:

  * the oracle's restatement of the extension (oracle/take_oracle.hpp: EnvLight, env_*) — the device code executed
    on the host is bit-identical to it, and the GPU matches it within the same bars as every other render;
  * analytic checks on the oracle and on the GPU: a diffuse plane under a constant sky of radiance c reflects
    rho * c; a constant map gives the same expectation as `background` = c; an empty scene shows the map itself;
  * structural checks: at most one map, bad image index and an all-black map are rejected.
"""
import numpy as np
import pytest

import oracle
from helpers import hostsim_render, rmse
from take_amd import capi, scenes
from take_amd import cdefs as D
from take_amd.scene import SceneData


def env_soup(n=300, res=32, env=(64, 32)):
    return scenes.soup_scene(n, res, res, spp=4, envmap=env)


def plane_under_sky(c, rho, use_env):
    """a big diffuse quad seen from above under a uniform sky; background = c when the map is not used"""
    sd = SceneData(width=24, height=24, lookfrom=(0.0, 2.0, 0.0), lookat=(0.0, 0.0, 0.0), up=(0.0, 0.0, -1.0), vfov=30.0,
                   background=(0.0, 0.0, 0.0) if use_env else (c, c, c), spp=64, max_depth=3)
    m = sd.add_material(D.MAT_DIFFUSE, (rho, rho, rho))
    pos, idx, nrm, uv = scenes._quad((0, 0, 0), (50, 0, 0), (0, 0, -50), (0, 1, 0))
    sd.add_mesh(pos, idx, m, normals=nrm, uvs=uv)
    if use_env:
        sd.add_envmap(np.full((16, 32, 3), c))
    return sd


def oracle_render(sd, precision, spp, max_depth, seed):
    osc = oracle.OracleScene(sd, precision=precision)
    img = osc.render(spp=spp, max_depth=max_depth, rng_mode=oracle.RNG_COUNTER, seed=seed)
    osc.close()
    return img


# ------------------------------------------------------------------ CPU: oracle and the device code on the host
@pytest.mark.parametrize("precision", [1, 0])
def test_device_code_on_host_equals_oracle_with_envmap(precision):
    sd = env_soup()
    want = oracle_render(sd, precision, 4, 6, 5)
    got, _ = hostsim_render(sd, precision, 4, 6, seed=5)
    assert want.mean() > 0.05
    assert np.array_equal(got.astype(np.float64), want)  # same libm on the host: bit for bit


def test_oracle_plane_under_uniform_sky_reflects_rho_c():
    c, rho = 2.0, 0.6
    img = oracle_render(plane_under_sky(c, rho, True), 1, 256, 3, 1)
    assert abs(img.mean() - rho * c) / (rho * c) < 0.02
    # the same expectation without the map: constant background, no light to sample (the reference's own path)
    ref = oracle_render(plane_under_sky(c, rho, False), 1, 256, 3, 1)
    assert abs(ref.mean() - rho * c) / (rho * c) < 0.02


def test_oracle_empty_scene_shows_the_map():
    sd = SceneData(width=16, height=8, lookfrom=(0.0, 0.0, 0.0), lookat=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), vfov=40.0,
                   background=(0.5, 0.5, 0.5), spp=1, max_depth=2)
    sd.add_material(D.MAT_DIFFUSE, (0.5, 0.5, 0.5))
    env = np.zeros((8, 16, 3))
    env[:4] = (0.25, 0.5, 1.0)   # upper hemisphere
    env[4:] = (3.0, 2.0, 1.0)    # lower hemisphere
    sd.add_envmap(env, scale=(2.0, 1.0, 1.0))
    img = oracle_render(sd, 1, 1, 2, 0)
    assert np.array_equal(img[0, 0], [0.5, 0.5, 1.0]) and np.array_equal(img[-1, -1], [6.0, 2.0, 1.0])
    assert np.array_equal(np.unique(img.reshape(-1, 3), axis=0), [[0.5, 0.5, 1.0], [6.0, 2.0, 1.0]])


# ------------------------------------------------------------------ GPU, through the C ABI
@pytest.mark.gpu
def test_gpu_envmap_matches_oracle_f64_and_f32():
    sd = env_soup(1000, 64, (128, 64))
    want = oracle_render(sd, 1, 4, 5, 3)
    sc = capi.Scene(sd, precision=D.TAKE_PRECISION_F64)
    got = sc.render(spp=4, max_depth=5, seed=3)
    sc.close()
    assert rmse(got, want) < 1e-6 and np.median(np.abs(got - want)) < 1e-12  # ocml vs glibc ulps only
    want32 = oracle_render(sd, 0, 8, 50, 3)
    sc = capi.Scene(sd, precision=D.TAKE_PRECISION_F32)
    got32 = sc.render(spp=8, max_depth=50, seed=3)
    sc.close()
    assert rmse(got32, want32) < 1e-3  # the north-star tolerance


@pytest.mark.gpu
def test_gpu_plane_under_uniform_sky_and_constant_map_equals_background():
    c, rho = 2.0, 0.6
    for use_env in (True, False):
        sc = capi.Scene(plane_under_sky(c, rho, use_env))
        img = sc.render(spp=256, max_depth=3, seed=1)
        sc.close()
        assert abs(img.mean() - rho * c) / (rho * c) < 0.02, use_env
    # a whole scene: uniform map vs uniform background agree in expectation (different estimators)
    a = scenes.soup_scene(2000, 48, 48, spp=1)
    a.background = (0.7, 0.7, 0.7)
    b = scenes.soup_scene(2000, 48, 48, spp=1)
    b.add_envmap(np.full((8, 16, 3), 0.7))
    imgs = []
    for sd in (a, b):
        sc = capi.Scene(sd)
        imgs.append(sc.render(spp=512, max_depth=50, seed=2))
        sc.close()
    assert abs(imgs[0].mean() - imgs[1].mean()) / imgs[0].mean() < 0.01


@pytest.mark.gpu
def test_gpu_envmap_scene_is_deterministic_and_builder_independent():
    sd = scenes.soup_scene(100_000, 256, 256, spp=1, envmap=(512, 256))
    a = capi.Scene(sd)
    b = capi.Scene(sd, builder=D.TAKE_BUILDER_DEVICE_LBVH)
    try:
        x = a.render(spp=2, max_depth=50, seed=6)
        assert np.array_equal(x, a.render(spp=2, max_depth=50, seed=6, samples_per_batch=1))
        assert np.array_equal(x, b.render(spp=2, max_depth=50, seed=6))
        assert np.isfinite(x).all() and x.min() >= 0
    finally:
        a.close()
        b.close()


@pytest.mark.gpu
def test_gpu_envmap_validation():
    sd = env_soup()
    sd.add_envmap(np.ones((4, 8, 3)))
    with pytest.raises(capi.TakeError, match="more than one environment-map"):
        capi.Scene(sd)
    sd = scenes.soup_scene(100, 16, 16, spp=1)
    sd.add_envmap(np.zeros((4, 8, 3)))
    with pytest.raises(capi.TakeError, match="no positive luminance"):
        capi.Scene(sd)
