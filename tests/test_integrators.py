"""The reference's other integrators (src/integrator/path_tracing.h:114 raw, :161 one-sample MIS, :274 one-sample MIS
with power-based light picking — defined upstream, called by nothing), selectable through TakeRenderOpts.integrator.

Chain of evidence: the oracle's restatements are bit-exact against the COMPILED REFERENCE on 21 tables
(tests/test_oracle_golden.py::test_integrator_variants_bit_exact; the power tables Scene::lights_power_pmf/_cdf,
which the reference never fills, come from its own light_power()); here the device code is compared with the oracle:
bit for bit when executed on the host (same libm), within the usual bars on the GPU.
"""
import numpy as np
import pytest

import oracle
from helpers import GOLDEN_SCENES, golden_scene, hostsim_render, rmse
from take_amd import cdefs as D
from take_amd.scene import Light

INTEGRATORS = [D.INTEGRATOR_RAW, D.INTEGRATOR_ONE_SAMPLE_MIS, D.INTEGRATOR_ONE_SAMPLE_MIS_POWER]


def with_point_light(name="cbox"):
    """a PointLight in the list (counted by the uniform pick, power 0): when it is picked the reference's iteration
    does nothing (`get_if<DiffuseAreaLight>` fails) and the loop index advances without a ray"""
    sd = golden_scene(name)
    sd.lights.append(Light(0, -1, (5.0, 5.0, 5.0), (0.0, 0.5, 0.0)))
    return sd


@pytest.mark.parametrize("name", GOLDEN_SCENES)
@pytest.mark.parametrize("precision", [1, 0])
@pytest.mark.parametrize("integrator", INTEGRATORS)
def test_device_code_on_host_equals_oracle(name, precision, integrator):
    sd = golden_scene(name)
    osc = oracle.OracleScene(sd, precision=precision)
    for depth in (0, 4, 50):
        want = osc.render(2, depth, seed=5, integrator=integrator)
        got, _ = hostsim_render(sd, precision, 2, depth, seed=5, integrator=integrator)
        assert np.array_equal(got.astype(np.float64), want), f"{name} depth {depth}"
    osc.close()


@pytest.mark.parametrize("integrator", [0] + INTEGRATORS)
def test_point_light_iterations_do_nothing(integrator):
    sd = with_point_light()
    osc = oracle.OracleScene(sd, precision=1)
    want = osc.render(4, 12, seed=2, integrator=integrator)
    osc.close()
    got, _ = hostsim_render(sd, 1, 4, 12, seed=2, integrator=integrator)
    assert np.array_equal(got, want)
    assert np.isfinite(want).all() and want.mean() > 0.01


def test_integrators_agree_in_expectation():
    """four estimators of the same integral (the raw one without NEE is the noisiest): image means within a few %"""
    sd = golden_scene("cbox")
    osc = oracle.OracleScene(sd, precision=1)
    means = [osc.render(64, 8, seed=1, integrator=i).mean() for i in (0, 1, 2, 3)]
    osc.close()
    assert max(means) / min(means) < 1.08, means


# ------------------------------------------------------------------ GPU, through the C ABI
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cbox", "mats", "meshlight"])
@pytest.mark.parametrize("integrator", INTEGRATORS)
def test_gpu_integrators_match_oracle(name, integrator):
    from take_amd import capi

    sd = golden_scene(name)
    osc = oracle.OracleScene(sd, precision=1)
    sc = capi.Scene(sd, precision=D.TAKE_PRECISION_F64)
    try:
        want = osc.render(4, 5, seed=7, integrator=integrator)
        got = sc.render(spp=4, max_depth=5, seed=7, integrator=integrator)
        d = np.abs(got - want).max(axis=2)
        assert np.median(d) < 1e-12 and (d < 1e-9).mean() >= 0.995, (name, integrator, (d < 1e-9).mean())
        want = osc.render(4, 50, seed=7, integrator=integrator)
        got = sc.render(spp=4, max_depth=50, seed=7, integrator=integrator)
        d = np.abs(got - want).max(axis=2)
        assert np.median(d) < 1e-12 and (d < 1e-9).mean() >= 0.99
    finally:
        osc.close()
        sc.close()
    o32 = oracle.OracleScene(sd, precision=0)
    s32 = capi.Scene(sd, precision=D.TAKE_PRECISION_F32)
    try:
        want = o32.render(16, 50, seed=7, integrator=integrator)
        got = s32.render(spp=16, max_depth=50, seed=7, integrator=integrator)
        assert rmse(got, want) < 2e-3, rmse(got, want)
        # sharded and batched like the default integrator
        assert np.array_equal(got, s32.render(spp=16, max_depth=50, seed=7, integrator=integrator, samples_per_batch=3))
    finally:
        o32.close()
        s32.close()


@pytest.mark.gpu
def test_gpu_point_light_and_envmap_rules():
    from take_amd import capi, scenes

    sd = with_point_light()
    osc = oracle.OracleScene(sd, precision=1)
    sc = capi.Scene(sd, precision=D.TAKE_PRECISION_F64)
    try:
        for integrator in (0, 1, 2, 3):
            want = osc.render(4, 6, seed=3, integrator=integrator)
            got = sc.render(spp=4, max_depth=6, seed=3, integrator=integrator)
            d = np.abs(got - want).max(axis=2)
            assert (d < 1e-9).mean() >= 0.995, integrator
    finally:
        osc.close()
        sc.close()
    env = capi.Scene(scenes.soup_scene(300, 32, 32, spp=1, envmap=(64, 32)))
    try:
        with pytest.raises(capi.TakeError) as e:
            env.render(spp=1, max_depth=2, integrator=D.INTEGRATOR_ONE_SAMPLE_MIS)
        assert e.value.code == D.TAKE_E_INVALID
        with pytest.raises(capi.TakeError):
            env.render(spp=1, max_depth=2, integrator=7)
    finally:
        env.close()
