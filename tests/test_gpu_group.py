"""GPU: several GPUs from one process (take_hip_group_*, the C++ host's counterpart of the reference's thread pool,
src/parallel.cpp:183-237).  The GPU box has one device, so the shards are logical — several replicas of the scene on
device 0, each rendering its strips on its own host thread, strips copied with hipMemcpyPeer (device 0 to device 0)
and placed by the assembly kernel: the code path of an 8-GPU node with every device index equal.  The assembled image
must be bit-identical to the one-piece render for every shard count, in both precisions, and through the drop-in binary.
"""
import os

import numpy as np
import pytest

from helpers import GOLD, golden_scene
from take_amd import capi, scenes
from take_amd import cdefs as D

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", [D.TAKE_PRECISION_F32, D.TAKE_PRECISION_F64])
def test_group_render_equals_single_scene(precision):
    sd = golden_scene("mats")  # 64 x 48: 12 strips
    one = capi.Scene(sd, precision=precision)
    want = one.render(spp=4, max_depth=8, seed=3)
    one.close()
    for n in (1, 2, 3, 5, 16):  # 16 > 12 strips: some shards own nothing
        g = capi.SceneGroup(sd, [0] * n, precision=precision)
        try:
            assert g.size() == n
            got = g.render(spp=4, max_depth=8, seed=3)
            assert np.array_equal(got, want), n
            assert np.array_equal(g.render(spp=4, max_depth=8, seed=3), want)  # buffers are reused correctly
        finally:
            g.close()


def test_group_eight_shards_of_the_100k_soup_and_counters():
    sd = scenes.soup_scene(100_000, 640, 360, spp=2, materials="mixed", envmap=(128, 64))
    one = capi.Scene(sd)
    want = one.render(spp=2, max_depth=50, seed=9)
    one.close()
    g = capi.SceneGroup(sd, [0] * 8)
    try:
        got = g.render(spp=2, max_depth=50, seed=9)
        assert np.array_equal(got, want)
        samples = sum(g.counters(k)["samples"] for k in range(8))
        assert samples == 640 * 360 * 2  # the shards partition the image
        rows = [g.counters(k)["samples"] // (640 * 2) for k in range(8)]
        assert max(rows) - min(rows) <= 4  # 90 strips of 4 rows over 8 shards: 11 or 12 strips each
    finally:
        g.close()


def test_group_rejects_bad_arguments():
    sd = golden_scene("cbox")
    with pytest.raises(capi.TakeError) as e:
        capi.SceneGroup(sd, [0, 99])
    assert e.value.code == D.TAKE_E_INVALID
    with pytest.raises(capi.TakeError):
        capi.SceneGroup(sd, [])


def test_dropin_binary_on_four_logical_gpus(tmp_path):
    from test_gpu_dropin import CLI, read_pfm, run_cli

    if not os.path.exists(CLI):
        pytest.skip("oracle/_ref/take_gpu was not built (needs the reference sources: authoring container)")
    xml = os.path.join(GOLD, "scenes", "cbox.xml")
    imgs = []
    for gpus in ("1", "4:0"):
        pfm = str(tmp_path / f"out_{gpus[0]}.pfm")
        r = run_cli(xml, str(tmp_path), 5, {"TAKE_HIP_DUMP_PFM": pfm, "TAKE_HIP_SEED": "7", "TAKE_HIP_GPUS": gpus})
        assert r.returncode == 0, r.stderr[-2000:]
        imgs.append(read_pfm(pfm))
    assert np.array_equal(imgs[0], imgs[1])
