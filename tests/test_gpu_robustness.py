"""GPU robustness of the C ABI: failure paths leave a scene usable, and the regimes the default bench reaches
(path state beyond 4 GiB, slot indices beyond 2^25) give the same image as small batches.

  * a render whose workspace cannot be allocated (fault injection at the 1st / 3rd / 6th allocation) returns
    TAKE_E_NOMEM, leaves the scene without a workspace (no stale capacity over null pointers) and the next render on
    the same scene is correct;
  * a 35 M-slot batch (4.5 GB of path state: byte offsets beyond 2^32) is bit-identical to 1-spp batches, on a
    mixed-material scene so that the material sort sees the large queue too;
  * rays with a negative tmin are rejected by the trace hooks (entry distances are ordered as unsigned bit patterns).
"""
import numpy as np
import pytest

from helpers import golden_scene, random_rays, rays_to_abi
from take_amd import capi, scenes
from take_amd import cdefs as D

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nth", [1, 3, 6])
def test_failed_workspace_allocation_leaves_scene_usable(nth, monkeypatch):
    """the nth allocation of the render workspace fails (fault injection, TAKE_HIP_FAIL_ALLOC: a real out-of-memory
    cannot be provoked reliably — the driver over-commits): path state = 1st, a queue = 3rd, the sort keys = 6th"""
    sd = golden_scene("cbox")  # 64 x 64
    sc = capi.Scene(sd, precision=D.TAKE_PRECISION_F32)
    try:
        want = sc.render(spp=2, max_depth=5, seed=4)
        monkeypatch.setenv("TAKE_HIP_FAIL_ALLOC", str(nth))
        with pytest.raises(capi.TakeError) as e:
            sc.render(spp=64, max_depth=5, seed=4, samples_per_batch=64)  # needs a larger workspace than the first render
        assert e.value.code == D.TAKE_E_NOMEM, str(e.value)
        monkeypatch.delenv("TAKE_HIP_FAIL_ALLOC")
        # smaller than the failed request: must allocate afresh, not reuse a stale capacity over released buffers
        got = sc.render(spp=2, max_depth=5, seed=4)
        assert np.array_equal(got, want)
        big = sc.render(spp=64, max_depth=5, seed=4, samples_per_batch=64)
        assert np.array_equal(big, sc.render(spp=64, max_depth=5, seed=4, samples_per_batch=1))
    finally:
        sc.close()


@pytest.mark.parametrize("precision,nth", [(D.TAKE_PRECISION_F32, 1), (D.TAKE_PRECISION_MIXED, 1), (D.TAKE_PRECISION_MIXED, 7)])
def test_unpinned_batch_shrinks_when_the_allocation_fails(precision, nth, monkeypatch):
    """memory taken by someone else between the free-memory query and the allocation (two processes on one device):
    a batch size the caller did not pin is halved until it fits, and the image is the same.  nth = 1: the path
    state; 7 (mixed precision): the f32 records beside the f64 ones."""
    sc = capi.Scene(golden_scene("cbox"), precision=precision)
    try:
        want = sc.render(spp=64, max_depth=5, seed=4, samples_per_batch=1)  # (framebuffer + a 1-spp workspace exist now)
        monkeypatch.setenv("TAKE_HIP_FAIL_ALLOC", str(nth))
        got = sc.render(spp=64, max_depth=5, seed=4)
        monkeypatch.delenv("TAKE_HIP_FAIL_ALLOC")
        assert np.array_equal(got, want)  # (without the retry the injected failure would have surfaced as TAKE_E_NOMEM)
    finally:
        sc.close()


def test_batch_beyond_4gib_of_path_state_is_batch_invariant():
    sd = scenes.soup_scene(100_000, 1920, 1080, spp=17, materials="mixed")
    sc = capi.Scene(sd, precision=D.TAKE_PRECISION_F32)
    try:
        a = sc.render(spp=17, max_depth=12, seed=6, samples_per_batch=17)  # 35.3 M slots x 128 B = 4.5 GB
        b = sc.render(spp=17, max_depth=12, seed=6, samples_per_batch=1)
        assert np.array_equal(a, b)
        assert np.isfinite(a).all() and a.mean() > 0.01
    finally:
        sc.close()


@pytest.mark.parametrize("precision", [D.TAKE_PRECISION_F32, D.TAKE_PRECISION_F64])
def test_negative_tmin_is_rejected(precision):
    sc = capi.Scene(golden_scene("cbox"), precision=precision)
    try:
        rays = random_rays(64, 3)
        rays[7, 6] = -1e-3
        with pytest.raises(capi.TakeError) as e:
            sc.trace_closest(rays_to_abi(rays, precision))
        assert e.value.code == D.TAKE_E_INVALID
        with pytest.raises(capi.TakeError):
            sc.trace_any(rays_to_abi(rays, precision))
        rays[7, 6] = 0.0
        assert len(sc.trace_closest(rays_to_abi(rays, precision))) == 64
    finally:
        sc.close()


def test_device_resident_rays_with_bad_tmin_start_at_zero():
    """take_hip_trace_closest_device cannot look at device-resident rays: tmin = -0.0, a negative tmin and NaN are
    clamped to +0 in the kernel (include/take_hip.h) instead of corrupting the order keys — the hits equal those of
    the same rays with tmin = 0"""
    import torch

    sc = capi.Scene(golden_scene("cbox"), precision=D.TAKE_PRECISION_F32)
    try:
        rays = random_rays(4096, 17)
        rays[:, 6] = 0.0
        want = sc.trace_closest(rays_to_abi(rays, 0))
        assert (want["shape_id"] >= 0).sum() > 2000
        flat = rays_to_abi(rays, 0).copy()  # (n, 8) float32: org3 tmin dir3 tmax
        flat[0::3, 3] = -0.0
        flat[1::3, 3] = -1e-3
        flat[2::3, 3] = np.nan
        d_rays = torch.from_numpy(flat.copy()).cuda()
        d_hits = torch.zeros((len(rays), 4), dtype=torch.float32, device="cuda")
        sc.trace_closest_device(d_rays.data_ptr(), len(rays), d_hits.data_ptr())
        torch.cuda.synchronize()
        got = np.ascontiguousarray(d_hits.cpu().numpy()).view(np.dtype([("shape_id", "<i4"), ("t", "<f4"), ("u", "<f4"), ("v", "<f4")])).reshape(-1)
        for f in ("shape_id", "t", "u", "v"):
            assert np.array_equal(got[f], want[f]), f
    finally:
        sc.close()
