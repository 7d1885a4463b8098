"""Progressive rendering (take_hip_render_accumulate): the accumulator stays in HBM between calls and
the samples continue the one-shot render's numbering, so any split of N samples into calls gives the one-shot image of
N samples BIT FOR BIT (same random streams, same order of the per-pixel additions, src/render.cpp:68-78)."""
import numpy as np
import pytest
import torch

from helpers import golden_scene
from take_amd import capi
from take_amd import cdefs as D

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", [D.TAKE_PRECISION_F64, D.TAKE_PRECISION_F32])
def test_accumulated_calls_equal_the_one_shot_render(precision):
    sd = golden_scene("mats")
    sc = capi.Scene(sd, precision=precision)
    tdtype = torch.float64 if precision == D.TAKE_PRECISION_F64 else torch.float32
    try:
        want = sc.render(spp=12, max_depth=6, seed=9)
        out = torch.zeros((sd.height, sd.width, 3), dtype=tdtype, device="cuda")
        means = []
        for k, more in enumerate((5, 4, 3)):
            n = sc.render_accumulate(out.data_ptr(), more, 6, seed=9, restart=(k == 0), samples_per_batch=2 if k == 1 else 0)
            means.append(out.cpu().numpy().copy())
        assert n == 12
        assert np.array_equal(means[-1], want)
        # the intermediate images are the one-shot renders of 5 and 9 samples
        assert np.array_equal(means[0], sc.render(spp=5, max_depth=6, seed=9))
        # (a one-shot render ended the sequence: the next accumulate call starts from zero even without restart)
        assert sc.render_accumulate(out.data_ptr(), 9, 6, seed=9) == 9
        assert np.array_equal(out.cpu().numpy(), means[1])
        # options that change within a sequence are refused; restart accepts them
        with pytest.raises(capi.TakeError) as e:
            sc.render_accumulate(out.data_ptr(), 1, 6, seed=10)
        assert e.value.code == D.TAKE_E_INVALID
        assert sc.render_accumulate(out.data_ptr(), 2, 6, seed=10, restart=True) == 2
        assert np.array_equal(out.cpu().numpy(), sc.render(spp=2, max_depth=6, seed=10))
    finally:
        sc.close()


def test_accumulate_on_a_strip_set():
    sd = golden_scene("cbox")
    sc = capi.Scene(sd, precision=D.TAKE_PRECISION_F32)
    try:
        want = sc.render(spp=6, max_depth=4, seed=2, strip_first=1, strip_stride=3)
        out = torch.zeros(want.shape, dtype=torch.float32, device="cuda")
        sc.render_accumulate(out.data_ptr(), 2, 4, seed=2, restart=True, strip_first=1, strip_stride=3)
        sc.render_accumulate(out.data_ptr(), 4, 4, seed=2, strip_first=1, strip_stride=3)
        assert np.array_equal(out.cpu().numpy(), want)
    finally:
        sc.close()
