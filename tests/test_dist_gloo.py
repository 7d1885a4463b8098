"""The N>1 path on CPU: world_size-2 (and 3) gloo process groups run take_amd.dist.gather_strips — the one
collective of the multi-GPU render — on strips rendered by tests/hostsim, and rank 0 must hold exactly the
single-process image."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    from helpers import golden_scene, hostsim_render
    from take_amd.dist import gather_strips, strip_rows

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sd = golden_scene("mats")  # 64 x 48 -> 12 strips of 4 rows
        part, _ = hostsim_render(sd, 0, 1, 3, seed=21, strip_first=rank, strip_stride=world)
        assert part.shape[0] == len(strip_rows(sd.height, rank, world))
        full = gather_strips(torch.from_numpy(part), sd.height, rank, world)
        dist.barrier()
        if rank == 0:
            q.put(full.numpy())
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 5])  # 12 strips: even over 2 and 3 ranks, ragged (3,3,2,2,2) over 5
def test_gather_strips_over_gloo(world):
    from helpers import golden_scene, hostsim_render

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    want, _ = hostsim_render(golden_scene("mats"), 0, 1, 3, seed=21)
    assert np.array_equal(got, want)


def _report_worker(rank, world, port, q):
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    from take_amd.dist import multi_gpu_report

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rep = multi_gpu_report(100.0 + rank, 1.5 * (rank + 1))
        dist.barrier()
        if rank == 0:
            q.put(rep)
    finally:
        dist.destroy_process_group()


def test_multi_gpu_report_fields_over_gloo():
    """the N > 1 bench line's `multi_gpu` object: backend, world size from the process group, one render and one
    gather time per rank"""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_report_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    rep = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert rep["backend"] == "gloo" and rep["world_size"] == world
    assert rep["per_rank_render_ms"] == [100.0, 101.0, 102.0]
    assert rep["per_rank_gather_ms"] == [1.5, 3.0, 4.5]


def test_multi_gpu_report_without_a_process_group():
    from take_amd.dist import multi_gpu_report

    rep = multi_gpu_report(5.0, 0.0)
    assert rep == {"backend": None, "world_size": 1, "per_rank_render_ms": [5.0], "per_rank_gather_ms": [0.0]}
