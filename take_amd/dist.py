"""Multi-GPU sharding of the render: one process per GPU, scene replicated, image rows sharded.

The reference's only parallelism is its tile loop (src/render.cpp:59-82, src/parallel.cpp:183-237): tiles write
disjoint pixels and read a const scene.  Here the unit handed to a GPU is a 4-row strip (a quarter of a row of the
reference's 16x16 tiles); rank r of N renders the strips s with s % N == r (interleaved for load balance).
The random stream of a sample depends only on (seed, pixel, sample), so the image is identical for every N.
There is no data-path collective: the only exchange is ONE gather of the finished strips to rank 0
(RCCL over xGMI under backend "nccl"; the same code runs over gloo on CPU tensors in the tests).
"""
import numpy as np
import torch
import torch.distributed as dist

TILE_ROWS = 4  # rows per strip (take_amd/csrc/tk_integrate.h: TILE_ROWS)


def strip_rows(height, first, stride):
    """image rows (increasing, row 0 = top) owned by strip set (first, stride) — the order libtake_hip writes them.
    Strips are counted from the bottom of the image, like the reference's tile rows (y = 0 is the bottom row)."""
    n_strips = (height + TILE_ROWS - 1) // TILE_ROWS
    ys = [y for s in range(first, n_strips, stride) for y in range(s * TILE_ROWS, min(height, (s + 1) * TILE_ROWS))]
    return np.array(sorted(height - 1 - y for y in ys), np.int64)


def max_rows(height, world):
    return max(len(strip_rows(height, r, world)) for r in range(world))


def gather_strips(local, height, rank, world, group=None, dst=0):
    """local: (rows_r, W, 3) tensor of this rank's rows (device or CPU).  Returns the (H, W, 3) image on `dst`,
    None elsewhere.  One collective: a gather of equal-sized (padded) strip buffers."""
    if world == 1:
        return local
    if local.is_cuda and dist.get_backend(group) == "gloo":
        local = local.cpu()  # rehearsal on CPU process groups: gloo has no device gather
    width = local.shape[1]
    m = max_rows(height, world)
    padded = torch.zeros((m, width, 3), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, gather_list=bufs, dst=dst, group=group)
    if rank != dst:
        return None
    full = torch.empty((height, width, 3), dtype=local.dtype, device=local.device)
    for r in range(world):
        rows = torch.as_tensor(strip_rows(height, r, world), device=local.device)
        full[rows] = bufs[r][: rows.numel()]
    return full


def render_sharded(scene, spp, max_depth, seed=0, ray_epsilon=0.0, samples_per_batch=0, group=None, out=None):
    """Render this rank's strips on its GPU (scene: take_amd.capi.Scene on the current device) and gather.
    Returns (image on rank 0 / None, local strip tensor)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rows = strip_rows(scene.sd.height, rank, world)
    tdtype = torch.float64 if scene.dtype == np.float64 else torch.float32
    if out is None:
        out = torch.empty((len(rows), scene.sd.width, 3), dtype=tdtype, device="cuda")
    scene.render_device(out.data_ptr(), spp, max_depth, seed=seed, ray_epsilon=ray_epsilon, strip_first=rank,
                        strip_stride=world, samples_per_batch=samples_per_batch,
                        stream=torch.cuda.current_stream().cuda_stream)
    return gather_strips(out, scene.sd.height, rank, world, group=group), out


def multi_gpu_report(render_ms, gather_ms, group=None):
    """What a multi-GPU bench line needs so that "did N ranks really take part, over which backend" can be read from it:
    backend, world size as the process group reports it, and every rank's own render / gather time (all-gathered:
    the list has one entry per rank that answered).  Works on any backend (gloo in the CPU tests, nccl = RCCL on GPUs)."""
    if not dist.is_initialized():
        return {"backend": None, "world_size": 1, "per_rank_render_ms": [float(render_ms)], "per_rank_gather_ms": [float(gather_ms)]}
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = "cuda" if backend == "nccl" else "cpu"
    mine = torch.tensor([float(render_ms), float(gather_ms)], dtype=torch.float64, device=dev)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine, group=group)
    return {"backend": backend, "world_size": world,
            "per_rank_render_ms": [float(t[0]) for t in every], "per_rank_gather_ms": [float(t[1]) for t in every]}
