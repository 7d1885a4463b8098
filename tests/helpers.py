"""Shared test helpers: golden scene loading, ray generators, the hostsim wrapper (tests/hostsim)."""
import ctypes as C
import os
import subprocess

import numpy as np

from take_amd import cdefs as D
from take_amd.scene import load_tkscene

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
GOLDEN_SCENES = ["cbox", "mats", "soup1k", "spherelight", "meshlight"]


def golden_scene(name):
    return load_tkscene(os.path.join(GOLD, "scenes", name + ".tkscene"))


_HOSTSIM = None


def hostsim():
    global _HOSTSIM
    if _HOSTSIM is None:
        d = os.path.join(HERE, "hostsim")
        subprocess.run(["make", "-C", d], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        L = C.CDLL(os.path.join(d, "libhostsim.so"))
        L.hostsim_last_error.restype = C.c_char_p
        L.hostsim_render.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
        L.hostsim_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        L.hostsim_check_qnodes.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        _HOSTSIM = L
    return _HOSTSIM


def render_opts(spp, max_depth, seed=0, ray_epsilon=0.0, strip_first=0, strip_stride=1, samples_per_batch=0,
                integrator=0):
    o = D.TakeRenderOpts()
    o.spp, o.max_depth, o.seed, o.ray_epsilon = spp, max_depth, seed, ray_epsilon
    o.strip_first, o.strip_stride, o.samples_per_batch = strip_first, strip_stride, samples_per_batch
    o.integrator = integrator
    return o


def n_local_rows(height, first, stride):
    from take_amd.dist import TILE_ROWS as T

    n_strips = (height + T - 1) // T
    return sum(min(height, (s + 1) * T) - s * T for s in range(first, n_strips, stride))


def hostsim_render(sd, precision, spp, max_depth, seed=0, ray_epsilon=0.0, strip_first=0, strip_stride=1,
                   samples_per_batch=0, integrator=0):
    desc, keep = sd.to_desc()
    o = render_opts(spp, max_depth, seed, ray_epsilon, strip_first, strip_stride, samples_per_batch, integrator)
    rows = n_local_rows(sd.height, strip_first, strip_stride)
    out = np.zeros((rows, sd.width, 3), np.float64 if precision == 1 else np.float32)
    stats = (C.c_uint64 * 7)()
    rc = hostsim().hostsim_render(C.addressof(desc), precision, C.addressof(o), out.ctypes.data, stats)
    if rc != 0:
        raise RuntimeError(hostsim().hostsim_last_error().decode())
    keys = ["closest", "shadow", "nodes", "prims", "max_stack", "bvh_nodes", "bvh_depth"]
    return out, dict(zip(keys, [int(x) for x in stats]))


def rays_to_abi(rays8, precision):
    """(n,8) org3 dir3 tmin tmax -> TakeRayF/D memory layout (org3 tmin dir3 tmax)"""
    r = np.asarray(rays8, np.float64)
    a = np.concatenate([r[:, 0:3], r[:, 6:7], r[:, 3:6], r[:, 7:8]], axis=1)
    return np.ascontiguousarray(a, np.float64 if precision == 1 else np.float32)


def hostsim_trace(sd, precision, rays8, any_hit=False):
    desc, keep = sd.to_desc()
    a = rays_to_abi(rays8, precision)
    hits = np.zeros((a.shape[0], 4), a.dtype)
    rc = hostsim().hostsim_trace(C.addressof(desc), precision, a.ctypes.data, a.shape[0], hits.ctypes.data, int(any_hit))
    if rc != 0:
        raise RuntimeError(hostsim().hostsim_last_error().decode())
    return hits


def random_rays(n, seed, camera_fraction=0.25, bounded_fraction=0.3, tmin=1e-4):
    """rays inside the [-1,1]^3 box of the golden scenes, a share of them from the camera position"""
    rng = np.random.default_rng(seed)
    o = rng.uniform(-0.95, 0.95, (n, 3))
    d = rng.normal(size=(n, 3))
    k = int(n * camera_fraction)
    o[:k] = np.array([0, 0, 3.9])
    d[:k] = rng.uniform(-0.3, 0.3, (k, 3)) + np.array([0, 0, -1.0])
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tmax = np.where(rng.uniform(size=(n, 1)) < bounded_fraction, rng.uniform(0.05, 2, (n, 1)), np.inf)
    return np.hstack([o, d, np.full((n, 1), tmin), tmax])


def rmse(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)))
