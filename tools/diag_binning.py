"""diagnostic: what spatially sorted rays would buy the closest-hit kernel (L2 locality).  Diffuse-bounce-like rays —
origins on random triangles of the soup, cosine-free random directions — traced in random order, and the SAME rays
sorted by the Morton code of their origin at several grid resolutions.  Kernel time from the library's HIP events."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from take_amd import capi, scenes  # noqa: E402

tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 25
sd = scenes.soup_scene(tris, 64, 64, spp=1)
sc = capi.Scene(sd)
rng = np.random.default_rng(1)
mesh = max(sd.meshes, key=lambda m: m.indices.shape[0])
f = rng.integers(0, mesh.indices.shape[0], n)
v = mesh.positions[mesh.indices[f]]  # (n, 3, 3)
b = rng.dirichlet((1, 1, 1), n)
org = (v * b[:, :, None]).sum(1).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.zeros((n, 8), np.float32)
rays[:, 0:3] = org
rays[:, 3] = 1e-4
rays[:, 4:7] = d
rays[:, 7] = np.inf


def morton(o, bits):
    lo, hi = o.min(0), o.max(0)
    q = np.minimum(((o - lo) / (hi - lo) * (1 << bits)).astype(np.uint32), (1 << bits) - 1)
    code = np.zeros(o.shape[0], np.uint64)
    for i in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> i) & 1).astype(np.uint64) << np.uint64(3 * i + a)
    return code


d_hits = torch.empty(n * 4, dtype=torch.float32, device="cuda")
sc.set_instrumentation(timing=True)


def run(order, label):
    d_rays = torch.from_numpy(np.ascontiguousarray(rays[order])).cuda()
    best = 1e9
    for _ in range(3):
        sc.trace_closest_device(d_rays.data_ptr(), n, d_hits.data_ptr())
        torch.cuda.synchronize()
        best = min(best, sc.counters()["ms_trace_closest"])
    print(f"{label:28s} {best:8.2f} ms  {n / best / 1e6:6.2f} Grays/s", flush=True)
    return best


base = run(np.arange(n), "random order")
for bits in (1, 2, 3, 4):
    t0 = time.time()
    order = np.argsort(morton(org, bits), kind="stable")
    run(order, f"sorted, {1 << bits}^3 cells")
sc.close()
