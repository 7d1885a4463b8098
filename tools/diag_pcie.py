"""diagnostic: the PCIe-inclusive rate of the default workload.  bench.py times take_hip_render_device (image left in
HBM); the reference-shaped entry point take_hip_render hands the image back in a host buffer.  Same render, both ways,
plus the one-off scene_create (host arrays -> HBM + BVH)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from take_amd import capi, scenes  # noqa: E402
from take_amd import cdefs as D  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sd = scenes.soup_scene(1_000_000, 1920, 1080, spp=spp, envmap=(2048, 1024))
t0 = time.time()
sc = capi.Scene(sd, precision=D.TAKE_PRECISION_MIXED, builder=D.TAKE_BUILDER_HOST_SAH)
t_create = time.time() - t0
out = torch.empty((1080, 1920, 3), dtype=torch.float64, device="cuda")
sc.render_device(out.data_ptr(), spp, 50, seed=0)  # warm-up (workspace allocation)
torch.cuda.synchronize()
res = {}
for label in ("device", "host", "device", "host"):
    t0 = time.time()
    if label == "device":
        sc.render_device(out.data_ptr(), spp, 50, seed=0)
        torch.cuda.synchronize()
    else:
        img = sc.render(spp=spp, max_depth=50, seed=0)
    res.setdefault(label, []).append(time.time() - t0)
n = 1920 * 1080 * spp
d, h = min(res["device"]), min(res["host"])
print(f"scene_create (host SAH, f64 + f32 sides): {t_create:.2f} s")
print(f"render, image left in HBM : {d * 1e3:9.1f} ms = {n / d / 1e6:.2f} Msamples/s")
print(f"render, image to the host : {h * 1e3:9.1f} ms = {n / h / 1e6:.2f} Msamples/s  (+{(h - d) * 1e3:.1f} ms for {out.numel() * 8 / 1e6:.0f} MB)")
print(f"one cold frame (scene_create + render to host): {n / (t_create + h) / 1e6:.2f} Msamples/s")
assert np.array_equal(img, out.cpu().numpy())
sc.close()
