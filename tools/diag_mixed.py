"""mixed precision: RMSE against the f64 render and throughput as a function of the number of exact bounces"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from take_amd import capi, scenes
from take_amd import cdefs as D

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tris = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
sd = scenes.soup_scene(tris, 1920, 1080, spp=spp, envmap=(2048, 1024))


def run(precision, exact=0):
    sc = capi.Scene(sd, precision=precision)
    sc.exact_bounces = exact
    tdt = torch.float32 if precision == D.TAKE_PRECISION_F32 else torch.float64
    out = torch.zeros((1080, 1920, 3), dtype=tdt, device="cuda")
    sc.render_device(out.data_ptr(), 1, 50, seed=0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    sc.render_device(out.data_ptr(), spp, 50, seed=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    img = out.double().cpu().numpy()
    sc.close()
    del out
    torch.cuda.empty_cache()
    return img, 1920 * 1080 * spp / dt / 1e6


ref, r64 = run(D.TAKE_PRECISION_F64)
print(f"f64: {r64:.1f} Msamples/s", flush=True)
i32, r32 = run(D.TAKE_PRECISION_F32)
print(f"f32: {r32:.1f} Msamples/s, rmse vs f64 {np.sqrt(((i32 - ref) ** 2).mean()):.3e}", flush=True)
for k in (1, 2, 3, 4, 6, 60):
    im, r = run(D.TAKE_PRECISION_MIXED, k)
    d = np.abs(im - ref).max(axis=2)
    print(f"mixed exact_bounces={k}: {r:.1f} Msamples/s, rmse vs f64 {np.sqrt(((im - ref) ** 2).mean()):.3e}, pixels within 1e-3: {(d < 1e-3).mean():.4f}, "
          f"identical: {np.array_equal(im, ref)}", flush=True)
