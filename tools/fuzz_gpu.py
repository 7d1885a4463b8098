"""Exploratory GPU fuzz (more seeds than the committed tests/test_fuzz_scenes.py): random scenes, hit tables vs exhaustive
search, f64 renders of all four integrators vs the oracle, f32 host-built vs device-built tree.  usage: fuzz_gpu.py [first_seed last_seed]"""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, oracle
from fuzz_scenes import random_scene, random_rays, add_random_instances
from helpers import rays_to_abi, rmse
from take_amd import capi
from take_amd import cdefs as D
bad=0
first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (13, 60)
for seed in range(first, last + 1):
    sd, scale = random_scene(seed, res=20)
    for precision in (1, 0):
        f64 = precision==1
        rays = random_rays(sd, scale, 20000, seed, 1e-7 if f64 else 1e-4)
        if not f64: rays = rays.astype(np.float32).astype(np.float64)
        osc = oracle.OracleScene(sd, precision=precision); want = osc.isect_brute(rays)
        sc = capi.Scene(sd, precision=precision)
        hits = sc.trace_closest(rays_to_abi(rays, precision)); hit = want[:,0]>=0
        ok = np.array_equal(hits["shape_id"]>=0, hit) and np.array_equal(hits["t"][hit].astype(np.float64), want[hit,1]) and np.array_equal(sc.trace_any(rays_to_abi(rays, precision)).astype(bool), hit)
        if not ok: bad+=1; print("TRACE MISMATCH", seed, precision)
        if f64:
            for integ in (0,1,2,3):
                img = osc.render(4, 6, seed=seed, integrator=integ); got = sc.render(spp=4, max_depth=6, seed=seed, integrator=integ)
                d = np.abs(got-img).max(axis=2); m = max(1.0, img.max())
                if not (np.median(d) < 1e-11*m and (d < 1e-8*m).mean() >= 0.97):
                    bad+=1; print("RENDER MISMATCH", seed, integ, float((d<1e-8*m).mean()), rmse(got,img))
        else:
            dev = capi.Scene(sd, precision=precision, builder=D.TAKE_BUILDER_DEVICE_LBVH)
            a = sc.render(spp=2, max_depth=6, seed=seed); b = dev.render(spp=2, max_depth=6, seed=seed)
            if not np.array_equal(a, b, equal_nan=True): bad+=1; print("BUILDER MISMATCH", seed, np.abs(a-b).max())
            dev.close()
        sc.close(); osc.close()
print("gpu fuzz done, mismatches:", bad)
