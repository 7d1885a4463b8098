"""diagnostic: GPU render vs oracle per pixel on the golden scenes (run on the GPU box)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
from helpers import GOLDEN_SCENES, golden_scene
from take_amd import capi

for name in GOLDEN_SCENES:
    sd = golden_scene(name)
    for prec in (1, 0):
        osc = oracle.OracleScene(sd, precision=prec)
        sc = capi.Scene(sd, precision=prec)
        for depth in (-1, 0, 1, 2, 50):
            want = osc.render(1, depth, rng_mode=oracle.RNG_COUNTER, seed=11)
            got = sc.render(spp=1, max_depth=depth, seed=11).astype(np.float64)
            d = np.abs(got - want).max(axis=2)
            bad = np.argwhere(d > 1e-6 * (1 + np.abs(want).max(axis=2)))
            print(f"{name} {'f64' if prec else 'f32'} depth {depth}: rmse {np.sqrt(((got-want)**2).mean()):.3e} bad pixels {len(bad)} / {d.size}")
            for (y, x) in bad[:4]:
                print("    px", y, x, "got", got[y, x], "want", want[y, x])
        osc.close(); sc.close()
