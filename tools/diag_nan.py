"""diagnostic: locate non-finite pixels of a GPU render of a fuzz scene and the bounce / material they come from"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from fuzz_scenes import random_scene  # noqa: E402
from take_amd import capi  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 12
sd, scale = random_scene(seed)
print("materials:", [(i, m.tag) for i, m in enumerate(sd.materials)])
sc = capi.Scene(sd, precision=1)
osc = oracle.OracleScene(sd, precision=1)
for depth in range(0, 5):
    for spp in (1, 4):
        got = sc.render(spp=spp, max_depth=depth, seed=seed)
        want = osc.render(spp, depth, seed=seed)
        bad = np.argwhere(~np.isfinite(got).all(axis=2))
        print("depth", depth, "spp", spp, "non-finite px", len(bad), bad[:4].tolist(), "oracle finite", bool(np.isfinite(want).all()))
        if len(bad) and spp == 1:
            y, x = bad[0]
            print("  gpu", got[y, x], "oracle", want[y, x])
sc.close()
osc.close()
