"""diagnostic: find pixels where GPU and oracle differ, then dump every sample's path state on GPU and in hostsim"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
from helpers import golden_scene, hostsim_render
from take_amd import capi
name, prec, depth, spp, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
sd = golden_scene(name)
osc = oracle.OracleScene(sd, precision=prec)
want = osc.render(spp, depth, rng_mode=oracle.RNG_COUNTER, seed=seed)
sc = capi.Scene(sd, precision=prec)
got = sc.render(spp=spp, max_depth=depth, seed=seed).astype(np.float64)
d = np.abs(got - want).max(axis=2)
bad = np.argwhere(d > 1e-9)
print("bad pixels", len(bad), bad[:5].tolist(), flush=True)
if len(bad):
    iy, ix = bad[0]
    p = (sd.height - 1 - iy) * sd.width + ix
    npix = sd.width * sd.height
    for s in range(spp):
        os.environ["TAKE_HIP_DUMP_SLOT"] = str(s * npix + p)
        print(f"== hostsim s{s}", file=sys.stderr, flush=True)
        hostsim_render(sd, prec, spp, depth, seed=seed)
        print(f"== gpu s{s}", file=sys.stderr, flush=True)
        sc.render(spp=spp, max_depth=depth, seed=seed)
