"""diagnostic: dump one path's state per kernel on the GPU and in hostsim (TAKE_HIP_DUMP_SLOT)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import golden_scene, hostsim_render
from take_amd import capi
name, prec, depth, iy, ix = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
sd = golden_scene(name)
slot = (sd.height - 1 - iy) * sd.width + ix
os.environ["TAKE_HIP_DUMP_SLOT"] = str(slot)
print("== hostsim", file=sys.stderr, flush=True)
hostsim_render(sd, prec, 1, depth, seed=11)
print("== gpu", file=sys.stderr, flush=True)
sc = capi.Scene(sd, precision=prec)
sc.render(spp=1, max_depth=depth, seed=11)
if len(sys.argv) > 6:
    os.environ["TAKE_HIP_NO_SORT"] = "1"
    print("== gpu no sort", file=sys.stderr, flush=True)
    sc.render(spp=1, max_depth=depth, seed=11)
