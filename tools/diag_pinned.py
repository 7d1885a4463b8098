"""scene_create with the device builder at 10M triangles: the caller's arrays pinned in place vs the pageable path"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TAKE_HIP_VERBOSE"] = "1"
from take_amd import capi, scenes
from take_amd import cdefs as D
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sd = scenes.soup_scene(n, 640, 360, spp=1)
for mode in ("1", "0", "1", "0"):
    os.environ["TAKE_HIP_PINNED_UPLOAD"] = mode
    t = time.time(); sc = capi.Scene(sd, builder=D.TAKE_BUILDER_DEVICE_LBVH); dt = time.time() - t
    print(f"pinned={mode}: scene_create {dt:.3f} s", flush=True)
    sc.close()
