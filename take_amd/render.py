"""Host-side mirror of the reference's `render()` (src/render.h:5, src/render.cpp:9-86).

Same argument convention — a parameter list holding a scene file and an optional `-max_depth D`
(default 50, src/render.cpp:14) — and the same result: the image in `Image3` layout (row 0 = top,
`img(x, y)` at [y, x, :]).  The scene file is a `.tkscene`, i.e. a reference `Scene` flattened by
take_amd/host/take_flatten.hpp after the reference's own XML parser ran (the parser is outside this path).
All rendering happens in libtake_hip.so on the current HIP device; without it the call raises.
"""
import numpy as np

from . import cdefs as D
from .capi import Scene
from .scene import SceneData, load_tkscene


def parse_params(params):
    """argument handling of src/render.cpp:14-23"""
    max_depth = 50
    filename = None
    i = 0
    while i < len(params):
        if params[i] == "-max_depth":
            i += 1
            max_depth = int(params[i])
        elif filename is None:
            filename = params[i]
        i += 1
    return filename, max_depth


def render(params, seed=0, precision=D.TAKE_PRECISION_F32, ray_epsilon=0.0):
    """render(["scene.tkscene", "-max_depth", "5"]) -> (H, W, 3) numpy image (float32, or float64 in f64 mode).
    An empty parameter list returns an empty image, as the reference does (src/render.cpp:10-12)."""
    if len(params) < 1:
        return np.zeros((0, 0, 3), np.float32)
    filename, max_depth = parse_params(list(params))
    sd = filename if isinstance(filename, SceneData) else load_tkscene(filename)
    scene = Scene(sd, precision=precision)
    try:
        return scene.render(spp=sd.spp, max_depth=max_depth, seed=seed, ray_epsilon=ray_epsilon)
    finally:
        scene.close()


def imwrite(path, img):
    """the reference's imwrite (src/image.cpp:135-177): `.exr` (half, as tinyexr writes it — take_amd/exr.py) or `.pfm`"""
    if str(path).endswith(".exr"):
        from .exr import write_exr

        write_exr(path, img)
    elif str(path).endswith(".pfm"):
        imwrite_pfm(path, img)
    else:
        raise ValueError(f"Unsupported image format: {path}")


def imwrite_pfm(path, img):
    """PFM as the reference writes it (src/image.cpp:145-153): header `PF\\nW H\\n-1\\n`, float32 RGB rows in
    Image3 order."""
    a = np.ascontiguousarray(img, np.float32)
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1\n" % (a.shape[1], a.shape[0]))
        f.write(a.tobytes())


def main(argv=None):
    """`python -m take_amd.render scene.tkscene [-max_depth D]` — the reference's main.cpp:9-27: render, then
    imwrite("image.exr") into the current directory."""
    import sys

    params = list(sys.argv[1:] if argv is None else argv)
    if "-t" in params:  # main.cpp:13-15: thread count of the CPU pool; meaningless here, accepted and dropped
        i = params.index("-t")
        del params[i:i + 2]
    if not params:
        return 0  # an empty image is not written (src/image.cpp:136-138)
    # render + float -> half + scanline packing on the device (take_hip_render_exr_scanlines); what the host adds is
    # the byte-serial rest of imwrite: ZIP pre-filter, deflate, header (take_amd/exr.py)
    from .exr import write_exr_scanlines

    filename, max_depth = parse_params(params)
    sd = load_tkscene(filename)
    scene = Scene(sd)
    try:
        write_exr_scanlines("image.exr", scene.render_exr_scanlines(spp=sd.spp, max_depth=max_depth, seed=0))
    finally:
        scene.close()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
