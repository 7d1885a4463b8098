"""diagnostic: parse -> first pixel at 10M triangles.  Writes configs[4]-sized soup as the binary PLY
the reference's scenes use and times, on the same (page-cached) file:
  * the reference's own parse_ply (oracle/_ref/ref_harness ply_time, if the harness travelled to this box),
  * a vectorised numpy parse (oracle/ply.py — what a careful host parser costs),
  * take_hip_mesh_from_ply_file (header on the host, body -> HBM, decode kernels),
then scene_create from the device-decoded mesh against scene_create from host arrays."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from oracle import ply as oply  # noqa: E402
from take_amd import capi, scenes  # noqa: E402
from take_amd import cdefs as D  # noqa: E402

n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
path = f"/tmp/soup_{n_tris}.ply"
sd = scenes.soup_scene(n_tris, 64, 64, spp=1)
soup = max(range(len(sd.meshes)), key=lambda i: sd.meshes[i].indices.shape[0])
host = sd.meshes[soup]
vert = np.zeros(len(host.positions), [("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
vert["x"], vert["y"], vert["z"] = host.positions[:, 0], host.positions[:, 1], host.positions[:, 2]
face = np.zeros(len(host.indices), [("n", "u1"), ("i", "<i4", 3)])
face["n"], face["i"] = 3, host.indices
hdr = "\n".join(["ply", "format binary_little_endian 1.0", f"element vertex {len(vert)}", "property float x", "property float y",
                 "property float z", f"element face {len(face)}", "property list uchar int vertex_indices", "end_header"]) + "\n"
with open(path, "wb") as f:
    f.write(hdr.encode()), f.write(vert.tobytes()), f.write(face.tobytes())
size = os.path.getsize(path)
print(f"{path}: {len(vert)} vertices, {len(face)} faces, {size / 1e6:.0f} MB", flush=True)
del vert, face

harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
if os.path.exists(harness):
    for _ in range(2):
        r = subprocess.run([harness, "ply_time", path], stdout=subprocess.PIPE, text=True)
        print("reference parse_ply (tinyply + host loops):", r.stdout.strip().split()[0], "s", flush=True)
for _ in range(2):
    t0 = time.time()
    m = oply.parse_ply(open(path, "rb").read())
    print(f"numpy host parse: {time.time() - t0:.3f} s", flush=True)
del m
capi.device_count()
for _ in range(3):
    t0 = time.time()
    dm = capi.DeviceMesh(path, material_id=host.material_id)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"device decode (mmap -> HBM -> kernels): {dt * 1e3:.1f} ms = {size / dt / 1e9:.1f} GB/s of file", flush=True)
    keep = dm
# the same mesh as a Mitsuba-serialized file (version 4, float; zlib level 1)
import struct  # noqa: E402
import zlib  # noqa: E402

spath = f"/tmp/soup_{n_tris}.serialized"
body = struct.pack("<I", 0x1000) + b"soup\0" + struct.pack("<QQ", len(host.positions), len(host.indices))
body += host.positions.astype("<f4").tobytes() + host.indices.astype("<i4").tobytes()
with open(spath, "wb") as f:
    f.write(struct.pack("<HH", 0x041C, 4) + zlib.compress(body, 1) + struct.pack("<QI", 0, 1))
print(f"{spath}: {os.path.getsize(spath) / 1e6:.0f} MB ({len(body) / 1e6:.0f} MB inflated)", flush=True)
del body
if os.path.exists(harness):
    r = subprocess.run([harness, "serialized_time", spath], stdout=subprocess.PIPE, text=True)
    print("reference parse_serialized (ZStream, three reads per vertex):", r.stdout.strip().split()[0], "s", flush=True)
for _ in range(2):
    t0 = time.time()
    sm = capi.DeviceMesh(spath, material_id=host.material_id)
    torch.cuda.synchronize()
    print(f"device decode of the serialized file (one-pass inflate on the host + kernels): {time.time() - t0:.3f} s", flush=True)
    sm.close()
os.remove(spath)
os.environ["TAKE_HIP_VERBOSE"] = "1"
sd_dev = scenes.soup_scene(8, 64, 64, spp=1)  # (the box + light; the soup mesh is swapped for the device one)
for label, mesh in (("host arrays", type(host)(host.positions.astype(np.float32).astype(np.float64), host.indices, host.material_id)),
                    ("device-decoded mesh", keep)):
    sdx = scenes.soup_scene(n_tris, 64, 64, spp=1)
    sdx.meshes[soup] = mesh
    t0 = time.time()
    sc = capi.Scene(sdx)
    print(f"scene_create from {label}: {time.time() - t0:.3f} s", flush=True)
    img = sc.render(spp=1, max_depth=4)
    print(f"  first frame mean {img.mean():.4f}", flush=True)
    sc.close()
os.remove(path)
