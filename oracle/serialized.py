"""serialized.py — TEST INFRASTRUCTURE: CPU restatement of the reference's Mitsuba-serialized mesh loader, in numpy.

What it restates: `TriangleMesh parse_serialized(filename, shape_index, to_world)`
(src/parse/parse_serialized.cpp:174-256) with its helpers — skip_to_idx (:117-133: the offset table at the end of the
file, u64 entries in format version 4, u32 in version 3, then a u32 count), the zlib stream behind the 4-byte
magic/version header (ZStream, :27-115), the head of the inflated stream (u32 flags, a NUL-terminated name in version 4,
u64 vertex and triangle counts) and the blocks: positions, normals (EHasNormals), uvs (EHasTexcoords), colours
(EHasColors, skipped), int triples — reals double iff EDoublePrecision (:212).  Positions go through xform_point,
normals through xform_normal(inverse(to_world)) exactly as in oracle/ply.py.

Pinned by tests/golden/serialized/*: files written by oracle/gen_golden.py and the arrays the reference's own parser
(oracle/_ref/ref_harness serialized) made of them.  Only tests/ may import this module."""
import struct
import zlib

import numpy as np

HAS_NORMALS, HAS_TEXCOORDS, HAS_COLORS, DOUBLE = 0x0001, 0x0002, 0x0008, 0x2000


def parse_serialized(data, shape_index=0, to_world=None, inv_to_world=None):
    buf = bytes(data)
    version = struct.unpack_from("<H", buf, 2)[0]
    at = 0
    if shape_index > 0:
        count = struct.unpack_from("<I", buf, len(buf) - 4)[0]
        if version == 4:
            at = struct.unpack_from("<Q", buf, len(buf) - 4 - 8 * (count - shape_index))[0]
        else:
            at = struct.unpack_from("<I", buf, len(buf) - 4 * (count - shape_index + 1))[0]
    s = zlib.decompressobj(15).decompress(buf[at + 4:])
    flags = struct.unpack_from("<I", s, 0)[0]
    o = 4
    if version == 4:
        o = s.index(b"\0", o) + 1
    nv, nf = struct.unpack_from("<QQ", s, o)
    o += 16
    t = np.dtype("<f8" if flags & DOUBLE else "<f4")

    def block(cols):
        nonlocal o
        a = np.frombuffer(s, t, nv * cols, o).reshape(nv, cols).astype(np.float64)
        o += nv * cols * t.itemsize
        return a

    p = block(3)
    X = np.eye(4) if to_world is None else np.asarray(to_world, np.float64).reshape(4, 4)
    Xi = np.eye(4) if inv_to_world is None else np.asarray(inv_to_world, np.float64).reshape(4, 4)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    h = [X[i, 0] * x + X[i, 1] * y + X[i, 2] * z + X[i, 3] for i in range(4)]
    inv_w = 1.0 / h[3]
    out = {"positions": np.stack([h[0] * inv_w, h[1] * inv_w, h[2] * inv_w], axis=1), "normals": None, "uvs": None}
    if flags & HAS_NORMALS:
        n = block(3)
        x, y, z = n[:, 0], n[:, 1], n[:, 2]
        m = [Xi[0, j] * x + Xi[1, j] * y + Xi[2, j] * z for j in range(3)]
        length = np.sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2])
        with np.errstate(divide="ignore", invalid="ignore"):
            inv_l = 1.0 / length
            out["normals"] = np.where((length <= 0)[:, None], 0.0, np.stack([m[0] * inv_l, m[1] * inv_l, m[2] * inv_l], axis=1))
    if flags & HAS_TEXCOORDS:
        out["uvs"] = block(2)
    if flags & HAS_COLORS:
        block(3)
    out["indices"] = np.frombuffer(s, "<i4", nf * 3, o).reshape(nf, 3).copy()
    return out
