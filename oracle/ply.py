"""ply.py — TEST INFRASTRUCTURE: CPU restatement of the reference's PLY loader, in numpy.

What it restates: `TriangleMesh parse_ply(filename, to_world)` (src/parse/parse_ply.cpp:9-123) for binary
little-endian files — tinyply's header grammar, then the reference's four loops: positions widened to double and pushed
through xform_point (src/transform.cpp:79-87), normals through xform_normal(inverse(to_world)) (src/transform.cpp:95-100
+ normalize, src/vector.h:250-257, Vector3 / Real = multiply by the reciprocal, src/vector.h:194-197), uvs widened,
the face list narrowed to int triples (three indices per face, whatever the count says: parse_ply.cpp:85-120).
numpy evaluates a + b + c + d left to right in IEEE double like the C++ does, so the arrays are bit-comparable.

Pinned by tests/golden/ply/*: files in every encoding the reference accepts and the arrays its own parser
(oracle/_ref/ref_harness ply) made of them (tests/test_oracle_golden.py::test_ply_oracle_matches_reference).
Only tests/ may import this module; the product decodes PLY files on the device (take_amd/csrc/tk_ply.h).
"""
import numpy as np

TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "<i2", "int16": "<i2", "ushort": "<u2",
         "uint16": "<u2", "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4", "float": "<f4", "float32": "<f4",
         "double": "<f8", "float64": "<f8"}


def read_header(buf):
    """-> (elements, header_bytes); elements = [(name, count, [(property name, dtype | (count dtype, item dtype))])]"""
    end = buf.find(b"end_header")
    if end < 0 or not buf.startswith(b"ply"):
        raise ValueError("not a PLY file")
    nl = buf.index(b"\n", end)
    elements = []
    for ln in buf[:nl].decode("ascii").splitlines()[1:]:
        w = ln.split()
        if not w or w[0] in ("comment", "obj_info", "end_header"):
            continue
        if w[0] == "format":
            if w[1] != "binary_little_endian":
                raise ValueError("unsupported encoding " + w[1])
        elif w[0] == "element":
            elements.append((w[1], int(w[2]), []))
        elif w[0] == "property":
            if w[1] == "list":
                elements[-1][2].append((w[4], (TYPES[w[2]], TYPES[w[3]])))
            else:
                elements[-1][2].append((w[2], TYPES[w[1]]))
        else:
            raise ValueError("unknown header line: " + ln)
    return elements, nl + 1


def parse_ply(data, to_world=None, inv_to_world=None):
    """-> dict(positions (nv,3) f64, indices (nf,3) i32, normals (nv,3) f64 | None, uvs (nv,2) f64 | None)"""
    buf = bytes(data)
    elements, off = read_header(buf)
    rows = {}
    for name, count, props in elements:
        fields = []
        for pname, t in props:
            if isinstance(t, tuple):  # list: three items per row (what parse_ply.cpp assumes), count kept for the check
                fields += [(pname + "#n", t[0]), (pname, t[1], 3)]
            else:
                fields.append((pname, t))
        dt = np.dtype(fields)
        rows[name] = np.frombuffer(buf, dt, count, off)
        off += count * dt.itemsize
        if name == "face":
            for pname, t in props:
                if isinstance(t, tuple) and not (rows[name][pname + "#n"] == 3).all():
                    raise ValueError("a face is not a triangle")
    v, f = rows["vertex"], rows["face"]
    X = np.eye(4) if to_world is None else np.asarray(to_world, np.float64).reshape(4, 4)
    Xi = np.eye(4) if inv_to_world is None else np.asarray(inv_to_world, np.float64).reshape(4, 4)
    x, y, z = (v[k].astype(np.float64) for k in "xyz")
    t = [X[i, 0] * x + X[i, 1] * y + X[i, 2] * z + X[i, 3] for i in range(4)]
    inv_w = 1.0 / t[3]
    out = {"positions": np.stack([t[0] * inv_w, t[1] * inv_w, t[2] * inv_w], axis=1), "normals": None, "uvs": None,
           "indices": np.ascontiguousarray(f["vertex_indices"].astype(np.int64).astype(np.int32))}
    names = v.dtype.names
    if all(k in names for k in ("nx", "ny", "nz")):
        x, y, z = (v[k].astype(np.float64) for k in ("nx", "ny", "nz"))
        n = [Xi[0, j] * x + Xi[1, j] * y + Xi[2, j] * z for j in range(3)]
        length = np.sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2])
        with np.errstate(divide="ignore", invalid="ignore"):
            inv_l = 1.0 / length
            out["normals"] = np.where((length <= 0)[:, None], 0.0, np.stack([n[0] * inv_l, n[1] * inv_l, n[2] * inv_l], axis=1))
    if "u" in names and "v" in names:
        out["uvs"] = np.stack([v["u"].astype(np.float64), v["v"].astype(np.float64)], axis=1)
    return out
