"""OpenEXR egress of the host mirror: the file the reference's `imwrite("image.exr", img)` produces
(src/image.cpp:155-176 -> tinyexr `SaveEXR(..., components = 3, save_as_fp16 = 1)`): one scanline part, channels
B, G, R as 16-bit half (rounded as that writer rounds — see float_to_half), increasing-y line order, ZIP compression in blocks of 16 scanlines
(no compression when both sides are < 16 pixels), data window = display window = the image.

Written from the OpenEXR file-layout documentation; pinned by tests/test_exr.py against a file written by the
reference's own imwrite (tests/golden/egress/): same header attributes that matter, identical half pixels.
`read_exr` is a small reader for exactly this family of files (scanline, NONE / ZIPS / ZIP, half or float channels)."""
import struct
import zlib

import numpy as np

MAGIC = 20000630
COMPRESSION = {"none": 0, "zips": 2, "zip": 3}
LINES_PER_BLOCK = {0: 1, 2: 1, 3: 16}


def _attr(name, typ, payload):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload


def _predict(raw):
    """OpenEXR's ZIP pre-filter: de-interleave even/odd bytes, then byte deltas (+128)"""
    b = np.frombuffer(raw, np.uint8)
    t = np.concatenate([b[0::2], b[1::2]]).astype(np.int16)
    d = t.copy()
    d[1:] = t[1:] - t[:-1] + 128
    return (d & 0xFF).astype(np.uint8).tobytes()


def _unpredict(buf):
    d = np.frombuffer(buf, np.uint8).astype(np.int64)
    d[1:] -= 128
    t = (np.cumsum(d) & 0xFF).astype(np.uint8)
    half = (len(t) + 1) // 2
    out = np.empty(len(t), np.uint8)
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def float_to_half(a):
    """float32 -> half bit patterns with the rounding of the reference's writer (measured on its files and on the
    ties in tests/golden/egress): the magnitude is rounded HALF UP on the first dropped mantissa bit alone — not to
    nearest-even — a carry may run into the exponent (up to inf); float denormals become zero; anything at or beyond
    2^16 becomes inf; results below the half normal range become half denormals by the same rule."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.int64)
    sign = (u >> 16) & 0x8000
    exp = (u >> 23) & 0xFF
    man = u & 0x7FFFFF
    newexp = exp - 112  # re-biased exponent
    out = np.zeros(u.shape, np.int64)
    normal = (exp != 0) & (exp != 255) & (newexp > 0) & (newexp < 31)
    out = np.where(normal, ((newexp << 10) | (man >> 13)) + ((man >> 12) & 1), out)
    out = np.where((exp != 255) & (newexp >= 31), 0x7C00, out)
    out = np.where(exp == 255, 0x7C00 | np.where(man != 0, 0x200, 0), out)
    sub = (exp != 0) & (newexp <= 0) & (14 - newexp <= 24)
    full = man | 0x800000
    sh = np.clip(14 - newexp, 1, 25)
    out = np.where(sub, (full >> sh) + ((full >> (sh - 1)) & 1), out)
    return (out | sign).astype(np.uint16)


def write_exr(path, img):
    """img: (H, W, 3) RGB float array in Image3 order (row 0 = top).  Writes what the reference's imwrite writes."""
    a = np.asarray(img, np.float32)
    assert a.ndim == 3 and a.shape[2] == 3
    half = float_to_half(a)  # (H, W, 3) bit patterns
    write_exr_scanlines(path, np.ascontiguousarray(half[:, :, ::-1].transpose(0, 2, 1)))  # per line: B, G, R


def write_exr_scanlines(path, scan):
    """scan: uint16 (H, 3, W) — per scanline the B, G, R half bit patterns, i.e. what libtake_hip's device-side
    egress (take_hip_pack_exr_scanlines / Scene.render_exr_scanlines) hands over.  What is left for the host is the
    byte-serial part of the reference's imwrite: ZIP pre-filter + deflate per 16-line block, header, offset table."""
    scan = np.ascontiguousarray(scan, np.uint16)
    assert scan.ndim == 3 and scan.shape[1] == 3
    h, w = scan.shape[0], scan.shape[2]
    comp = COMPRESSION["none"] if (w < 16 and h < 16) else COMPRESSION["zip"]
    chlist = b"".join(n + b"\0" + struct.pack("<iBBBBii", 1, 0, 0, 0, 0, 1, 1) for n in (b"B", b"G", b"R")) + b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    header = struct.pack("<iI", MAGIC, 2)
    header += _attr("channels", "chlist", chlist)
    header += _attr("compression", "compression", struct.pack("<B", comp))
    header += _attr("dataWindow", "box2i", box)
    header += _attr("displayWindow", "box2i", box)
    header += _attr("lineOrder", "lineOrder", b"\0")
    header += _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    header += _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0))
    header += _attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    header += b"\0"
    lines = LINES_PER_BLOCK[comp]
    chunks = []
    for y0 in range(0, h, lines):
        raw = scan[y0:y0 + lines].astype("<u2").tobytes()
        data = raw
        if comp != 0:
            z = zlib.compress(_predict(raw))
            if len(z) < len(raw):
                data = z
        chunks.append(struct.pack("<ii", y0, len(data)) + data)
    offset = len(header) + 8 * len(chunks)
    table = b""
    for c in chunks:
        table += struct.pack("<Q", offset)
        offset += len(c)
    with open(path, "wb") as f:
        f.write(header + table + b"".join(chunks))


def read_exr(path):
    """-> (channels: dict name -> (H, W) array of float16/float32, header: dict of raw attributes)"""
    buf = open(path, "rb").read()
    magic, version = struct.unpack_from("<iI", buf, 0)
    assert magic == MAGIC and (version & 0xFF) == 2 and not (version & 0x1E00), "single-part scanline files only"
    pos = 8
    hdr = {}
    while buf[pos] != 0:
        e = buf.index(b"\0", pos)
        name = buf[pos:e].decode()
        pos = e + 1
        e = buf.index(b"\0", pos)
        typ = buf[pos:e].decode()
        pos = e + 1
        (size,) = struct.unpack_from("<i", buf, pos)
        pos += 4
        hdr[name] = (typ, buf[pos:pos + size])
        pos += size
    pos += 1
    channels = []
    ch = hdr["channels"][1]
    p = 0
    while ch[p] != 0:
        e = ch.index(b"\0", p)
        name = ch[p:e].decode()
        ptype, _, _, _, _, xs, ys = struct.unpack_from("<iBBBBii", ch, e + 1)
        assert xs == 1 and ys == 1 and ptype in (1, 2)
        channels.append((name, np.float16 if ptype == 1 else np.float32))
        p = e + 1 + 16
    comp = hdr["compression"][1][0]
    x0, y0, x1, y1 = struct.unpack("<iiii", hdr["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    lines = LINES_PER_BLOCK[comp]
    n_chunks = (h + lines - 1) // lines
    offsets = struct.unpack_from("<%dQ" % n_chunks, buf, pos)
    out = {name: np.zeros((h, w), dt) for name, dt in channels}
    line_bytes = sum(np.dtype(dt).itemsize for _, dt in channels) * w
    for off in offsets:
        y, size = struct.unpack_from("<ii", buf, off)
        data = buf[off + 8:off + 8 + size]
        n = min(lines, y1 - y + 1)
        if comp != 0 and size < n * line_bytes:
            data = _unpredict(zlib.decompress(data))
        p = 0
        for r in range(n):
            for name, dt in channels:
                nb = np.dtype(dt).itemsize * w
                out[name][y - y0 + r] = np.frombuffer(data, dt, w, p)
                p += nb
    return out, hdr
