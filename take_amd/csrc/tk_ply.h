// PLY / serialized -> device mesh arrays ("direct PLY/serialized -> device buffers").
//
// What it replaces: src/parse/parse_ply.cpp:9-123 — tinyply reads the file on the host, then four host loops widen
// x/y/z, u/v, nx/ny/nz to Real, push positions through xform_point(to_world) (src/transform.cpp:79-87), normals through
// xform_normal(inverse(to_world)) (src/transform.cpp:95-100) and narrow the face list to Vector3i.  Here the host reads
// ONLY the text header (parse_header: a few hundred bytes); the binary body goes to HBM as it lies in the file and two
// kernels do the widening, the transforms and the index narrowing, one vertex / one face per lane.  The arithmetic is the
// reference's, operation for operation, in double (-ffp-contract=off): the arrays are bit-identical to what parse_ply
// fills (tests/test_gpu_ply.py against the reference's own parser through oracle/_ref).
//
// Scope = what the reference's call can read, minus the host-only encodings: binary_little_endian files whose
// `vertex` element has scalar properties only and whose `face` element holds the list `vertex_indices` (any count /
// index type tinyply accepts) with every face a triangle — parse_ply.cpp:85-120 reads three indices per face whatever
// the list says, so anything else is garbage upstream and an error here.  ascii / big-endian files, and files with a
// list property ahead of the data, are TAKE_E_INVALID "unsupported": the caller keeps its host parser for those.
#pragma once
#include <cstdint>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "tk_common.h"

namespace tk {
namespace ply {

enum ScalarType : int32_t { T_I8 = 0, T_U8, T_I16, T_U16, T_I32, T_U32, T_F32, T_F64, T_NONE };

TK_HD int type_size(int32_t t) {
    return (t == T_I8 || t == T_U8) ? 1 : (t == T_I16 || t == T_U16) ? 2 : (t == T_F64) ? 8 : 4;
}

// what the header says about the two elements parse_ply reads; offsets are bytes from the start of the FILE
struct Layout {
    int64_t n_vertices, n_faces;
    int64_t vertex_off, face_off;  // first byte of the element's rows
    int64_t end_off;               // one past the last byte the decode reads (file must be at least this long)
    int32_t vertex_stride;         // bytes per vertex row
    int32_t face_stride;           // bytes per face row, GIVEN three indices per face (checked on the device)
    int32_t pos_type, pos_off[3];  // T_F32 / T_F64 (parse_ply.cpp:39-52 fills positions for these two only)
    int32_t nrm_type, nrm_off[3];  // T_NONE: the file has no nx/ny/nz
    int32_t uv_type, uv_off[2];    // T_NONE: no u/v
    int64_t nrm_base, uv_base;     // first byte of the rows the normals / uvs are read from, and their strides: the
    int32_t nrm_stride, uv_stride; // vertex rows in a PLY file (interleaved), blocks of their own in a serialized mesh
    int32_t count_type, index_type;  // count_type T_NONE: rows of three indices without a count (serialized mesh)
    int32_t list_off;              // byte offset of the list's count inside a face row
    int32_t header_bytes;
};

inline int32_t type_of(const std::string &s) {
    static const struct { const char *a, *b; int32_t t; } names[] = {
        {"char", "int8", T_I8},   {"uchar", "uint8", T_U8},   {"short", "int16", T_I16},  {"ushort", "uint16", T_U16},
        {"int", "int32", T_I32},  {"uint", "uint32", T_U32},  {"float", "float32", T_F32}, {"double", "float64", T_F64}};
    for (const auto &n : names)
        if (s == n.a || s == n.b) return n.t;
    return T_NONE;
}

// -> "" or what is wrong / unsupported.  Reads at most the header.
inline std::string parse_header(const uint8_t *p, size_t n, Layout &L) {
    std::memset(&L, 0, sizeof(L));
    L.pos_type = L.nrm_type = L.uv_type = L.count_type = L.index_type = T_NONE;
    // the header is text lines up to and including "end_header\n"
    static const char END[] = "end_header";
    size_t hdr = 0;
    {
        const size_t lim = std::min<size_t>(n, (size_t)1 << 20);
        size_t line = 0;
        bool found = false;
        for (size_t i = 0; i < lim; i++) {
            if (p[i] != '\n') continue;
            size_t e = i;
            if (e > line && p[e - 1] == '\r') e--;
            if (e - line == sizeof(END) - 1 && std::memcmp(p + line, END, sizeof(END) - 1) == 0) {
                hdr = i + 1, found = true;
                break;
            }
            line = i + 1;
        }
        if (!found) return "not a PLY file: no end_header line";
    }
    L.header_bytes = (int32_t)hdr;
    std::istringstream in(std::string((const char *)p, hdr));
    std::string ln;
    if (!std::getline(in, ln)) return "empty PLY header";
    if (!ln.empty() && ln.back() == '\r') ln.pop_back();
    if (ln != "ply") return "not a PLY file: first line is not `ply`";
    struct Prop { std::string name; int32_t type, count_type; bool list; };
    struct Elem { std::string name; int64_t count; std::vector<Prop> props; };
    std::vector<Elem> elems;
    bool have_format = false;
    while (std::getline(in, ln)) {
        if (!ln.empty() && ln.back() == '\r') ln.pop_back();
        std::istringstream ls(ln);
        std::string kw;
        if (!(ls >> kw)) continue;
        if (kw == "format") {
            std::string f, v;
            ls >> f >> v;
            if (f != "binary_little_endian") return "unsupported PLY encoding `" + f + "` (the device decode reads binary_little_endian)";
            have_format = true;
        } else if (kw == "comment" || kw == "obj_info") {
        } else if (kw == "element") {
            Elem e;
            long long c = -1;
            if (!(ls >> e.name >> c) || c < 0) return "malformed element line: " + ln;
            e.count = c;
            elems.push_back(e);
        } else if (kw == "property") {
            if (elems.empty()) return "property before any element";
            Prop pr{};
            std::string t;
            ls >> t;
            if (t == "list") {
                std::string ct, it;
                if (!(ls >> ct >> it >> pr.name)) return "malformed property line: " + ln;
                pr.list = true, pr.count_type = type_of(ct), pr.type = type_of(it);
                if (pr.count_type == T_NONE || pr.count_type == T_F32 || pr.count_type == T_F64) return "bad list count type: " + ln;
            } else {
                if (!(ls >> pr.name)) return "malformed property line: " + ln;
                pr.type = type_of(t);
            }
            if (pr.type == T_NONE) return "unknown property type: " + ln;
            elems.back().props.push_back(pr);
        } else if (kw == "end_header") {
            break;
        } else {
            return "unknown header keyword: " + ln;
        }
    }
    if (!have_format) return "PLY header has no format line";
    int64_t off = (int64_t)hdr;
    bool have_v = false, have_f = false;
    for (const Elem &e : elems) {
        if (e.name == "vertex" && !have_v) {
            int32_t stride = 0;
            int32_t seen[8];  // x y z nx ny nz u v -> offset
            int32_t types[8];
            for (int k = 0; k < 8; k++) seen[k] = -1, types[k] = T_NONE;
            static const char *want[8] = {"x", "y", "z", "nx", "ny", "nz", "u", "v"};
            for (const Prop &pr : e.props) {
                if (pr.list) return "unsupported: list property `" + pr.name + "` in the vertex element";
                for (int k = 0; k < 8; k++)
                    if (pr.name == want[k] && seen[k] < 0) seen[k] = stride, types[k] = pr.type;
                stride += type_size(pr.type);
            }
            // (request_properties_from_element needs every key of a request and one type per request: tinyply)
            if (seen[0] < 0 || seen[1] < 0 || seen[2] < 0) return "vertex positions not found";
            if (types[0] != types[1] || types[0] != types[2]) return "x / y / z have different types";
            if (types[0] != T_F32 && types[0] != T_F64) return "unsupported: x / y / z are neither float nor double (the reference leaves such positions unset)";
            L.pos_type = types[0];
            for (int k = 0; k < 3; k++) L.pos_off[k] = seen[k];
            if (seen[3] >= 0 && seen[4] >= 0 && seen[5] >= 0) {
                if (types[3] != types[4] || types[3] != types[5]) return "nx / ny / nz have different types";
                if (types[3] != T_F32 && types[3] != T_F64) return "unsupported: nx / ny / nz are neither float nor double";
                L.nrm_type = types[3];
                for (int k = 0; k < 3; k++) L.nrm_off[k] = seen[3 + k];
            }
            if (seen[6] >= 0 && seen[7] >= 0) {
                if (types[6] != types[7]) return "u / v have different types";
                if (types[6] != T_F32 && types[6] != T_F64) return "unsupported: u / v are neither float nor double";
                L.uv_type = types[6];
                for (int k = 0; k < 2; k++) L.uv_off[k] = seen[6 + k];
            }
            L.n_vertices = e.count, L.vertex_off = off, L.vertex_stride = stride;
            L.nrm_base = L.uv_base = off, L.nrm_stride = L.uv_stride = stride;
            off += e.count * (int64_t)stride;
            have_v = true;
        } else if (e.name == "face" && !have_f) {
            int32_t stride = 0;
            bool found = false, named = false;
            for (const Prop &pr : e.props) named = named || (pr.list && pr.name == "vertex_indices");
            if (!named) return "vertex indices not found (no list property `vertex_indices` in the face element)";
            for (const Prop &pr : e.props) {
                if (pr.list && pr.name == "vertex_indices" && !found) {
                    if (pr.type == T_F32 || pr.type == T_F64) return "vertex_indices is a list of reals";
                    found = true;
                    L.list_off = stride, L.count_type = pr.count_type, L.index_type = pr.type;
                    stride += type_size(pr.count_type) + 3 * type_size(pr.type);
                } else if (pr.list) {
                    return "unsupported: a second list property `" + pr.name + "` in the face element";
                } else {
                    stride += type_size(pr.type);
                }
            }
            L.n_faces = e.count, L.face_off = off, L.face_stride = stride;
            off += e.count * (int64_t)stride;
            have_f = true;
        } else {
            if (have_v && have_f) break;  // nothing behind the two elements is read
            int64_t stride = 0;
            for (const Prop &pr : e.props) {
                if (pr.list && e.count > 0) return "unsupported: list property in element `" + e.name + "` ahead of the mesh data";
                stride += type_size(pr.type);
            }
            off += e.count * stride;
        }
    }
    if (!have_v) return "vertex positions not found (no vertex element)";
    if (!have_f) return "vertex indices not found (no face element)";
    L.end_off = std::max(L.vertex_off + L.n_vertices * (int64_t)L.vertex_stride, L.face_off + L.n_faces * (int64_t)L.face_stride);
    if ((uint64_t)L.end_off > n) return "PLY file is shorter than its header says (" + std::to_string(n) + " bytes, need " + std::to_string(L.end_off) + ")";
    if (L.n_vertices >= ((int64_t)1 << 31) || L.n_faces >= ((int64_t)1 << 31) / 3) return "mesh too large for 32-bit vertex indices";
    return "";
}

// ---- Mitsuba's serialized mesh format (src/parse/parse_serialized.cpp:174-256) ------------------------------------
// File: per sub-mesh [u16 magic][u16 version 3|4][zlib stream]; at the end of the file one offset per sub-mesh (u64 in
// version 4, u32 in version 3) and a u32 count (skip_to_idx, parse_serialized.cpp:117-133).  Inflated stream: u32 flags,
// (version 4: a NUL-terminated name), u64 vertex count, u64 triangle count, then BLOCKS — positions (3 reals per
// vertex), normals if EHasNormals, uvs (2) if EHasTexcoords, colours (3, ignored) if EHasColors, 3 ints per triangle.
// Reals are double iff EDoublePrecision is set (parse_serialized.cpp:212: the single-precision flag is not looked at).
enum SerializedFlags : uint32_t { S_HAS_NORMALS = 0x0001, S_HAS_TEXCOORDS = 0x0002, S_HAS_COLORS = 0x0008, S_DOUBLE = 0x2000 };

// Layout of the blocks behind the counts; offsets are bytes from the first byte of the position block
inline void serialized_layout(uint32_t flags, int64_t nv, int64_t nf, Layout &L) {
    std::memset(&L, 0, sizeof(L));
    const int32_t t = (flags & S_DOUBLE) ? T_F64 : T_F32, sz = type_size(t);
    int64_t off = 0;
    L.n_vertices = nv, L.n_faces = nf;
    L.pos_type = t, L.vertex_off = 0, L.vertex_stride = 3 * sz;
    for (int k = 0; k < 3; k++) L.pos_off[k] = k * sz, L.nrm_off[k] = k * sz;
    L.uv_off[0] = 0, L.uv_off[1] = sz;
    off += nv * 3 * (int64_t)sz;
    L.nrm_type = L.uv_type = T_NONE;
    if (flags & S_HAS_NORMALS) L.nrm_type = t, L.nrm_base = off, L.nrm_stride = 3 * sz, off += nv * 3 * (int64_t)sz;
    if (flags & S_HAS_TEXCOORDS) L.uv_type = t, L.uv_base = off, L.uv_stride = 2 * sz, off += nv * 2 * (int64_t)sz;
    if (flags & S_HAS_COLORS) off += nv * 3 * (int64_t)sz;
    L.count_type = T_NONE, L.index_type = T_I32, L.list_off = 0, L.face_off = off, L.face_stride = 12;
    off += nf * 12;
    L.end_off = off;
}

// one scalar of a row, widened to double / narrowed to int32 the way the reference's casts do; rows are not aligned
// (a 13-byte face row is the common case), so bytes are gathered — neighbouring lanes read neighbouring rows
TK_HD double load_real(const uint8_t *p, int32_t type) {
    if (type == T_F32) {
        float f;
        __builtin_memcpy(&f, p, 4);
        return (double)f;
    }
    double d;
    __builtin_memcpy(&d, p, 8);
    return d;
}
TK_HD int64_t load_int(const uint8_t *p, int32_t type) {
    switch (type) {
    case T_I8: return (int8_t)p[0];
    case T_U8: return p[0];
    case T_I16: { int16_t v; __builtin_memcpy(&v, p, 2); return v; }
    case T_U16: { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
    case T_I32: { int32_t v; __builtin_memcpy(&v, p, 4); return v; }
    default: { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
    }
}

struct Mat4 {
    double m[16];  // row-major: m[4 * i + j] = Matrix4x4(i, j)
};

#if defined(__HIPCC__)
// positions: xform_point (src/transform.cpp:79-87) — homogeneous multiply, then times 1 / w
// normals:   xform_normal (src/transform.cpp:95-100) with the INVERSE matrix, transposed access, then normalize()
//            (src/vector.h:250-257: zero vector when the length is not positive)
__global__ void k_ply_vertices(const uint8_t *file, Layout L, Mat4 X, Mat4 Xi, double *pos, double *nrm, double *uv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.n_vertices) return;
    const uint8_t *row = file + L.vertex_off + i * (int64_t)L.vertex_stride;
    {
        const double x = load_real(row + L.pos_off[0], L.pos_type), y = load_real(row + L.pos_off[1], L.pos_type),
                     z = load_real(row + L.pos_off[2], L.pos_type);
        const double *m = X.m;
        const double tx = m[0] * x + m[1] * y + m[2] * z + m[3];
        const double ty = m[4] * x + m[5] * y + m[6] * z + m[7];
        const double tz = m[8] * x + m[9] * y + m[10] * z + m[11];
        const double tw = m[12] * x + m[13] * y + m[14] * z + m[15];
        const double inv_w = 1.0 / tw;
        pos[3 * i + 0] = tx * inv_w, pos[3 * i + 1] = ty * inv_w, pos[3 * i + 2] = tz * inv_w;
    }
    if (nrm) {
        const uint8_t *nrow = file + L.nrm_base + i * (int64_t)L.nrm_stride;
        const double x = load_real(nrow + L.nrm_off[0], L.nrm_type), y = load_real(nrow + L.nrm_off[1], L.nrm_type),
                     z = load_real(nrow + L.nrm_off[2], L.nrm_type);
        const double *m = Xi.m;
        const double nx = m[0] * x + m[4] * y + m[8] * z;
        const double ny = m[1] * x + m[5] * y + m[9] * z;
        const double nz = m[2] * x + m[6] * y + m[10] * z;
        const double l = sqrt(nx * nx + ny * ny + nz * nz);
        const double inv_l = 1.0 / l;  // (Vector3 / Real multiplies by the reciprocal: src/vector.h:194-197)
        if (l <= 0) nrm[3 * i + 0] = 0, nrm[3 * i + 1] = 0, nrm[3 * i + 2] = 0;
        else nrm[3 * i + 0] = nx * inv_l, nrm[3 * i + 1] = ny * inv_l, nrm[3 * i + 2] = nz * inv_l;
    }
    if (uv) {
        const uint8_t *urow = file + L.uv_base + i * (int64_t)L.uv_stride;
        uv[2 * i + 0] = load_real(urow + L.uv_off[0], L.uv_type);
        uv[2 * i + 1] = load_real(urow + L.uv_off[1], L.uv_type);
    }
}

// faces: three indices per face, narrowed to int (parse_ply.cpp:85-120).  status[0] |= 1: a face is not a triangle
// (the row stride assumed three indices: nothing behind that face can be trusted); |= 2: an index outside the
// vertex array (the reference would read out of bounds later; scene_create would reject it too)
__global__ void k_ply_faces(const uint8_t *file, Layout L, int32_t *idx, int32_t *status) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L.n_faces) return;
    const uint8_t *row = file + L.face_off + i * (int64_t)L.face_stride + L.list_off;
    const bool counted = L.count_type != T_NONE;
    int32_t bad = (counted && load_int(row, L.count_type) != 3) ? 1 : 0;
    const int cs = counted ? type_size(L.count_type) : 0, is = type_size(L.index_type);
    for (int k = 0; k < 3; k++) {
        const int32_t v = (int32_t)load_int(row + cs + k * is, L.index_type);
        if (v < 0 || v >= L.n_vertices) bad |= 2;
        idx[3 * i + k] = v;
    }
    if (bad) atomicOr(status, bad);
}
#endif

}  // namespace ply
}  // namespace tk
