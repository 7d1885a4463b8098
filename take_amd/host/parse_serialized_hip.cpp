// parse_serialized_hip.cpp — the reference's `TriangleMesh parse_serialized(const fs::path&, int, const Matrix4x4&)`
// (src/parse/parse_serialized.h:9-11, src/parse/parse_serialized.cpp:174-256) with the decode on the MI355X.
//
// A maintainer of TaKe links this file in place of src/parse/parse_serialized.cpp: parse_scene.cpp
// calls it unchanged.  take_hip_mesh_from_serialized_file finds the sub-mesh, inflates its zlib stream in one pass on the
// host (the reference pulls it through ZStream::read three scalars per vertex) and decodes the blocks on the device;
// the arrays come back into the reference's own `TriangleMesh` members, bit-identical to what the host loops fill
// (tests/test_serialized.py, tests/golden/serialized).  Every variant the reference reads is covered (versions 3 and 4,
// float / double, normals / uvs / colours, sub-mesh index), so there is no fall-back to the host parser here.
// Compiled, in the authoring container only, by `make -C oracle gpu_cli` into oracle/_ref/take_gpu.
#include <cstdint>
#include <string>

#include "matrix.h"
#include "parse/parse_serialized.h"
#include "utils/flexception.h"

#include "take_hip.h"

TriangleMesh parse_serialized(const fs::path &filename, int shape_index, const Matrix4x4 &to_world) {
    static_assert(sizeof(Vector3) == 3 * sizeof(double) && sizeof(Vector2) == 2 * sizeof(double) && sizeof(Vector3i) == 3 * sizeof(int32_t),
                  "TriangleMesh members are plain arrays of Real / int (src/vector.h): the download writes into them");
    const Matrix4x4 inv = inverse(to_world);  // what parse_serialized.cpp:227 pushes the normals through
    double xw[16], xi[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) xw[4 * i + j] = to_world(i, j), xi[4 * i + j] = inv(i, j);
    TakeMesh dm;
    if (take_hip_mesh_from_serialized_file(filename.string().c_str(), shape_index, xw, xi, 0, &dm) != TAKE_OK)
        Error(std::string("parse_serialized: ") + take_hip_last_error() + " in " + filename.string());
    TriangleMesh mesh;
    mesh.positions.resize((size_t)dm.n_vertices);
    mesh.indices.resize((size_t)dm.n_faces);
    if (dm.normals) mesh.normals.resize((size_t)dm.n_vertices);
    if (dm.uvs) mesh.uvs.resize((size_t)dm.n_vertices);
    const int rc = take_hip_mesh_download(&dm, (double *)mesh.positions.data(), (int32_t *)mesh.indices.data(),
                                          dm.normals ? (double *)mesh.normals.data() : nullptr, dm.uvs ? (double *)mesh.uvs.data() : nullptr);
    take_hip_mesh_release(&dm);
    if (rc != TAKE_OK) Error(std::string("parse_serialized: ") + take_hip_last_error());
    return mesh;
}
