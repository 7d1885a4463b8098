// tk_scene.h — device-side scene layout (what lives in HBM) and the path-state SoA.
//
// Layout rules:
//  * BVH: 4-wide nodes in breadth-first order (top levels = array prefix).  The render paths of both precisions read
//    the 64-byte compressed form (QNode4: four 16-byte slots, planes on a 15-bit grid) — a lane loads its ray's
//    node with 4 x dwordx4, one 64-byte fabric request per node; the full-width form (Node4<R>) is the
//    builder's output and the fall-back for scenes the grid is too coarse for.
//  * Primitives are stored in leaf order as pre-transformed 64-byte records (v0, e1, e2 | centre, radius, then the
//    shading side of the primitive) so a leaf is one contiguous run and a hit costs the shade kernel one line.
//  * Everything else the shading stage needs (vertex normals, uvs, materials, lights, texels) is touched once per
//    bounce, not per node.
//  * Path state is one 128-byte record per path (f32): queues are compacted every bounce, so live slots are
//    scattered and a lane should consume whole cache lines (see PathState below).
#pragma once

#include "tk_common.h"

namespace tk {

constexpr int32_t CHILD_EMPTY = (int32_t)0x80000000;
constexpr int MAX_LEAF = 4;
constexpr int MAX_STACK_ENTRIES_W8 = 192;  // the same for the 8-wide tree (7 entries per level + 1)
constexpr int MAX_STACK_ENTRIES = 96;  // deepest traversal stack the trace kernels provide (LDS levels + spill area): per tree level 3 entries (4-wide nodes) or 7 (8-wide), + 1

// child word: >= 0 interior node index; < 0 (and != CHILD_EMPTY) leaf: -(1 + first*4 + (count-1)), first < 2^28, i.e.
// leaf words lie in [-2^30, -1]; the words between CHILD_EMPTY and -2^30 are instance references (two-level scenes:
// the leaf of the top-level tree that holds instance `id`): CHILD_EMPTY + 1 + id.
TK_HD int32_t make_leaf(int32_t first, int32_t count) { return -(1 + first * MAX_LEAF + (count - 1)); }
TK_HD int32_t leaf_first(int32_t c) { return (-c - 1) / MAX_LEAF; }
TK_HD int32_t leaf_count(int32_t c) { return ((-c - 1) % MAX_LEAF) + 1; }
constexpr int32_t INSTANCE_WORD_END = -(1 << 30);  // instance words are < this (and > CHILD_EMPTY)
TK_HD int32_t make_instance_word(int32_t id) { return (int32_t)(0x80000001u + (uint32_t)id); }
TK_HD bool is_instance_word(int32_t c) { return c < INSTANCE_WORD_END && c != CHILD_EMPTY; }
TK_HD int32_t instance_of_word(int32_t c) { return (int32_t)((uint32_t)c - 0x80000001u); }

// One child slot of a wide node: box + child word.  32 B in f32: the four lanes of a traversal quad each load one
// slot (2 x dwordx4), i.e. one wave instruction reads whole 128-B lines, 4 lanes per line.
template <class R> struct alignas(16) NodeChild {
    R bmin[3];
    R bmax[3];
    int32_t child;
    int32_t pad;
};
template <class R, int W> struct alignas(16) NodeW {
    NodeChild<R> c[W];
};
template <class R> using Node4 = NodeW<R, 4>;
template <class R> using Node8 = NodeW<R, 8>;  // (builder-side only: the 8-wide tree is traversed in its compressed form)
static_assert(sizeof(NodeChild<float>) == 32 && sizeof(Node4<float>) == 128, "one L2 line per f32 node");

// Compressed wide node of the render paths: 64 B, two per cache line.  The child boxes are stored on ONE 15-bit
// grid that spans the whole scene: plane = grid_lo[a] + (Q_BIAS + q) * grid_step[a]  (q in 0..Q_MAX,
// DeviceScene::grid_*), rounded outwards, so the decoded box contains the true one (tk_bvh.h: quantise_nodes).  A cell
// is 3e-5 of the scene extent — far below a primitive's size unless the scene mixes scales by more than ~1e4, in which
// case the builder keeps the full-width nodes (tk_host_scene.h).
// Why 15 bits and the bias: a plane coordinate becomes a float with ONE byte permute (v_perm_b32) and no conversion —
// the bits 0x48000000 | q << 8 are the float 4 * (Q_BIAS + q) for q < 2^15 (exponent 2^17, q in mantissa bits 8..22) —
// so the slab test of a slot is 6 permutes + 3 packed fmas (round 2: 3 rotates + 6 conversions + 3 packed fmas on a
// 16-bit grid).  The bias is folded into grid_lo: grid_lo lies one grid extent below the scene's lower corner.
// Why: the traversal is bound by the vector L1, which pays one line look-up per ray and per load instruction, and
// by VALU issue at the same time (profiles/r01_tcp_counters.txt, r01_ubench_gather.txt).  A slot is
// 16 B: a lane reads its ray's node with 4 x dwordx4 from one line (the pair build: two slots per lane, no exchange),
// and because the grid is global the ray is moved into grid space once per ray, not once per node.
constexpr int Q_MAX = 32767;    // largest plane coordinate on the grid
constexpr int Q_BIAS = 32768;   // plane = grid_lo + (Q_BIAS + q) * grid_step
struct QChild {
    uint32_t q[3];  // axis a: lo | hi << 16, both in 0..Q_MAX
    int32_t child;  // child word (node index / leaf word / CHILD_EMPTY)
};
template <int W> struct alignas(16 * W) QNodeW {
    QChild c[W];
};
using QNode4 = QNodeW<4>;
// 8-wide compressed node: 128 B = one L2 line = one fabric request per visit, a third fewer visits per ray than the
// 4-wide tree.  Slot s holds the child that lies on the side (s & 1 ? high : low) of x, (s & 2) of y,
// (s & 4) of z of the node (tk_bvh.h assigns children to slots by an auction on their centres): a ray visits the slots
// in the order s ^ octant — front to back as far as one permutation per octant can tell — so a step needs no
// ranking of eight entry distances, only their minimum.
using QNode8 = QNodeW<8>;
static_assert(sizeof(QNode4) == 64 && sizeof(QNode8) == 128, "64- and 128-byte compressed nodes");
// a quantisation grid as the builders make it (tk_bvh.h: make_qgrid): delta = the outward slack every plane gets
// before it is snapped to the grid
struct QGrid {
    float lo[3], step[3];
    double delta[3];
};
// outward snap of the interval [l, h] (already widened by delta) to plane coordinates of axis a; exact in double (a
// float plus a 17-bit multiple of a float), the same on the host and on the device
TK_HD void qgrid_snap(const QGrid &g, int a, double l, double h, long long &ql, long long &qh) {
    const double base = (double)g.lo[a], step = (double)g.step[a], p = base + (double)Q_BIAS * step;
    ql = (long long)floor((l - p) / step);
    while (base + (double)(Q_BIAS + ql) * step > l) ql--;
    qh = (long long)ceil((h - p) / step);
    while (base + (double)(Q_BIAS + qh) * step < h) qh++;
    ql = ql < 0 ? 0 : ql, qh = qh > Q_MAX ? Q_MAX : qh;  // no-ops by construction of the grid
}

constexpr int32_t PRIM_TRIANGLE = 0;
constexpr int32_t PRIM_SPHERE = 1;
// triangle: a = v0.xyz, e1.xyz, e2.xyz (e = v_k - v0 in R arithmetic, as the reference computes per test,
// src/shape.cpp:52-53); sphere: a[0..2] = centre, a[3] = radius.  meta = kind | material tag << 8.
// The last four words are the shading side of the primitive — what the shade kernel needs right after a hit, without
// the prim -> shape -> mesh chain of dependent loads — in the same 64-byte record (f32), so a hit costs the shade
// kernel one line for geometry and shading data together; the trace kernels read only the first 44 bytes.
template <class R> struct alignas(16) PrimRec {
    R a[9];
    int32_t shape_id;
    int32_t meta;
    int32_t material;
    int32_t area_light;
    int32_t nidx;  // first of the face's 3 entries in face_idx (units: faces), or -1 when the mesh has neither vertex
                   // normals nor uvs (then nothing else is read)
    int32_t mesh;  // mesh id (valid when nidx >= 0)
    int32_t pad[sizeof(R) == 4 ? 1 : 0];
};
static_assert(sizeof(PrimRec<float>) == 64, "64-byte f32 primitive record: two per cache line, never straddling");
static_assert(sizeof(PrimRec<double>) == 96, "f64 primitive record");
constexpr int PRIM_TEST_BYTES = 48;  // what a primitive test reads of a record (3 x 16 B): the algorithmic bytes per test

constexpr int32_t META_HAS_ATTR = 1 << 16;  // PrimRec::meta flag: vertex normals and/or uvs exist

struct ShapeInfo {  // indexed by shape id
    int32_t mesh;   // >= 0 mesh id (triangle); < 0: a sphere (-(1 + sphere id))
    int32_t face;   // face id within the mesh
    int32_t material;
    int32_t area_light;
};
struct MeshInfo {
    int32_t fbase;   // first face of this mesh in face_idx (units: faces)
    int32_t nbase;   // first vertex normal (units: vertices), -1 if the mesh has none
    int32_t uvbase;  // first uv, -1 if none
    int32_t material;
};
template <class R> struct MaterialRec {
    int32_t tag, tex_kind, tex_image, pad;
    R color[3];
    R uscale, vscale, uoffset, voffset;
    R p0, p1;
    R p[TAKE_MATERIAL_PARAMS];  // every TakeMaterial::param (the Burley tags read these; p0 = p[0], p1 = p[1])
};
struct ImageInfo {
    int32_t width, height;
    int64_t offset;  // first texel (units: texels) in `texels`
};
// Environment-map light (TakeLight kind 2: an EXTENSION — the reference has only a constant background, SURVEY.md
// §0; BASELINE configs[2] asks for env-map IBL with importance sampling).  Equirectangular image, y up:
// v = acos(d.y) / pi (row 0 = zenith), u = atan2(d.z, d.x) / 2pi + 1/2; radiance = texel (nearest) * scale, so
// that the piecewise-constant sampling density below is exact.  marginal: height+1 row CDF values of
// luminance * sin(theta_row); conditional: per row width+1 column CDF values (both start at 0 and end at 1).
template <class R> struct EnvMap {
    int32_t light;   // index of the env light in `lights`, -1 = none (then `background` is used)
    int32_t width, height;
    int32_t n_guide_m;  // entries of the guide tables (powers of two: k = int(xi * n) is then exact), see below
    int32_t n_guide_c, pad_;
    int64_t texel0;  // first texel of the image in `texels`
    R scale[3];
    const R *marginal;
    const R *conditional;
    // guide tables: entry k of a guide = the CDF interval that holds k / ENV_GUIDE_*; a look-up starts its bisection
    // between two neighbouring entries (0-1 steps instead of 10-11), with the same result
    const int32_t *guide_m;  // n_guide_m + 1 entries
    const int32_t *guide_c;  // height rows of n_guide_c + 1 entries
};
constexpr int ENV_GUIDE_M_MAX = 1 << 16, ENV_GUIDE_C_MAX = 1 << 13;  // (sizes: tk_host_scene.h; TAKE_HIP_ENV_GUIDE=<m>,<c> overrides them for A/B runs)
template <class R> struct LightRec {
    int32_t kind;      // 0 point, 1 diffuse area, 2 environment map (see EnvMap)
    int32_t shape_id;  // -1 for point lights
    int32_t is_sphere;
    int32_t pad;
    R intensity[3];
    R v[9];  // triangle: v0 v1 v2 | sphere: centre, radius
    R n[9];  // triangle: vertex normals n0 n1 n2
};

// ---- instancing (EXTENSION, TakeInstance): a placement of a prototype mesh = one leaf of the top-level tree.
// What the trace kernel reads when a ray enters the instance: world -> object transform, the root of the mesh's own
// BVH (node indices and leaf ranges are global: all trees share `nodes` / `prims`) and, for compressed nodes, that
// tree's quantisation grid.
template <class R> struct alignas(16) InstTrace {
    R inv[12];          // world -> object, 3x4 row-major (t along a ray is the same number in both spaces)
    float grid_lo[3], grid_step[3];
    int32_t root_child;
    int32_t pad;
};
// What the shade kernel reads for a hit inside an instance: object -> world linear part (edges of the hit triangle),
// the transposed inverse for normals comes from InstTrace::inv; the material of the placement.
template <class R> struct alignas(16) InstShade {
    R fwd[9];           // linear part of object -> world, row-major
    int32_t material;   // material of this placement (already resolved: never -1)
    int32_t tag;        // its tag (the material sort reads it)
    int32_t shape_base; // shape id of the instance's face 0
    int32_t pad;
};

template <class R> struct CameraRec {
    R u[3], v[3], w[3], lookfrom[3];
    R viewport_width, viewport_height;
    int32_t width, height;
};

// All device pointers of one scene.  Passed to kernels by value.
template <class R> struct DeviceScene {
    const Node4<R> *nodes;
    const QNode4 *qnodes;  // compressed copy of nodes (same indices); nullptr = traverse the full-width nodes (or qnodes8)
    const QNode8 *qnodes8; // the 8-wide compressed tree (then nodes and qnodes are null)
    float grid_lo[3], grid_step[3];  // the quantisation grid of qnodes: plane = grid_lo + (Q_BIAS + q) * grid_step
    const PrimRec<R> *prims;
    int32_t root_child;  // child word of the root (a leaf word when the scene has <= MAX_LEAF shapes)
    int32_t n_nodes;
    const ShapeInfo *shapes;
    const MeshInfo *meshes;
    const int32_t *face_idx;  // 3 local vertex ids per face, all meshes concatenated
    const R *normals;         // 3 per vertex, meshes with normals concatenated
    const R *uvs;             // 2 per vertex, meshes with uvs concatenated
    const MaterialRec<R> *materials;
    const ImageInfo *images;
    const R *texels;  // 3 per texel
    const InstTrace<R> *inst_trace;  // two-level scenes (TakeInstance), else null
    const InstShade<R> *inst_shade;
    int32_t n_instances;
    const LightRec<R> *lights;
    const R *light_pmf;  // n_lights: power of each light / total (reference Scene::lights_power_pmf, scene.h:28)
    const R *light_cdf;  // n_lights + 1: running sum from 0 (Scene::lights_power_cdf); integrator 3 only
    int32_t n_lights;
    int32_t n_shapes;
    R background[3];
    EnvMap<R> env;
    CameraRec<R> cam;
};

// ---- path state: one PATH_REC-word record per path slot (128 B in f32, 256 B in f64).
// Queues are compacted every round, so after a few bounces the live slots are scattered: with one array per
// component every 4-byte access of a wave touched its own cache line (measured: the shade kernel moved 5x its
// algorithmic bytes).  With a record per path a lane consumes whole lines.  The first 64 B hold what the
// closest-hit kernel touches (ray, hit record), the rest what only the shade / shadow kernels need.
enum StateR {
    S_OX = 0, S_OY, S_OZ,      // ray origin = position of the current vertex
    S_DX, S_DY, S_DZ,          // ray direction (extend ray)
    S_HT, S_HU, S_HV,          // closest-hit record of the extend ray
    S_PDF = 12,                // pdf of the pending BSDF sample
    S_TX, S_TY, S_TZ,          // throughput
    S_LX, S_LY, S_LZ,          // radiance of this sample so far
    S_ST,                      // shadow-ray tmax
    S_FX, S_FY, S_FZ,          // FG of the pending BSDF sample
    S_SX, S_SY, S_SZ,          // shadow-ray direction
    S_CX, S_CY, S_CZ,          // throughput * C1: added to radiance if the shadow ray is unoccluded
    S_NUM_R = 29
};
enum StateI { S_HIT = 9, S_CTR = 10, S_FLAGS = 11, S_INST = 29, S_OCC = 30, S_CONV = 31, S_NUM_I = 6 };  // integer words of the same record
// S_HIT: index of the hit primitive (leaf order), -1 = the extend ray missed
// S_INST: instance the hit primitive was reached through (two-level scenes only; -1 = none)
// S_OCC: counting mode only (an instrument): the primitive that occluded this slot's previous shadow ray
// S_CONV: mixed precision, f64 records only: 1 = this path went on in the f32 record of its slot (its radiance is the
//         sum of the two records'); cleared by round 0 of every batch (same cache line as the radiance words)
constexpr int PATH_REC = 32;
constexpr int32_t FLAG_SPECULAR = 1;

template <class R> struct PathState {
    R *r;            // PATH_REC * slots
    int64_t stride;  // slots allocated (not used for addressing)
    TK_HD R &R_(int c, int64_t s) const { return r[s * PATH_REC + c]; }
    TK_HD int32_t &I_(int c, int64_t s) const { return *reinterpret_cast<int32_t *>(&r[s * PATH_REC + c]); }
};

// queue bookkeeping words (one small device array)
enum QueueWord {
    Q_N_EXT0, Q_N_EXT1,   // sizes of the two extend queues (ping-pong)
    Q_N_SHADOW,           // size of the shadow queue
    Q_HEAD_CLOSEST,       // work-fetch head of the closest-hit kernel
    Q_HEAD_SHADOW,        // work-fetch head of the shadow kernel
    Q_HEAD_SHADE,
    Q_NUM_WORDS = 16
};

// instrumentation counters (device, 64-bit)
enum CounterWord {
    C_NODE_VISITS, C_PRIM_TESTS, C_RAYS_CLOSEST, C_RAYS_SHADOW, C_BOUNCES, C_LEAF_VISITS, C_WAVE_NODE_STEPS,
    C_WAVE_LEAF_STEPS, C_WAIT_SLOTS, C_IDLE_SLOTS,
    C_RAYS_CLOSEST_TAIL,  // mixed-precision renders: the extend rays of the f32 rounds (also counted in C_RAYS_CLOSEST)
    C_OCC_CACHE_HITS,     // counting mode: shadow rays that the previous occluder of their path slot occludes as well
    C_NUM_WORDS = 12
};

}  // namespace tk
