/*
 * take_hip.h — C ABI of the MI355X path-tracing core (libtake_hip.so).
 *
 * This is the drop-in boundary for TaKe's hot path.  The reference has no FFI; the
 * seam is the C++ function `Image3 render(const std::vector<std::string>&)`
 * (reference src/render.h:5).  Everything between `build_bvh(scene)`
 * (src/render.cpp:49) and the end of the `parallel_for` tile loop
 * (src/render.cpp:59-82) is replaced by the calls below; the XML parser, the
 * `Scene` aggregate (src/scene.h:13-33) and `imwrite` (src/image.cpp:135) stay
 * on the host.  The flattening of a reference `Scene` into a `TakeSceneDesc` is
 * `take_amd/host/take_flatten.hpp`; INTEGRATION.md shows the 20-line patch to
 * src/render.cpp.
 *
 * Conventions: plain C, no exceptions cross the boundary.  Every entry point
 * returns 0 on success or a negative TAKE_E_* code and records a message that
 * `take_hip_last_error()` returns (thread-local).  Host arrays in a
 * `TakeSceneDesc` are caller-owned and only read during `take_hip_scene_create`.
 * Handles are library-owned.  Calls are synchronous unless they take a stream.
 * One scene handle lives on one GPU (the current HIP device at create time); a
 * multi-GPU job is one process per GPU, each with its own handle (scene
 * replicated), rows sharded with `strip_first/strip_stride`.
 */
#ifndef TAKE_HIP_H
#define TAKE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TAKE_HIP_ABI_VERSION 5 /* 2: TakeSceneDesc.instances, TakeRenderOpts.integrator (was reserved); 3: take_hip_render_accumulate; 4: TAKE_PRECISION_MIXED, TakeRenderOpts.exact_bounces; 5: TakeMesh.flags (was reserved), take_hip_mesh_from_ply / _from_serialized, TakeBuildOpts.instances (was reserved) */

/* error codes */
#define TAKE_OK 0
#define TAKE_E_INVALID (-1)   /* bad argument / malformed scene description   */
#define TAKE_E_DEVICE (-2)    /* HIP runtime error (message holds hipGetErrorString) */
#define TAKE_E_NO_GPU (-3)    /* no HIP device: the library never falls back to the CPU */
#define TAKE_E_NOMEM (-4)

/* Material tags: the alternative index of reference `Material`
 * (src/material.h:82-93), in declaration order. */
enum TakeMaterialTag {
    TAKE_MAT_DIFFUSE = 0,
    TAKE_MAT_MIRROR = 1,
    TAKE_MAT_PLASTIC = 2,
    TAKE_MAT_PHONG = 3,
    TAKE_MAT_BLINN_PHONG = 4,
    TAKE_MAT_BLINN_PHONG_MICROFACET = 5,
    TAKE_MAT_DISNEY_DIFFUSE = 6,
    TAKE_MAT_DISNEY_METAL = 7,
    TAKE_MAT_DISNEY_GLASS = 8,
    TAKE_MAT_DISNEY_CLEARCOAT = 9,
    TAKE_MAT_DISNEY_SHEEN = 10,
    TAKE_MAT_DISNEY_BSDF = 11,
    /* EXTENSION (not reference behaviour): the lobes the reference's README promises where its
     * src/materials/disney_{metal,glass,clearcoat,sheen,bsdf}.inl hold Lambert clones — Burley's
     * model in the five-lobe form of UCSD CSE 272 homework 1; specification: DESIGN.md §4d and
     * oracle/take_burley.hpp.  Tag = the reference alternative's tag + 5; a scene made from reference
     * materials gets them through TakeBuildOpts.burley_lobes. */
    TAKE_MAT_BURLEY_METAL = 12,
    TAKE_MAT_BURLEY_GLASS = 13,
    TAKE_MAT_BURLEY_CLEARCOAT = 14,
    TAKE_MAT_BURLEY_SHEEN = 15,
    TAKE_MAT_BURLEY_BSDF = 16,
    TAKE_MAT_COUNT = 17
};

/* reference `Texture = variant<ConstTexture, ImageTexture>` (src/texture.h:16-27) */
typedef struct TakeTexture {
    int32_t kind;      /* 0 = ConstTexture, 1 = ImageTexture */
    int32_t image_id;  /* index into TakeSceneDesc.images (ImageTexture::texture_id) */
    double value[3];   /* ConstTexture::value */
    double uscale, vscale, uoffset, voffset;
} TakeTexture;

/* One alternative of reference `Material`.  `param[]` = the alternative's scalar members in
 * declaration order (src/material.h:7-80), after `reflectance`:
 *   MIRROR, PLASTIC:                 eta
 *   PHONG, BLINN_PHONG, _MICROFACET: exponent
 *   DISNEY_DIFFUSE:                  roughness, subsurface
 *   DISNEY_METAL / BURLEY_METAL:     roughness, anisotropic
 *   DISNEY_GLASS / BURLEY_GLASS:     roughness, anisotropic, eta
 *   DISNEY_CLEARCOAT / BURLEY_CLEARCOAT: clearcoat_gloss
 *   DISNEY_SHEEN / BURLEY_SHEEN:     sheen_tint
 *   DISNEY_BSDF / BURLEY_BSDF:       specular_transmission, metallic, subsurface, specular, roughness,
 *                                    specular_tint, anisotropic, sheen, sheen_tint, clearcoat,
 *                                    clearcoat_gloss, eta
 * Tags 7..11 ignore their parameters, as the reference does (Lambert clones; CLEARCOAT evaluates
 * to zero); tags 12..16 use them.                                                              */
#define TAKE_MATERIAL_PARAMS 12
typedef struct TakeMaterial {
    int32_t tag;
    int32_t reserved;
    TakeTexture reflectance;
    double param[TAKE_MATERIAL_PARAMS];
} TakeMaterial;

/* reference `Image3` (src/image.h:13-39): texel (x,y) at data[(y*width+x)*3 + c] */
typedef struct TakeImage3 {
    int32_t width, height;
    const double *data;
} TakeImage3;

/* reference `TriangleMesh` (src/shape.h:13-18) */
typedef struct TakeMesh {
    int64_t n_vertices;
    int64_t n_faces;
    const double *positions; /* n_vertices * 3 */
    const int32_t *indices;  /* n_faces * 3 */
    const double *normals;   /* n_vertices * 3, or NULL (mesh.normals.empty()) */
    const double *uvs;       /* n_vertices * 2, or NULL (mesh.uvs.empty())     */
    int32_t material_id;
    int32_t flags;           /* 0, or TAKE_MESH_DEVICE_ARRAYS: the four arrays live in device memory (a mesh that
                                take_hip_mesh_from_ply decoded there) */
} TakeMesh;
#define TAKE_MESH_DEVICE_ARRAYS 1

/* reference `Sphere` (src/shape.h:20-23) */
typedef struct TakeSphere {
    double center[3];
    double radius;
    int32_t material_id;
    int32_t reserved;
} TakeSphere;

/* reference `Light = variant<PointLight, DiffuseAreaLight>` (src/light.h:9-19) */
typedef struct TakeLight {
    int32_t kind;      /* 0 = PointLight (counts toward N, contributes nothing:
                          src/integrator/path_tracing.h:33), 1 = DiffuseAreaLight,
                          2 = environment map (EXTENSION, not in the reference: equirectangular
                          image, y up, row 0 = zenith; importance-sampled by luminance*sin(theta),
                          seen by rays that leave the scene instead of `background`; at most one) */
    int32_t shape_id;  /* kind 1: DiffuseAreaLight::shape_id (index into the shape arrays);
                          kind 2: index into `images`; `intensity` then scales the image   */
    double intensity[3];
    double position[3];
} TakeLight;

/* EXTENSION (BASELINE configs[4]: "10M-triangle instanced scene"; the reference has no instancing):
 * one placement of a prototype mesh.  The mesh is NOT copied: the instance is a leaf of the top-level BVH that holds
 * a transform, and a ray entering it is moved into the mesh's object space (two-level traversal).  The result is
 * specified as that of the same geometry flattened to world space (xform applied to the vertices), to fp rounding.
 * Instanced triangles are not emitters.  Their shape ids follow the ordinary shapes: n_shapes + (sum of the face
 * counts of the preceding instances) + face.                                                               */
typedef struct TakeInstance {
    int32_t mesh_id;      /* index into TakeSceneDesc.meshes: the prototype (it need not appear in the shape arrays) */
    int32_t material_id;  /* material of this placement; -1: the mesh's own */
    double xform[12];     /* object -> world, 3x4 row-major affine: world = M[:, :3] * p + M[:, 3]; invertible */
} TakeInstance;

/* reference `Camera` (src/camera.h:5-11) */
typedef struct TakeCamera {
    int32_t width, height;
    double lookfrom[3], lookat[3], up[3];
    double vfov;
} TakeCamera;

/* Flattened reference `Scene` (src/scene.h:13-33).  The shape arrays are
 * `scene.shapes` in order (index = BVH primitive id = DiffuseAreaLight::shape_id),
 * as SoA:  kind 0 = Sphere (ref = index into spheres), 1 = Triangle (ref = mesh id,
 * face = face id).  area_light = ShapeBase::area_light_id (-1 if none).           */
typedef struct TakeSceneDesc {
    TakeCamera camera;
    double background[3];
    int32_t n_meshes;
    int32_t n_spheres;
    const TakeMesh *meshes;
    const TakeSphere *spheres;
    int64_t n_shapes;
    const int32_t *shape_kind;
    const int32_t *shape_ref;
    const int32_t *shape_face;
    const int32_t *shape_area_light;
    int32_t n_lights;
    int32_t n_materials;
    const TakeLight *lights;
    const TakeMaterial *materials;
    int32_t n_images;
    int32_t reserved;
    const TakeImage3 *images;
    int64_t n_instances;            /* extension, see TakeInstance; 0 = none */
    const TakeInstance *instances;
} TakeSceneDesc;

#define TAKE_PRECISION_F32 0
#define TAKE_PRECISION_F64 1
/* Mixed precision: the scene is resident in both arithmetics; a render computes the first `exact_bounces` rounds of
 * every path — camera ray and the next bounces: the ones whose hit / miss decisions carry the most radiance — in
 * double on the f64 scene, converts the surviving paths' records to float and finishes them on the f32 scene.  Images
 * are double.  Why: the f32 path differs from the reference's arithmetic by whole samples wherever a discrete decision
 * flips (~1.7 % of the paths on the 1M-triangle soup: per-pixel RMSE 1.8e-3 at 256 spp, outside the north-star's
 * 1e-3), and a flip costs what the path still carries, which falls off geometrically with the bounce. */
#define TAKE_PRECISION_MIXED 2
#define TAKE_DEFAULT_EXACT_BOUNCES 3

/* scene_create options */
typedef struct TakeBuildOpts {
    int32_t precision;     /* TAKE_PRECISION_F32 (production) or _F64 (parity mode) */
    int32_t bvh_threads;   /* host threads for the BVH build; <=0: hardware_concurrency */
    int32_t max_leaf_size; /* primitives per leaf, 1..4; <=0: default (1) */
    int32_t builder;       /* TAKE_BUILDER_AUTO (0): host SAH below TAKE_AUTO_DEVICE_BUILD_SHAPES shapes, device
                              LBVH from there on (f32 scenes without instances);
                              TAKE_BUILDER_DEVICE_LBVH: primitive records, Morton-order tree and its compression are
                              made on the GPU straight from the caller's mesh arrays (10M triangles: 0.2 s);
                              TAKE_BUILDER_HOST_SAH: binned SAH on the host (10M triangles: 6 s; traversal 2-6 %
                              faster).  Results do not depend on the builder (conservative box tests).          */
    int32_t burley_lobes;  /* 0: materials as given (tags 7..11 behave as the reference's stubs do);
                              1: tags 7..11 are taken as 12..16 — the scene's Disney materials get real lobes */
    int32_t instances;     /* what scene_create does with TakeSceneDesc.instances:
                              TAKE_INSTANCES_TWO_LEVEL (0): prototypes stay single, placements are leaves of a top-level
                              BVH (1000 x 10k triangles: 26 MB);
                              TAKE_INSTANCES_FLATTEN: every placement is expanded to world-space triangles before the
                              build — the geometry the instanced render is specified to equal, 1.1 GB for the same
                              scene, traversed a third faster (one tree resolves the overlap of the placements that
                              a two-level tree must descend into one by one).  Shape ids are the same either way. */
} TakeBuildOpts;
#define TAKE_INSTANCES_TWO_LEVEL 0
#define TAKE_INSTANCES_FLATTEN 1
#define TAKE_BUILDER_AUTO 0
#define TAKE_BUILDER_DEVICE_LBVH 1
#define TAKE_BUILDER_HOST_SAH 2
#define TAKE_AUTO_DEVICE_BUILD_SHAPES 4000000

/* render options: `scene.options` (src/scene.h:8-11) + what the reference
 * hard-codes in src/render.cpp. */
typedef struct TakeRenderOpts {
    int32_t spp;           /* RenderOptions::spp                                      */
    int32_t max_depth;     /* RenderOptions::max_depth; loop is i <= max_depth         */
    uint64_t seed;         /* global seed of the counter RNG (replaces the unseedable
                              std::random_device at src/render.cpp:60)                 */
    double ray_epsilon;    /* ray tmin / shadow-ray shortening (c_EPSILON's role at
                              src/render.cpp:75, path_tracing.h:53,79); <=0: default
                              (1e-7 in f64 as the reference, 1e-4 in f32 — the
                              reference's own commented alternative, src/take.h:31)    */
    int32_t strip_first;   /* multi-GPU: this rank renders the 4-row strips s with     */
    int32_t strip_stride;  /*   s % strip_stride == strip_first (1-GPU: 0 and 1)       */
    int32_t samples_per_batch; /* samples per pixel in flight at once; <=0: auto       */
    int32_t integrator;    /* which of the reference's integrators (src/integrator/path_tracing.h):
                              0 path_tracing (:5, multi-sample MIS — what render() calls; the default),
                              1 path_tracing_raw (:114), 2 path_tracing_one_sample_MIS (:161),
                              3 path_tracing_one_sample_MIS_power (:274, lights picked by power:
                              src/light.cpp:9-30).  1..3 are defined upstream but called by nothing there;
                              they do not know the environment-map extension (TAKE_E_INVALID with one) */
    int32_t exact_bounces; /* TAKE_PRECISION_MIXED scenes: rounds computed in double before the paths continue in
                              float (<= 0: TAKE_DEFAULT_EXACT_BOUNCES); ignored by f32 / f64 scenes             */
    int32_t reserved;
} TakeRenderOpts;
#define TAKE_INTEGRATOR_PATH_MIS 0
#define TAKE_INTEGRATOR_RAW 1
#define TAKE_INTEGRATOR_ONE_SAMPLE_MIS 2
#define TAKE_INTEGRATOR_ONE_SAMPLE_MIS_POWER 3

typedef struct TakeScene TakeScene; /* opaque */

/* ray / hit records of the trace hooks (test + traversal-only benchmark surface;
 * counterpart of scene_intersect / scene_occluded, src/scene.cpp:25-64).
 * Real-typed views: f32 scenes take/return the float fields, f64 scenes the doubles.
 * Contract of every trace hook: tmin >= 0 (the traversal orders entry distances through their bit patterns, which
 * needs non-negative values).  The host entry points (take_hip_trace_closest / _any) refuse a ray with tmin < 0 or
 * NaN with TAKE_E_INVALID; the *_device entry points, which cannot look at device-resident rays, start such a ray —
 * and one with tmin = -0.0 — at tmin = 0. */
typedef struct TakeRayF {
    float org[3], tmin, dir[3], tmax;
} TakeRayF;
typedef struct TakeRayD {
    double org[3], tmin, dir[3], tmax;
} TakeRayD;
typedef struct TakeHitF {
    int32_t shape_id; /* -1 = miss */
    float t, u, v;
} TakeHitF;
typedef struct TakeHitD {
    int32_t shape_id;
    int32_t reserved;
    double t, u, v;
} TakeHitD;

/* work counters of the most recent take_hip_render / trace call on the scene */
typedef struct TakeCounters {
    uint64_t samples;        /* camera paths                                      */
    uint64_t rays_closest;   /* scene_intersect calls                             */
    uint64_t rays_shadow;    /* scene_occluded calls                              */
    uint64_t node_visits;    /* wide-BVH interior nodes fetched (counting mode)   */
    uint64_t prim_tests;     /* ray/triangle + ray/sphere tests (counting mode)   */
    uint64_t bounces;        /* shade-kernel path iterations                      */
    double ms_trace_closest; /* summed HIP-event time of the closest-hit kernel   */
    double ms_trace_shadow;
    double ms_shade;
    double ms_other;
    double ms_total;         /* whole render, first kernel to last                */
    uint64_t launches_trace_closest;
    uint64_t launches_trace_shadow;
    uint64_t node_bytes;     /* bytes of one interior-node fetch in this layout   */
    uint64_t prim_bytes;     /* bytes of one primitive record                     */
    uint64_t leaf_visits;    /* leaves fetched (counting mode)                    */
    uint64_t wave_node_steps;/* wave-level node-phase iterations (counting mode): node_visits / (16 * this) is
                                the fraction of the 16 ray slots of a wave doing useful work in a node step */
    uint64_t wave_leaf_steps;/* wave-level leaf-phase iterations (counting mode)  */
    /* mixed-precision renders: the share of rays_closest / ms_trace_closest / launches_trace_closest that belongs to the
     * f32 rounds (the f32 instance of the closest-hit kernel); zero for f32 and f64 scenes */
    uint64_t rays_closest_f32;
    double ms_trace_closest_f32;
    uint64_t launches_trace_closest_f32;
} TakeCounters;

const char *take_hip_last_error(void);
int take_hip_abi_version(void);
/* number of visible HIP devices, or TAKE_E_NO_GPU */
int take_hip_device_count(void);

/* Replaces build_bvh(scene) (src/scene.cpp:4-23, src/bvh.cpp:8-45) + upload. */
int take_hip_scene_create(const TakeSceneDesc *desc, const TakeBuildOpts *opts, TakeScene **out);
int take_hip_scene_destroy(TakeScene *scene);

/* Replaces the parallel_for tile loop of render() (src/render.cpp:59-82) and all it
 * calls.  rgb_out: this rank's rows only, compacted in increasing image-row order
 * (n_rows(strip_first, strip_stride) * width * 3 Real), already flipped as
 * `img(x, height-y-1)` (src/render.cpp:78) and divided by spp.  f32 scenes write
 * float, f64 scenes write double.  `take_hip_render_rows` returns how many rows. */
int take_hip_render(TakeScene *scene, const TakeRenderOpts *opts, void *rgb_out_host);
/* Same, output left in device memory (d_rgb_out = device pointer), enqueued on
 * `stream` (hipStream_t, NULL = default stream); returns after enqueue + sync. */
int take_hip_render_device(TakeScene *scene, const TakeRenderOpts *opts, void *d_rgb_out,
                           void *stream);
/* Progressive / interactive rendering: the per-pixel accumulate of the tile loop
 * (src/render.cpp:68-78) kept resident in HBM between calls.  Renders opts->spp MORE samples per pixel on top of
 * what the scene has accumulated since the last call with restart != 0 (or since a one-shot take_hip_render*, which
 * ends a sequence), and writes the mean over ALL samples so far to d_rgb_out (device memory, same layout as
 * take_hip_render_device).  The samples continue the one-shot render's numbering — same random streams, same order of
 * the additions — so after calls with a, b, c.. samples the image equals take_hip_render(spp = a + b + c ..) BIT FOR
 * BIT.  seed, max_depth, integrator, ray_epsilon and the strip set must stay the same within a sequence
 * (TAKE_E_INVALID otherwise).  take_hip_accumulated_samples: samples per pixel in the accumulator. */
int take_hip_render_accumulate(TakeScene *scene, const TakeRenderOpts *opts, int32_t restart, void *d_rgb_out,
                               void *stream);
int64_t take_hip_accumulated_samples(const TakeScene *scene);
/* Egress on the device: the conversion half of the reference's imwrite("image.exr") (src/image.cpp:155-176 ->
 * tinyexr SaveEXR(components 3, fp16)).  d_rgb: height * width * 3 Real in device memory (row 0 = top, as
 * take_hip_render_device leaves it; precision says float or double) -> d_out: uint16 [height][3][width], per
 * scanline the channels B, G, R as half bit patterns rounded as that writer rounds (half-up on the first dropped
 * bit) — the bytes of an EXR scanline block before its ZIP pre-filter; the host deflates and frames them
 * (take_amd/exr.py: write_exr_scanlines).  Enqueued on `stream`, returns after it has completed. */
int take_hip_pack_exr_scanlines(const void *d_rgb, int32_t precision, int32_t width, int32_t height, uint16_t *d_out,
                                void *stream);
/* Render the whole image (strip_first / strip_stride are ignored) and hand back those scanlines in host memory
 * (height * 3 * width uint16): the float framebuffer never leaves the device. */
int take_hip_render_exr_scanlines(TakeScene *scene, const TakeRenderOpts *opts, uint16_t *out_host);

/* rows this rank owns / their image-row indices (rows_out may be NULL) */
int take_hip_render_rows(const TakeScene *scene, int32_t strip_first, int32_t strip_stride,
                         int32_t *rows_out);

/* Kernel-level hooks: n rays in host memory -> n hits in host memory. */
int take_hip_trace_closest(TakeScene *scene, const void *rays, int64_t n, void *hits);
int take_hip_trace_any(TakeScene *scene, const void *rays, int64_t n, int32_t *occluded);
/* Device-resident variants for the traversal-only benchmark: rays/hits are device
 * pointers; `count_mode` != 0 runs the instrumented kernel that fills node_visits /
 * prim_tests (never timed). */
int take_hip_trace_closest_device(TakeScene *scene, const void *d_rays, int64_t n, void *d_hits,
                                  int32_t count_mode, void *stream);

/* ---- several GPUs from ONE process (the C++ host of the reference is a single process: main.cpp -> render()).
 * Replaces the thread pool of src/parallel.cpp:183-237 for the drop-in: the scene is built once per device
 * (replicated), device k of n renders the 4-row strips s with s % n == k on its own host thread, and the strips are
 * gathered on the first device with hipMemcpyPeerAsync (xGMI between the GPUs of a node) — the only exchange.  The
 * image is identical for every n (pixel keys do not depend on the sharding).  `devices` lists the HIP devices to use
 * (NULL: 0 .. n_gpus-1); a device may appear more than once (logical shards of one GPU: how the sharding is tested
 * on a one-GPU box).  (A multi-process job — one process per GPU, RCCL gather — uses plain scene handles with
 * strip_first / strip_stride instead: take_amd/dist.py, bench.py.)                                               */
typedef struct TakeSceneGroup TakeSceneGroup; /* opaque */
int take_hip_group_create(const TakeSceneDesc *desc, const TakeBuildOpts *opts, int32_t n_gpus, const int32_t *devices,
                          TakeSceneGroup **out);
int take_hip_group_destroy(TakeSceneGroup *group);
/* Whole image (height * width * 3 Real, row 0 = top) into host memory / into memory of the group's first device.
 * opts->strip_first / strip_stride are ignored (the group shards by itself). */
int take_hip_group_render(TakeSceneGroup *group, const TakeRenderOpts *opts, void *rgb_out_host);
int take_hip_group_render_device(TakeSceneGroup *group, const TakeRenderOpts *opts, void *d_rgb_out);
int take_hip_group_size(const TakeSceneGroup *group);
/* counters of shard k's last render (ms_total etc. per device: load balance) */
int take_hip_group_get_counters(const TakeSceneGroup *group, int32_t k, TakeCounters *out);

int take_hip_get_counters(const TakeScene *scene, TakeCounters *out);
/* enable per-kernel HIP-event timing + counting mode for subsequent renders
 * (bit 0 = event timing, bit 1 = visit counters; both off by default) */
int take_hip_set_instrumentation(TakeScene *scene, int32_t flags);

/* Test hook: run the device shading functions on the rows of one of the reference's golden tables
 * (tests/golden/tables, column layouts of oracle/ref_harness.cpp).  kind: 0 material (27 -> 14 columns),
 * 1 light (30 -> 9), 2 texture (6 -> 3), 3 to_world (6 -> 3), 4 hemisphere_cos (1 -> 4).  `rnd` holds 8 doubles
 * per row: the first random_real() draws of the mt19937 stream the reference used for that row. */
int take_hip_debug_table(int32_t kind, int32_t precision, const double *in, int64_t n, int32_t in_cols,
                         const double *rnd, double *out, int32_t out_cols);

/* ---- PLY -> device mesh arrays ------------------------------------------------------------
 * Replaces src/parse/parse_ply.cpp:9-123 (`TriangleMesh parse_ply(filename, to_world)`) for binary_little_endian
 * files: the host reads only the text header, the binary body goes to HBM as it lies in the file and kernels do what
 * the reference's host loops do — widen x/y/z, nx/ny/nz, u/v to double, xform_point(to_world) on positions
 * (src/transform.cpp:79-87), xform_normal(inverse(to_world)) on normals (src/transform.cpp:95-100), narrow the face
 * list to int triples.  The arrays are bit-identical to the reference's `TriangleMesh` members.
 * `to_world` / `inv_to_world`: the reference's Matrix4x4, row-major (m[4*i+j] = M(i,j)); NULL = identity.  The caller
 * passes the inverse it already has (`inverse(to_world)`, src/matrix.h:81) — only meshes with normals read it.
 * On success `*out` describes a mesh whose arrays are DEVICE memory owned by the library (flags =
 * TAKE_MESH_DEVICE_ARRAYS): put it into a TakeSceneDesc like any other mesh, and give it back with
 * take_hip_mesh_release once every scene_create that uses it has returned.
 * Not decodable here (TAKE_E_INVALID, message starts with "unsupported"): ascii / big-endian files, list properties in
 * the vertex element or ahead of the mesh data — the caller keeps its host parser for those.  Faces that are not
 * triangles or index past the vertex array are TAKE_E_INVALID (the reference reads three indices per face
 * unconditionally, parse_ply.cpp:85-120). */
typedef struct TakePlyLayout {
    int64_t n_vertices, n_faces;
    int64_t vertex_offset, face_offset; /* bytes from the start of the file */
    int32_t vertex_stride, face_stride; /* bytes per row (faces: with three indices) */
    int32_t has_normals, has_uvs;
    int32_t position_is_f64, index_bytes;
    int32_t header_bytes, reserved;
} TakePlyLayout;
/* header only; needs no GPU */
int take_hip_ply_layout(const void *file_bytes, size_t n_bytes, TakePlyLayout *out);
int take_hip_mesh_from_ply(const void *file_bytes, size_t n_bytes, const double *to_world, const double *inv_to_world,
                           int32_t material_id, TakeMesh *out);
/* the same on a file (memory-mapped, so the body is read once, by the copy to the device) */
int take_hip_mesh_from_ply_file(const char *path, const double *to_world, const double *inv_to_world,
                                int32_t material_id, TakeMesh *out);
/* Mitsuba's serialized mesh format — replaces `TriangleMesh parse_serialized(filename, shape_index, to_world)`
 * (src/parse/parse_serialized.cpp:174-256).  The zlib stream is inflated on the host (a serial job) in ONE pass into one
 * buffer — the reference pulls it through ZStream::read three scalars per vertex — and the blocks (positions, normals,
 * uvs, [colours: skipped], index triples; float or double by EDoublePrecision) are decoded on the device by the kernels
 * of the PLY path, with the same transforms.  Versions 3 and 4, any sub-mesh (`shape_index`, offset table at the end of
 * the file: skip_to_idx, :117-133).  Arrays bit-identical to the reference's. */
int take_hip_mesh_from_serialized(const void *file_bytes, size_t n_bytes, int32_t shape_index, const double *to_world,
                                  const double *inv_to_world, int32_t material_id, TakeMesh *out);
int take_hip_mesh_from_serialized_file(const char *path, int32_t shape_index, const double *to_world,
                                       const double *inv_to_world, int32_t material_id, TakeMesh *out);
/* copy a device-array mesh to host arrays the caller sized from n_vertices / n_faces (NULL = skip that array) */
int take_hip_mesh_download(const TakeMesh *mesh, double *positions, int32_t *indices, double *normals, double *uvs);
int take_hip_mesh_release(TakeMesh *mesh);

/* BVH introspection (tests / DESIGN figures): node count, primitive count, depth */
int take_hip_scene_stats(const TakeScene *scene, int64_t *n_nodes, int64_t *n_prims,
                         int32_t *depth, int64_t *device_bytes);

#ifdef __cplusplus
}
#endif
#endif /* TAKE_HIP_H */
