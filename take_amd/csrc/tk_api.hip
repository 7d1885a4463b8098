// tk_api.hip — implementation of the C ABI of include/take_hip.h: scene upload, the wavefront render loop,
// the trace hooks.  Host code here only orchestrates: every per-sample operation runs in the kernels of
// tk_kernels.h.  There is no CPU rendering path in this library: without a HIP device every entry point
// returns TAKE_E_NO_GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <memory>
#include <zlib.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "take_hip.h"
#include "tk_host_scene.h"
#include "tk_build_gpu.h"
#include "tk_kernels.h"
#include "tk_ply.h"

using namespace tk;

namespace {

thread_local std::string g_error;
int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(TAKE_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));             \
    } while (0)

// Fault injection for the allocation-failure tests (tests/test_gpu_robustness.py): TAKE_HIP_FAIL_ALLOC=<k> makes the
// k-th device allocation after the variable was (re)set fail with hipErrorOutOfMemory.  A real out-of-memory cannot
// be provoked reliably from a test: the driver over-commits, a 300 GB request on a 288 GB device succeeded.
// (allocations happen on several host threads at once — one per shard of a scene group — hence the lock)
inline bool inject_alloc_failure() {
    static std::mutex mu;
    static std::string seen;
    static long calls = 0;
    const char *e = std::getenv("TAKE_HIP_FAIL_ALLOC");
    std::lock_guard<std::mutex> lock(mu);
    if (!e || !*e) {
        seen.clear();
        return false;
    }
    if (seen != e) seen = e, calls = 0;
    return ++calls == std::atol(e);
}

// Host -> device copies of the caller's large arrays (mesh positions, shape arrays) for the device-side scene build
// (SURVEY.md §8(f)2): the pages are pinned IN PLACE (hipHostRegister) so that the DMA engine reads the caller's memory
// directly — no bounce through the runtime's staging buffers — and the copies of all arrays are in flight together;
// the registrations are dropped once the stream has drained.  Arrays below 4 MiB, and memory that cannot be
// registered, take the ordinary pageable path.  TAKE_HIP_PINNED_UPLOAD=0 turns the registration off (A/B runs).
struct PinnedUploads {
    hipStream_t stream = nullptr;
    std::vector<void *> regs;
    size_t pinned_bytes = 0, plain_bytes = 0;
    bool enabled = !(std::getenv("TAKE_HIP_PINNED_UPLOAD") && std::atoi(std::getenv("TAKE_HIP_PINNED_UPLOAD")) == 0);
    hipError_t copy(void *dst, const void *src, size_t bytes) {
        if (bytes == 0) return hipSuccess;
        if (enabled && bytes >= ((size_t)4 << 20)) {
            if (hipHostRegister(const_cast<void *>(src), bytes, hipHostRegisterDefault) == hipSuccess) {
                regs.push_back(const_cast<void *>(src));
                pinned_bytes += bytes;
            } else {
                (void)hipGetLastError();  // (already registered, read-only mapping, ...): pageable copy
                plain_bytes += bytes;
            }
        } else {
            plain_bytes += bytes;
        }
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream);
    }
    hipError_t finish() {
        const hipError_t e = hipStreamSynchronize(stream);
        for (void *p : regs) (void)hipHostUnregister(p);
        regs.clear();
        return e;
    }
    ~PinnedUploads() { (void)finish(); }
};

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        release();  // p = nullptr, n = 0: the state a failed allocation leaves behind
        if (count == 0) return hipSuccess;
        const hipError_t e = inject_alloc_failure() ? hipErrorOutOfMemory : hipMalloc((void **)&p, count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();  // the error is reported through the return value, not left sticky
            return e;
        }
        n = count;
        return hipSuccess;
    }
    hipError_t upload(const std::vector<T> &v) {
        hipError_t e = alloc(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    size_t bytes() const { return n * sizeof(T); }
};

struct EventPool {
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    hipEvent_t get() {
        if (used == ev.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;  // callers treat a null event as a failed timing call
            ev.push_back(e);
        }
        return ev[used++];
    }
    void reset() { used = 0; }
    void destroy() {
        for (auto e : ev) (void)hipEventDestroy(e);
        ev.clear();
        used = 0;
    }
};

enum TimedKernel { TK_CLOSEST, TK_SHADOW, TK_SHADE, TK_OTHER, TK_CLOSEST_TAIL /* mixed precision: closest hits of the f32 rounds */, TK_NUM };

template <class R> struct SceneT {
    HostScene<R> host;  // kept: cheap relative to HBM copies, used for stats
    DevBuf<Node4<R>> nodes;
    DevBuf<QNode4> qnodes;
    DevBuf<QNode8> qnodes8;
    DevBuf<PrimRec<R>> prims;
    DevBuf<ShapeInfo> shapes;
    DevBuf<MeshInfo> meshes;
    DevBuf<int32_t> face_idx;
    DevBuf<R> normals, uvs, texels;
    DevBuf<MaterialRec<R>> materials;
    DevBuf<ImageInfo> images;
    DevBuf<LightRec<R>> lights;
    DevBuf<R> light_pmf, light_cdf;
    DevBuf<InstTrace<R>> inst_trace;
    DevBuf<InstShade<R>> inst_shade;
    DevBuf<R> env_marginal, env_conditional;
    DevBuf<int32_t> env_guide_m, env_guide_c;
    DeviceScene<R> dev{};
    // render workspace (grown on demand)
    DevBuf<R> state_r;
    DevBuf<int32_t> queue[2], shadow_queue, sorted_queue;
    DevBuf<uint8_t> sort_keys;            // one key byte per queue entry (material sort)
    DevBuf<int32_t> sort_hist, sort_base;  // [key][wave] counts and their exclusive scan
    DevBuf<R> accum, out;
    DevBuf<int32_t> qwords;  // Q_NUM_WORDS + 2 * N_SORT_KEYS
    DevBuf<unsigned long long> counters;
    DevBuf<unsigned long long> spill;
    int64_t capacity = 0;  // path slots allocated
    int trace_grid = 0;
    int group = TQ_GROUP;      // lanes per ray of the trace kernel (one instantiated size)
    bool built_on_device = false;
    int64_t spill_stride = 0;  // ray groups in the persistent trace grid

    size_t scene_bytes() const {
        return nodes.bytes() + qnodes.bytes() + qnodes8.bytes() + prims.bytes() + shapes.bytes() + meshes.bytes() + face_idx.bytes() + normals.bytes() +
               uvs.bytes() + texels.bytes() + materials.bytes() + images.bytes() + lights.bytes() + inst_trace.bytes() + inst_shade.bytes();
    }
    void release() {
        nodes.release(), qnodes.release(), qnodes8.release(), prims.release(), shapes.release(), meshes.release(), face_idx.release();
        normals.release(), uvs.release(), texels.release(), materials.release(), images.release(), lights.release();
        env_marginal.release(), env_conditional.release(), env_guide_m.release(), env_guide_c.release();
        light_pmf.release(), light_cdf.release(), inst_trace.release(), inst_shade.release();
        state_r.release(), queue[0].release(), queue[1].release(), shadow_queue.release();
        sorted_queue.release(), sort_keys.release(), sort_hist.release(), sort_base.release(), accum.release(), out.release(), qwords.release(), counters.release(), spill.release();
    }
};

}  // namespace

struct TakeScene {
    int precision = TAKE_PRECISION_F32;
    int device = 0;
    int num_cus = 256;
    // progressive rendering (take_hip_render_accumulate): samples per pixel summed in `accum` so far, under which options
    int64_t acc_samples = 0;
    TakeRenderOpts acc_opts{};
    int mem_share = 1;  // scenes of one group on this device: each sizes its path-state batch for 1/mem_share of the free HBM
    int instrumentation = 0;
    SceneT<float> f;
    SceneT<double> d;
    TakeCounters counters{};
    EventPool events;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> timed;
    // queue lengths read back WITHOUT stalling the launch loop: a ring of pinned words + events (render_impl)
    static constexpr int POLL_RING = 64;
    int32_t *poll_host = nullptr;  // POLL_RING pinned words
    hipEvent_t poll_ev[POLL_RING] = {};
};

namespace {

// records, images and trace hooks of a mixed-precision scene are the f64 ones (its f32 side finishes the paths)
inline bool is_f64(const TakeScene *s) { return s->precision != TAKE_PRECISION_F32; }

template <class R> SceneT<R> &pick(TakeScene *s);
template <> SceneT<float> &pick<float>(TakeScene *s) { return s->f; }
template <> SceneT<double> &pick<double>(TakeScene *s) { return s->d; }

// BVH build on the device (tk_build_gpu.h).  In: sc.prims uploaded in SHAPE order.  Out: the records in
// leaf order, sc.nodes or sc.qnodes, host-side stats and grid.  Returns TAKE_OK, an error, or 1 = "use the host
// builder" (tree deeper than the traversal stack allows: long runs of equal Morton codes).
int build_bvh_device(SceneT<float> &sc, int max_leaf, bool compressed_ok, bool compressed_forced) {
    using namespace lbvh;
    HostScene<float> &h = sc.host;
    const int n = (int)sc.prims.n;
    // default 1 primitive per leaf: two Morton neighbours need not be close, and a leaf box around both costs more
    // primitive tests than the extra node (1M soup, 16 spp: 1 / 2 / 4 per leaf = 55.2 / 38.3 / 30.0 Msamples/s)
    const int leaf_size = std::max(1, std::min(max_leaf > 0 ? max_leaf : 1, MAX_LEAF));
    const int n_leaves = (n + leaf_size - 1) / leaf_size;
    if (n_leaves < 2) return 1;
    hipStream_t stream = nullptr;
    const dim3 blk(BLK);
    auto grid = [](int64_t items) { return dim3((unsigned)((items + BLK - 1) / BLK)); };

    DevBuf<Box> pb, lbox, ibox;
    DevBuf<uint64_t> keys, keys_s, lkey;
    DevBuf<uint32_t> vals, vals_s;
    DevBuf<int> scene_ord, parent_i, parent_l, flag, frontier[2], lvl;
    DevBuf<int2> child;
    DevBuf<char> temp;
    DevBuf<double> acc;
    struct Cleanup {
        std::function<void()> f;
        ~Cleanup() { f(); }
    } cleanup{[&] {
        pb.release(), lbox.release(), ibox.release(), keys.release(), vals.release(), keys_s.release(), vals_s.release();
        lkey.release(), scene_ord.release(), parent_i.release(), parent_l.release(), flag.release();
        frontier[0].release(), frontier[1].release(), lvl.release(), child.release(), temp.release(), acc.release();
    }};
    HIP_TRY(pb.alloc(n));
    HIP_TRY(keys.alloc(n));
    HIP_TRY(vals.alloc(n));
    HIP_TRY(keys_s.alloc(n));
    HIP_TRY(vals_s.alloc(n));
    HIP_TRY(scene_ord.alloc(6));
    const int ord_init[6] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN, INT32_MIN};
    HIP_TRY(hipMemcpy(scene_ord.p, ord_init, sizeof(ord_init), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_prim_boxes, grid(n), blk, 0, stream, sc.prims.p, n, pb.p, scene_ord.p);
    hipLaunchKernelGGL(k_morton, grid(n), blk, 0, stream, pb.p, n, scene_ord.p, keys.p, vals.p);
    size_t temp_bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys.p, keys_s.p, vals.p, vals_s.p, (size_t)n, 0, 63, stream));
    HIP_TRY(temp.alloc(temp_bytes));
    HIP_TRY(rocprim::radix_sort_pairs(temp.p, temp_bytes, keys.p, keys_s.p, vals.p, vals_s.p, (size_t)n, 0, 63, stream));

    HIP_TRY(lbox.alloc(n_leaves));
    HIP_TRY(lkey.alloc(n_leaves));
    HIP_TRY(ibox.alloc(n_leaves));
    HIP_TRY(child.alloc(n_leaves));
    HIP_TRY(parent_i.alloc(n_leaves));
    HIP_TRY(parent_l.alloc(n_leaves));
    HIP_TRY(flag.alloc(n_leaves));
    HIP_TRY(hipMemsetAsync(flag.p, 0, flag.bytes(), stream));
    hipLaunchKernelGGL(k_leaves, grid(n_leaves), blk, 0, stream, pb.p, keys_s.p, vals_s.p, n, leaf_size, n_leaves, lbox.p, lkey.p);
    hipLaunchKernelGGL(k_hierarchy, grid(n_leaves - 1), blk, 0, stream, lkey.p, n_leaves, child.p, parent_i.p, parent_l.p);
    hipLaunchKernelGGL(k_refit, grid(n_leaves), blk, 0, stream, n_leaves, child.p, parent_i.p, parent_l.p, lbox.p, ibox.p, flag.p);

    // collapse to 4-wide nodes, breadth-first, one launch per level
    HIP_TRY(sc.nodes.alloc(n_leaves));
    HIP_TRY(frontier[0].alloc(n_leaves));
    HIP_TRY(frontier[1].alloc(n_leaves));
    HIP_TRY(lvl.alloc(MAX_LEVELS + 2));
    HIP_TRY(hipMemsetAsync(lvl.p, 0, lvl.bytes(), stream));
    hipLaunchKernelGGL(k_fill_int, dim3(1), blk, 0, stream, lvl.p, 1, 1);           // one node on level 0 ...
    hipLaunchKernelGGL(k_fill_int, dim3(1), blk, 0, stream, frontier[0].p, 1, 0);   // ... made from BVH2 node 0
    const int cgrid = std::max(1, std::min((n_leaves + BLK - 1) / BLK, 2048));
    for (int level = 0; level < MAX_LEVELS; level++)
        hipLaunchKernelGGL(k_collapse, dim3(cgrid), blk, 0, stream, level, frontier[level & 1].p, frontier[(level + 1) & 1].p,
                           lvl.p, child.p, ibox.p, lbox.p, leaf_size, n, sc.nodes.p);
    int lvl_h[MAX_LEVELS + 2];
    int ord_h[6];
    HIP_TRY(hipMemcpyAsync(lvl_h, lvl.p, sizeof(lvl_h), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(ord_h, scene_ord.p, sizeof(ord_h), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (lvl_h[MAX_LEVELS] != 0) {
        sc.nodes.release();
        return 1;  // deeper than the traversal stack allows
    }
    int64_t n_nodes = 0;
    int depth = 0;
    for (int k = 0; k < MAX_LEVELS; k++)
        if (lvl_h[k] > 0) n_nodes += lvl_h[k], depth = k + 1;
    h.stats = WideBvhStats{};
    h.stats.n_nodes = n_nodes, h.stats.n_prims = n, h.stats.depth = depth;
    h.root_child = 0;
    sc.nodes.n = (size_t)n_nodes;  // the tail of the allocation is unused

    // compressed nodes on the scene grid (same fall-back rule as the host path)
    h.q_inflation = 1.0;
    bool use_q = false;
    if (compressed_ok) {
        double lo[3], hi[3];
        for (int a = 0; a < 3; a++) lo[a] = ord2f(ord_h[a]), hi[a] = ord2f(ord_h[3 + a]);
        const QGrid g = make_qgrid(lo, hi);
        HIP_TRY(sc.qnodes.alloc((size_t)n_nodes));
        HIP_TRY(acc.alloc(2));
        HIP_TRY(hipMemsetAsync(acc.p, 0, acc.bytes(), stream));
        hipLaunchKernelGGL(k_quantise, grid(n_nodes), blk, 0, stream, sc.nodes.p, (int)n_nodes, g, sc.qnodes.p, acc.p);
        double acc_h[2] = {0, 0};
        HIP_TRY(hipMemcpy(acc_h, acc.p, sizeof(acc_h), hipMemcpyDeviceToHost));
        h.q_inflation = acc_h[1] > 0 ? acc_h[0] / acc_h[1] : 1.0;
        use_q = compressed_forced || h.q_inflation <= 1.10;
        for (int a = 0; a < 3; a++) h.grid_lo[a] = g.lo[a], h.grid_step[a] = g.step[a];
        if (use_q) sc.nodes.release();
        else sc.qnodes.release();
    }
    // records into leaf order
    DevBuf<PrimRec<float>> prims_sorted;
    HIP_TRY(prims_sorted.alloc(n));
    hipLaunchKernelGGL((k_permute<PrimRec<float>>), grid(n), blk, 0, stream, sc.prims.p, vals_s.p, n, prims_sorted.p);
    HIP_TRY(hipStreamSynchronize(stream));
    sc.prims.release();
    sc.prims = prims_sorted;  // DevBuf is a plain handle: ownership moves
    HIP_TRY(hipGetLastError());
    return TAKE_OK;
}
// TAKE_INSTANCES_FLATTEN: the description with every placement expanded to a world-space mesh of its own — the geometry
// an instanced render is specified to equal (TakeInstance, include/take_hip.h).  Placement i becomes mesh n_meshes + i:
// positions M[:, :3] p + M[:, 3] and normals n^T L^-1 (not re-normalised: interpolation commutes with the linear map
// only then; the interpolated normal is normalised at the hit) in double, on `threads` host threads; the prototype's
// index and uv arrays are shared, not copied.  The shape arrays grow by the placements' faces in placement order, so
// shape ids are the two-level scene's (n_shapes + faces of the preceding placements + face).
struct FlattenedInstances {
    std::vector<TakeMesh> meshes;
    std::vector<std::vector<double>> arrays;
    std::vector<int32_t> kind, ref, face, area_light;
    int expand(TakeSceneDesc &d, int threads) {
        if (d.n_instances <= 0) return TAKE_OK;
        if (!d.instances) return fail(TAKE_E_INVALID, "n_instances > 0 but instances is null");
        int64_t extra = 0;
        for (int64_t i = 0; i < d.n_instances; i++) {
            const TakeInstance &in = d.instances[i];
            if (in.mesh_id < 0 || in.mesh_id >= d.n_meshes) return fail(TAKE_E_INVALID, "instance " + std::to_string(i) + ": bad mesh index");
            const TakeMesh &m = d.meshes[in.mesh_id];
            if (m.flags & TAKE_MESH_DEVICE_ARRAYS) return fail(TAKE_E_INVALID, "instance " + std::to_string(i) + ": flattening reads the prototype on the host; it is a device-array mesh");
            if (m.n_vertices < 0 || m.n_faces < 0 || (m.n_faces > 0 && (!m.positions || !m.indices))) return fail(TAKE_E_INVALID, "instance " + std::to_string(i) + ": bad prototype mesh");
            if (in.material_id >= d.n_materials) return fail(TAKE_E_INVALID, "instance " + std::to_string(i) + ": bad material index");
            extra += m.n_faces;
        }
        if (d.n_shapes + extra >= ((int64_t)1 << 31) || (int64_t)d.n_meshes + d.n_instances >= ((int64_t)1 << 31))
            return fail(TAKE_E_INVALID, "flattened scene too large (" + std::to_string(d.n_shapes + extra) + " shapes)");
        meshes.assign(d.meshes, d.meshes + d.n_meshes);
        meshes.resize((size_t)d.n_meshes + (size_t)d.n_instances);
        arrays.resize(2 * (size_t)d.n_instances);
        std::string err;
        std::mutex mu;
        auto work = [&](int64_t lo, int64_t hi) {
            try {
            for (int64_t i = lo; i < hi; i++) {
                const TakeInstance &in = d.instances[i];
                const TakeMesh &m = d.meshes[in.mesh_id];
                const double *x = in.xform;
                const double a00 = x[0], a01 = x[1], a02 = x[2], a10 = x[4], a11 = x[5], a12 = x[6], a20 = x[8], a21 = x[9], a22 = x[10];
                std::vector<double> &pos = arrays[2 * (size_t)i], &nrm = arrays[2 * (size_t)i + 1];
                pos.resize(3 * (size_t)m.n_vertices);
                for (int64_t v = 0; v < m.n_vertices; v++) {
                    const double px = m.positions[3 * v], py = m.positions[3 * v + 1], pz = m.positions[3 * v + 2];
                    pos[3 * v + 0] = a00 * px + a01 * py + a02 * pz + x[3];
                    pos[3 * v + 1] = a10 * px + a11 * py + a12 * pz + x[7];
                    pos[3 * v + 2] = a20 * px + a21 * py + a22 * pz + x[11];
                }
                if (m.normals) {
                    const double det = a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20);
                    if (!(std::fabs(det) > 1e-300)) {
                        std::lock_guard<std::mutex> lock(mu);
                        err = "instance " + std::to_string(i) + ": singular transform";
                        return;
                    }
                    const double inv[9] = {(a11 * a22 - a12 * a21) / det, (a02 * a21 - a01 * a22) / det, (a01 * a12 - a02 * a11) / det,
                                           (a12 * a20 - a10 * a22) / det, (a00 * a22 - a02 * a20) / det, (a02 * a10 - a00 * a12) / det,
                                           (a10 * a21 - a11 * a20) / det, (a01 * a20 - a00 * a21) / det, (a00 * a11 - a01 * a10) / det};
                    nrm.resize(3 * (size_t)m.n_vertices);
                    for (int64_t v = 0; v < m.n_vertices; v++) {
                        const double nx = m.normals[3 * v], ny = m.normals[3 * v + 1], nz = m.normals[3 * v + 2];
                        nrm[3 * v + 0] = nx * inv[0] + ny * inv[3] + nz * inv[6];  // (n^T L^-1)
                        nrm[3 * v + 1] = nx * inv[1] + ny * inv[4] + nz * inv[7];
                        nrm[3 * v + 2] = nx * inv[2] + ny * inv[5] + nz * inv[8];
                    }
                }
                TakeMesh &o = meshes[(size_t)d.n_meshes + (size_t)i];
                o = m;
                o.positions = pos.data();
                o.normals = m.normals ? nrm.data() : nullptr;
                o.material_id = in.material_id >= 0 ? in.material_id : m.material_id;
            }
            } catch (const std::exception &) {  // (an exception must not leave a worker thread)
                std::lock_guard<std::mutex> lock(mu);
                err = "out of host memory while flattening the instances";
            }
        };
        const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(threads, d.n_instances));
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; t++) pool.emplace_back(work, d.n_instances * t / nt, d.n_instances * (t + 1) / nt);
        for (auto &th : pool) th.join();
        if (!err.empty()) return fail(TAKE_E_INVALID, err);
        const size_t n0 = (size_t)d.n_shapes, n1 = n0 + (size_t)extra;
        kind.resize(n1), ref.resize(n1), face.resize(n1), area_light.resize(n1);
        if (n0) {
            std::memcpy(kind.data(), d.shape_kind, n0 * 4), std::memcpy(ref.data(), d.shape_ref, n0 * 4);
            std::memcpy(face.data(), d.shape_face, n0 * 4), std::memcpy(area_light.data(), d.shape_area_light, n0 * 4);
        }
        size_t at = n0;
        for (int64_t i = 0; i < d.n_instances; i++) {
            const int64_t nf = d.meshes[d.instances[i].mesh_id].n_faces;
            for (int64_t k = 0; k < nf; k++, at++) kind[at] = 1, ref[at] = (int32_t)(d.n_meshes + i), face[at] = (int32_t)k, area_light[at] = -1;
        }
        d.meshes = meshes.data(), d.n_meshes = (int32_t)meshes.size();
        d.shape_kind = kind.data(), d.shape_ref = ref.data(), d.shape_face = face.data(), d.shape_area_light = area_light.data();
        d.n_shapes = (int64_t)n1;
        d.n_instances = 0, d.instances = nullptr;
        return TAKE_OK;
    }
};

// Device-array meshes (TAKE_MESH_DEVICE_ARRAYS, take_hip_mesh_from_ply) in a scene description: the host side of the
// build — index validation, the face / normal / uv tables, the SAH builder — reads host copies, staged here.
struct StagedMeshes {
    bool any = false;
    std::vector<TakeMesh> meshes;            // what the build sees (d.meshes points here)
    std::vector<const double *> d_positions;  // per mesh: its device positions while they have not been staged
    std::vector<std::vector<double>> reals;
    std::vector<std::vector<int32_t>> ints;
    hipError_t real(const double *&p, size_t n) {
        if (!p || n == 0) return hipSuccess;
        reals.emplace_back(n);
        const hipError_t e = hipMemcpy(reals.back().data(), p, n * sizeof(double), hipMemcpyDeviceToHost);
        p = reals.back().data();
        return e;
    }
    // all_positions: the host builder will run (it reads every vertex).  Otherwise only the meshes an area light
    // sits on bring their positions to the host (the light records are made there); the device build copies the
    // others device-to-device.
    int stage(TakeSceneDesc &d, bool all_positions) {
        for (int i = 0; i < d.n_meshes; i++) any = any || (d.meshes && (d.meshes[i].flags & TAKE_MESH_DEVICE_ARRAYS));
        if (!any) return TAKE_OK;
        meshes.assign(d.meshes, d.meshes + d.n_meshes);
        d_positions.assign((size_t)d.n_meshes, nullptr);
        std::vector<char> emissive((size_t)d.n_meshes, 0);
        for (int i = 0; i < d.n_lights; i++) {
            const TakeLight &l = d.lights[i];
            if (l.kind != 1 || l.shape_id < 0 || l.shape_id >= d.n_shapes || d.shape_kind[l.shape_id] != 1) continue;
            const int32_t mi = d.shape_ref[l.shape_id];
            if (mi >= 0 && mi < d.n_meshes) emissive[mi] = 1;
        }
        for (int i = 0; i < d.n_meshes; i++) {
            TakeMesh &m = meshes[i];
            if (!(m.flags & TAKE_MESH_DEVICE_ARRAYS)) continue;
            if (m.n_vertices < 0 || m.n_faces < 0) return fail(TAKE_E_INVALID, "negative mesh size");
            if (all_positions || emissive[i]) HIP_TRY(real(m.positions, 3 * (size_t)m.n_vertices));
            else d_positions[i] = m.positions;
            HIP_TRY(real(m.normals, 3 * (size_t)m.n_vertices));
            HIP_TRY(real(m.uvs, 2 * (size_t)m.n_vertices));
            if (m.indices && m.n_faces > 0) {
                ints.emplace_back(3 * (size_t)m.n_faces);
                HIP_TRY(hipMemcpy(ints.back().data(), m.indices, ints.back().size() * sizeof(int32_t), hipMemcpyDeviceToHost));
                m.indices = ints.back().data();
            }
            m.flags &= ~TAKE_MESH_DEVICE_ARRAYS;
        }
        d.meshes = meshes.data();
        return TAKE_OK;
    }
    // the device build gave up (tree too deep): the host builder needs every vertex after all
    int ensure_positions() {
        for (size_t i = 0; i < meshes.size(); i++) {
            if (!d_positions[i]) continue;
            HIP_TRY(real(meshes[i].positions, 3 * (size_t)meshes[i].n_vertices));
            d_positions[i] = nullptr;
        }
        return TAKE_OK;
    }
};
thread_local StagedMeshes *t_staged = nullptr;  // set by scene_create while upload_scene runs on a staged description

// Primitive records on the device from the caller's arrays (tk_build_gpu.h::k_make_prims): the mesh positions go up as
// they are (double, one copy per mesh, no host staging), the face indices are the validated concatenation the shading
// side keeps anyway (sc.face_idx, uploaded here), the four shape arrays go up as they are.
// device_positions: per mesh, positions that are in device memory already (a mesh take_hip_mesh_from_ply decoded; the
// description then holds host copies of what the host side validates and tabulates, not of these), or null
int make_prims_on_device(SceneT<float> &sc, const TakeSceneDesc &d, const double *const *device_positions) {
    using namespace lbvh;
    const int n = (int)d.n_shapes;
    HostScene<float> &h = sc.host;
    std::vector<MeshSrc> ms(d.n_meshes);
    int64_t nv = 0;
    for (int i = 0; i < d.n_meshes; i++) {
        const MeshInfo &mi = h.meshes[i];
        ms[i] = MeshSrc{nv, mi.fbase, mi.material, h.materials[mi.material].tag, (mi.nbase >= 0 || mi.uvbase >= 0) ? 1 : 0};
        nv += d.meshes[i].n_vertices;
    }
    std::vector<SphereSrc> ss(d.n_spheres);
    for (int i = 0; i < d.n_spheres; i++) {
        const TakeSphere &s = d.spheres[i];
        ss[i] = SphereSrc{{s.center[0], s.center[1], s.center[2]}, s.radius, s.material_id, h.materials[s.material_id].tag};
    }
    DevBuf<double> d_pos;
    DevBuf<int32_t> d_kind, d_ref, d_face, d_al;
    DevBuf<MeshSrc> d_ms;
    DevBuf<SphereSrc> d_ss;
    struct Cleanup {
        std::function<void()> f;
        ~Cleanup() { f(); }
    } cleanup{[&] { d_pos.release(), d_kind.release(), d_ref.release(), d_face.release(), d_al.release(), d_ms.release(), d_ss.release(); }};
    HIP_TRY(d_pos.alloc(3 * (size_t)std::max<int64_t>(nv, 1)));
    PinnedUploads pin;
    for (int i = 0; i < d.n_meshes; i++) {
        if (d.meshes[i].n_vertices <= 0) continue;
        const size_t bytes = sizeof(double) * 3 * (size_t)d.meshes[i].n_vertices;
        // a mesh decoded on the device (take_hip_mesh_from_ply): its positions never were on the host
        if (device_positions && device_positions[i])
            HIP_TRY(hipMemcpyAsync(d_pos.p + 3 * ms[i].pos_off, device_positions[i], bytes, hipMemcpyDeviceToDevice, pin.stream));
        else
            HIP_TRY(pin.copy(d_pos.p + 3 * ms[i].pos_off, d.meshes[i].positions, bytes));
    }
    HIP_TRY(sc.face_idx.upload(h.face_idx));
    auto up = [&](DevBuf<int32_t> &b, const int32_t *src) -> hipError_t {
        hipError_t e = b.alloc((size_t)n);
        return e != hipSuccess ? e : pin.copy(b.p, src, sizeof(int32_t) * (size_t)n);
    };
    HIP_TRY(up(d_kind, d.shape_kind));
    HIP_TRY(up(d_ref, d.shape_ref));
    HIP_TRY(up(d_face, d.shape_face));
    HIP_TRY(up(d_al, d.shape_area_light));
    HIP_TRY(d_ms.upload(ms));
    HIP_TRY(d_ss.upload(ss));
    HIP_TRY(sc.prims.alloc((size_t)n));
    hipLaunchKernelGGL(k_make_prims, dim3((unsigned)((n + BLK - 1) / BLK)), dim3(BLK), 0, nullptr, d_kind.p, d_ref.p, d_face.p, d_al.p,
                       d_ms.p, d_pos.p, sc.face_idx.p, d_ss.p, n, sc.prims.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(pin.finish());  // (the kernel is behind the copies on the same stream)
    if (std::getenv("TAKE_HIP_VERBOSE"))
        std::fprintf(stderr, "[take_hip] scene_create: uploads pinned in place %.1f MB, pageable %.1f MB\n", pin.pinned_bytes / 1e6, pin.plain_bytes / 1e6);
    return TAKE_OK;
}
template <class R> int make_prims_on_device(SceneT<R> &, const TakeSceneDesc &, const double *const *) { return 1; }

// phase timer of scene_create (TAKE_HIP_VERBOSE=1 prints the phases to stderr)
struct PhaseClock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    bool on = std::getenv("TAKE_HIP_VERBOSE") != nullptr;
    void lap(const char *what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[take_hip] scene_create: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

template <class R> int upload_scene(TakeScene *ts, const TakeSceneDesc &desc, const TakeBuildOpts &opts) {
    SceneT<R> &sc = pick<R>(ts);
    PhaseClock clock;
    int threads = opts.bvh_threads > 0 ? opts.bvh_threads : (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    int max_leaf = opts.max_leaf_size;
    if (max_leaf <= 0 && std::getenv("TAKE_HIP_MAX_LEAF")) max_leaf = std::atoi(std::getenv("TAKE_HIP_MAX_LEAF"));  // tuning knob
    // lanes per ray: one (TQ_GROUP).  Round 1 measured quad 143.6 / pair 121.5 / one ray per lane 128.8 ms of closest-hit
    // time per 8.3 M samples on full-width nodes; on the 64-byte nodes one ray per lane is 9 % (f32) and 22 % (f64)
    // ahead of the pair kernel.  The other group sizes stay behind -DTQ_GROUP for comparison builds.
    sc.group = TQ_GROUP;
    const char *fmt_env = std::getenv("TAKE_HIP_NODES");
    const std::string fmt = fmt_env ? fmt_env : "";
    // device build: f32 scenes with enough primitives to make a tree; otherwise (and as its fall-back) the host SAH build
    // builder: AUTO = host SAH (best trees) up to 4M shapes, device LBVH beyond (f32): at 10M triangles the host build
    // is 6 s of setup against 0.2 s, for 2-6 % of traversal speed
    const bool want_device = opts.builder == TAKE_BUILDER_DEVICE_LBVH ||
                             (opts.builder == TAKE_BUILDER_AUTO && desc.n_shapes >= TAKE_AUTO_DEVICE_BUILD_SHAPES);
    bool on_device = want_device && sizeof(R) == 4 && desc.n_shapes >= 8 && desc.n_instances == 0;
    std::string err = prepare_scene<R>(desc, max_leaf, threads, sc.host, on_device ? PREP_TABLES : PREP_ALL, opts.burley_lobes != 0);
    if (!err.empty()) return fail(TAKE_E_INVALID, err);
    clock.lap(on_device ? "host validation + tables" : "host records + SAH build");
    HostScene<R> &h = sc.host;
    bool use_q = false;
    if (on_device) {
        int rc = TAKE_OK;
        if constexpr (sizeof(R) == 4) {
            rc = make_prims_on_device(sc, desc, t_staged && t_staged->any ? t_staged->d_positions.data() : nullptr);
            clock.lap("mesh arrays -> HBM, records");
            if (!rc) rc = build_bvh_device(sc, max_leaf, sc.group <= 2 && fmt != "wide", fmt == "q16");
        } else {
            rc = 1;
        }
        if (rc == 1) {  // not buildable on the device (tree too deep): do it on the host after all
            on_device = false;
            if (t_staged && t_staged->any) {
                const int rs = t_staged->ensure_positions();
                if (rs) return rs;
            }
            err = prepare_scene<R>(desc, max_leaf, threads, sc.host, PREP_ALL, opts.burley_lobes != 0);
            if (!err.empty()) return fail(TAKE_E_INVALID, err);
        } else if (rc != TAKE_OK) {
            return rc;
        } else {
            use_q = sc.qnodes.p != nullptr;
        }
    }
    if (!on_device) {
        HIP_TRY(sc.prims.upload(h.prims));
        clock.lap("primitive records -> HBM");
    }
    if (!on_device) {
        use_q = sc.group <= 2 && (!h.qnodes.empty() || !h.qnodes8.empty());  // compressed nodes (not in the quad kernel)
        if (!h.qnodes8.empty()) HIP_TRY(sc.qnodes8.upload(h.qnodes8));
        else if (use_q) HIP_TRY(sc.qnodes.upload(h.qnodes));
        else HIP_TRY(sc.nodes.upload(h.nodes));
    }
    sc.built_on_device = on_device;
    clock.lap(on_device ? "device LBVH build" : "nodes -> HBM");
    // the trace kernels address nodes and primitive records with 32-bit byte offsets (full-rate integer math)
    {
        const uint64_t node_bytes = (uint64_t)h.stats.n_nodes * (sc.qnodes8.p ? sizeof(QNode8) : (use_q ? sizeof(QNode4) : sizeof(Node4<R>)));
        const uint64_t prim_bytes = (uint64_t)sc.prims.n * sizeof(PrimRec<R>);
        if (node_bytes >= (1ull << 32) || prim_bytes >= (1ull << 32))
            return fail(TAKE_E_INVALID, "scene too large for the 32-bit record offsets of the trace kernels (" +
                                            std::to_string(sc.prims.n) + " primitives, " + std::to_string(h.stats.n_nodes) + " nodes)");
    }
    // (ShapeInfo stays on the host: every kernel reads the shading side of a primitive from its own record)
    HIP_TRY(sc.meshes.upload(h.meshes));
    if (!on_device) HIP_TRY(sc.face_idx.upload(h.face_idx));  // (device build: already there, k_make_prims read it)
    HIP_TRY(sc.normals.upload(h.normals));
    HIP_TRY(sc.uvs.upload(h.uvs));
    HIP_TRY(sc.texels.upload(h.texels));
    HIP_TRY(sc.materials.upload(h.materials));
    HIP_TRY(sc.images.upload(h.images));
    HIP_TRY(sc.lights.upload(h.lights));
    HIP_TRY(sc.light_pmf.upload(h.light_pmf));
    HIP_TRY(sc.light_cdf.upload(h.light_cdf));
    HIP_TRY(sc.inst_trace.upload(h.inst_trace));
    HIP_TRY(sc.inst_shade.upload(h.inst_shade));
    HIP_TRY(sc.env_marginal.upload(h.env_marginal));
    HIP_TRY(sc.env_conditional.upload(h.env_conditional));
    HIP_TRY(sc.env_guide_m.upload(h.env_guide_m));
    HIP_TRY(sc.env_guide_c.upload(h.env_guide_c));
    DeviceScene<R> &d = sc.dev;
    d = h.view();
    d.n_nodes = (int32_t)h.stats.n_nodes;
    d.nodes = sc.nodes.p;
    d.qnodes = use_q ? sc.qnodes.p : nullptr;
    d.qnodes8 = sc.qnodes8.p;
    d.prims = sc.prims.p;
    d.shapes = nullptr;
    d.meshes = sc.meshes.p;
    d.face_idx = sc.face_idx.p;
    d.normals = sc.normals.p;
    d.uvs = sc.uvs.p;
    d.texels = sc.texels.p;
    d.materials = sc.materials.p;
    d.images = sc.images.p;
    d.lights = sc.lights.p;
    d.inst_trace = sc.inst_trace.p;
    d.inst_shade = sc.inst_shade.p;
    d.light_pmf = sc.light_pmf.p;
    d.light_cdf = sc.light_cdf.p;
    d.env.marginal = sc.env_marginal.p;
    d.env.conditional = sc.env_conditional.p;
    d.env.guide_m = sc.env_guide_m.p;
    d.env.guide_c = sc.env_guide_c.p;
    HIP_TRY(sc.qwords.alloc(Q_NUM_WORDS + 2 * N_SORT_KEYS));
    HIP_TRY(hipMemset(sc.qwords.p, 0, sc.qwords.bytes()));
    HIP_TRY(sc.counters.alloc(C_NUM_WORDS));
    HIP_TRY(hipMemset(sc.counters.p, 0, sc.counters.bytes()));
    // persistent trace grid: resident blocks of the heaviest trace kernel x CUs
    int per_cu = 0;
    const bool w8 = sc.qnodes8.p != nullptr;
    const int groups_per_block = GroupGeom<TQ_GROUP>::GROUPS;
    const int spill_levels = w8 ? GroupGeom<TQ_GROUP == 1 ? 1 : TQ_GROUP, TQ_GROUP == 1 ? 8 : 4>::SPILL : GroupGeom<TQ_GROUP>::SPILL;
    if constexpr (TQ_GROUP == 1) {
        if (w8 && sc.inst_trace.n) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_group<R, 1, false, false, PathIo<R>, true, true, 8>, TQ_BLOCK, 0));
        else if (w8) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_group<R, 1, false, false, PathIo<R>, true, false, 8>, TQ_BLOCK, 0));
    }
    if (!w8) {
        if (use_q) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_group<R, TQ_GROUP, false, false, PathIo<R>, true>, TQ_BLOCK, 0));
        else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_group<R, TQ_GROUP, false, false, PathIo<R>>, TQ_BLOCK, 0));
        if (sc.inst_trace.n) {  // two-level scenes run the INST instances: size the persistent grid for them
            if (use_q) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_group<R, TQ_GROUP, false, false, PathIo<R>, true, true>, TQ_BLOCK, 0));
            else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_trace_group<R, TQ_GROUP, false, false, PathIo<R>, false, true>, TQ_BLOCK, 0));
        }
    }
    per_cu = std::max(1, std::min(per_cu, 8));
    if (const char *e = std::getenv("TAKE_HIP_TRACE_BLOCKS")) per_cu = std::max(1, std::min(per_cu, std::atoi(e)));  // experiment: leave room for a concurrent kernel
    sc.trace_grid = ts->num_cus * per_cu;
    sc.spill_stride = (int64_t)sc.trace_grid * groups_per_block;
    HIP_TRY(sc.spill.alloc((size_t)sc.spill_stride * spill_levels));
    clock.lap("shading tables -> HBM, grid");
    // everything the kernels read is in HBM now; the host keeps the small tables (camera, material tags, tree
    // statistics) and drops the copies of the large arrays (1.1 GB at 10M triangles)
    h.nodes = {}, h.qnodes = {}, h.qnodes8 = {}, h.nodes8 = {}, h.prims = {}, h.shapes = {}, h.face_idx = {}, h.normals = {}, h.uvs = {}, h.texels = {};
    h.inst_trace = {}, h.inst_shade = {};
    return TAKE_OK;
}

// Path-state / queue / framebuffer workspace of a scene, grown on demand.  A failed allocation leaves the scene
// WITHOUT a workspace (capacity 0, every buffer released) and returns TAKE_E_NOMEM: the next render allocates afresh
// instead of trusting a stale capacity over null pointers.
template <class R> void release_workspace(SceneT<R> &sc) {
    sc.state_r.release(), sc.queue[0].release(), sc.queue[1].release(), sc.shadow_queue.release();
    sc.sorted_queue.release(), sc.sort_keys.release();
    sc.capacity = 0;
}
template <class R> int ensure_workspace(SceneT<R> &sc, int64_t slots, int64_t npix) {
    if (slots > sc.capacity) {
        release_workspace(sc);
        const bool ok = sc.state_r.alloc((size_t)PATH_REC * slots) == hipSuccess && sc.queue[0].alloc(slots) == hipSuccess &&
                        sc.queue[1].alloc(slots) == hipSuccess && sc.shadow_queue.alloc(slots) == hipSuccess &&
                        sc.sorted_queue.alloc(slots) == hipSuccess && sc.sort_keys.alloc(slots) == hipSuccess;
        if (!ok) {
            release_workspace(sc);
            return fail(TAKE_E_NOMEM, "out of device memory for " + std::to_string(slots) + " path slots (" +
                                          std::to_string((size_t)slots * (PATH_REC * sizeof(R) + 17) >> 20) + " MiB)");
        }
        sc.capacity = slots;
    }
    if ((int64_t)sc.accum.n < 3 * npix) {
        if (sc.accum.alloc(3 * npix) != hipSuccess || sc.out.alloc(3 * npix) != hipSuccess) {
            sc.accum.release(), sc.out.release();
            return fail(TAKE_E_NOMEM, "out of device memory for the framebuffer");
        }
    }
    return TAKE_OK;
}

__global__ void k_prep(int32_t *q, int next) {
    const int t = threadIdx.x;
    if (t == 0) {
        q[Q_HEAD_CLOSEST] = 0;
        q[Q_HEAD_SHADOW] = 0;
        q[Q_N_SHADOW] = 0;
        q[next ? Q_N_EXT1 : Q_N_EXT0] = 0;
    }
    if (t < 2 * N_SORT_KEYS) q[Q_NUM_WORDS + t] = 0;
}
__global__ void k_set_word(int32_t *q, int word, int32_t value) { q[word] = value; }

int rows_of(int height, int first, int stride, int32_t *rows_out) {
    const int n_strips = (height + TILE_ROWS - 1) / TILE_ROWS;
    int n = 0;
    std::vector<int> ys;
    for (int s = first; s < n_strips; s += stride)
        for (int y = s * TILE_ROWS; y < std::min(height, (s + 1) * TILE_ROWS); y++) ys.push_back(y);
    n = (int)ys.size();
    if (rows_out)
        for (int j = 0; j < n; j++) rows_out[j] = height - 1 - ys[n - 1 - j];  // increasing image row
    return n;
}

struct Timer {
    TakeScene *ts;
    hipStream_t stream;
    bool on;
    hipError_t err = hipSuccess;  // first failure of an event call; render_impl reports it instead of bogus times
    void begin(int which) {
        if (!on) return;
        hipEvent_t a = ts->events.get(), b = ts->events.get();
        const hipError_t e = (a && b) ? hipEventRecord(a, stream) : hipErrorOutOfMemory;
        if (e != hipSuccess && err == hipSuccess) err = e;
        ts->timed.push_back({which, {a, b}});
    }
    void end() {
        if (!on) return;
        const hipEvent_t b = ts->timed.back().second.second;
        const hipError_t e = b ? hipEventRecord(b, stream) : hipErrorOutOfMemory;
        if (e != hipSuccess && err == hipSuccess) err = e;
    }
};

// launch the trace kernel instance for (lanes per ray, any-hit, counting)
template <class R, class Io>
void launch_trace(int group, bool any, bool count, dim3 grid, hipStream_t stream, const DeviceScene<R> &dev, const Io &io,
                  const int32_t *n_ptr, int32_t n_direct, int32_t *head, unsigned long long *counters, int counter_word,
                  StackSpill spill) {
#define TK_LAUNCH(A, C, Q, I, W)                                                                                              \
    hipLaunchKernelGGL((k_trace_group<R, TQ_GROUP, A, C, Io, Q, I, W>), grid, dim3(TQ_BLOCK), 0, stream, dev, io, n_ptr, n_direct, head, \
                       counters, counter_word, spill)
#define TK_LAUNCH_AC(Q, I, W)                          \
    do {                                            \
        if (any && count) TK_LAUNCH(true, true, Q, I, W);       \
        else if (any) TK_LAUNCH(true, false, Q, I, W);          \
        else if (count) TK_LAUNCH(false, true, Q, I, W);        \
        else TK_LAUNCH(false, false, Q, I, W);                  \
    } while (0)
    (void)group;  // one instantiated group size (TQ_GROUP)
    const bool q = dev.qnodes != nullptr, two_level = dev.inst_trace != nullptr;
    if constexpr (TQ_GROUP == 1) {
        if (dev.qnodes8 != nullptr) {  // the 8-wide tree
            if (two_level) TK_LAUNCH_AC(true, true, 8);
            else TK_LAUNCH_AC(true, false, 8);
            return;
        }
    }
    if (q && two_level) TK_LAUNCH_AC(true, true, 4);
    else if (q) TK_LAUNCH_AC(true, false, 4);
    else if (two_level) TK_LAUNCH_AC(false, true, 4);
    else TK_LAUNCH_AC(false, false, 4);
#undef TK_LAUNCH_AC
#undef TK_LAUNCH
}

template <class R> struct ShadeArgs {
    DeviceScene<R> dev;
    RenderParams<R> rp;
    PathState<R> st;
    const int32_t *queue;
    const int32_t *n_cur;
    const int32_t *tag_count;
    int32_t *next_queue, *n_next, *shadow_queue, *n_shadow;
    int k;
    unsigned long long *counters;
    int grid;
    hipStream_t stream;
    float *to_f32;  // mixed precision, last exact round: the f32 records the continuing paths are converted into (else null)
};
template <class R, int TAG> void launch_shade_tag(const ShadeArgs<R> &a) {
    if (a.rp.integrator != 0)
        hipLaunchKernelGGL((k_shade<R, TAG, true>), dim3(a.grid), dim3(BLOCK), 0, a.stream, a.dev, a.rp, a.st, a.queue, a.n_cur,
                           a.tag_count, a.next_queue, a.n_next, a.shadow_queue, a.n_shadow, a.k, a.counters, a.to_f32);
    else
        hipLaunchKernelGGL((k_shade<R, TAG, false>), dim3(a.grid), dim3(BLOCK), 0, a.stream, a.dev, a.rp, a.st, a.queue, a.n_cur,
                           a.tag_count, a.next_queue, a.n_next, a.shadow_queue, a.n_shadow, a.k, a.counters, a.to_f32);
}
template <class R> void launch_shade(int tag, const ShadeArgs<R> &a) {
    switch (tag) {
        case 0: launch_shade_tag<R, 0>(a); break;
        case 1: launch_shade_tag<R, 1>(a); break;
        case 2: launch_shade_tag<R, 2>(a); break;
        case 3: launch_shade_tag<R, 3>(a); break;
        case 4: launch_shade_tag<R, 4>(a); break;
        case 5: launch_shade_tag<R, 5>(a); break;
        case 6: launch_shade_tag<R, 6>(a); break;
        case 7: launch_shade_tag<R, 7>(a); break;
        case 8: launch_shade_tag<R, 8>(a); break;
        case 9: launch_shade_tag<R, 9>(a); break;
        case 10: launch_shade_tag<R, 10>(a); break;
        case 11: launch_shade_tag<R, 11>(a); break;
        case 12: launch_shade_tag<R, 12>(a); break;
        case 13: launch_shade_tag<R, 13>(a); break;
        case 14: launch_shade_tag<R, 14>(a); break;
        case 15: launch_shade_tag<R, 15>(a); break;
        case 16: launch_shade_tag<R, 16>(a); break;
        default: launch_shade_tag<R, TAG_MISS>(a); break;
    }
}

// Debug aid (TAKE_HIP_DUMP_SLOT=<slot>): print one path's state after every kernel of a round.
template <class R> void dump_slot(const PathState<R> &st, int64_t slot, const char *tag, int k, hipStream_t stream) {
    (void)hipStreamSynchronize(stream);
    std::fprintf(stderr, "[slot %lld] k=%d %s R:", (long long)slot, k, tag);
    R rec[PATH_REC];
    (void)hipMemcpy(rec, st.r + slot * PATH_REC, sizeof rec, hipMemcpyDeviceToHost);
    for (int c = 0; c < PATH_REC; c++)
        if (c != S_HIT && c != S_CTR && c != S_FLAGS) std::fprintf(stderr, " %.17g", (double)rec[c]);
    std::fprintf(stderr, " I:");
    for (int c : {(int)S_HIT, (int)S_CTR, (int)S_FLAGS}) std::fprintf(stderr, " %d", *reinterpret_cast<int32_t *>(&rec[c]));
    std::fprintf(stderr, "\n");
}

// The shared buffers of a render's rounds: queues, queue words, sort scratch, counters.  They belong to the scene whose
// precision owns the workspace (mixed-precision renders: the f64 scene's; the f32 rounds use them too — slot numbers
// and queue words do not depend on the precision of the records they point to).
struct RoundWs {
    int32_t *q;  // queue words + tag counts
    int32_t *queue[2], *shadow_queue, *sorted_queue;
    uint8_t *sort_keys;
    int32_t *sort_hist, *sort_base;
    unsigned long long *counters;
    int wide_grid;
};
// One round k of a batch on the records of precision RR: closest hits of the extend queue, material sort, shade,
// shadow rays.  (Everything is enqueued; nothing waits.)
// Round 0 of the default integrator on the default node format: no generate pass (CameraIo).  The counting instances,
// the other integrators (their shade rounds read the initial flag word) and the other node formats keep k_generate.
// TAKE_HIP_CAMERA_FUSED=0 turns it off (A/B runs).
template <class RR> bool camera_fused(const SceneT<RR> &sc, const RenderParams<RR> &rp, bool counting) {
    static const bool enabled = !(std::getenv("TAKE_HIP_CAMERA_FUSED") && std::atoi(std::getenv("TAKE_HIP_CAMERA_FUSED")) == 0);
    return enabled && TQ_GROUP == 1 && !counting && rp.integrator == 0 && sc.dev.qnodes != nullptr && sc.dev.qnodes8 == nullptr;
}

template <class RR>
void launch_round(TakeScene *ts, SceneT<RR> &sc, const RoundWs &ws, PathState<RR> st, const RenderParams<RR> &rp, int k, int64_t n_bound,
                  Timer &tm, bool counting, bool sort_materials, hipStream_t stream, int64_t dump, int64_t slots, bool tail = false,
                  float *to_f32 = nullptr) {
    int32_t *q = ws.q;
    int32_t *tag_count = q + Q_NUM_WORDS;
    const int cur = k & 1, next = cur ^ 1;
    int32_t *n_cur = q + (cur ? Q_N_EXT1 : Q_N_EXT0), *n_next = q + (next ? Q_N_EXT1 : Q_N_EXT0);
    StackSpill spill{sc.spill.p, sc.spill_stride};
    const PathIo<RR> io_ext{sc.dev.prims, st, ws.queue[cur], rp.ray_eps}, io_shadow{sc.dev.prims, st, ws.shadow_queue, rp.ray_eps};
    // persistent trace grid, cut down when the queue (bounded by n_bound) cannot fill it: one block per 128 rays
    const dim3 tgrid((unsigned)std::max<int64_t>(1, std::min<int64_t>(sc.trace_grid, (n_bound + 127) / 128)));
    hipLaunchKernelGGL(k_prep, dim3(1), dim3(64), 0, stream, q, next);
    tm.begin(tail ? TK_CLOSEST_TAIL : TK_CLOSEST);
    if (k == 0 && camera_fused<RR>(sc, rp, counting)) {
        // (the camera rays are made by the launch that traces them: CameraIo, tk_kernels.h)
        CameraIo<RR> io_cam;
        static_cast<PathIo<RR> &>(io_cam) = io_ext;
        io_cam.cam = sc.dev.cam, io_cam.rp = rp;
        if (sc.dev.inst_trace)
            hipLaunchKernelGGL((k_trace_group<RR, TQ_GROUP, false, false, CameraIo<RR>, true, true, 4>), tgrid, dim3(TQ_BLOCK), 0, stream, sc.dev,
                               io_cam, n_cur, 0, q + Q_HEAD_CLOSEST, ws.counters, (int)C_RAYS_CLOSEST, spill);
        else
            hipLaunchKernelGGL((k_trace_group<RR, TQ_GROUP, false, false, CameraIo<RR>, true, false, 4>), tgrid, dim3(TQ_BLOCK), 0, stream, sc.dev,
                               io_cam, n_cur, 0, q + Q_HEAD_CLOSEST, ws.counters, (int)C_RAYS_CLOSEST, spill);
    } else {
        launch_trace<RR>(sc.group, false, counting, tgrid, stream, sc.dev, io_ext, n_cur, 0, q + Q_HEAD_CLOSEST, ws.counters,
                         tail ? (int)C_RAYS_CLOSEST_TAIL : (int)C_RAYS_CLOSEST, spill);
    }
    tm.end();
    if (dump >= 0 && dump < slots) dump_slot(st, dump, "after trace_closest", k, stream);
    const int32_t *shade_in = ws.queue[cur];
    if (sort_materials) {
        tm.begin(TK_OTHER);
        // every wave of the sort gets >= 512 entries of the (bounded) queue: the one-block scan walks
        // 13 x waves counters, which must not dominate small rounds (it was 40 % of a 256x256 render)
        const int sort_grid = (int)std::max<int64_t>(1, std::min<int64_t>(ws.wide_grid, (n_bound + 2047) / 2048));
        hipLaunchKernelGGL((k_sort_count<RR>), dim3(sort_grid), dim3(BLOCK), 0, stream, sc.dev.prims, sc.dev.inst_shade, st,
                           ws.queue[cur], n_cur, ws.sort_keys, ws.sort_hist);
        hipLaunchKernelGGL(k_sort_scan, dim3(1), dim3(SORT_SCAN_THREADS), 0, stream, ws.sort_hist, ws.sort_base, tag_count,
                           sort_grid * (BLOCK / WAVE));
        hipLaunchKernelGGL(k_sort_scatter, dim3(sort_grid), dim3(BLOCK), 0, stream, ws.queue[cur], n_cur, ws.sort_keys, ws.sort_base,
                           ws.sorted_queue);
        tm.end();
        shade_in = ws.sorted_queue;
    }
    tm.begin(TK_SHADE);
    {
        const int shade_grid = (int)((n_bound + BLOCK - 1) / BLOCK);
        ShadeArgs<RR> sa{sc.dev, rp, st, shade_in, n_cur, sort_materials ? tag_count : nullptr, ws.queue[next],
                         n_next, ws.shadow_queue, q + Q_N_SHADOW, k, ws.counters, shade_grid, stream, to_f32};
        if (sort_materials) {
            // one specialised launch per material tag present in the scene + the miss segment
            for (int t = 0; t < TAKE_MAT_COUNT; t++)
                if (sc.host.tag_mask & (1u << t)) launch_shade<RR>(t, sa);
            launch_shade<RR>(TAG_MISS, sa);
        } else {
            launch_shade<RR>(sc.host.single_tag, sa);
        }
    }
    tm.end();
    if (dump >= 0 && dump < slots) dump_slot(st, dump, "after shade", k, stream);
    if (k <= rp.max_depth && rp.integrator == 0) {  // integrators 1..3 trace no shadow rays
        tm.begin(TK_SHADOW);
        launch_trace<RR>(sc.group, true, counting, tgrid, stream, sc.dev, io_shadow, q + Q_N_SHADOW, 0, q + Q_HEAD_SHADOW, ws.counters,
                         (int)C_RAYS_SHADOW, spill);
        tm.end();
        if (dump >= 0 && dump < slots) dump_slot(st, dump, "after trace_shadow", k, stream);
    }
}

// first_sample / keep_accum: progressive rendering — the samples of this call are numbered from first_sample (their
// random streams are those of a one-shot render's samples first_sample .. first_sample + spp - 1), keep_accum adds them
// to what `accum` holds instead of starting from zero, and the image is the mean over first_sample + spp samples.
template <class R> int render_impl(TakeScene *ts, const TakeRenderOpts &o, void *d_out, hipStream_t stream,
                                   int64_t first_sample = 0, bool keep_accum = false) {
    SceneT<R> &sc = pick<R>(ts);
    if (!keep_accum) ts->acc_samples = 0;  // (a one-shot render overwrites the accumulator: a progressive sequence ends)
    const int W = sc.host.cam.width, H = sc.host.cam.height;
    if (o.spp <= 0) return fail(TAKE_E_INVALID, "spp must be positive");
    if (o.max_depth < -1) return fail(TAKE_E_INVALID, "max_depth must be >= -1");
    if (o.integrator < 0 || o.integrator > 3) return fail(TAKE_E_INVALID, "unknown integrator");
    if (o.integrator != 0 && sc.host.env.light >= 0)
        return fail(TAKE_E_INVALID, "integrators 1..3 are the reference's own: they do not know the environment-map extension");
    const int stride = o.strip_stride > 0 ? o.strip_stride : 1;
    const int first = o.strip_first;
    if (first < 0 || first >= stride) return fail(TAKE_E_INVALID, "strip_first must be in [0, strip_stride)");
    const int n_rows = rows_of(H, first, stride, nullptr);
    const int64_t npix = (int64_t)n_rows * W;
    ts->counters = TakeCounters{};
    ts->counters.node_bytes = sc.dev.qnodes8 ? sizeof(QNode8) : (sc.dev.qnodes ? sizeof(QNode4) : sizeof(Node4<R>));
    ts->counters.prim_bytes = PRIM_TEST_BYTES * (int)(sizeof(R) / 4);
    if (ts->precision == TAKE_PRECISION_MIXED) ts->counters.prim_bytes = PRIM_TEST_BYTES;  // (most rounds read the f32 records)
    if (npix == 0) return TAKE_OK;
    if (npix >= ((int64_t)1 << 30)) return fail(TAKE_E_INVALID, "image too large");

    // paths in flight per batch: up to 512 Mi (69 GB of f32 path state + 9 GB of queues) — bigger batches keep the
    // persistent trace grid full for more of each bounce (measured on the 1M-triangle scene, spp per batch 8 / 16 / 32 /
    // 64 / 128 / 256 = 53.2 / 57.7 / 60.3 / 61.9 | 64.5 / 64.9 / 65.5 Msamples/s), and a 288 GB device has the room;
    // capped at four fifths of what is free now (round 3: it was half — a mixed-precision render, 418 B per path, then
    // needed two batches for 256 spp at 1920x1080 and lost ~1 % to the second set of thin late rounds)
    int64_t target = (int64_t)512 << 20;
    {
        size_t free_b = 0, total_b = 0;
        // (mixed precision: every slot has an f32 record beside its f64 one)
        const int64_t per_path = (int64_t)PATH_REC * (int64_t)(sizeof(R) + (ts->precision == TAKE_PRECISION_MIXED ? sizeof(float) : 0)) +
                                 4 * (int64_t)sizeof(int32_t);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const int64_t have = (int64_t)sc.capacity * per_path;  // already allocated by an earlier render
            // (shards of a scene group that share a device size their batches concurrently: each takes its share)
            target = std::min<int64_t>(target, std::max<int64_t>((int64_t)1 << 20, ((int64_t)free_b / std::max(1, ts->mem_share) + have) / 5 * 4 / per_path));
        }
    }
    int spb = o.samples_per_batch > 0 ? o.samples_per_batch : (int)std::max<int64_t>(1, target / npix);
    spb = std::min(spb, o.spp);
    while ((int64_t)spb * npix >= ((int64_t)1 << 31) - (1 << 26)) spb--;
    // The free-memory figure above is a snapshot: another process on the device (or another host thread) may take the
    // memory before the allocation lands.  A batch size the caller did not pin is then halved until it fits — the
    // image does not depend on it (a sample's random stream is a function of seed, pixel and sample index only).
    const bool mixed_records = sizeof(R) == 8 && ts->precision == TAKE_PRECISION_MIXED;
    int64_t slots = 0;
    int rc = TAKE_OK;
    for (;;) {
        slots = (int64_t)spb * npix;
        rc = ensure_workspace(sc, slots, npix);
        if (rc == TAKE_OK && mixed_records && (int64_t)ts->f.state_r.n < (int64_t)PATH_REC * slots &&
            ts->f.state_r.alloc((size_t)PATH_REC * slots) != hipSuccess) {
            ts->f.state_r.release();
            release_workspace(sc);
            rc = fail(TAKE_E_NOMEM, "out of device memory for the f32 path records of a mixed-precision render (" + std::to_string(slots) + " path slots)");
        }
        if (rc != TAKE_E_NOMEM || spb == 1 || o.samples_per_batch > 0) break;
        (void)hipGetLastError();
        spb = (spb + 1) / 2;
    }
    if (rc) return rc;

    PathState<R> st{sc.state_r.p, sc.capacity};
    RenderParams<R> rp{};
    rp.width = W, rp.height = H, rp.n_local_rows = n_rows, rp.npix = (int32_t)npix;
    rp.inv_npix = 1.0 / (double)npix, rp.inv_width = 1.0 / (double)W;
    rp.strip_first = first, rp.strip_stride = stride;
    rp.spp = o.spp, rp.max_depth = o.max_depth, rp.seed = o.seed, rp.integrator = o.integrator;
    rp.ray_eps = o.ray_epsilon > 0 ? R(o.ray_epsilon) : (sizeof(R) == 8 ? R(1e-7) : R(1e-4));

    const bool timing = (ts->instrumentation & 1) != 0;
    const bool counting = (ts->instrumentation & 2) != 0;
    const bool sort_materials = sc.host.n_material_tags > 1;
    const char *dump_env = std::getenv("TAKE_HIP_DUMP_SLOT");
    const int64_t dump = dump_env ? std::atoll(dump_env) : -1;
    ts->events.reset();
    ts->timed.clear();
    Timer tm{ts, stream, timing};
    int32_t *q = sc.qwords.p;
    const int wide_grid = (int)std::min<int64_t>((slots + BLOCK - 1) / BLOCK, (int64_t)ts->num_cus * 8);
    const int pix_grid = (int)std::min<int64_t>((npix + BLOCK - 1) / BLOCK, (int64_t)ts->num_cus * 8);

    if (sort_materials) {
        const size_t need = (size_t)N_SORT_KEYS * wide_grid * (BLOCK / WAVE);
        if (sc.sort_hist.n != need) {
            HIP_TRY(sc.sort_hist.alloc(need));
            HIP_TRY(sc.sort_base.alloc(need));
        }
    }
    const RoundWs ws{q, {sc.queue[0].p, sc.queue[1].p}, sc.shadow_queue.p, sc.sorted_queue.p, sc.sort_keys.p, sc.sort_hist.p, sc.sort_base.p,
                     sc.counters.p, wide_grid};
    // mixed precision (TAKE_PRECISION_MIXED): rounds k < exact_rounds on the f64 records and scene, the rest on f32
    // records of the same slots and the f32 scene
    bool mixed = false;
    int exact_rounds = 0;
    PathState<float> st32{nullptr, 0};
    RenderParams<float> rp32{};
    if constexpr (sizeof(R) == 8) {
        mixed = ts->precision == TAKE_PRECISION_MIXED;
        if (mixed) {
            exact_rounds = o.exact_bounces > 0 ? o.exact_bounces : TAKE_DEFAULT_EXACT_BOUNCES;
            if (o.integrator != 0) return fail(TAKE_E_INVALID, "mixed precision renders the reference's path_tracing (integrator 0) only");
            st32 = PathState<float>{ts->f.state_r.p, slots};
            rp32.width = rp.width, rp32.height = rp.height, rp32.n_local_rows = rp.n_local_rows, rp32.npix = rp.npix;
            rp32.inv_npix = rp.inv_npix, rp32.inv_width = rp.inv_width;
            rp32.strip_first = rp.strip_first, rp32.strip_stride = rp.strip_stride, rp32.spp = rp.spp, rp32.max_depth = rp.max_depth;
            rp32.integrator = rp.integrator, rp32.seed = rp.seed;
            rp32.ray_eps = o.ray_epsilon > 0 ? (float)o.ray_epsilon : 1e-4f;
        }
    }
    if (!keep_accum) HIP_TRY(hipMemsetAsync(sc.accum.p, 0, sizeof(R) * 3 * npix, stream));
    HIP_TRY(hipMemsetAsync(sc.counters.p, 0, sc.counters.bytes(), stream));
    hipEvent_t ev_begin = ts->events.get(), ev_end = ts->events.get();
    HIP_TRY(hipEventRecord(ev_begin, stream));
    if (!ts->poll_host) {
        HIP_TRY(hipHostMalloc((void **)&ts->poll_host, sizeof(int32_t) * TakeScene::POLL_RING, hipHostMallocDefault));
        for (int i = 0; i < TakeScene::POLL_RING; i++) HIP_TRY(hipEventCreateWithFlags(&ts->poll_ev[i], hipEventDisableTiming));
    }
    int64_t poll_issued = 0, poll_done = 0;

    for (int s0 = 0; s0 < o.spp; s0 += spb) {
        const int nb = std::min(spb, o.spp - s0);
        const int64_t n = (int64_t)nb * npix;
        rp.s0 = (int32_t)first_sample + s0;
        rp.spb = nb;
        rp32.s0 = rp.s0, rp32.spb = nb;
        tm.begin(TK_OTHER);
        if (camera_fused<R>(sc, rp, counting)) hipLaunchKernelGGL(k_iota, dim3(wide_grid), dim3(BLOCK), 0, stream, sc.queue[0].p, n);
        else hipLaunchKernelGGL((k_generate<R>), dim3(wide_grid), dim3(BLOCK), 0, stream, sc.dev, rp, st, sc.queue[0].p, n);
        hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, stream, q, (int)Q_N_EXT0, (int32_t)n);
        tm.end();
        const int rounds = o.max_depth + 2;
        int64_t n_bound = n;  // upper bound of the extend-queue length (queues only shrink)
        for (int k = 0; k < rounds; k++) {
            const int next = (k & 1) ^ 1;
            int32_t *n_next = q + (next ? Q_N_EXT1 : Q_N_EXT0);
            if constexpr (sizeof(R) == 8) {
                if (mixed && k == exact_rounds && !TK_SHADE_RECORD) {
                    // mixed precision: the paths still alive continue on f32 records (and the f32 scene) from here on
                    // (with TK_SHADE_RECORD the last exact shade round has written them already: k_shade, to_f32)
                    tm.begin(TK_OTHER);
                    hipLaunchKernelGGL(k_convert_state, dim3(wide_grid), dim3(BLOCK), 0, stream, st, st32, ws.queue[k & 1],
                                       q + ((k & 1) ? Q_N_EXT1 : Q_N_EXT0));
                    tm.end();
                }
                if (mixed && k >= exact_rounds) launch_round<float>(ts, ts->f, ws, st32, rp32, k, n_bound, tm, counting, sort_materials, stream, -1, slots, true);
                else launch_round<R>(ts, sc, ws, st, rp, k, n_bound, tm, counting, sort_materials, stream, dump, slots, false,
                                     (mixed && TK_SHADE_RECORD && k == exact_rounds - 1) ? st32.r : nullptr);
            } else {
                launch_round<R>(ts, sc, ws, st, rp, k, n_bound, tm, counting, sort_materials, stream, dump, slots);
            }
            // Queue length of the next round, read back asynchronously (pinned word + event, polled — the launch loop
            // never waits for the GPU): any value that has arrived bounds the grids of all later rounds (queues only
            // shrink), and a zero ends the launching.  (Round 1 blocked on a stream sync every 4 rounds: with the
            // ~370 launches of a small render that was a third of its 12 ms.)
            if (k + 1 < rounds && poll_issued - poll_done < TakeScene::POLL_RING) {
                const int slot = poll_issued % TakeScene::POLL_RING;
                HIP_TRY(hipMemcpyAsync(ts->poll_host + slot, n_next, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipEventRecord(ts->poll_ev[slot], stream));
                poll_issued++;
            }
            bool finished = false;
            while (poll_done < poll_issued) {
                const int slot = poll_done % TakeScene::POLL_RING;
                hipError_t q = hipEventQuery(ts->poll_ev[slot]);
                if (q == hipErrorNotReady) {
                    (void)hipGetLastError();  // "not ready" is an answer, not an error: keep it out of the sticky state
                    // stay at most 8 rounds ahead of the GPU: enough queued work that it never idles, close enough
                    // that a batch whose paths have all ended stops being launched
                    if (poll_issued - poll_done < 8) break;
                    q = hipEventSynchronize(ts->poll_ev[slot]);
                }
                HIP_TRY(q);
                const int32_t alive = ts->poll_host[slot];
                poll_done++;
                n_bound = std::min<int64_t>(n_bound, alive);
                if (alive == 0) finished = true;
            }
            if (finished) break;
        }
        // (outstanding read-backs of this batch complete with the stream; the ring indices just move on)
        poll_done = poll_issued;
        tm.begin(TK_OTHER);
        if constexpr (sizeof(R) == 8) {
            if (mixed) hipLaunchKernelGGL(k_accumulate_mixed, dim3(pix_grid), dim3(BLOCK), 0, stream, st, st32, sc.accum.p, (int32_t)npix, nb);
            else hipLaunchKernelGGL((k_accumulate<R>), dim3(pix_grid), dim3(BLOCK), 0, stream, st, sc.accum.p, (int32_t)npix, nb);
        } else {
            hipLaunchKernelGGL((k_accumulate<R>), dim3(pix_grid), dim3(BLOCK), 0, stream, st, sc.accum.p, (int32_t)npix, nb);
        }
        tm.end();
    }
    tm.begin(TK_OTHER);
    hipLaunchKernelGGL((k_resolve<R>), dim3(pix_grid), dim3(BLOCK), 0, stream, sc.accum.p, (R *)d_out, W, n_rows,
                       (int32_t)first_sample + o.spp);
    tm.end();
    HIP_TRY(hipEventRecord(ev_end, stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    if (tm.err != hipSuccess) return fail(TAKE_E_DEVICE, std::string("kernel timing events: ") + hipGetErrorString(tm.err));

    unsigned long long c[C_NUM_WORDS];
    HIP_TRY(hipMemcpy(c, sc.counters.p, sizeof c, hipMemcpyDeviceToHost));
    TakeCounters &tc = ts->counters;
    tc.samples = (uint64_t)npix * (uint64_t)o.spp;
    tc.rays_closest = c[C_RAYS_CLOSEST] + c[C_RAYS_CLOSEST_TAIL];
    tc.rays_closest_f32 = c[C_RAYS_CLOSEST_TAIL];
    tc.rays_shadow = c[C_RAYS_SHADOW];
    tc.node_visits = c[C_NODE_VISITS];
    tc.prim_tests = c[C_PRIM_TESTS];
    tc.bounces = c[C_BOUNCES];
    tc.leaf_visits = c[C_LEAF_VISITS];
    tc.wave_node_steps = c[C_WAVE_NODE_STEPS];
    tc.wave_leaf_steps = c[C_WAVE_LEAF_STEPS];
    if (std::getenv("TAKE_HIP_VERBOSE"))
        std::fprintf(stderr, "[take_hip] node-step ray slots: waiting-at-leaf %llu idle %llu running %llu; shadow rays the slot's previous occluder stops again: %llu of %llu\n",
                     c[C_WAIT_SLOTS], c[C_IDLE_SLOTS], c[C_NODE_VISITS], c[C_OCC_CACHE_HITS], c[C_RAYS_SHADOW]);
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, ev_begin, ev_end));
    tc.ms_total = ms;
    double acc[TK_NUM] = {0, 0, 0, 0, 0};
    for (auto &t : ts->timed) {
        float m = 0;
        if (hipEventElapsedTime(&m, t.second.first, t.second.second) == hipSuccess) acc[t.first] += m;
        if (t.first == TK_CLOSEST || t.first == TK_CLOSEST_TAIL) tc.launches_trace_closest++;
        if (t.first == TK_CLOSEST_TAIL) tc.launches_trace_closest_f32++;
        if (t.first == TK_SHADOW) tc.launches_trace_shadow++;
    }
    tc.ms_trace_closest = acc[TK_CLOSEST] + acc[TK_CLOSEST_TAIL];
    tc.ms_trace_closest_f32 = acc[TK_CLOSEST_TAIL];
    tc.ms_trace_shadow = acc[TK_SHADOW];
    tc.ms_shade = acc[TK_SHADE];
    tc.ms_other = acc[TK_OTHER];
    return TAKE_OK;
}

template <class R>
int trace_impl(TakeScene *ts, const void *d_rays, int64_t n, void *d_hits, int32_t *d_occ, bool any, bool count,
               hipStream_t stream) {
    SceneT<R> &sc = pick<R>(ts);
    if (n < 0 || n >= ((int64_t)1 << 31) - (1 << 26)) return fail(TAKE_E_INVALID, "ray count out of range");
    StackSpill spill{sc.spill.p, sc.spill_stride};
    int32_t *q = sc.qwords.p;
    HIP_TRY(hipMemsetAsync(q + Q_HEAD_CLOSEST, 0, sizeof(int32_t), stream));
    HIP_TRY(hipMemsetAsync(sc.counters.p, 0, sc.counters.bytes(), stream));
    hipEvent_t a = nullptr, b = nullptr;
    ts->events.reset();
    a = ts->events.get(), b = ts->events.get();
    HIP_TRY(hipEventRecord(a, stream));
    const HookIo<R> io{sc.dev.prims, (const RayAoS<R> *)d_rays, (HitAoS<R> *)d_hits, d_occ, sc.dev.inst_shade};
    launch_trace<R>(sc.group, any, count, dim3(sc.trace_grid), stream, sc.dev, io, nullptr, (int32_t)n, q + Q_HEAD_CLOSEST,
                    sc.counters.p, -1, spill);
    HIP_TRY(hipEventRecord(b, stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    unsigned long long c[C_NUM_WORDS];
    HIP_TRY(hipMemcpy(c, sc.counters.p, sizeof c, hipMemcpyDeviceToHost));
    ts->counters = TakeCounters{};
    ts->counters.node_bytes = sc.dev.qnodes8 ? sizeof(QNode8) : (sc.dev.qnodes ? sizeof(QNode4) : sizeof(Node4<R>));
    ts->counters.prim_bytes = PRIM_TEST_BYTES * (int)(sizeof(R) / 4);
    (any ? ts->counters.rays_shadow : ts->counters.rays_closest) = (uint64_t)n;
    ts->counters.node_visits = c[C_NODE_VISITS];
    ts->counters.prim_tests = c[C_PRIM_TESTS];
    ts->counters.leaf_visits = c[C_LEAF_VISITS];
    ts->counters.wave_node_steps = c[C_WAVE_NODE_STEPS];
    ts->counters.wave_leaf_steps = c[C_WAVE_LEAF_STEPS];
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, a, b));
    (any ? ts->counters.ms_trace_shadow : ts->counters.ms_trace_closest) = ms;
    (any ? ts->counters.launches_trace_shadow : ts->counters.launches_trace_closest) = 1;
    ts->counters.ms_total = ms;
    return TAKE_OK;
}

template <class R> int trace_host(TakeScene *ts, const void *rays, int64_t n, void *hits, int32_t *occ, bool any) {
    if (n == 0) return TAKE_OK;
    // entry distances are ordered through their bit patterns (non-negative floats): a ray must start at tmin >= 0
    for (int64_t i = 0; i < n; i++) {
        const RayAoS<R> &q = ((const RayAoS<R> *)rays)[i];
        if (!(q.tmin >= R(0))) return fail(TAKE_E_INVALID, "ray " + std::to_string(i) + ": tmin must be >= 0");
    }
    DevBuf<RayAoS<R>> d_rays;
    DevBuf<HitAoS<R>> d_hits;
    DevBuf<int32_t> d_occ;
    HIP_TRY(d_rays.alloc(n));
    int rc = TAKE_OK;
    do {
        if (hipMemcpy(d_rays.p, rays, n * sizeof(RayAoS<R>), hipMemcpyHostToDevice) != hipSuccess) {
            rc = fail(TAKE_E_DEVICE, "ray upload failed");
            break;
        }
        if (any ? d_occ.alloc(n) != hipSuccess : d_hits.alloc(n) != hipSuccess) {
            rc = fail(TAKE_E_NOMEM, "hit buffer allocation failed");
            break;
        }
        rc = trace_impl<R>(ts, d_rays.p, n, d_hits.p, d_occ.p, any, false, nullptr);
        if (rc) break;
        hipError_t e = any ? hipMemcpy(occ, d_occ.p, n * sizeof(int32_t), hipMemcpyDeviceToHost)
                           : hipMemcpy(hits, d_hits.p, n * sizeof(HitAoS<R>), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(TAKE_E_DEVICE, "hit download failed");
    } while (0);
    d_rays.release(), d_hits.release(), d_occ.release();
    return rc;
}

// A scene lives on the device that was current when it was created; every entry point that touches it makes that
// device current for the duration of the call and restores the caller's afterwards.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) ok = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};
#define TAKE_ON_DEVICE(ts)                                                                            \
    DeviceGuard guard_((ts)->device);                                                                 \
    if (!guard_.ok) return fail(TAKE_E_DEVICE, "cannot make the scene's device current")

int check_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(TAKE_E_NO_GPU, "no HIP device visible: libtake_hip has no CPU path");
    return n;
}

}  // namespace

// ---- PLY -> device mesh arrays (tk_ply.h) --------------------------------------------------------------------------
namespace {
void fill_layout(const ply::Layout &L, TakePlyLayout *o) {
    std::memset(o, 0, sizeof(*o));
    o->n_vertices = L.n_vertices, o->n_faces = L.n_faces;
    o->vertex_offset = L.vertex_off, o->face_offset = L.face_off;
    o->vertex_stride = L.vertex_stride, o->face_stride = L.face_stride;
    o->has_normals = L.nrm_type != ply::T_NONE, o->has_uvs = L.uv_type != ply::T_NONE;
    o->position_is_f64 = L.pos_type == ply::T_F64, o->index_bytes = ply::type_size(L.index_type);
    o->header_bytes = L.header_bytes;
}

// a file, memory-mapped read-only (the body is read once: by the copy to the device, or by the inflater)
struct MappedFile {
    void *p = nullptr;
    size_t n = 0;
    std::string err;
    explicit MappedFile(const char *path) {
        const int fd = open(path, O_RDONLY);
        if (fd < 0) {
            err = std::string("cannot open ") + path;
            return;
        }
        struct stat sb;
        if (fstat(fd, &sb) != 0 || sb.st_size <= 0) {
            close(fd);
            err = std::string("cannot read ") + path;
            return;
        }
        void *q = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (q == MAP_FAILED) {
            err = std::string("cannot map ") + path;
            return;
        }
        (void)madvise(q, (size_t)sb.st_size, MADV_SEQUENTIAL);
        p = q, n = (size_t)sb.st_size;
    }
    ~MappedFile() {
        if (p) munmap(p, n);
    }
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
};

void free_mesh_arrays(TakeMesh *m) {
    if (m->positions) (void)hipFree(const_cast<double *>(m->positions));
    if (m->indices) (void)hipFree(const_cast<int32_t *>(m->indices));
    if (m->normals) (void)hipFree(const_cast<double *>(m->normals));
    if (m->uvs) (void)hipFree(const_cast<double *>(m->uvs));
    std::memset(m, 0, sizeof(*m));
}

// body (host) -> HBM, then the two decode kernels (tk_ply.h); D's offsets are relative to `host_body`
int decode_mesh_body(const uint8_t *host_body, const ply::Layout &D, const double *to_world, const double *inv_to_world,
                     int32_t material_id, const char *what, TakeMesh *out) {
    ply::Mat4 X, Xi;
    static const double I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(X.m, to_world ? to_world : I, sizeof(I));
    std::memcpy(Xi.m, inv_to_world ? inv_to_world : I, sizeof(I));
    if (to_world && !inv_to_world && D.nrm_type != ply::T_NONE)
        return fail(TAKE_E_INVALID, "the file has normals: pass inverse(to_world) along with to_world");
    DevBuf<uint8_t> body;
    DevBuf<int32_t> status;
    TakeMesh m{};
    m.n_vertices = D.n_vertices, m.n_faces = D.n_faces, m.material_id = material_id, m.flags = TAKE_MESH_DEVICE_ARRAYS;
    auto bail = [&](int rc) {
        body.release(), status.release();
        free_mesh_arrays(&m);
        return rc;
    };
    auto dmalloc = [&](auto *&p, size_t count) -> bool {
        void *q = nullptr;
        if (count == 0) return true;
        if (inject_alloc_failure() || hipMalloc(&q, count * sizeof(*p)) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        p = (std::remove_reference_t<decltype(p)>)q;
        return true;
    };
    double *pos = nullptr, *nrm = nullptr, *uv = nullptr;
    int32_t *idx = nullptr;
    const bool ok = body.alloc((size_t)std::max<int64_t>(D.end_off, 1)) == hipSuccess && status.alloc(1) == hipSuccess &&
                    dmalloc(pos, 3 * (size_t)D.n_vertices) && dmalloc(idx, 3 * (size_t)D.n_faces) &&
                    (D.nrm_type == ply::T_NONE || dmalloc(nrm, 3 * (size_t)D.n_vertices)) &&
                    (D.uv_type == ply::T_NONE || dmalloc(uv, 2 * (size_t)D.n_vertices));
    m.positions = pos, m.indices = idx, m.normals = nrm, m.uvs = uv;
    if (!ok) return bail(fail(TAKE_E_NOMEM, "out of device memory for a " + std::to_string(D.n_faces) + "-face " + what + " mesh"));
    {
        PinnedUploads pin;
        hipError_t e = pin.copy(body.p, host_body, (size_t)D.end_off);
        if (e == hipSuccess) e = hipMemsetAsync(status.p, 0, sizeof(int32_t), pin.stream);
        constexpr int BLK = 256;
        if (e == hipSuccess && D.n_vertices > 0)
            hipLaunchKernelGGL(ply::k_ply_vertices, dim3((unsigned)((D.n_vertices + BLK - 1) / BLK)), dim3(BLK), 0, pin.stream, body.p, D, X, Xi, pos, nrm, uv);
        if (e == hipSuccess && D.n_faces > 0)
            hipLaunchKernelGGL(ply::k_ply_faces, dim3((unsigned)((D.n_faces + BLK - 1) / BLK)), dim3(BLK), 0, pin.stream, body.p, D, idx, status.p);
        if (e == hipSuccess) e = hipGetLastError();
        int32_t st = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&st, status.p, sizeof(st), hipMemcpyDeviceToHost, pin.stream);
        if (e == hipSuccess) e = pin.finish();
        if (e != hipSuccess) return bail(fail(TAKE_E_DEVICE, std::string(what) + " decode: " + hipGetErrorString(e)));
        if (st & 1) return bail(fail(TAKE_E_INVALID, std::string("a face of the ") + what + " file is not a triangle (the reference reads three indices per face)"));
        if (st & 2) return bail(fail(TAKE_E_INVALID, std::string("a face of the ") + what + " file indexes past its vertex array"));
    }
    body.release(), status.release();
    *out = m;
    return TAKE_OK;
}
}  // namespace

extern "C" {

const char *take_hip_last_error(void) { return g_error.c_str(); }
int take_hip_abi_version(void) { return TAKE_HIP_ABI_VERSION; }
int take_hip_device_count(void) { return check_device(); }

int take_hip_ply_layout(const void *file_bytes, size_t n_bytes, TakePlyLayout *out) {
    if (!file_bytes || !out) return fail(TAKE_E_INVALID, "null argument");
    ply::Layout L;
    const std::string err = ply::parse_header((const uint8_t *)file_bytes, n_bytes, L);
    if (!err.empty()) return fail(TAKE_E_INVALID, err);
    fill_layout(L, out);
    return TAKE_OK;
}

int take_hip_mesh_from_ply(const void *file_bytes, size_t n_bytes, const double *to_world, const double *inv_to_world,
                           int32_t material_id, TakeMesh *out) {
    if (!file_bytes || !out) return fail(TAKE_E_INVALID, "null argument");
    std::memset(out, 0, sizeof(*out));
    ply::Layout L;
    const std::string err = ply::parse_header((const uint8_t *)file_bytes, n_bytes, L);
    if (!err.empty()) return fail(TAKE_E_INVALID, err);
    const int nd = check_device();
    if (nd < 0) return nd;
    // the body as it lies in the file: one copy, from the first to the last byte the two elements span
    const int64_t lo = std::min(L.vertex_off, L.face_off);
    ply::Layout D = L;  // offsets relative to the copied span
    D.vertex_off -= lo, D.face_off -= lo, D.nrm_base -= lo, D.uv_base -= lo, D.end_off -= lo;
    return decode_mesh_body((const uint8_t *)file_bytes + lo, D, to_world, inv_to_world, material_id, "PLY", out);
}

// ---- Mitsuba serialized meshes (src/parse/parse_serialized.cpp:174-256): inflate on the host, decode on the device --
namespace {
struct Inflater {
    z_stream z{};
    bool open = false;
    const uint8_t *src;
    size_t left;
    Inflater(const uint8_t *p, size_t n) : src(p), left(n) {
        open = inflateInit2(&z, 15) == Z_OK;  // (windowBits 15: parse_serialized.cpp:47)
    }
    ~Inflater() {
        if (open) inflateEnd(&z);
    }
    // exactly `size` inflated bytes into dst, or what is wrong (the messages of ZStream::read, parse_serialized.cpp:60-104)
    const char *read(void *dst, size_t size) {
        uint8_t *out = (uint8_t *)dst;
        while (size > 0) {
            if (z.avail_in == 0) {
                const size_t take = std::min<size_t>(left, (size_t)1 << 30);
                if (take == 0) return "read less data than expected";
                z.next_in = const_cast<uint8_t *>(src), z.avail_in = (uInt)take;
                src += take, left -= take;
            }
            const size_t want = std::min<size_t>(size, (size_t)1 << 30);
            z.next_out = out, z.avail_out = (uInt)want;
            const int rv = inflate(&z, Z_NO_FLUSH);
            if (rv == Z_STREAM_ERROR) return "inflate(): stream error";
            if (rv == Z_NEED_DICT) return "inflate(): need dictionary";
            if (rv == Z_DATA_ERROR) return "inflate(): data error";
            if (rv == Z_MEM_ERROR) return "inflate(): memory error";
            const size_t got = want - z.avail_out;
            out += got, size -= got;
            if (size > 0 && rv == Z_STREAM_END) return "inflate(): attempting to read past the end of the stream";
            if (got == 0 && rv == Z_BUF_ERROR && left == 0 && z.avail_in == 0) return "read less data than expected";
        }
        return nullptr;
    }
};
}  // namespace

int take_hip_mesh_from_serialized(const void *file_bytes, size_t n_bytes, int32_t shape_index, const double *to_world,
                                  const double *inv_to_world, int32_t material_id, TakeMesh *out) {
    if (!file_bytes || !out) return fail(TAKE_E_INVALID, "null argument");
    std::memset(out, 0, sizeof(*out));
    const uint8_t *f = (const uint8_t *)file_bytes;
    if (n_bytes < 4) return fail(TAKE_E_INVALID, "not a serialized mesh file: shorter than its header");
    uint16_t version = 0;
    std::memcpy(&version, f + 2, 2);  // (the magic number in front of it is ignored: parse_serialized.cpp:178)
    if (version != 3 && version != 4) return fail(TAKE_E_INVALID, "serialized mesh: unknown format version " + std::to_string(version));
    size_t at = 0;
    if (shape_index > 0) {  // skip_to_idx (parse_serialized.cpp:117-133): the offset table at the end of the file
        uint32_t count = 0;
        std::memcpy(&count, f + n_bytes - 4, 4);
        const size_t esz = version == 4 ? 8 : 4;
        if ((uint64_t)shape_index >= count || n_bytes < 4 + esz * (size_t)count)
            return fail(TAKE_E_INVALID, "serialized mesh: shape index " + std::to_string(shape_index) + " of " + std::to_string(count));
        uint64_t off = 0;
        std::memcpy(&off, f + n_bytes - 4 - esz * ((size_t)count - (size_t)shape_index), esz);
        if (off + 4 > n_bytes) return fail(TAKE_E_INVALID, "serialized mesh: sub-mesh offset past the end of the file");
        at = (size_t)off;
    } else if (shape_index < 0) {
        return fail(TAKE_E_INVALID, "serialized mesh: negative shape index");
    }
    Inflater z(f + at + 4, n_bytes - at - 4);
    if (!z.open) return fail(TAKE_E_DEVICE, "could not initialize zlib");
    uint32_t flags = 0;
    uint64_t nv = 0, nf = 0;
    const char *bad = z.read(&flags, 4);
    if (!bad && version == 4) {  // the mesh's name, NUL-terminated
        char c = 1;
        while (!bad && c != 0) bad = z.read(&c, 1);
    }
    if (!bad) bad = z.read(&nv, 8);
    if (!bad) bad = z.read(&nf, 8);
    if (bad) return fail(TAKE_E_INVALID, std::string("serialized mesh: ") + bad);
    if (nv >= ((uint64_t)1 << 31) || nf >= ((uint64_t)1 << 31) / 3) return fail(TAKE_E_INVALID, "serialized mesh too large for 32-bit vertex indices");
    ply::Layout L;
    ply::serialized_layout(flags, (int64_t)nv, (int64_t)nf, L);
    const int nd = check_device();
    if (nd < 0) return nd;
    // the blocks, inflated once into one host buffer; the kernels read them where the stream put them
    std::unique_ptr<uint8_t[]> body(new (std::nothrow) uint8_t[(size_t)std::max<int64_t>(L.end_off, 1)]);
    if (!body) return fail(TAKE_E_NOMEM, "out of host memory for the inflated mesh");
    bad = z.read(body.get(), (size_t)L.end_off);
    if (bad) return fail(TAKE_E_INVALID, std::string("serialized mesh: ") + bad);
    return decode_mesh_body(body.get(), L, to_world, inv_to_world, material_id, "serialized", out);
}

int take_hip_mesh_from_ply_file(const char *path, const double *to_world, const double *inv_to_world, int32_t material_id, TakeMesh *out) {
    if (!path || !out) return fail(TAKE_E_INVALID, "null argument");
    std::memset(out, 0, sizeof(*out));
    MappedFile mf(path);
    if (!mf.p) return fail(TAKE_E_INVALID, mf.err);
    return take_hip_mesh_from_ply(mf.p, mf.n, to_world, inv_to_world, material_id, out);
}

int take_hip_mesh_from_serialized_file(const char *path, int32_t shape_index, const double *to_world, const double *inv_to_world,
                                       int32_t material_id, TakeMesh *out) {
    if (!path || !out) return fail(TAKE_E_INVALID, "null argument");
    std::memset(out, 0, sizeof(*out));
    MappedFile mf(path);
    if (!mf.p) return fail(TAKE_E_INVALID, mf.err);
    return take_hip_mesh_from_serialized(mf.p, mf.n, shape_index, to_world, inv_to_world, material_id, out);
}

int take_hip_mesh_download(const TakeMesh *m, double *positions, int32_t *indices, double *normals, double *uvs) {
    if (!m) return fail(TAKE_E_INVALID, "null mesh");
    if (!(m->flags & TAKE_MESH_DEVICE_ARRAYS)) return fail(TAKE_E_INVALID, "not a device-array mesh");
    auto down = [&](void *dst, const void *src, size_t bytes) -> hipError_t {
        return (dst && src && bytes) ? hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) : hipSuccess;
    };
    HIP_TRY(down(positions, m->positions, sizeof(double) * 3 * (size_t)m->n_vertices));
    HIP_TRY(down(indices, m->indices, sizeof(int32_t) * 3 * (size_t)m->n_faces));
    HIP_TRY(down(normals, m->normals, sizeof(double) * 3 * (size_t)m->n_vertices));
    HIP_TRY(down(uvs, m->uvs, sizeof(double) * 2 * (size_t)m->n_vertices));
    return TAKE_OK;
}

int take_hip_mesh_release(TakeMesh *m) {
    if (!m) return TAKE_OK;
    if (m->flags & TAKE_MESH_DEVICE_ARRAYS) free_mesh_arrays(m);
    return TAKE_OK;
}

int take_hip_scene_create(const TakeSceneDesc *desc, const TakeBuildOpts *opts, TakeScene **out) {
    if (!desc || !out) return fail(TAKE_E_INVALID, "null argument");
    *out = nullptr;
    int nd = check_device();
    if (nd < 0) return nd;
    TakeBuildOpts o{};
    if (opts) o = *opts;
    if (o.precision != TAKE_PRECISION_F32 && o.precision != TAKE_PRECISION_F64 && o.precision != TAKE_PRECISION_MIXED)
        return fail(TAKE_E_INVALID, "unknown precision");
    if (o.builder < TAKE_BUILDER_AUTO || o.builder > TAKE_BUILDER_HOST_SAH) return fail(TAKE_E_INVALID, "unknown builder");
    if (o.instances != TAKE_INSTANCES_TWO_LEVEL && o.instances != TAKE_INSTANCES_FLATTEN) return fail(TAKE_E_INVALID, "unknown instance mode");
    TakeScene *ts = new (std::nothrow) TakeScene();
    if (!ts) return fail(TAKE_E_NOMEM, "out of host memory");
    ts->precision = o.precision;
    hipDeviceProp_t prop;
    if (hipGetDevice(&ts->device) != hipSuccess || hipGetDeviceProperties(&prop, ts->device) != hipSuccess) {
        delete ts;
        return fail(TAKE_E_DEVICE, "cannot query the HIP device");
    }
    ts->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    int rc;
    try {
        // device-array meshes (take_hip_mesh_from_ply): the host side of the build — index validation, the face / normal /
        // uv tables, the SAH builder below TAKE_AUTO_DEVICE_BUILD_SHAPES shapes — reads host copies; the device build
        // takes the positions where they are
        StagedMeshes staged;
        FlattenedInstances flat;
        TakeSceneDesc local = *desc;
        if (o.instances == TAKE_INSTANCES_FLATTEN) {
            int threads = o.bvh_threads > 0 ? o.bvh_threads : (int)std::thread::hardware_concurrency();
            const int rf = flat.expand(local, std::max(1, threads));
            if (rf) {
                delete ts;
                return rf;
            }
        }
        desc = &local;  // (from here on: the description as it will be built)
        const bool device_build = o.precision == TAKE_PRECISION_F32 && desc->n_shapes >= 8 && desc->n_instances == 0 &&
                                  (o.builder == TAKE_BUILDER_DEVICE_LBVH || (o.builder == TAKE_BUILDER_AUTO && desc->n_shapes >= TAKE_AUTO_DEVICE_BUILD_SHAPES));
        rc = staged.stage(local, !device_build);
        struct Reset {
            ~Reset() { t_staged = nullptr; }
        } reset;
        t_staged = &staged;
        if (!rc) rc = o.precision != TAKE_PRECISION_F32 ? upload_scene<double>(ts, local, o) : upload_scene<float>(ts, local, o);
        if (!rc && o.precision == TAKE_PRECISION_MIXED) rc = upload_scene<float>(ts, local, o);  // the same scene in f32 beside it
    } catch (const std::bad_alloc &) {
        rc = fail(TAKE_E_NOMEM, "out of host memory while preparing the scene");
    } catch (const std::exception &e) {
        rc = fail(TAKE_E_INVALID, e.what());
    }
    if (rc) {
        ts->f.release();
        ts->d.release();
        delete ts;
        return rc;
    }
    *out = ts;
    return TAKE_OK;
}

int take_hip_scene_destroy(TakeScene *ts) {
    if (!ts) return TAKE_OK;
    DeviceGuard guard_(ts->device);
    ts->f.release();
    ts->d.release();
    ts->events.destroy();
    if (ts->poll_host) {
        (void)hipHostFree(ts->poll_host);
        for (auto e : ts->poll_ev)
            if (e) (void)hipEventDestroy(e);
    }
    delete ts;
    return TAKE_OK;
}

int take_hip_render_rows(const TakeScene *ts, int32_t strip_first, int32_t strip_stride, int32_t *rows_out) {
    if (!ts) return fail(TAKE_E_INVALID, "null scene");
    if (strip_stride <= 0 || strip_first < 0 || strip_first >= strip_stride)
        return fail(TAKE_E_INVALID, "strip_first must be in [0, strip_stride)");
    const int H = is_f64(ts) ? ts->d.host.cam.height : ts->f.host.cam.height;
    return rows_of(H, strip_first, strip_stride, rows_out);
}

int take_hip_render_device(TakeScene *ts, const TakeRenderOpts *opts, void *d_rgb_out, void *stream) {
    if (!ts || !opts || !d_rgb_out) return fail(TAKE_E_INVALID, "null argument");
    TAKE_ON_DEVICE(ts);
    if (is_f64(ts)) return render_impl<double>(ts, *opts, d_rgb_out, (hipStream_t)stream);
    return render_impl<float>(ts, *opts, d_rgb_out, (hipStream_t)stream);
}

// Progressive rendering (the per-pixel accumulate of src/render.cpp:68-78 kept resident between calls).
int take_hip_render_accumulate(TakeScene *ts, const TakeRenderOpts *opts, int32_t restart, void *d_rgb_out, void *stream) {
    if (!ts || !opts || !d_rgb_out) return fail(TAKE_E_INVALID, "null argument");
    TAKE_ON_DEVICE(ts);
    const bool f64 = is_f64(ts);
    const TakeRenderOpts &a = ts->acc_opts;
    const bool fresh = restart != 0 || ts->acc_samples == 0;
    if (!fresh && (a.seed != opts->seed || a.max_depth != opts->max_depth || a.integrator != opts->integrator ||
                   a.strip_first != opts->strip_first || a.strip_stride != opts->strip_stride || a.ray_epsilon != opts->ray_epsilon))
        return fail(TAKE_E_INVALID, "take_hip_render_accumulate: options differ from the ones the accumulated samples were "
                                    "rendered with (seed, max_depth, integrator, strips, ray_epsilon): pass restart = 1");
    const int64_t first = fresh ? 0 : ts->acc_samples;
    if (first + (int64_t)opts->spp >= ((int64_t)1 << 31)) return fail(TAKE_E_INVALID, "too many accumulated samples");
    // (a workspace grown for a bigger batch keeps the accumulator: ensure_workspace only ever enlarges it, and the
    // strip set — hence the pixel count — is fixed for the sequence)
    const int rc = f64 ? render_impl<double>(ts, *opts, d_rgb_out, (hipStream_t)stream, first, !fresh)
                       : render_impl<float>(ts, *opts, d_rgb_out, (hipStream_t)stream, first, !fresh);
    if (rc) {
        ts->acc_samples = 0;  // the accumulator may hold a partial batch: the sequence has to restart
        return rc;
    }
    ts->acc_samples = first + opts->spp;
    ts->acc_opts = *opts;
    return TAKE_OK;
}
int64_t take_hip_accumulated_samples(const TakeScene *ts) { return ts ? ts->acc_samples : 0; }

int take_hip_render(TakeScene *ts, const TakeRenderOpts *opts, void *rgb_out_host) {
    if (!ts || !opts || !rgb_out_host) return fail(TAKE_E_INVALID, "null argument");
    TAKE_ON_DEVICE(ts);
    const bool f64 = is_f64(ts);
    const int W = f64 ? ts->d.host.cam.width : ts->f.host.cam.width;
    const int stride = opts->strip_stride > 0 ? opts->strip_stride : 1;
    if (opts->strip_first < 0 || opts->strip_first >= stride)
        return fail(TAKE_E_INVALID, "strip_first must be in [0, strip_stride)");
    const int rows = take_hip_render_rows(ts, opts->strip_first, stride, nullptr);
    if (rows < 0) return rows;
    const size_t bytes = (size_t)rows * W * 3 * (f64 ? 8 : 4);
    if (bytes == 0) return TAKE_OK;
    // render into the scene's own output buffer, then copy out
    int rc;
    if (f64) {
        rc = ensure_workspace(ts->d, 0, (int64_t)rows * W);
        if (!rc) rc = render_impl<double>(ts, *opts, ts->d.out.p, nullptr);
        if (!rc) HIP_TRY(hipMemcpy(rgb_out_host, ts->d.out.p, bytes, hipMemcpyDeviceToHost));
    } else {
        rc = ensure_workspace(ts->f, 0, (int64_t)rows * W);
        if (!rc) rc = render_impl<float>(ts, *opts, ts->f.out.p, nullptr);
        if (!rc) HIP_TRY(hipMemcpy(rgb_out_host, ts->f.out.p, bytes, hipMemcpyDeviceToHost));
    }
    return rc;
}

int take_hip_pack_exr_scanlines(const void *d_rgb, int32_t precision, int32_t width, int32_t height, uint16_t *d_out, void *stream) {
    if (!d_rgb || !d_out || width <= 0 || height <= 0) return fail(TAKE_E_INVALID, "bad argument");
    if (precision != TAKE_PRECISION_F32 && precision != TAKE_PRECISION_F64) return fail(TAKE_E_INVALID, "unknown precision");
    const int nd = check_device();
    if (nd < 0) return nd;
    const int64_t total = (int64_t)width * height * 3;
    const dim3 grid((unsigned)std::min<int64_t>((total + BLOCK - 1) / BLOCK, 8192));
    if (precision == TAKE_PRECISION_F64)
        hipLaunchKernelGGL((k_pack_exr<double>), grid, dim3(BLOCK), 0, (hipStream_t)stream, (const double *)d_rgb, width, height, d_out);
    else
        hipLaunchKernelGGL((k_pack_exr<float>), grid, dim3(BLOCK), 0, (hipStream_t)stream, (const float *)d_rgb, width, height, d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return TAKE_OK;
}

int take_hip_render_exr_scanlines(TakeScene *ts, const TakeRenderOpts *opts, uint16_t *out_host) {
    if (!ts || !opts || !out_host) return fail(TAKE_E_INVALID, "null argument");
    TAKE_ON_DEVICE(ts);
    const bool f64 = is_f64(ts);
    const int W = f64 ? ts->d.host.cam.width : ts->f.host.cam.width, H = f64 ? ts->d.host.cam.height : ts->f.host.cam.height;
    TakeRenderOpts o = *opts;
    o.strip_first = 0, o.strip_stride = 1;
    int rc = f64 ? ensure_workspace(ts->d, 0, (int64_t)W * H) : ensure_workspace(ts->f, 0, (int64_t)W * H);
    if (rc) return rc;
    void *d_img = f64 ? (void *)ts->d.out.p : (void *)ts->f.out.p;
    rc = f64 ? render_impl<double>(ts, o, d_img, nullptr) : render_impl<float>(ts, o, d_img, nullptr);
    if (rc) return rc;
    DevBuf<uint16_t> halves;
    if (halves.alloc((size_t)W * H * 3) != hipSuccess) return fail(TAKE_E_NOMEM, "out of device memory for the scanline buffer");
    rc = take_hip_pack_exr_scanlines(d_img, ts->precision, W, H, halves.p, nullptr);
    if (!rc && hipMemcpy(out_host, halves.p, halves.bytes(), hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(TAKE_E_DEVICE, "scanline download failed");
    halves.release();
    return rc;
}

int take_hip_trace_closest(TakeScene *ts, const void *rays, int64_t n, void *hits) {
    if (!ts || (n > 0 && (!rays || !hits))) return fail(TAKE_E_INVALID, "null argument");
    TAKE_ON_DEVICE(ts);
    return is_f64(ts) ? trace_host<double>(ts, rays, n, hits, nullptr, false)
                                               : trace_host<float>(ts, rays, n, hits, nullptr, false);
}
int take_hip_trace_any(TakeScene *ts, const void *rays, int64_t n, int32_t *occluded) {
    if (!ts || (n > 0 && (!rays || !occluded))) return fail(TAKE_E_INVALID, "null argument");
    TAKE_ON_DEVICE(ts);
    return is_f64(ts) ? trace_host<double>(ts, rays, n, nullptr, occluded, true)
                                               : trace_host<float>(ts, rays, n, nullptr, occluded, true);
}
int take_hip_trace_closest_device(TakeScene *ts, const void *d_rays, int64_t n, void *d_hits, int32_t count_mode,
                                  void *stream) {
    if (!ts || (n > 0 && (!d_rays || !d_hits))) return fail(TAKE_E_INVALID, "null argument");
    if (n == 0) return TAKE_OK;
    TAKE_ON_DEVICE(ts);
    return is_f64(ts)
               ? trace_impl<double>(ts, d_rays, n, d_hits, nullptr, false, count_mode != 0, (hipStream_t)stream)
               : trace_impl<float>(ts, d_rays, n, d_hits, nullptr, false, count_mode != 0, (hipStream_t)stream);
}

int take_hip_debug_table(int32_t kind, int32_t precision, const double *in, int64_t n, int32_t in_cols,
                         const double *rnd, double *out, int32_t out_cols) {
    if (!in || !out || !rnd || n < 0) return fail(TAKE_E_INVALID, "null argument");
    int nd = check_device();
    if (nd < 0) return nd;
    if (n == 0) return TAKE_OK;
    DevBuf<double> d_in, d_rnd, d_out;
    DevBuf<ImageInfo> d_img;
    DevBuf<float> d_texf;
    DevBuf<double> d_texd;
    // the fixed 5x4 image the reference harness used for the material / texture tables (oracle/ref_harness.cpp)
    std::vector<float> tf(60);
    std::vector<double> td(60);
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 5; x++) {
            const double c[3] = {0.1 + 0.15 * x + 0.01 * y, 0.9 - 0.2 * y + 0.02 * x, 0.3 + 0.05 * ((x * 3 + y * 7) % 5)};
            for (int a = 0; a < 3; a++) td[3 * (y * 5 + x) + a] = c[a], tf[3 * (y * 5 + x) + a] = (float)c[a];
        }
    std::vector<ImageInfo> img{ImageInfo{5, 4, 0}};
    int rc = TAKE_OK;
    do {
        if (d_in.alloc((size_t)n * in_cols) != hipSuccess || d_rnd.alloc((size_t)n * TAB_RND) != hipSuccess ||
            d_out.alloc((size_t)n * out_cols) != hipSuccess || d_img.upload(img) != hipSuccess ||
            d_texf.upload(tf) != hipSuccess || d_texd.upload(td) != hipSuccess) {
            rc = fail(TAKE_E_NOMEM, "debug table allocation failed");
            break;
        }
        if (hipMemcpy(d_in.p, in, d_in.bytes(), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_rnd.p, rnd, d_rnd.bytes(), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset(d_out.p, 0, d_out.bytes()) != hipSuccess) {
            rc = fail(TAKE_E_DEVICE, "debug table upload failed");
            break;
        }
        const dim3 g((unsigned)((n + BLOCK - 1) / BLOCK)), b(BLOCK);
        if (precision == TAKE_PRECISION_F64) {
            DeviceScene<double> sc{};
            sc.images = d_img.p, sc.texels = d_texd.p;
            hipLaunchKernelGGL((k_debug_table<double>), g, b, 0, nullptr, sc, kind, d_in.p, d_rnd.p, n, d_out.p);
        } else {
            DeviceScene<float> sc{};
            sc.images = d_img.p, sc.texels = d_texf.p;
            hipLaunchKernelGGL((k_debug_table<float>), g, b, 0, nullptr, sc, kind, d_in.p, d_rnd.p, n, d_out.p);
        }
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) {
            rc = fail(TAKE_E_DEVICE, "debug table kernel failed");
            break;
        }
        if (hipMemcpy(out, d_out.p, d_out.bytes(), hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(TAKE_E_DEVICE, "debug table download failed");
    } while (0);
    d_in.release(), d_rnd.release(), d_out.release(), d_img.release(), d_texf.release(), d_texd.release();
    return rc;
}

// ------------------------------------------------------------------------------------------------ scene groups
}  // extern "C"

namespace {
// A replica of `src` on `device`: every device array is copied peer to peer (xGMI between the GPUs of a node), the
// small host tables by value — the scene is prepared and its tree built ONCE per group, whichever builder made it.
template <class T> int peer_copy(DevBuf<T> &dst, int dst_dev, const DevBuf<T> &src, int src_dev) {
    if (dst.alloc(src.n) != hipSuccess) return fail(TAKE_E_NOMEM, "out of device memory for a scene replica");
    if (src.n && hipMemcpyPeer(dst.p, dst_dev, src.p, src_dev, src.bytes()) != hipSuccess)
        return fail(TAKE_E_DEVICE, "hipMemcpyPeer of a scene array failed");
    return TAKE_OK;
}
template <class R> int replicate_t(const SceneT<R> &a, int a_dev, int a_cus, SceneT<R> &b, int b_dev, int b_cus) {
    int rc = TAKE_OK;
#define TK_COPY(member) if (!rc) rc = peer_copy(b.member, b_dev, a.member, a_dev)
    TK_COPY(nodes); TK_COPY(qnodes); TK_COPY(qnodes8); TK_COPY(prims); TK_COPY(meshes); TK_COPY(face_idx); TK_COPY(normals); TK_COPY(uvs);
    TK_COPY(texels); TK_COPY(materials); TK_COPY(images); TK_COPY(lights); TK_COPY(light_pmf); TK_COPY(light_cdf);
    TK_COPY(inst_trace); TK_COPY(inst_shade); TK_COPY(env_marginal); TK_COPY(env_conditional); TK_COPY(env_guide_m);
    TK_COPY(env_guide_c);
#undef TK_COPY
    if (rc) return rc;
    // small host tables (the large vectors were dropped after the upload)
    b.host.cam = a.host.cam;
    b.host.env = a.host.env;
    b.host.stats = a.host.stats;
    b.host.n_material_tags = a.host.n_material_tags, b.host.tag_mask = a.host.tag_mask, b.host.single_tag = a.host.single_tag;
    b.host.q_inflation = a.host.q_inflation, b.host.root_child = a.host.root_child, b.host.node_width = a.host.node_width;
    b.host.n_blas = a.host.n_blas, b.host.blas_nodes = a.host.blas_nodes, b.host.blas_prims = a.host.blas_prims;
    for (int k = 0; k < 3; k++) b.host.grid_lo[k] = a.host.grid_lo[k], b.host.grid_step[k] = a.host.grid_step[k], b.host.background[k] = a.host.background[k];
    DeviceScene<R> &d = b.dev;
    d = a.dev;  // the plain values; then the pointers of this device
    d.nodes = b.nodes.p, d.qnodes = a.dev.qnodes ? b.qnodes.p : nullptr, d.qnodes8 = b.qnodes8.p, d.prims = b.prims.p, d.shapes = nullptr;
    d.meshes = b.meshes.p, d.face_idx = b.face_idx.p, d.normals = b.normals.p, d.uvs = b.uvs.p, d.texels = b.texels.p;
    d.materials = b.materials.p, d.images = b.images.p, d.lights = b.lights.p, d.light_pmf = b.light_pmf.p, d.light_cdf = b.light_cdf.p;
    d.inst_trace = b.inst_trace.p, d.inst_shade = b.inst_shade.p;
    d.env.marginal = b.env_marginal.p, d.env.conditional = b.env_conditional.p, d.env.guide_m = b.env_guide_m.p, d.env.guide_c = b.env_guide_c.p;
    // the persistent trace grid of THIS device: blocks per CU are a property of the kernels (the same code object on
    // every device), the CU count is the replica device's own
    const int per_cu = std::max(1, a.trace_grid / std::max(1, a_cus));
    const int64_t groups_per_block = a.trace_grid > 0 ? a.spill_stride / a.trace_grid : 0;
    const int64_t spill_levels = a.spill_stride > 0 ? (int64_t)a.spill.n / a.spill_stride : 0;
    b.group = a.group, b.built_on_device = a.built_on_device, b.trace_grid = per_cu * b_cus, b.spill_stride = (int64_t)b.trace_grid * groups_per_block;
    if (b.qwords.alloc(a.qwords.n) != hipSuccess || b.counters.alloc(a.counters.n) != hipSuccess ||
        b.spill.alloc((size_t)(b.spill_stride * spill_levels)) != hipSuccess)
        return fail(TAKE_E_NOMEM, "out of device memory for a scene replica");
    HIP_TRY(hipMemset(b.qwords.p, 0, b.qwords.bytes()));
    HIP_TRY(hipMemset(b.counters.p, 0, b.counters.bytes()));
    return TAKE_OK;
}
// -> a new scene handle on `device` (made current for the call), equal to `src`
int replicate_scene(const TakeScene *src, int device, TakeScene **out) {
    *out = nullptr;
    TakeScene *ts = new (std::nothrow) TakeScene();
    if (!ts) return fail(TAKE_E_NOMEM, "out of host memory");
    ts->precision = src->precision, ts->device = device, ts->num_cus = src->num_cus, ts->instrumentation = 0;
    DeviceGuard guard(device);
    int rc = guard.ok ? TAKE_OK : fail(TAKE_E_DEVICE, "cannot make the replica's device current");
    if (!rc) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ts->num_cus = cus;
        (void)hipGetLastError();
    }
    if (!rc) rc = is_f64(src) ? replicate_t(src->d, src->device, src->num_cus, ts->d, device, ts->num_cus)
                              : replicate_t(src->f, src->device, src->num_cus, ts->f, device, ts->num_cus);
    if (!rc && src->precision == TAKE_PRECISION_MIXED) rc = replicate_t(src->f, src->device, src->num_cus, ts->f, device, ts->num_cus);
    if (rc) {
        ts->f.release(), ts->d.release();
        delete ts;
        return rc;
    }
    *out = ts;
    return TAKE_OK;
}
}  // namespace

struct TakeSceneGroup {
    std::vector<TakeScene *> scenes;  // one per shard, each on its device
    std::vector<void *> staging;      // on the first device: shard k's compact rows (k > 0), copied peer to peer
    std::vector<int32_t *> d_rows;    // on the first device: image row of each compact row of shard k
    std::vector<int> n_rows;
    void *d_full = nullptr;           // on the first device: the assembled image (take_hip_group_render)
    int width = 0, height = 0;
    bool f64 = false;
};

namespace {
// compact rows of one shard -> their rows of the full image
template <class R>
__global__ void __launch_bounds__(BLOCK) k_place_rows(const R *__restrict__ src, const int32_t *__restrict__ rows, int n_rows,
                                                      int row_words, R *dst) {
    const int64_t total = (int64_t)n_rows * row_words;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * BLOCK) {
        const int r = (int)(i / row_words), c = (int)(i % row_words);
        dst[(int64_t)rows[r] * row_words + c] = src[i];
    }
}

void group_release(TakeSceneGroup *g) {
    if (!g) return;
    if (!g->scenes.empty() && g->scenes[0]) {
        DeviceGuard guard(g->scenes[0]->device);
        for (void *p : g->staging)
            if (p) (void)hipFree(p);
        for (int32_t *p : g->d_rows)
            if (p) (void)hipFree(p);
        if (g->d_full) (void)hipFree(g->d_full);
    }
    for (TakeScene *ts : g->scenes) take_hip_scene_destroy(ts);
    delete g;
}

int group_render(TakeSceneGroup *g, const TakeRenderOpts &opts, void *d_out) {
    const int n = (int)g->scenes.size();
    const size_t esz = g->f64 ? 8 : 4;
    const int row_words = g->width * 3;
    // every shard renders its strips on its own device, from its own host thread
    std::vector<int> rc(n, TAKE_OK);
    std::vector<std::string> err(n);
    std::vector<std::thread> pool;
    for (int k = 0; k < n; k++)
        pool.emplace_back([&, k] {
            TakeScene *ts = g->scenes[k];
            TakeRenderOpts o = opts;
            o.strip_first = k, o.strip_stride = n;
            if (g->n_rows[k] == 0) return;
            DeviceGuard guard(ts->device);
            if (!guard.ok) {
                rc[k] = TAKE_E_DEVICE, err[k] = "cannot make the shard's device current";
                return;
            }
            int r = g->f64 ? ensure_workspace(ts->d, 0, (int64_t)g->n_rows[k] * g->width)
                           : ensure_workspace(ts->f, 0, (int64_t)g->n_rows[k] * g->width);
            if (!r) r = g->f64 ? render_impl<double>(ts, o, ts->d.out.p, nullptr) : render_impl<float>(ts, o, ts->f.out.p, nullptr);
            if (!r && k > 0) {  // the one exchange: this shard's rows to the first device
                const hipError_t e = hipMemcpyPeer(g->staging[k], g->scenes[0]->device, g->f64 ? (void *)ts->d.out.p : (void *)ts->f.out.p,
                                                   ts->device, (size_t)g->n_rows[k] * row_words * esz);
                if (e != hipSuccess) r = TAKE_E_DEVICE, g_error = std::string("hipMemcpyPeer: ") + hipGetErrorString(e);
            }
            rc[k] = r;
            if (r) err[k] = g_error;  // g_error is thread-local: hand the message to the caller's thread
        });
    for (auto &t : pool) t.join();
    for (int k = 0; k < n; k++)
        if (rc[k]) return fail(rc[k], "shard " + std::to_string(k) + ": " + err[k]);
    // assemble on the first device
    DeviceGuard guard(g->scenes[0]->device);
    if (!guard.ok) return fail(TAKE_E_DEVICE, "cannot make the first device current");
    for (int k = 0; k < n; k++) {
        if (g->n_rows[k] == 0) continue;
        const void *src = k == 0 ? (g->f64 ? (void *)g->scenes[0]->d.out.p : (void *)g->scenes[0]->f.out.p) : g->staging[k];
        const int64_t total = (int64_t)g->n_rows[k] * row_words;
        const dim3 grid((unsigned)std::min<int64_t>((total + BLOCK - 1) / BLOCK, 4096));
        if (g->f64) hipLaunchKernelGGL((k_place_rows<double>), grid, dim3(BLOCK), 0, nullptr, (const double *)src, g->d_rows[k], g->n_rows[k], row_words, (double *)d_out);
        else hipLaunchKernelGGL((k_place_rows<float>), grid, dim3(BLOCK), 0, nullptr, (const float *)src, g->d_rows[k], g->n_rows[k], row_words, (float *)d_out);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return TAKE_OK;
}
}  // namespace

extern "C" {

int take_hip_group_create(const TakeSceneDesc *desc, const TakeBuildOpts *opts, int32_t n_gpus, const int32_t *devices,
                          TakeSceneGroup **out) {
    if (!desc || !out) return fail(TAKE_E_INVALID, "null argument");
    *out = nullptr;
    const int nd = check_device();
    if (nd < 0) return nd;
    if (n_gpus <= 0 || n_gpus > 64) return fail(TAKE_E_INVALID, "n_gpus must be in 1..64");
    for (int k = 0; k < n_gpus; k++) {
        const int dev = devices ? devices[k] : k;
        if (dev < 0 || dev >= nd) return fail(TAKE_E_INVALID, "device " + std::to_string(dev) + " of shard " + std::to_string(k) + " is not visible (" + std::to_string(nd) + " devices)");
    }
    TakeSceneGroup *g = new (std::nothrow) TakeSceneGroup();
    if (!g) return fail(TAKE_E_NOMEM, "out of host memory");
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = TAKE_OK;
    for (int k = 0; k < n_gpus && !rc; k++) {
        const int dev = devices ? devices[k] : k;
        if (hipSetDevice(dev) != hipSuccess) {
            rc = fail(TAKE_E_DEVICE, "hipSetDevice failed");
            break;
        }
        TakeScene *ts = nullptr;
        // the first shard prepares and builds the scene; the others are peer-to-peer copies of its device arrays
        rc = k == 0 ? take_hip_scene_create(desc, opts, &ts) : replicate_scene(g->scenes[0], dev, &ts);
        if (!rc) g->scenes.push_back(ts);
    }
    if (!rc) {
        for (TakeScene *x : g->scenes) {  // shards that share a device share its free memory
            int share = 0;
            for (TakeScene *y : g->scenes) share += y->device == x->device;
            x->mem_share = share;
        }
        TakeScene *t0 = g->scenes[0];
        g->f64 = is_f64(t0);
        g->width = g->f64 ? t0->d.host.cam.width : t0->f.host.cam.width;
        g->height = g->f64 ? t0->d.host.cam.height : t0->f.host.cam.height;
        const size_t esz = g->f64 ? 8 : 4;
        g->staging.assign(n_gpus, nullptr), g->d_rows.assign(n_gpus, nullptr), g->n_rows.assign(n_gpus, 0);
        if (hipSetDevice(t0->device) != hipSuccess) rc = fail(TAKE_E_DEVICE, "hipSetDevice failed");
        for (int k = 0; k < n_gpus && !rc; k++) {
            std::vector<int32_t> rows((size_t)g->height);
            const int nr = rows_of(g->height, k, n_gpus, rows.data());
            g->n_rows[k] = nr;
            if (nr == 0) continue;
            if (hipMalloc((void **)&g->d_rows[k], (size_t)nr * sizeof(int32_t)) != hipSuccess ||
                hipMemcpy(g->d_rows[k], rows.data(), (size_t)nr * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess ||
                (k > 0 && hipMalloc(&g->staging[k], (size_t)nr * g->width * 3 * esz) != hipSuccess))
                rc = fail(TAKE_E_NOMEM, "out of device memory for the strip staging buffers");
            if (!rc && k > 0 && g->scenes[k]->device != t0->device) {
                // direct peer access if the fabric offers it (hipMemcpyPeer works either way)
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, t0->device, g->scenes[k]->device) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(g->scenes[k]->device, 0);
                (void)hipGetLastError();
            }
        }
    }
    (void)hipSetDevice(prev);
    if (rc) {
        const std::string msg = g_error;
        group_release(g);
        return fail(rc, msg);
    }
    *out = g;
    return TAKE_OK;
}

int take_hip_group_destroy(TakeSceneGroup *g) {
    group_release(g);
    return TAKE_OK;
}
int take_hip_group_size(const TakeSceneGroup *g) { return g ? (int)g->scenes.size() : fail(TAKE_E_INVALID, "null group"); }

int take_hip_group_render_device(TakeSceneGroup *g, const TakeRenderOpts *opts, void *d_rgb_out) {
    if (!g || !opts || !d_rgb_out) return fail(TAKE_E_INVALID, "null argument");
    return group_render(g, *opts, d_rgb_out);
}

int take_hip_group_render(TakeSceneGroup *g, const TakeRenderOpts *opts, void *rgb_out_host) {
    if (!g || !opts || !rgb_out_host) return fail(TAKE_E_INVALID, "null argument");
    const size_t bytes = (size_t)g->width * g->height * 3 * (g->f64 ? 8 : 4);
    DeviceGuard guard(g->scenes[0]->device);
    if (!guard.ok) return fail(TAKE_E_DEVICE, "cannot make the first device current");
    if (!g->d_full && hipMalloc(&g->d_full, bytes) != hipSuccess) {
        g->d_full = nullptr;
        return fail(TAKE_E_NOMEM, "out of device memory for the assembled image");
    }
    const int rc = group_render(g, *opts, g->d_full);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(rgb_out_host, g->d_full, bytes, hipMemcpyDeviceToHost));
    return TAKE_OK;
}

int take_hip_group_get_counters(const TakeSceneGroup *g, int32_t k, TakeCounters *out) {
    if (!g || !out || k < 0 || k >= (int)g->scenes.size()) return fail(TAKE_E_INVALID, "bad argument");
    *out = g->scenes[k]->counters;
    return TAKE_OK;
}

int take_hip_get_counters(const TakeScene *ts, TakeCounters *out) {
    if (!ts || !out) return fail(TAKE_E_INVALID, "null argument");
    *out = ts->counters;
    return TAKE_OK;
}
int take_hip_set_instrumentation(TakeScene *ts, int32_t flags) {
    if (!ts) return fail(TAKE_E_INVALID, "null scene");
    ts->instrumentation = flags;
    return TAKE_OK;
}
int take_hip_scene_stats(const TakeScene *ts, int64_t *n_nodes, int64_t *n_prims, int32_t *depth,
                         int64_t *device_bytes) {
    if (!ts) return fail(TAKE_E_INVALID, "null scene");
    const bool f64 = is_f64(ts);
    const WideBvhStats &s = f64 ? ts->d.host.stats : ts->f.host.stats;
    if (n_nodes) *n_nodes = s.n_nodes;
    if (n_prims) *n_prims = s.n_prims;
    if (depth) *depth = s.depth;
    if (device_bytes) *device_bytes = (int64_t)((f64 ? ts->d.scene_bytes() : 0) + (ts->precision != TAKE_PRECISION_F64 ? ts->f.scene_bytes() : 0));
    return TAKE_OK;
}

}  // extern "C"
