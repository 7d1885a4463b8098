// render_hip.cpp — the reference's `Image3 render(const std::vector<std::string>&)` (src/render.h:5) on the MI355X.
//
// A maintainer of TaKe drops this file in place of src/render.cpp and links libtake_hip.so;
// everything else of the reference — main.cpp, the XML/OBJ/PLY parsers, Scene and its variants, imwrite — is used
// as it is.  It is written against the reference's own headers and is compiled, in the authoring container only,
// by `make -C oracle gpu_cli` into oracle/_ref/take_gpu (reference main.cpp + parser + this file), which the
// GPU tests run on the golden XML scenes (tests/test_gpu_dropin.py).
//
// Same argument convention as src/render.cpp:9-23: params[0] is parsed as the scene file, "-max_depth N" may
// appear anywhere after it (default 50), an empty list returns an empty image.  Same messages on stdout for the
// three timed phases.  Errors of the GPU library surface through Error() like the parser's.
// Settings that the reference does not have come from the environment, so that main.cpp stays untouched:
//   TAKE_HIP_PRECISION=f64   render in double (default f32)
//   TAKE_HIP_SEED=<n>        global seed (the reference seeds from std::random_device, src/render.cpp:60)
//   TAKE_HIP_DUMP_PFM=<file> also write the float image through the reference's imwrite (tests)
//   TAKE_HIP_GPUS=<n>        render on the first n GPUs of the node from this one process (take_hip_group_*: scene
//                            replicated, 4-row strips dealt round-robin, strips gathered peer to peer on GPU 0) —
//                            the counterpart of the reference's `-t <threads>` (src/parallel.cpp:183-237);
//                            "n x d" forms like TAKE_HIP_GPUS=4:0 put all 4 shards on device 0 (tests)
#include <chrono>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "image.h"
#include "parse/parse_scene.h"
#include "render.h"
#include "scene.h"
#include "utils/flexception.h"

#include "take_flatten.hpp"
#include "take_hip.h"

namespace {

struct Stopwatch {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double lap() {
        const auto t1 = std::chrono::steady_clock::now();
        const double s = std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
        return s;
    }
};

[[noreturn]] void gpu_error(const char *what) { Error(std::string("take_hip: ") + what + ": " + take_hip_last_error()); }

}  // namespace

Image3 render(const std::vector<std::string> &params) {
    if (params.empty()) return Image3(0, 0);
    int max_depth = 50;
    for (size_t i = 1; i + 1 < params.size(); i++)
        if (params[i] == "-max_depth") max_depth = std::stoi(params[i + 1]);

    Stopwatch sw;
    std::cout << "Parsing and constructing scene " << params[0] << "." << std::endl;
    Scene scene = parse_scene(params[0]);
    scene.options.max_depth = max_depth;
    std::cout << "Scene parsing done. Took " << sw.lap() << " seconds." << std::endl;

    // was: build_bvh(scene)                                                              src/render.cpp:49
    std::cout << "Building BVH..." << std::endl;
    take_hip::FlatScene flat;  // owns the small tables; mesh arrays are referenced in place
    take_hip::flatten_scene(scene, flat);
    const char *prec = std::getenv("TAKE_HIP_PRECISION");
    // TAKE_HIP_PRECISION: f32 (default) | f64 (the reference's Real) | mixed (first bounces f64, the rest f32; images double)
    const bool mixed = prec && std::string(prec) == "mixed";
    const bool f64 = mixed || (prec && std::string(prec) == "f64");
    TakeBuildOpts bo{};
    bo.precision = mixed ? TAKE_PRECISION_MIXED : (f64 ? TAKE_PRECISION_F64 : TAKE_PRECISION_F32);
    // TAKE_HIP_BURLEY=1: the scene's disney* materials get real lobes instead of upstream's Lambert clones (extension)
    if (const char *b = std::getenv("TAKE_HIP_BURLEY")) bo.burley_lobes = std::atoi(b) != 0;
    // TAKE_HIP_GPUS: "<n>" or "<n>:<device>" (all shards on one device)
    int n_gpus = 1, one_device = -1;
    if (const char *g = std::getenv("TAKE_HIP_GPUS")) {
        n_gpus = std::max(1, std::atoi(g));
        if (const char *c = std::strchr(g, ':')) one_device = std::atoi(c + 1);
    }
    std::vector<int32_t> devices(n_gpus);
    for (int k = 0; k < n_gpus; k++) devices[k] = one_device >= 0 ? one_device : k;
    TakeSceneGroup *gpu = nullptr;
    if (take_hip_group_create(&flat.desc, &bo, n_gpus, devices.data(), &gpu) != TAKE_OK) gpu_error("group_create");
    std::cout << "Finish building BVH. Took " << sw.lap() << " seconds." << std::endl;

    // was: parallel_for over 16x16 tiles, path_tracing per sample                        src/render.cpp:59-82
    std::cout << "Rendering..." << std::endl;
    TakeRenderOpts ro{};
    ro.spp = scene.options.spp;
    ro.max_depth = scene.options.max_depth;
    ro.seed = std::getenv("TAKE_HIP_SEED") ? std::strtoull(std::getenv("TAKE_HIP_SEED"), nullptr, 10) : 0;
    Image3 img(scene.camera.width, scene.camera.height);
    const size_t n = (size_t)img.width * img.height;
    int rc;
    if (f64) {
        static_assert(sizeof(Vector3) == 3 * sizeof(double), "Image3 is tightly packed double RGB");
        rc = take_hip_group_render(gpu, &ro, img.data.data());  // Image3 order: row 0 = top (src/render.cpp:78)
    } else {
        std::vector<float> rgb(3 * n);
        rc = take_hip_group_render(gpu, &ro, rgb.data());
        for (size_t i = 0; i < n; i++) img.data[i] = Vector3{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
    }
    if (rc != TAKE_OK) {
        take_hip_group_destroy(gpu);
        gpu_error("render");
    }
    take_hip_group_destroy(gpu);
    std::cout << std::endl << "Finish building rendering. Took " << sw.lap() << " seconds." << std::endl;
    if (const char *dump = std::getenv("TAKE_HIP_DUMP_PFM")) imwrite(dump, img);
    return img;
}
