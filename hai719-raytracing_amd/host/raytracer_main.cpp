// Headless driver: what the reference does on key 'r' (main.cpp:321-325 ->
// ray_trace_from_camera(), main.cpp:200-263), with the GLUT window replaced by
// command-line arguments.  Scene set-up stays in host C++; the per-pixel x
// per-sample loop runs on the GPU behind hrt_render().
//
//   raytracer [--scene cornell_box|cornell_mesh|random_spheres|mesh_in_box|backrooms_pool]
//             [--w 850] [--h 480] [--spp 20] [--seed 1] [--out ./rendu.ppm] [--assets DIR] [--gpu 0]
//             [--kd gpu]                         build the KD-trees' split search on the GPU (hrt_kd_build_gpu; the same trees)
//             [--gpus N | --devices 0,1,2,...]   image tiles across several GPUs of this node (hrt_multi_*; an ordinal may repeat)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/hrt.h"
#include "scene.h"

using namespace hrt_host;

static unsigned int SCREENWIDTH = 850, SCREENHEIGHT = 480;  // main.cpp:52-53
static unsigned int nsamples = 20;                          // DEFAULT_NSAMPLES, Constants.h:10
static Scene scene;
static hrt_scene *device_scene = nullptr;
static uint64_t seed = 1;
static std::string out_path = "./rendu.ppm";

// Drop-in for ray_trace_from_camera(): same inputs (current scene, nsamples, window size, camera),
// same output file and quantisation; returns non-zero instead of printing-and-returning on failure.
static hrt_multi *multi = nullptr;  // --gpus / --devices: the same frame, tiles across several GPUs

static int ray_trace_from_camera() {
    const unsigned w = SCREENWIDTH, h = SCREENHEIGHT;
    std::vector<float> image((size_t)w * h * 3, 0.f);
    const hrt_camera cam = default_camera((float)w / (float)h);
    std::cout << "Ray tracing a " << w << " x " << h << " image on the GPU using " << nsamples
              << " samples per pixel" << std::endl;
    hrt_stats st;
    int rc = multi ? hrt_multi_render(multi, &cam, w, h, nsamples, seed, HRT_FLAG_GAMMA, image.data(), &st)
                   : hrt_render(device_scene, &cam, w, h, nsamples, seed, HRT_FLAG_GAMMA, image.data(), &st);
    if (rc != HRT_OK) {
        std::cout << "hrt_render failed: " << hrt_last_error() << std::endl;
        return rc;
    }
    std::cout << "  Done in " << st.total_ms / 1000.0 << " seconds (kernel " << st.kernel_ms << " ms, "
              << (double)st.samples / st.kernel_ms / 1e3 << " Msamples/s)" << std::endl;
    rc = hrt_write_ppm(out_path.c_str(), image.data(), w, h);
    if (rc != HRT_OK) std::cout << hrt_last_error() << std::endl;
    return rc;
}

int main(int argc, char **argv) {
    std::string name = "cornell_box", assets = "assets";
    int gpu = 0;
    bool kd_on_gpu = false;
    std::vector<int> devices;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i], v = argv[i + 1];
        if (k == "--scene") name = v;
        else if (k == "--w") SCREENWIDTH = (unsigned)atoi(v.c_str());
        else if (k == "--h") SCREENHEIGHT = (unsigned)atoi(v.c_str());
        else if (k == "--spp") nsamples = (unsigned)atoi(v.c_str());
        else if (k == "--seed") seed = strtoull(v.c_str(), nullptr, 10);
        else if (k == "--out") out_path = v;
        else if (k == "--assets") assets = v;
        else if (k == "--gpu") gpu = atoi(v.c_str());
        else if (k == "--kd") kd_on_gpu = v == "gpu";
        else if (k == "--gpus") { devices.clear(); for (int d = 0; d < atoi(v.c_str()); ++d) devices.push_back(d); }
        else if (k == "--devices") {
            devices.clear();
            for (size_t a = 0; a < v.size();) { size_t b = v.find(',', a); if (b == std::string::npos) b = v.size(); devices.push_back(atoi(v.substr(a, b - a).c_str())); a = b + 1; }
        }
        else { std::cerr << "unknown option " << k << std::endl; return 2; }
    }
    scene.asset_root = assets;
    if (!scene.setup_by_name(name, (float)SCREENWIDTH / (float)SCREENHEIGHT, seed)) {
        std::cerr << scene.error << std::endl;
        return EXIT_FAILURE;  // the reference exit()s on a missing mesh (Mesh.cpp:12-13)
    }
    if (kd_on_gpu) {  // the trees' split search on the device (hrt_kd_build_gpu): the same trees, built before the scene is uploaded
        if (hrt_init(devices.empty() ? gpu : devices[0]) != HRT_OK) { std::cerr << hrt_last_error() << std::endl; return EXIT_FAILURE; }
        scene.kd_params.builder = hrt_kd_build_gpu;
    }
    std::unique_ptr<FlatScene> flat = scene.flatten();
    const int up = devices.empty() ? (hrt_init(gpu) != HRT_OK ? HRT_ERR_DEVICE : hrt_scene_create(&flat->desc, &device_scene))
                                   : hrt_multi_create(&flat->desc, (uint32_t)devices.size(), devices.data(), &multi);
    if (up != HRT_OK) {
        std::cerr << hrt_last_error() << std::endl;
        return EXIT_FAILURE;
    }
    if (multi) {  // which gather the tiles take to slot 0, and anything creation fell back from
        const std::string note = hrt_last_error();
        std::cout << "Image tiles across " << devices.size() << " GPU slot(s), gather: " << hrt_multi_gather(multi)
                  << (note.empty() ? "" : " (" + note + ")") << std::endl;
    }
    int rc = ray_trace_from_camera();  // the 'r' key
    hrt_scene_destroy(device_scene);
    hrt_multi_destroy(multi);
    hrt_shutdown();
    return rc == HRT_OK ? 0 : EXIT_FAILURE;
}
