#!/bin/bash
# AddressSanitizer + UBSan pass over the HOST layer (CPU build only): every built-in scene is set up and flattened
# (OFF / PPM loaders, SAH KD build with its worker threads, reference-tree analysis, rope flattening), twice.
# Run from the repo root in the build container:  bash tools/host_asan.sh
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
cat > $T/main.cpp <<'CPP'
#include "hrt_host.h"
#include <cstdio>
int main() {
    const char *names[] = {"cornell_box", "cornell_mesh", "random_spheres", "mesh_in_box", "backrooms_pool", "single_sphere", "single_square", "mesh",
                           "rt_in_a_weekend", "debug_refraction", "flamingo", "raccoon", "flamingo_pond", "flamingo_lake"};
    for (int round = 0; round < 2; ++round)
        for (const char *n : names) {
            hrt_host_scene *s = nullptr;
            if (hrt_host_scene_new("assets", &s) != 0) { std::printf("new failed: %s\n", hrt_host_last_error()); return 1; }
            if (hrt_host_scene_setup(s, n, 16.f / 9.f, 1) != 0) { std::printf("%s: setup failed: %s\n", n, hrt_host_last_error()); return 1; }
            const hrt_scene_desc *d = nullptr;
            if (hrt_host_scene_flatten(s, &d) != 0 || !d) { std::printf("%s: flatten failed: %s\n", n, hrt_host_last_error()); return 1; }
            unsigned long tris = 0, units = 0, exc = 0;
            for (unsigned m = 0; m < d->n_meshes; ++m) { tris += d->meshes[m].n_triangles; units += d->meshes[m].n_kd_units; exc += d->meshes[m].n_exceptions; }
            if (round == 0) std::printf("%-18s spheres %u quads %u meshes %u triangles %lu kd units %lu exceptions %lu\n", n, d->n_spheres, d->n_quads, d->n_meshes, tris, units, exc);
            hrt_host_scene_free(s);
        }
    std::printf("host layer: all scenes set up and flattened twice under ASan + UBSan\n");
    return 0;
}
CPP
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -pthread -o $T/host_asan $T/main.cpp hai719-raytracing_amd/host/host_api.cpp hai719-raytracing_amd/host/kdtree.cpp hai719-raytracing_amd/host/mesh.cpp hai719-raytracing_amd/host/ref_tree.cpp hai719-raytracing_amd/host/scene.cpp hai719-raytracing_amd/host/scene_demo.cpp hai719-raytracing_amd/host/scene_pool.cpp
ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 $T/host_asan
rm -rf $T
