// Host Scene: the caller-side container of the trace path.
//
// Interface mirrors /root/reference/src/Scene.h:57-188 (containers, addBox,
// load_texture / load_normal_map / loadSkybox, computeKDTrees) and the
// setup_* functions that BASELINE.json's configs need.  The reference keeps
// its containers private and traces them in place; here the only consumer is
// flatten(), which produces the hrt_scene_desc handed over the C ABI.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "../../include/hrt.h"
#include "geom.h"
#include "kdtree.h"
#include "ref_tree.h"

namespace hrt_host {

// Owns every array a hrt_scene_desc points to.
struct FlatScene {
    hrt_scene_desc desc{};
    std::vector<hrt_material> materials;
    std::vector<hrt_sphere> spheres;
    std::vector<hrt_quad> quads;
    std::vector<hrt_mesh> meshes;
    std::vector<hrt_light> lights;
    std::vector<hrt_image> images;
    std::vector<std::vector<uint8_t>> image_bytes;
    std::vector<std::vector<float>> mesh_positions, mesh_vcolors, mesh_fcolors;
    std::vector<std::vector<uint32_t>> mesh_indices;
    std::vector<FlatKDTree> trees;
    std::vector<RefTreeAnalysis> ref_analysis;  // irregular triangles per mesh (ref_tree.h)
};

class Scene {
public:
    std::vector<Mesh> meshes;
    std::vector<Sphere> spheres;
    std::vector<Square> squares;
    std::vector<Light> lights;
    std::vector<ppmLoader::ImageRGB> textures;
    std::vector<ppmLoader::ImageRGB> normals;
    ppmLoader::ImageRGB skybox;
    bool dark_sky = true;

    std::string asset_root = ".";  // directory holding img/ and mesh/ (the reference uses the cwd)
    std::string error;             // first I/O failure, empty if none
    KDBuildParams kd_params;

    void clear();
    void addBox(const std::vector<Material> &materials, const bool faces[6], const Vec3 &pos,
                const Vec3 rotation, float size = 1.f, bool facing_out = true);  // Scene.h:92-146
    void loadSkybox(const std::string &filename);
    int load_texture(const std::string &filename);
    int load_normal_map(const std::string &filename);
    bool load_mesh(Mesh &m, const std::string &filename);  // loadOFF + error capture

    // --- scenes of BASELINE.json configs ---
    void setup_cornell_box(float aspect_ratio);                       // Scene.h:421-619 (cfg 1)
    void setup_cornell_mesh(float aspect_ratio,                       // cfg 2: SURVEY.md 8(c) recipe
                            const std::string &off = "mesh/flamingo_lowpoly.off");
    void setup_random_spheres(uint64_t seed = 1);                     // Scene.h:829-924 (cfg 3), own seeded PRNG (N13)
    void setup_mesh_in_box(float aspect_ratio,                        // cfg 4: triceratops in the Cornell walls
                           const std::string &off = "mesh/triceratops.off");
    void setup_backrooms_pool();                                      // Scene.h:1329-1882 (cfg 5)
    // demo scenes outside BASELINE.json's configs, host/scene_demo.cpp
    void setup_single_sphere();      // Scene.h:358-382
    void setup_single_square();      // Scene.h:384-419
    void setup_mesh();               // Scene.h:714-827
    void setup_rt_in_a_weekend();    // Scene.h:621-712
    void setup_debug_refraction();   // Scene.h:926-998
    void setup_flamingo();           // Scene.h:1000-1078
    void setup_raccoon();            // Scene.h:1080-1207
    void setup_flamingo_pond();      // Scene.h:1209-1262
    void setup_flamingo_lake();      // Scene.h:1264-1327
    // Runs a setup by name: "cornell_box", "cornell_mesh", "random_spheres", "mesh_in_box", "backrooms_pool".
    bool setup_by_name(const std::string &name, float aspect_ratio, uint64_t seed);

    // Builds the flattened KD-trees (Scene::computeKDTrees, Scene.h:352-356) and the description.
    std::unique_ptr<FlatScene> flatten() const;
};

// Default camera of the reference: Camera.cpp:24-37 + main.cpp:418 => eye (0,0,6.1), -Z, fovy 45.
hrt_camera default_camera(float aspect_ratio);

}  // namespace hrt_host
