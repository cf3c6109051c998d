#include "scene.h"

#include <cstdlib>
#include <cstring>
#include <iostream>

namespace hrt_host {

void Scene::clear() {
    meshes.clear();
    spheres.clear();
    squares.clear();
    lights.clear();
    textures.clear();
    normals.clear();
}

void Scene::loadSkybox(const std::string &filename) {
    ppmLoader::load_ppm(skybox, asset_root + "/" + filename);
}
int Scene::load_texture(const std::string &filename) {
    ppmLoader::ImageRGB img;
    if (!ppmLoader::load_ppm(img, asset_root + "/" + filename) && error.empty())
        error = "cannot read texture " + filename;
    textures.push_back(std::move(img));
    return (int)textures.size() - 1;
}
int Scene::load_normal_map(const std::string &filename) {
    ppmLoader::ImageRGB img;
    if (!ppmLoader::load_ppm(img, asset_root + "/" + filename) && error.empty())
        error = "cannot read normal map " + filename;
    normals.push_back(std::move(img));
    return (int)normals.size() - 1;
}
bool Scene::load_mesh(Mesh &m, const std::string &filename) {
    if (m.loadOFF(asset_root + "/" + filename)) return true;
    if (error.empty()) error = "cannot read mesh " + filename;
    return false;
}

// Scene.h:92-146.  `rotation` is accepted and ignored, and every face is a
// 1 x 1 quad whatever `size` says (setQuad re-normalises the edge vectors and
// multiplies by width = height = 1): both are reference behaviour (N12).
void Scene::addBox(const std::vector<Material> &mats, const bool faces[6], const Vec3 &pos,
                   const Vec3 /*rotation*/, float size, bool facing_out) {
    const Vec3 corner = Vec3((float)(-size / 2.));
    const Vec3 right(size, 0.f, 0.f), up(0.f, 0.f, size);
    struct FaceTurn { bool rx; float ax; bool ry; float ay; };
    static const FaceTurn turn[6] = {
        {false, 0.f, false, 0.f},   // bottom
        {true, 180.f, false, 0.f},  // top
        {true, 90.f, false, 0.f},   // front
        {true, -90.f, false, 0.f},  // back
        {true, 90.f, true, 90.f},   // left
        {true, 90.f, true, -90.f},  // right
    };
    const size_t first = squares.size();
    for (int f = 0; f < 6; ++f) {
        if (!faces[f]) continue;
        squares.emplace_back();
        Square &q = squares.back();
        q.setQuad(corner, right, up, 1.f, 1.f);
        if (turn[f].rx) q.rotate_x(turn[f].ax);
        if (turn[f].ry) q.rotate_y(turn[f].ay);
    }
    for (size_t i = first; i < squares.size(); ++i) {
        squares[i].translate(pos);
        squares[i].build_arrays();
        if (!facing_out) squares[i].m_normal *= -1.f;  // no effect on tracing: intersect() rebuilds the normal
        squares[i].material = mats[i - first];
    }
}

namespace {

// The unit quad every wall starts from: setQuad((-1,-1,0),(1,0,0),(0,1,0),2,2).
Square &new_wall(std::vector<Square> &squares, float y0 = -1.f) {
    squares.emplace_back();
    Square &s = squares.back();
    s.setQuad(Vec3(-1.f, y0, 0.f), Vec3(1.f, 0.f, 0.f), Vec3(0.f, 1.f, 0.f), 2.f, 2.f);
    return s;
}

void shiny(Material &m, const Vec3 &albedo, const Vec3 &spec, double shininess) {
    m.diffuse_material = albedo;
    m.specular_material = spec;
    m.shininess = shininess;
}

void image_textured(Material &m, int tex, int nmap) {
    m.texture_type = Texture_Image;
    m.set_texture(tex);
    m.set_normals(nmap);
}

}  // namespace

// ---------------------------------------------------------------------------
// cfg 1 -- Scene.h:421-619: 5-face emissive light fixture (addBox), six walls,
// a glass and a mirror sphere, no point light.
// ---------------------------------------------------------------------------
void Scene::setup_cornell_box(float aspect_ratio) {
    clear();
    skybox = ppmLoader::ImageRGB();
    const int brick = load_texture("img/planeTextures/brickwall.ppm");
    const int brick_n = load_normal_map("img/normalMaps/brickwall_normal.ppm");
    const int floor_n = load_normal_map("img/normalMaps/n1.ppm");
    const int sand = load_texture("img/planeTextures/sand.ppm");
    (void)load_normal_map("img/normalMaps/water_normal.ppm");  // loaded, never bound (Scene.h:428)

    {   // light fixture: emissive bottom + four white sides around (0,1.95,0)
        Material white;
        shiny(white, Vec3(0.9f), Vec3(1.f), 16);
        Material lamp;
        lamp.emissive = true;
        lamp.light_color = Vec3(1.f);
        lamp.light_intensity = 60.f;
        std::vector<Material> mats{lamp, white, white, white, white};
        const bool faces[6] = {true, false, true, true, true, true};
        addBox(mats, faces, Vec3(0.f, 1.95f, 0.f), Vec3(45.f), 1.f, false);
    }
    const float a = aspect_ratio;
    const float wide = (float)(2. * a);            // 2.*aspect_ratio
    const float side_z = (float)(-2. * (-a));      // -2.*(-aspect_ratio)
    {   // back wall
        Square &s = new_wall(squares);
        s.scale(Vec3(wide, 2.f, 1.f));
        s.translate(Vec3(0.f, 0.f, -2.f));
        s.build_arrays();
        shiny(s.material, Vec3(1.f, 1.f, 1.f), Vec3(1.f, 1.f, 1.f), 16);
        image_textured(s.material, brick, brick_n);
        s.material.texture_scale_x = (float)(1. * a);
        s.material.texture_scale_y = 1.f;
    }
    {   // left wall (red)
        Square &s = new_wall(squares);
        s.rotate_x(180.f);
        s.scale(Vec3(2.f, 2.f, 1.f));
        s.translate(Vec3(0.f, 0.f, side_z));
        s.rotate_y(90.f);
        s.build_arrays();
        shiny(s.material, Vec3(1.f, 0.f, 0.f), Vec3(1.f, 0.f, 0.f), 16);
        image_textured(s.material, brick, brick_n);
    }
    {   // right wall (green)
        Square &s = new_wall(squares);
        s.rotate_x(180.f);
        s.translate(Vec3(0.f, 0.f, side_z));
        s.scale(Vec3(2.f, 2.f, 1.f));
        s.rotate_y(-90.f);
        s.build_arrays();
        shiny(s.material, Vec3(0.f, 1.f, 0.f), Vec3(0.f, 1.f, 0.f), 16);
        image_textured(s.material, brick, brick_n);
    }
    {   // floor
        Square &s = new_wall(squares);
        s.translate(Vec3(0.f, 0.f, -2.f));
        s.scale(Vec3(wide, 2.f, 1.f));
        s.rotate_x(-90.f);
        s.build_arrays();
        shiny(s.material, Vec3((float)(246. / 255.), (float)(204. / 255.), (float)(162. / 255.)), Vec3(1.f, 1.f, 1.f), 1);
        image_textured(s.material, sand, floor_n);
    }
    {   // ceiling (checkerboard)
        Square &s = new_wall(squares);
        s.translate(Vec3(0.f, 0.f, -2.f));
        s.scale(Vec3(wide, 2.f, 1.f));
        s.rotate_x(90.f);
        s.build_arrays();
        shiny(s.material, Vec3(1.f, 1.f, 1.f), Vec3(1.f, 1.f, 1.f), 16);
        s.material.texture_type = Texture_Checkerboard;
        s.material.checkerboard_color1 = Vec3(0.95f);
        s.material.checkerboard_color2 = Vec3(0.5f);
        s.material.texture_scale_x = (float)(8. * a);
        s.material.texture_scale_y = 8.f;
    }
    {   // front wall (faces away from the camera: culled for camera rays, N8)
        Square &s = new_wall(squares);
        s.translate(Vec3(0.f, 0.f, -2.f));
        s.scale(Vec3(wide, 2.f, 1.f));
        s.rotate_y(180.f);
        s.build_arrays();
        shiny(s.material, Vec3(1.f, 1.f, 1.f), Vec3(1.f, 1.f, 1.f), 16);
        image_textured(s.material, brick, brick_n);
    }
    {   // glass sphere
        spheres.emplace_back(Vec3(1.0f, -1.25f, 0.5f), 0.75f);
        Material &m = spheres.back().material;
        m.type = Material_Glass;
        shiny(m, Vec3(1.f), Vec3(1.f), 16);
        m.transparency = 1.0f;
        m.index_medium = 1.4f;
    }
    {   // mirror sphere
        spheres.emplace_back(Vec3(-1.0f, -1.25f, -0.5f), 0.75f);
        Material &m = spheres.back().material;
        m.type = Material_Mirror;
        shiny(m, Vec3(0.7f), Vec3(1.f, 1.f, 1.f), 16);
        m.transparency = 0.f;
        m.index_medium = 0.f;
    }
}

// ---------------------------------------------------------------------------
// cfg 2 -- Cornell box + a mesh standing on the floor between the spheres.
// Not a reference setup_*: BASELINE.json config 2, transform from SURVEY 8(c).
// ---------------------------------------------------------------------------
void Scene::setup_cornell_mesh(float aspect_ratio, const std::string &off) {
    setup_cornell_box(aspect_ratio);
    meshes.emplace_back();
    Mesh &m = meshes.back();
    if (!load_mesh(m, off)) { meshes.pop_back(); return; }
    m.centerAndScaleToUnit();
    m.scale(Vec3(1.9f));
    m.rotate_x(90.f);
    m.rotate_y(180.f);
    m.rotate_z(180.f);
    m.computeAABB();
    m.translate(Vec3(0.f, -2.f - m.aabb.p0[1] + 0.001f, -0.6f));
    m.build_arrays();
    shiny(m.material, Vec3((float)(237. / 255.), (float)(149. / 255.), (float)(218. / 255.)), Vec3(1.f), 16);
}

// ---------------------------------------------------------------------------
// cfg 4 -- the Cornell walls (spheres removed) around triceratops.off.
// ---------------------------------------------------------------------------
void Scene::setup_mesh_in_box(float aspect_ratio, const std::string &off) {
    setup_cornell_box(aspect_ratio);
    spheres.clear();
    meshes.emplace_back();
    Mesh &m = meshes.back();
    if (!load_mesh(m, off)) { meshes.pop_back(); return; }
    m.centerAndScaleToUnit();
    m.scale(Vec3(1.6f));
    m.rotate_y(30.f);
    m.computeAABB();
    m.translate(Vec3(0.f, -2.f - m.aabb.p0[1] + 0.001f, -0.3f));
    m.build_arrays();
    shiny(m.material, Vec3(0.55f, 0.75f, 0.35f), Vec3(1.f), 16);
}

namespace {
// Scene.h:892-922 draws from rand() (unseeded) and a time-seeded mt19937, so
// the reference scene differs on every run (N13).  This is the build's own
// generator: 32-bit xorshift-multiply counter hash, documented in DESIGN.md.
struct SceneRng {
    uint32_t key, ctr = 0;
    explicit SceneRng(uint64_t seed) : key((uint32_t)(seed * 0x9E3779B97F4A7C15ull >> 32) ^ (uint32_t)seed) {}
    uint32_t next_u32() {
        uint32_t x = key + (ctr++) * 0x9E3779B9u;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        return x;
    }
    float unit() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }
    float range(float lo, float hi) { return lo + (hi - lo) * unit(); }  // random_float(min,max), Functions.cpp:10-12
};
}  // namespace

// ---------------------------------------------------------------------------
// cfg 3 -- Scene.h:829-924: floor quad, 3 fixed + 79 random spheres with
// vertical motion blur, one spherical light, gradient sky.
// ---------------------------------------------------------------------------
void Scene::setup_random_spheres(uint64_t seed) {
    clear();
    skybox = ppmLoader::ImageRGB();
    dark_sky = false;
    const int nSpheres = 79;
    {
        lights.emplace_back();
        Light &l = lights.back();
        l.pos = Vec3(-1.0f, 8.f, 2.0f);
        l.radius = 1.5f;
        l.powerCorrection = 2.f;
        l.material = Vec3(1.f, 1.f, 1.f);
    }
    {   // floor
        Square &s = new_wall(squares, -0.2f);
        s.translate(Vec3(0.f, 0.f, -4.f));
        s.scale(Vec3(100.f, 100.f, 1.f));
        s.rotate_x(-90.f);
        s.build_arrays();
        s.material.diffuse_material = Vec3(0.8f, 0.8f, 0.f);
        s.material.specular_material = Vec3(1.f, 1.f, 1.f);
    }
    auto fixed = [&](Vec3 c, float r, MaterialType t, double shin) {
        spheres.emplace_back(c, r);
        Material &m = spheres.back().material;
        m.type = t;
        shiny(m, Vec3(0.8f), Vec3(0.8f), shin);
    };
    fixed(Vec3(-3.f, 0.f, -22.f), 4.f, Material_Mirror, 32);
    fixed(Vec3(4.f, -2.f, -15.f), 2.f, Material_Mirror, 32);
    fixed(Vec3(-1.f, -2.5f, -8.f), 1.5f, Material_Glass, 20);

    SceneRng rng(seed);
    for (int i = 0; i < nSpheres; ++i) {
        const float height = rng.range(0.25f, 1.f);
        const float radius = rng.range(0.25f, 1.5f);
        const int type = (int)(rng.next_u32() % 3u);
        const float cx = rng.range(-30.f, 30.f);
        const float cz = rng.range(-50.f, -2.f);
        spheres.emplace_back(Vec3(cx, -4 + radius + height, cz), radius);
        Material &m = spheres.back().material;
        auto rgb = [&]() { float r = rng.unit(), g = rng.unit(), b = rng.unit(); return Vec3(r, g, b); };
        switch (type) {
            case 0:
                m.type = Material_Mirror;
                m.diffuse_material = rgb();
                m.specular_material = rgb();
                m.shininess = rng.range(32.f, 100.f);
                break;
            case 1:
                m.type = Material_Glass;
                m.diffuse_material = Vec3(rng.range(0.7f, 1.f));
                m.specular_material = Vec3(rng.range(0.7f, 1.f));
                m.shininess = rng.range(32.f, 70.f);
                m.transparency = rng.range(0.7f, 1.f);
                m.index_medium = rng.range(1.f, 2.f);
                break;
            default:
                m.diffuse_material = rgb();
                m.specular_material = rgb();
                m.shininess = rng.range(0.f, 30.f);
                break;
        }
        m.motion_blur_translation = Vec3(0.f, height, 0.f);
    }
}

bool Scene::setup_by_name(const std::string &name, float aspect_ratio, uint64_t seed) {
    error.clear();
    if (name == "cornell_box") setup_cornell_box(aspect_ratio);
    else if (name == "cornell_mesh") setup_cornell_mesh(aspect_ratio);
    else if (name == "random_spheres") setup_random_spheres(seed);
    else if (name == "mesh_in_box") setup_mesh_in_box(aspect_ratio);
    else if (name == "backrooms_pool") setup_backrooms_pool();
    else if (name == "single_sphere") setup_single_sphere();
    else if (name == "single_square") setup_single_square();
    else if (name == "mesh") setup_mesh();
    else if (name == "rt_in_a_weekend") setup_rt_in_a_weekend();
    else if (name == "debug_refraction") setup_debug_refraction();
    else if (name == "flamingo") setup_flamingo();
    else if (name == "raccoon") setup_raccoon();
    else if (name == "flamingo_pond") setup_flamingo_pond();
    else if (name == "flamingo_lake") setup_flamingo_lake();
    else { error = "unknown scene '" + name + "'"; return false; }
    return error.empty();
}

// ---------------------------------------------------------------------------
// flatten: objects -> hrt_scene_desc.  Material i belongs to object i in the
// order spheres, squares, meshes (each reference object owns its Material).
// ---------------------------------------------------------------------------
static hrt_material flat_material(const Material &m, int n_textures) {
    hrt_material o;
    std::memset(&o, 0, sizeof(o));
    for (int k = 0; k < 3; ++k) {
        o.albedo[k] = m.diffuse_material[k];
        o.checker1[k] = m.checkerboard_color1[k];
        o.checker2[k] = m.checkerboard_color2[k];
        o.light_color[k] = m.light_color[k];
        o.motion[k] = m.motion_blur_translation[k];
    }
    o.transparency = m.transparency;
    o.index_medium = m.index_medium;
    o.type = (int32_t)m.type;
    o.texture_type = (int32_t)m.texture_type;
    o.tex_scale_x = m.texture_scale_x;
    o.tex_scale_y = m.texture_scale_y;
    o.emissive = m.emissive ? 1 : 0;
    o.light_intensity = m.light_intensity;
    o.image = m.image;                                              // textures come first in images[]
    o.normal_map = m.has_normal_map ? n_textures + m.normals : -1;  // then the normal maps
    return o;
}

std::unique_ptr<FlatScene> Scene::flatten() const {
    std::unique_ptr<FlatScene> fs(new FlatScene());
    FlatScene &f = *fs;
    const int n_tex = (int)textures.size();

    auto push_image = [&](const ppmLoader::ImageRGB &img) {
        f.image_bytes.emplace_back((size_t)std::max(img.w, 0) * (size_t)std::max(img.h, 0) * 3);
        std::vector<uint8_t> &bytes = f.image_bytes.back();
        for (size_t i = 0; i < bytes.size() / 3; ++i) {
            bytes[3 * i] = img.data[i].r; bytes[3 * i + 1] = img.data[i].g; bytes[3 * i + 2] = img.data[i].b;
        }
        hrt_image hi;
        hi.w = img.w; hi.h = img.h; hi.rgb = nullptr;
        f.images.push_back(hi);
    };
    for (const auto &t : textures) push_image(t);
    for (const auto &n : normals) push_image(n);
    int sky = -1;
    if (skybox.w >= 1 && skybox.h >= 1) { push_image(skybox); sky = (int)f.images.size() - 1; }
    for (size_t i = 0; i < f.images.size(); ++i) f.images[i].rgb = f.image_bytes[i].data();

    for (const Sphere &s : spheres) {
        hrt_sphere o;
        for (int k = 0; k < 3; ++k) o.center[k] = s.m_center[k];
        o.radius = s.m_radius;
        o.material = (int32_t)f.materials.size();
        f.materials.push_back(flat_material(s.material, n_tex));
        f.spheres.push_back(o);
    }
    for (const Square &q : squares) {
        hrt_quad o;
        for (int k = 0; k < 3; ++k) {
            o.v0[k] = q.vertices[0].position[k];
            o.v1[k] = q.vertices[1].position[k];
            o.v3[k] = q.vertices[3].position[k];
            o.tangent[k] = q.m_right_vector[k];
            o.bitangent[k] = q.m_up_vector[k];
        }
        o.material = (int32_t)f.materials.size();
        f.materials.push_back(flat_material(q.material, n_tex));
        f.quads.push_back(o);
    }
    const size_t nm = meshes.size();
    f.mesh_positions.resize(nm); f.mesh_vcolors.resize(nm); f.mesh_fcolors.resize(nm);
    f.mesh_indices.resize(nm); f.trees.resize(nm); f.ref_analysis.resize(nm);
    for (size_t mi = 0; mi < nm; ++mi) {
        const Mesh &m = meshes[mi];
        hrt_mesh o;
        std::memset(&o, 0, sizeof(o));
        std::vector<float> &pos = f.mesh_positions[mi];
        std::vector<uint32_t> &idx = f.mesh_indices[mi];
        pos.resize(3 * m.vertices.size());
        std::vector<float> scaled(pos.size());
        for (size_t v = 0; v < m.vertices.size(); ++v)
            for (int k = 0; k < 3; ++k) {
                pos[3 * v + k] = m.vertices[v].position[k];
                scaled[3 * v + k] = m.vertices[v].position[k] * HRT_TRIANGLE_SCALING;  // KDTree.cpp:38-40
            }
        idx.resize(3 * m.triangles.size());
        for (size_t t = 0; t < m.triangles.size(); ++t)
            for (int k = 0; k < 3; ++k) idx[3 * t + k] = m.triangles[t][k];
        if (m.colorType == ColorType_Vertex) {
            f.mesh_vcolors[mi].resize(3 * m.vertColors.size());
            for (size_t v = 0; v < m.vertColors.size(); ++v)
                for (int k = 0; k < 3; ++k) f.mesh_vcolors[mi][3 * v + k] = m.vertColors[v][k];
        } else if (m.colorType == ColorType_Face) {
            f.mesh_fcolors[mi].resize(3 * m.faceColors.size());
            for (size_t t = 0; t < m.faceColors.size(); ++t)
                for (int k = 0; k < 3; ++k) f.mesh_fcolors[mi][3 * t + k] = m.faceColors[t][k];
        }
        // Mesh::computeKDTree (Mesh.cpp:107-110): recompute the box, then build.
        Mesh boxed = m;
        boxed.computeAABB();
        const float amin[3] = {boxed.aabb.p0[0], boxed.aabb.p0[1], boxed.aabb.p0[2]}, amax[3] = {boxed.aabb.p1[0], boxed.aabb.p1[1], boxed.aabb.p1[2]};
        // What the reference's own tree does to this mesh that no other tree would reproduce (ref_tree.h): dropped
        // triangles and slivers stay out of the SAH tree and travel as (triangle, reference leaf box) exceptions.
        if (std::getenv("HRT_ABL_NO_EXCEPTIONS")) f.ref_analysis[mi].irregular.assign(m.triangles.size(), 0);  // timing experiments only: not parity-safe
        else f.ref_analysis[mi] = analyse_reference_tree(pos.data(), (uint32_t)m.vertices.size(), idx.data(), (uint32_t)m.triangles.size(),
                                                         amin, amax);
        f.trees[mi] = build_flat_kdtree(scaled.data(), (uint32_t)m.vertices.size(), idx.data(),
                                        (uint32_t)m.triangles.size(), kd_params, f.ref_analysis[mi].irregular.data());
        o.n_vertices = (uint32_t)m.vertices.size();
        o.n_triangles = (uint32_t)m.triangles.size();
        o.positions = pos.data();
        o.indices = idx.data();
        o.color_type = (int32_t)m.colorType;
        o.vert_colors = f.mesh_vcolors[mi].empty() ? nullptr : f.mesh_vcolors[mi].data();
        o.face_colors = f.mesh_fcolors[mi].empty() ? nullptr : f.mesh_fcolors[mi].data();
        for (int k = 0; k < 3; ++k) { o.aabb_min[k] = boxed.aabb.p0[k]; o.aabb_max[k] = boxed.aabb.p1[k]; }
        o.material = (int32_t)f.materials.size();
        f.materials.push_back(flat_material(m.material, n_tex));
        o.kd_root = f.trees[mi].root;
        for (int k = 0; k < 3; ++k) { o.kd_min[k] = f.trees[mi].root_lo[k]; o.kd_max[k] = f.trees[mi].root_hi[k]; }
        o.n_kd_units = (uint32_t)f.trees[mi].units.size();
        o.kd_units = f.trees[mi].units.data();
        o.n_leaf_tris = (uint32_t)f.trees[mi].leaf_tris.size();
        o.leaf_tris = f.trees[mi].leaf_tris.data();
        o.n_exceptions = (uint32_t)f.ref_analysis[mi].exceptions.size();
        o.exceptions = f.ref_analysis[mi].exceptions.empty() ? nullptr : f.ref_analysis[mi].exceptions.data();
        f.meshes.push_back(o);
    }
    for (const Light &l : lights) {
        hrt_light o;
        for (int k = 0; k < 3; ++k) { o.pos[k] = l.pos[k]; o.color[k] = l.material[k]; }
        o.radius = l.radius;
        f.lights.push_back(o);
    }
    hrt_scene_desc &d = f.desc;
    d.n_materials = (uint32_t)f.materials.size(); d.materials = f.materials.data();
    d.n_spheres = (uint32_t)f.spheres.size();     d.spheres = f.spheres.data();
    d.n_quads = (uint32_t)f.quads.size();         d.quads = f.quads.data();
    d.n_meshes = (uint32_t)f.meshes.size();       d.meshes = f.meshes.data();
    d.n_lights = (uint32_t)f.lights.size();       d.lights = f.lights.data();
    d.n_images = (uint32_t)f.images.size();       d.images = f.images.data();
    d.dark_sky = dark_sky ? 1 : 0;
    d.skybox_image = sky;
    return fs;
}

hrt_camera default_camera(float aspect_ratio) {
    hrt_camera c;
    std::memset(&c, 0, sizeof(c));
    c.eye[0] = 0.f; c.eye[1] = 0.f; c.eye[2] = 6.1f;  // -(-3.1) + zoom 3 (main.cpp:418, Camera.cpp:37,130)
    c.right[0] = 1.f; c.up[1] = 1.f; c.forward[2] = -1.f;
    c.fovy_deg = 45.f;
    c.aspect = aspect_ratio;
    c.znear = 4.1f;
    c.zfar = 10000.f;
    return c;
}

}  // namespace hrt_host
