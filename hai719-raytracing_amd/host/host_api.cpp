// C ABI over the host scene layer (include/hrt_host.h).
#include <cstring>
#include <memory>
#include <string>

#include "../../include/hrt_host.h"
#include "scene.h"

using namespace hrt_host;

struct hrt_host_scene {
    Scene scene;
    std::unique_ptr<FlatScene> flat;
};

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

static Material to_material(const hrt_material *m) {
    Material o;
    if (!m) return o;
    o.diffuse_material = Vec3(m->albedo[0], m->albedo[1], m->albedo[2]);
    o.transparency = m->transparency;
    o.index_medium = m->index_medium;
    o.type = (MaterialType)m->type;
    o.texture_type = (TextureType)m->texture_type;
    o.checkerboard_color1 = Vec3(m->checker1[0], m->checker1[1], m->checker1[2]);
    o.checkerboard_color2 = Vec3(m->checker2[0], m->checker2[1], m->checker2[2]);
    o.texture_scale_x = m->tex_scale_x;
    o.texture_scale_y = m->tex_scale_y;
    o.emissive = m->emissive != 0;
    o.light_color = Vec3(m->light_color[0], m->light_color[1], m->light_color[2]);
    o.light_intensity = m->light_intensity;
    if (m->image >= 0) o.set_texture(m->image);
    if (m->normal_map >= 0) o.set_normals(m->normal_map);
    o.motion_blur_translation = Vec3(m->motion[0], m->motion[1], m->motion[2]);
    return o;
}

static ppmLoader::ImageRGB to_image(int32_t w, int32_t h, const uint8_t *rgb) {
    ppmLoader::ImageRGB img;
    img.w = w;
    img.h = h;
    img.data.resize((size_t)w * h);
    for (size_t i = 0; i < img.data.size(); ++i) img.data[i] = ppmLoader::RGB{rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
    return img;
}

extern "C" {

const char *hrt_host_last_error(void) { return g_err.c_str(); }

int hrt_host_scene_new(const char *asset_root, hrt_host_scene **out) {
    if (!out) return fail(HRT_ERR_INVALID, "hrt_host_scene_new: out is NULL");
    try {
        hrt_host_scene *s = new hrt_host_scene();
        if (asset_root && *asset_root) s->scene.asset_root = asset_root;
        *out = s;
        return HRT_OK;
    } catch (const std::exception &e) {
        return fail(HRT_ERR_STATE, e.what());
    }
}

void hrt_host_scene_free(hrt_host_scene *s) { delete s; }

int hrt_host_scene_setup(hrt_host_scene *s, const char *name, float aspect_ratio, uint64_t seed) {
    if (!s || !name) return fail(HRT_ERR_INVALID, "hrt_host_scene_setup: NULL argument");
    try {
        s->flat.reset();
        if (!s->scene.setup_by_name(name, aspect_ratio, seed)) {
            const bool unknown = s->scene.error.rfind("unknown scene", 0) == 0;
            return fail(unknown ? HRT_ERR_INVALID : HRT_ERR_IO, s->scene.error);
        }
        return HRT_OK;
    } catch (const std::exception &e) {
        return fail(HRT_ERR_STATE, e.what());
    }
}

int hrt_host_scene_clear(hrt_host_scene *s) {
    if (!s) return fail(HRT_ERR_INVALID, "NULL scene");
    s->scene.clear();
    s->scene.skybox = ppmLoader::ImageRGB();
    s->scene.dark_sky = true;
    s->flat.reset();
    return HRT_OK;
}

int hrt_host_scene_add_texture(hrt_host_scene *s, int32_t w, int32_t h, const uint8_t *rgb) {
    if (!s || !rgb || w < 1 || h < 1) return fail(HRT_ERR_INVALID, "add_texture: bad argument");
    s->scene.textures.push_back(to_image(w, h, rgb));
    return (int)s->scene.textures.size() - 1;
}
int hrt_host_scene_add_normal_map(hrt_host_scene *s, int32_t w, int32_t h, const uint8_t *rgb) {
    if (!s || !rgb || w < 1 || h < 1) return fail(HRT_ERR_INVALID, "add_normal_map: bad argument");
    s->scene.normals.push_back(to_image(w, h, rgb));
    return (int)s->scene.normals.size() - 1;
}

int hrt_host_scene_set_skybox(hrt_host_scene *s, int32_t w, int32_t h, const uint8_t *rgb) {
    if (!s) return fail(HRT_ERR_INVALID, "set_skybox: NULL scene");
    if (!rgb || w < 1 || h < 1) { s->scene.skybox = ppmLoader::ImageRGB(); return HRT_OK; }  // no skybox: dark_sky decides
    s->scene.skybox = to_image(w, h, rgb);
    return HRT_OK;
}

int hrt_host_scene_add_sphere(hrt_host_scene *s, const float c[3], float radius, const hrt_material *m) {
    if (!s || !c) return fail(HRT_ERR_INVALID, "add_sphere: NULL argument");
    s->scene.spheres.emplace_back(Vec3(c[0], c[1], c[2]), radius);
    s->scene.spheres.back().material = to_material(m);
    return HRT_OK;
}

int hrt_host_scene_add_quad(hrt_host_scene *s, const float bl[3], const float r[3], const float u[3],
                            float width, float height, const hrt_material *m) {
    if (!s || !bl || !r || !u) return fail(HRT_ERR_INVALID, "add_quad: NULL argument");
    s->scene.squares.emplace_back();
    Square &q = s->scene.squares.back();
    q.setQuad(Vec3(bl[0], bl[1], bl[2]), Vec3(r[0], r[1], r[2]), Vec3(u[0], u[1], u[2]), width, height);
    q.build_arrays();
    q.material = to_material(m);
    return HRT_OK;
}

int hrt_host_scene_add_quad_ex(hrt_host_scene *s, const float v0[3], const float v1[3], const float v2[3], const float v3[3],
                               const float tangent[3], const float bitangent[3], const hrt_material *m) {
    if (!s || !v0 || !v1 || !v2 || !v3 || !tangent || !bitangent) return fail(HRT_ERR_INVALID, "add_quad_ex: NULL argument");
    s->scene.squares.emplace_back();
    Square &q = s->scene.squares.back();
    // the four vertices as they stand (after any rotate / scale / translate the caller applied), and the tangent frame as
    // setQuad left it -- later transforms do NOT update m_right_vector / m_up_vector (Square.h:35-45, Scene.h:284)
    q.vertices.resize(4);
    const float *v[4] = {v0, v1, v2, v3};
    for (int k = 0; k < 4; ++k) q.vertices[k].position = Vec3(v[k][0], v[k][1], v[k][2]);
    q.triangles.resize(2);
    q.triangles[0][0] = 0; q.triangles[0][1] = 1; q.triangles[0][2] = 2;
    q.triangles[1][0] = 0; q.triangles[1][1] = 2; q.triangles[1][2] = 3;
    q.m_bottom_left = q.vertices[0].position;
    q.m_right_vector = Vec3(tangent[0], tangent[1], tangent[2]);
    q.m_up_vector = Vec3(bitangent[0], bitangent[1], bitangent[2]);
    q.build_arrays();
    q.material = to_material(m);
    return HRT_OK;
}

int hrt_host_scene_add_mesh(hrt_host_scene *s, const float *positions, uint32_t nv, const uint32_t *indices,
                            uint32_t nt, const float *face_colors, const hrt_material *m) {
    return hrt_host_scene_add_mesh_ex(s, positions, nv, indices, nt, nullptr, face_colors, face_colors ? HRT_COLOR_FACE : HRT_COLOR_NONE, m);
}

int hrt_host_scene_add_mesh_ex(hrt_host_scene *s, const float *positions, uint32_t nv, const uint32_t *indices, uint32_t nt,
                               const float *vert_colors, const float *face_colors, int32_t color_type, const hrt_material *m) {
    if (!s || (!positions && nv) || (!indices && nt)) return fail(HRT_ERR_INVALID, "add_mesh: NULL argument");
    if (color_type < HRT_COLOR_VERTEX || color_type > HRT_COLOR_NONE) return fail(HRT_ERR_INVALID, "add_mesh: bad color_type");
    for (uint32_t i = 0; i < 3 * nt; ++i)
        if (indices[i] >= nv) return fail(HRT_ERR_INVALID, "add_mesh: vertex index out of range");
    s->scene.meshes.emplace_back();
    Mesh &mesh = s->scene.meshes.back();
    mesh.vertices.resize(nv);
    for (uint32_t v = 0; v < nv; ++v) mesh.vertices[v].position = Vec3(positions[3 * v], positions[3 * v + 1], positions[3 * v + 2]);
    mesh.triangles.resize(nt);
    for (uint32_t t = 0; t < nt; ++t) {
        for (int k = 0; k < 3; ++k) mesh.triangles[t][k] = indices[3 * t + k];
        mesh.triangles[t][3] = t;
    }
    // Mesh::colorType decides what Scene::rayTraceRecursive reads (Scene.h:288-299); an array that is not given falls back
    // to the material's albedo, as a mesh loaded from a plain OFF file does
    mesh.colorType = ColorType_None;
    if (color_type == HRT_COLOR_FACE && face_colors) {
        mesh.colorType = ColorType_Face;
        mesh.faceColors.resize(nt);
        for (uint32_t t = 0; t < nt; ++t) mesh.faceColors[t] = Vec3(face_colors[3 * t], face_colors[3 * t + 1], face_colors[3 * t + 2]);
    } else if (color_type == HRT_COLOR_VERTEX && vert_colors) {
        mesh.colorType = ColorType_Vertex;
        mesh.vertColors.resize(nv);
        for (uint32_t v = 0; v < nv; ++v) mesh.vertColors[v] = Vec3(vert_colors[3 * v], vert_colors[3 * v + 1], vert_colors[3 * v + 2]);
    }
    mesh.build_arrays();
    mesh.material = to_material(m);
    return HRT_OK;
}

int hrt_host_scene_add_mesh_off(hrt_host_scene *s, const char *rel, const hrt_material *m) {
    if (!s || !rel) return fail(HRT_ERR_INVALID, "add_mesh_off: NULL argument");
    s->scene.meshes.emplace_back();
    Mesh &mesh = s->scene.meshes.back();
    if (!s->scene.load_mesh(mesh, rel)) {
        s->scene.meshes.pop_back();
        return fail(HRT_ERR_IO, s->scene.error);
    }
    mesh.build_arrays();
    mesh.material = to_material(m);
    return HRT_OK;
}

int hrt_host_scene_add_light(hrt_host_scene *s, const float pos[3], float radius, const float color[3]) {
    if (!s || !pos || !color) return fail(HRT_ERR_INVALID, "add_light: NULL argument");
    s->scene.lights.emplace_back();
    Light &l = s->scene.lights.back();
    l.pos = Vec3(pos[0], pos[1], pos[2]);
    l.radius = radius;
    l.material = Vec3(color[0], color[1], color[2]);
    return HRT_OK;
}

int hrt_host_scene_set_sky(hrt_host_scene *s, int32_t dark_sky) {
    if (!s) return fail(HRT_ERR_INVALID, "NULL scene");
    s->scene.dark_sky = dark_sky != 0;
    return HRT_OK;
}

int hrt_host_scene_set_kd_params(hrt_host_scene *s, uint32_t leaf_max, uint32_t max_depth) {
    if (!s) return fail(HRT_ERR_INVALID, "NULL scene");
    if (leaf_max) s->scene.kd_params.leaf_max = leaf_max;
    s->scene.kd_params.max_depth = max_depth;
    return HRT_OK;
}

int hrt_host_scene_set_kd_builder(hrt_host_scene *s, hrt_kd_builder_fn fn, void *user) {
    if (!s) return fail(HRT_ERR_INVALID, "NULL scene");
    s->scene.kd_params.builder = fn;
    s->scene.kd_params.builder_user = user;
    return HRT_OK;
}

int hrt_host_scene_flatten(hrt_host_scene *s, const hrt_scene_desc **out) {
    if (!s || !out) return fail(HRT_ERR_INVALID, "flatten: NULL argument");
    try {
        s->flat = s->scene.flatten();
        *out = &s->flat->desc;
        return HRT_OK;
    } catch (const std::exception &e) {
        return fail(HRT_ERR_STATE, e.what());
    }
}

int hrt_host_scene_irregular_stats(hrt_host_scene *s, uint32_t mi, uint32_t out[8]) {
    if (!s || !out || !s->flat || mi >= s->flat->ref_analysis.size()) return fail(HRT_ERR_INVALID, "irregular_stats: bad argument");
    const hrt_host::RefTreeAnalysis &a = s->flat->ref_analysis[mi];
    uint32_t n = 0;
    for (uint8_t b : a.irregular) n += b ? 1u : 0u;
    out[0] = n; out[1] = a.n_slivers; out[2] = a.n_dropped; out[3] = a.n_pairs; out[4] = a.ref_leaves; out[5] = a.ref_depth; out[6] = a.n_dead; out[7] = (uint32_t)a.exceptions.size();
    return HRT_OK;
}

int hrt_host_scene_kd_stats(hrt_host_scene *s, uint32_t mi, uint32_t out[6]) {
    if (!s || !out || !s->flat || mi >= s->flat->trees.size()) return fail(HRT_ERR_INVALID, "kd_stats: bad argument");
    const FlatKDTree &t = s->flat->trees[mi];
    out[0] = t.n_inner; out[1] = t.n_leaves; out[2] = t.n_empty_leaves; out[3] = t.depth;
    out[4] = (uint32_t)t.leaf_tris.size(); out[5] = (uint32_t)t.units.size();
    return HRT_OK;
}

void hrt_host_default_camera(float aspect_ratio, hrt_camera *out) {
    if (out) *out = default_camera(aspect_ratio);
}

}  // extern "C"

// PPM loader probe: loads `path` with the host loader and returns its size and the FNV-1a hash of
// its RGB bytes (parity check against the reference's imageLoader.cpp, tests/golden/ref_ppm.json).
extern "C" int hrt_host_ppm_info(const char *path, int32_t *w, int32_t *h, uint64_t *fnv1a) {
    if (!path || !w || !h || !fnv1a) return fail(HRT_ERR_INVALID, "hrt_host_ppm_info: NULL argument");
    ppmLoader::ImageRGB img;
    if (!ppmLoader::load_ppm(img, path)) return fail(HRT_ERR_IO, std::string("cannot read ") + path);
    *w = img.w;
    *h = img.h;
    uint64_t sum = 1469598103934665603ull;
    for (const ppmLoader::RGB &px : img.data) {
        sum = (sum ^ px.r) * 1099511628211ull;
        sum = (sum ^ px.g) * 1099511628211ull;
        sum = (sum ^ px.b) * 1099511628211ull;
    }
    *fnv1a = sum;
    return HRT_OK;
}
