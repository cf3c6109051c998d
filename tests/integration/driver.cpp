// TEST INFRASTRUCTURE: fills a mock reference Scene (ref_mock.h) the way a setup_* of the reference would -- textured,
// normal-mapped square that is ROTATED after setQuad, a moving sphere, a vertex-coloured mesh, a light -- and runs the
// binding code of INTEGRATION.md path A on it.
//   driver flatten        toHrt() + hrt_host_scene_flatten, prints what arrived in the hrt_scene_desc (no GPU needed)
//   driver render out.bin ray_trace_from_camera() as INTEGRATION.md writes it (needs a GPU); dumps the image
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <iostream>
#include <string>

#include "ref_mock.h"

struct MockBuilder {
    static void fill(Scene &s) {
        s.textures.resize(1);
        s.textures[0].w = 4; s.textures[0].h = 2; s.textures[0].data.resize(8);
        for (int i = 0; i < 8; ++i) s.textures[0].data[i] = ppmLoader::RGB{(unsigned char)(30 * i), (unsigned char)(255 - 20 * i), (unsigned char)(7 * i)};
        s.normals.resize(1);
        s.normals[0].w = 2; s.normals[0].h = 2; s.normals[0].data.resize(4);
        for (int i = 0; i < 4; ++i) s.normals[0].data[i] = ppmLoader::RGB{(unsigned char)(120 + 5 * i), (unsigned char)(130 - 3 * i), 250};
        // a square as setQuad((-1,-1,-2), (1,0,0), (0,1,0), 2, 3) leaves it, then rotated by 90 degrees about y: the vertices
        // move, m_right_vector / m_up_vector do not (Square.h:35-45, Mesh.h:198-204)
        s.squares.resize(1);
        Square &q = s.squares[0];
        q.m_right_vector = Vec3(2, 0, 0); q.m_up_vector = Vec3(0, 3, 0); q.m_bottom_left = Vec3(-1, -1, -2);
        const float P[4][3] = {{-1, -1, -2}, {1, -1, -2}, {1, 2, -2}, {-1, 2, -2}};
        q.vertices.resize(4);
        for (int k = 0; k < 4; ++k) q.vertices[k].position = Vec3(P[k][2], P[k][1], -P[k][0]);  // (x,y,z) -> (z,y,-x)
        q.material.diffuse_material = Vec3(0.9f, 0.8f, 0.7f);
        q.material.texture_type = Texture_Image; q.material.image = &s.textures[0];
        q.material.has_normal_map = true; q.material.normals = &s.normals[0];
        q.material.texture_scale_x = 2.f; q.material.texture_scale_y = 0.5f;
        s.spheres.resize(1);
        s.spheres[0].m_center = Vec3(0.5f, 0.25f, -1.f); s.spheres[0].m_radius = 0.75f;
        s.spheres[0].material.type = Material_Glass; s.spheres[0].material.index_medium = 1.4f; s.spheres[0].material.transparency = 0.9f;
        s.spheres[0].material.motion_blur_translation = Vec3(0.f, 0.5f, 0.f);
        s.meshes.resize(1);
        Mesh &m = s.meshes[0];
        const float V[4][3] = {{0, 0, 0}, {1, 0, 0}, {0.5f, 1, 0.2f}, {0.5f, 0.3f, 1}};
        const unsigned T[4][3] = {{0, 2, 1}, {0, 1, 3}, {1, 2, 3}, {0, 3, 2}};
        m.vertices.resize(4); m.triangles.resize(4); m.vertColors.resize(4);
        for (int k = 0; k < 4; ++k) {
            m.vertices[k].position = Vec3(V[k][0] - 2.f, V[k][1] - 1.f, V[k][2] - 1.5f);
            m.vertColors[k] = Vec3(0.25f * k, 1.f - 0.25f * k, 0.5f);
            for (int j = 0; j < 3; ++j) m.triangles[k][j] = T[k][j];
            m.triangles[k][3] = k;
        }
        m.colorType = ColorType_Vertex;
        m.material.diffuse_material = Vec3(0.2f, 0.9f, 0.3f);
        s.lights.resize(1);
        s.lights[0].pos = Vec3(0.f, 3.f, 2.f); s.lights[0].radius = 1.5f; s.lights[0].material = Vec3(1, 0.9f, 0.8f);
        s.dark_sky = false;
    }
};

// ---- what main.cpp provides around ray_trace_from_camera() (main.cpp:51-66, 200-214): mocked ------------------------
typedef double GLdouble;
enum { GLUT_WINDOW_WIDTH, GLUT_WINDOW_HEIGHT };
static int g_w = 96, g_h = 54;
static int glutGet(int what) { return what == GLUT_WINDOW_WIDTH ? g_w : g_h; }
struct Camera { void apply() {} } camera;
struct MatrixUtilities {
    GLdouble modelviewInverse[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 6.1, 1};  // the default pose (main.cpp:418, Camera.cpp:37)
    void updated() {}
    void updateMatrices() {}
} matrixUtilities;
static std::vector<Scene> scenes(1);
static unsigned int selected_scene = 0;
static unsigned int nsamples = 4;
static std::vector<Vec3> g_last_image;

#include "ray_trace_from_camera.inc"   // <- the new body of main.cpp:200-263, extracted verbatim from INTEGRATION.md

int main(int argc, char **argv) {
    MockBuilder::fill(scenes[0]);
    const std::string mode = argc > 1 ? argv[1] : "flatten";
    if (mode == "flatten") {
        hrt_host_scene *hs = scenes[0].toHrt();
        if (!hs) { std::printf("toHrt failed: %s\n", hrt_host_last_error()); return 1; }
        const hrt_scene_desc *d = nullptr;
        if (hrt_host_scene_flatten(hs, &d) != HRT_OK) { std::printf("flatten failed: %s\n", hrt_host_last_error()); return 1; }
        std::printf("counts %u %u %u %u %u %u %d\n", d->n_materials, d->n_spheres, d->n_quads, d->n_meshes, d->n_lights, d->n_images, d->dark_sky);
        const hrt_quad &q = d->quads[0];
        std::printf("quad v0 %g %g %g v1 %g %g %g v3 %g %g %g T %g %g %g B %g %g %g\n", q.v0[0], q.v0[1], q.v0[2], q.v1[0], q.v1[1], q.v1[2],
                    q.v3[0], q.v3[1], q.v3[2], q.tangent[0], q.tangent[1], q.tangent[2], q.bitangent[0], q.bitangent[1], q.bitangent[2]);
        const hrt_material &qm = d->materials[q.material];
        std::printf("quadmat tex %d image %d nmap %d scale %g %g albedo %g\n", qm.texture_type, qm.image, qm.normal_map, qm.tex_scale_x, qm.tex_scale_y, qm.albedo[0]);
        const hrt_material &sm = d->materials[d->spheres[0].material];
        std::printf("sphere %g %g type %d eta %g motion %g\n", d->spheres[0].center[0], d->spheres[0].radius, sm.type, sm.index_medium, sm.motion[1]);
        const hrt_mesh &m = d->meshes[0];
        std::printf("mesh %u %u color_type %d vc %g leaf %u box %g %g\n", m.n_vertices, m.n_triangles, m.color_type, m.vert_colors ? m.vert_colors[3] : -1.f,
                    m.n_leaf_tris, m.aabb_min[0], m.aabb_max[1]);
        std::printf("light %g %g %g\n", d->lights[0].pos[1], d->lights[0].radius, d->lights[0].color[2]);
        std::printf("image0 %d %d %u image1 %d %d\n", d->images[0].w, d->images[0].h, (unsigned)d->images[0].rgb[3], d->images[1].w, d->images[1].h);
        hrt_host_scene_free(hs);
        return 0;
    }
    ray_trace_from_camera();
    if (g_last_image.empty()) return 2;
    if (argc > 2) {
        FILE *f = std::fopen(argv[2], "wb");
        std::fwrite(&g_last_image[0], sizeof(Vec3), g_last_image.size(), f);
        std::fclose(f);
    }
    return 0;
}
