// cfg 5 -- Scene::setup_backrooms_pool, /root/reference/src/Scene.h:1329-1882: 28 squares (12 emissive
// panels, a glass water surface with a normal map, tiled walls), 2 spheres (the flamingo's eye), 3 meshes.
// Written as a table of per-square transforms instead of the reference's 28 copy-pasted blocks; every
// square starts from setQuad((-1,-0.2,0),(1,0,0),(0,1,0),2,2) and translate(0,0,-2) like there.
//
// Substitutions forced by blobs that are missing from the reference checkout:
//   mesh/flamingo_float_colored.off -> mesh/flamingo_float.off (same geometry family, no vertex
//   colours: the material albedo shows); img/textures/sky.ppm -> no skybox (dark_sky stays true).
#include "scene.h"

namespace hrt_host {
namespace {

enum Op { SX, RX, RY, RZ, TR, END };  // scale / rotate_x / rotate_y / rotate_z / translate
struct Step { Op op; float a, b, c; };
enum Look { PANEL, WATER, TILES, PLAIN };
struct PoolSquare {
    Step steps[4];
    Look look;
    float sx, sy;  // texture scale (TILES)
};

const PoolSquare kSquares[] = {
    // four ceiling light panels
    {{{SX, .5f, .5f, 1}, {RX, 90}, {TR, 0, 2.95f, -12.75f}, {END}}, PANEL, 1, 1},
    {{{SX, .5f, .5f, 1}, {RX, 90}, {TR, 0, 2.95f, -8.75f}, {END}}, PANEL, 1, 1},
    {{{SX, .5f, .5f, 1}, {RX, 90}, {TR, 0, 2.95f, -4.75f}, {END}}, PANEL, 1, 1},
    {{{SX, .5f, .5f, 1}, {RX, 90}, {TR, 0, 2.95f, -0.75f}, {END}}, PANEL, 1, 1},
    // water surface, pool floor, ceiling
    {{{SX, 4, 8, 1}, {RX, -90}, {TR, 0, -0.75f, 0}, {END}}, WATER, 1, 1},
    {{{SX, 4, 8, 1}, {RX, -90}, {TR, 0, -1, 0}, {END}}, TILES, 1, 2},
    {{{SX, 4, 8, 1}, {RX, 90}, {TR, 0, 3, -12.75f}, {END}}, PLAIN, 1, 1},
    // basin walls and upper walls
    {{{SX, .5f, 8, 1}, {RX, -90}, {RZ, 90}, {TR, 2, -2.5f, 0}}, TILES, .25f, 2},
    {{{SX, 2, 8, 1}, {RX, -90}, {RZ, 90}, {TR, 2, 4, 0}}, TILES, 1, 2},
    {{{SX, 2, 8, 1}, {RX, -90}, {RZ, -90}, {TR, -2, 4, 0}}, TILES, 1, 2},
    {{{SX, .5f, 8, 1}, {RX, -90}, {RZ, -90}, {TR, -2, -2.5f, 0}}, TILES, .25f, 2},
    // side walkways: floor / ceiling strips (the right ceiling strip appears twice in the reference)
    {{{SX, 1, 8, 1}, {RX, -90}, {TR, 5, 0, 0}, {END}}, TILES, 1, 2},
    {{{SX, 1, 8, 1}, {RX, 90}, {TR, 5, 0, -12.75f}, {END}}, TILES, 1, 2},
    {{{SX, 1, 8, 1}, {RX, -90}, {TR, -5, 0, 0}, {END}}, TILES, 1, 2},
    {{{SX, 1, 8, 1}, {RX, 90}, {TR, 5, 0, -12.75f}, {END}}, TILES, 1, 2},
    {{{SX, 1, 8, 1}, {RX, 90}, {TR, -5, 0, -12.75f}, {END}}, TILES, 1, 2},
    // right middle wall and its four light panels
    {{{SX, 8, 2, 1}, {RY, -90}, {TR, 4, -1.6f, -6.4f}, {END}}, TILES, 2, 1},
    {{{SX, .5f, .5f, 1}, {RY, -90}, {TR, 3.95f, .9f, -0.75f}, {END}}, PANEL, 1, 1},
    {{{SX, .5f, .5f, 1}, {RY, -90}, {TR, 3.95f, .9f, -4.75f}, {END}}, PANEL, 1, 1},
    {{{SX, .5f, .5f, 1}, {RY, -90}, {TR, 3.95f, .9f, -8.75f}, {END}}, PANEL, 1, 1},
    {{{SX, .5f, .5f, 1}, {RY, -90}, {TR, 3.95f, .9f, -12.75f}, {END}}, PANEL, 1, 1},
    // left middle wall and its four light panels
    {{{SX, 8, 2, 1}, {RY, 90}, {TR, -4, -1.6f, -6.4f}, {END}}, TILES, 2, 1},
    {{{SX, .5f, .5f, 1}, {RY, 90}, {TR, -3.95f, .8f, -0.75f}, {END}}, PANEL, 1, 1},
    {{{SX, .5f, .5f, 1}, {RY, 90}, {TR, -3.95f, .8f, -4.75f}, {END}}, PANEL, 1, 1},
    {{{SX, .5f, .5f, 1}, {RY, 90}, {TR, -3.95f, .8f, -8.75f}, {END}}, PANEL, 1, 1},
    {{{SX, .5f, .5f, 1}, {RY, 90}, {TR, -3.95f, .8f, -12.75f}, {END}}, PANEL, 1, 1},
    // front and back walls
    {{{SX, 8, 8, 1}, {RX, -180}, {TR, 0, 4, 0}, {END}}, TILES, 2, 2},
    {{{SX, 8, 8, 1}, {TR, 0, -3, -12}, {END}, {END}}, TILES, 2, 2},
};

void apply(Mesh &m, const Step &st) {
    switch (st.op) {
        case SX: m.scale(Vec3(st.a, st.b, st.c)); break;
        case RX: m.rotate_x(st.a); break;
        case RY: m.rotate_y(st.a); break;
        case RZ: m.rotate_z(st.a); break;
        case TR: m.translate(Vec3(st.a, st.b, st.c)); break;
        default: break;
    }
}

}  // namespace

void Scene::setup_backrooms_pool() {
    clear();
    skybox = ppmLoader::ImageRGB();  // img/textures/sky.ppm is not in the reference checkout
    dark_sky = true;
    const int tiles = load_texture("img/planeTextures/white_pool_tiles.ppm");
    const int tiles_n = load_normal_map("img/normalMaps/pool_tiles_normal.ppm");
    const int water_n = load_normal_map("img/normalMaps/water_normal.ppm");
    const float lights_intensity = 30.f;

    for (const PoolSquare &ps : kSquares) {
        squares.emplace_back();
        Square &s = squares.back();
        s.setQuad(Vec3(-1.f, -0.2f, 0.f), Vec3(1.f, 0.f, 0.f), Vec3(0.f, 1.f, 0.f), 2.f, 2.f);
        s.translate(Vec3(0.f, 0.f, -2.f));
        for (const Step &st : ps.steps) {
            if (st.op == END) break;
            apply(s, st);
        }
        s.build_arrays();
        Material &m = s.material;
        m.specular_material = Vec3(1.f, 1.f, 1.f);
        m.shininess = 16;
        switch (ps.look) {
            case PANEL:
                m.diffuse_material = Vec3(1.f);
                m.emissive = true;
                m.light_intensity = lights_intensity;
                m.light_color = Vec3(1.f);
                break;
            case WATER:
                m.diffuse_material = Vec3((float)(170. / 255.), (float)(213. / 255.), (float)(219. / 255.));
                m.type = Material_Glass;
                m.transparency = 0.99f;
                m.texture_type = Texture_None;
                m.set_normals(water_n);
                break;
            case TILES:
                m.diffuse_material = Vec3(0.1f, 0.5f, 0.1f);
                m.texture_type = Texture_Image;
                m.texture_scale_x = ps.sx;
                m.texture_scale_y = ps.sy;
                m.set_texture(tiles);
                m.set_normals(tiles_n);
                break;
            case PLAIN:
                m.diffuse_material = Vec3(0.8f);
                break;
        }
    }
    {   // flamingo (uncoloured substitute, see the header comment)
        meshes.emplace_back();
        Mesh &m = meshes.back();
        if (!load_mesh(m, "mesh/flamingo_float.off")) { meshes.pop_back(); return; }
        m.centerAndScaleToUnit();
        m.rotate_x(0.f);
        m.rotate_y(225.f);
        m.translate(Vec3(-0.5f, -1.35f, -2.f));
        m.scale(Vec3(1.8f));
        m.build_arrays();
        m.colorType = ColorType_None;
        m.material.diffuse_material = Vec3((float)(237. / 255.), (float)(149. / 255.), (float)(218. / 255.));
        m.material.specular_material = Vec3(1.f);
        m.material.shininess = 6.;
    }
    {   // the flamingo's eye: white ball + black pupil
        spheres.emplace_back(Vec3(0.05f, -1.4f, -3.1f), 0.05f);
        spheres.back().material.diffuse_material = Vec3(1.f);
        spheres.emplace_back(Vec3(0.05f, -1.4f, -3.05f), 0.01f);
        spheres.back().material.diffuse_material = Vec3(0.f);
    }
    {   // rubber duck (COFF vertex colours)
        meshes.emplace_back();
        Mesh &m = meshes.back();
        if (!load_mesh(m, "mesh/rubber_duck_colored.off")) { meshes.pop_back(); return; }
        m.centerAndScaleToUnit();
        m.rotate_y(-35.f);
        m.translate(Vec3(2.f, -1.65f, -2.f));
        m.scale(Vec3(1.3f));
        m.build_arrays();
        m.material.diffuse_material = Vec3(1.f, 1.f, 0.f);
        m.material.specular_material = Vec3(1.f);
        m.material.shininess = 6.;
    }
    {   // pool ladder (mirror)
        meshes.emplace_back();
        Mesh &m = meshes.back();
        if (!load_mesh(m, "mesh/pool_ladder.off")) { meshes.pop_back(); return; }
        m.centerAndScaleToUnit();
        m.rotate_y(90.f);
        m.translate(Vec3(-3.f, -1.445f, -3.f));
        m.scale(Vec3(1.3f));
        m.build_arrays();
        m.material.type = Material_Mirror;
        m.material.diffuse_material = Vec3(0.5f, 0.5f, 0.5f);
        m.material.specular_material = Vec3(1.f);
        m.material.shininess = 6.;
    }
}

}  // namespace hrt_host
