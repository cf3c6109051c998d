// TEST INFRASTRUCTURE: a mock of the PUBLIC MEMBER NAMES of the reference's scene classes -- only what the binding code
// of INTEGRATION.md (path A) touches -- so that the code a maintainer would paste into the reference can be compiled and
// run here, where the reference itself cannot be built (its headers include <GL/glut.h>).  Declarations only; nothing of
// the reference's behaviour is restated.  Members and the reference lines that declare them:
//   Vec3 operator[]                          src/Vec3.h:12-30
//   ppmLoader::RGB / ImageRGB                src/imageLoader.h:13-23
//   Material (data members)                  src/Material.h:23-47
//   MeshVertex / MeshTriangle / Mesh         src/Mesh.h:32-66, 109-125
//   Sphere m_center, m_radius                src/Sphere.h:38-40
//   Square m_right_vector, m_up_vector       src/Square.h:21-24
//   Light material, pos, radius              src/Scene.h:28-41
//   Scene's private containers               src/Scene.h:57-66
#pragma once
#include <cstdint>
#include <vector>

struct Vec3 {
    float mVals[3] = {0, 0, 0};
    Vec3() {}
    Vec3(float x, float y, float z) { mVals[0] = x; mVals[1] = y; mVals[2] = z; }
    float &operator[](unsigned c) { return mVals[c]; }
    float operator[](unsigned c) const { return mVals[c]; }
};
namespace ppmLoader {
struct RGB { unsigned char r, g, b; };
struct ImageRGB { int w, h; std::vector<RGB> data; };
}
enum MaterialType { Material_Diffuse_Blinn_Phong, Material_Glass, Material_Mirror };
enum TextureType { Texture_None, Texture_Checkerboard, Texture_Image };
struct Material {
    Vec3 ambient_material, diffuse_material, specular_material;
    double shininess = 0;
    Vec3 motion_blur_translation;
    float index_medium = 1.f, transparency = 0.f;
    MaterialType type = Material_Diffuse_Blinn_Phong;
    TextureType texture_type = Texture_None;
    Vec3 checkerboard_color1, checkerboard_color2;
    float texture_scale_x = 1.f, texture_scale_y = 1.f;
    bool emissive = false;
    Vec3 light_color;
    float light_intensity = 0.f;
    ppmLoader::ImageRGB *image = nullptr, *normals = nullptr;
    bool has_normal_map = false;
};
struct MeshVertex { Vec3 position, normal; float u = 0, v = 0; };
struct MeshTriangle {
    unsigned int v[4] = {0, 0, 0, 0};
    unsigned int &operator[](unsigned i) { return v[i]; }
    unsigned int operator[](unsigned i) const { return v[i]; }
};
enum ColorType { ColorType_Vertex, ColorType_Face, ColorType_None };
class Mesh {
public:
    std::vector<MeshVertex> vertices;
    std::vector<MeshTriangle> triangles;
    std::vector<Vec3> vertColors, faceColors;
    ColorType colorType = ColorType_None;
    Material material;
};
class Sphere : public Mesh { public: Vec3 m_center; float m_radius = 0.f; };
class Square : public Mesh { public: Vec3 m_normal, m_bottom_left, m_right_vector, m_up_vector; };
struct Light { Vec3 material; Vec3 pos; float radius = 0.f; };

#include "hrt.h"
#include "hrt_host.h"

class Scene {
    std::vector<Mesh> meshes;
    std::vector<Sphere> spheres;
    std::vector<Square> squares;
    std::vector<Light> lights;
    std::vector<ppmLoader::ImageRGB> textures;
    std::vector<ppmLoader::ImageRGB> normals;
    ppmLoader::ImageRGB skybox;
    bool dark_sky = true;

public:
#include "scene_to_hrt.inc"   // <- the method INTEGRATION.md tells the maintainer to add, extracted verbatim by the test
    friend struct MockBuilder;  // mock only: lets the test driver fill the private containers
};
