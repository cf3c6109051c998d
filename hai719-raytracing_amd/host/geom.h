// Host-side scene objects of the ray-trace path (C++, no GPU code).
//
// Mirrors the *interface* of the reference's scene layer so that scene set-up
// code written against the reference reads the same here:
//   Vec3 / Mat3        /root/reference/src/Vec3.h:12-114, 125-293
//   Material           /root/reference/src/Material.h:23-61
//   Mesh               /root/reference/src/Mesh.h:70-280, Mesh.cpp:9-117
//   Sphere             /root/reference/src/Sphere.h:42-47
//   Square             /root/reference/src/Square.h:20-63
//   Light              /root/reference/src/Scene.h:28-41
// Only what feeds the trace path is carried: no GL arrays, no draw().
// All arithmetic is fp32 in the reference's evaluation order (this file is
// compiled with -ffp-contract=off) so the flattened arrays are the ones the
// reference would hold.
#pragma once

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

namespace hrt_host {

struct Vec3 {
    float v[3];
    Vec3() : v{0.f, 0.f, 0.f} {}
    Vec3(float x, float y, float z) : v{x, y, z} {}
    Vec3(float f) : v{f, f, f} {}  // implicit on purpose (Vec3.h:21)
    float &operator[](unsigned i) { return v[i]; }
    float operator[](unsigned i) const { return v[i]; }
    float squareLength() const { return v[0] * v[0] + v[1] * v[1] + v[2] * v[2]; }
    float length() const { return std::sqrt(squareLength()); }
    void normalize() {
        float L = length();
        v[0] /= L; v[1] /= L; v[2] /= L;
    }
    static float dot(const Vec3 &a, const Vec3 &b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
    static Vec3 cross(const Vec3 &a, const Vec3 &b) {
        return Vec3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
    }
    void operator+=(const Vec3 &o) { v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; }
    void operator-=(const Vec3 &o) { v[0] -= o[0]; v[1] -= o[1]; v[2] -= o[2]; }
    void operator*=(float s) { v[0] *= s; v[1] *= s; v[2] *= s; }
    void operator/=(float s) { v[0] /= s; v[1] /= s; v[2] /= s; }
};
inline Vec3 operator+(const Vec3 &a, const Vec3 &b) { return Vec3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline Vec3 operator-(const Vec3 &a, const Vec3 &b) { return Vec3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline Vec3 operator*(float a, const Vec3 &b) { return Vec3(a * b[0], a * b[1], a * b[2]); }
inline Vec3 operator*(const Vec3 &b, float a) { return Vec3(a * b[0], a * b[1], a * b[2]); }
inline Vec3 operator/(const Vec3 &a, float b) { return Vec3(a[0] / b, a[1] / b, a[2] / b); }

// Row-major 3x3 (Vec3.h:125-293); only M*p is needed by the transforms.
struct Mat3 {
    float m[9];
    Mat3(float a, float b, float c, float d, float e, float f, float g, float h, float i)
        : m{a, b, c, d, e, f, g, h, i} {}
    Vec3 operator*(const Vec3 &p) const {
        return Vec3(m[0] * p[0] + m[1] * p[1] + m[2] * p[2],
                    m[3] * p[0] + m[4] * p[1] + m[5] * p[2],
                    m[6] * p[0] + m[7] * p[1] + m[8] * p[2]);
    }
};

enum MaterialType { Material_Diffuse_Blinn_Phong, Material_Glass, Material_Mirror };
enum TextureType { Texture_None, Texture_Checkerboard, Texture_Image };
enum ColorType { ColorType_Vertex, ColorType_Face, ColorType_None };

namespace ppmLoader {
struct RGB { unsigned char r, g, b; };
struct ImageRGB {
    int w = 0, h = 0;  // the reference leaves these unset on a failed load (N10): defined as empty here
    std::vector<RGB> data;
};
// imageLoader.cpp:21-103.  Returns false (and prints, like the reference) when the file cannot be read.
bool load_ppm(ImageRGB &img, const std::string &name);
}  // namespace ppmLoader

// Material.h:23-61.  Members the integrator never reads are kept as plain data
// so set-up code can assign them; image/normals are indices into the owning
// Scene's texture / normal-map tables (the reference keeps raw pointers into
// those vectors, Scene.h:504-505).
struct Material {
    Vec3 ambient_material;
    Vec3 diffuse_material;
    Vec3 specular_material;
    double shininess = 0.;
    Vec3 motion_blur_translation = Vec3(0.f);
    float index_medium = 1.0f;
    float transparency = 0.f;
    MaterialType type = Material_Diffuse_Blinn_Phong;
    TextureType texture_type = Texture_None;
    Vec3 checkerboard_color1;
    Vec3 checkerboard_color2;
    float texture_scale_x = 1.f;
    float texture_scale_y = 1.f;
    bool emissive = false;  // indeterminate in the reference (Material.cpp:5-11); defined false
    Vec3 light_color;
    float light_intensity = 0.f;
    int image = -1;    // index into Scene::textures
    int normals = -1;  // index into Scene::normals
    bool has_normal_map = false;
    void set_texture(int texture_index) { image = texture_index; }
    void set_normals(int normal_map_index) { normals = normal_map_index; has_normal_map = true; }
};

struct AABB {
    Vec3 p0 = Vec3(FLT_MAX), p1 = Vec3(-FLT_MAX);
};

struct MeshVertex {
    Vec3 position;
};
struct MeshTriangle {
    unsigned int v[4] = {0, 0, 0, 0};  // 3 vertex ids + the triangle's own index (Mesh.h:60)
    unsigned int &operator[](unsigned i) { return v[i]; }
    unsigned int operator[](unsigned i) const { return v[i]; }
};

struct FlatKDTree;  // kdtree.h

class Mesh {
public:
    std::vector<MeshVertex> vertices;
    std::vector<MeshTriangle> triangles;
    std::vector<Vec3> vertColors;
    std::vector<Vec3> faceColors;
    ColorType colorType = ColorType_None;
    AABB aabb;
    Material material;

    virtual ~Mesh() {}
    // Mesh.cpp:9-69; the reference exit()s on a missing file, this returns false and the C ABI reports HRT_ERR_IO.
    bool loadOFF(const std::string &filename);
    void centerAndScaleToUnit();              // Mesh.cpp:87-105
    void computeAABB();                       // Mesh.h:143-157 (FLT_MIN seed kept)
    void build_arrays() { computeAABB(); }    // Mesh.h:134-141 minus the GL arrays
    void translate(const Vec3 &t);            // Mesh.h:173-177
    void apply_transformation_matrix(const Mat3 &m);
    void scale(const Vec3 &s);                // Mesh.h:186-191
    void rotate_x(float angle_deg);           // Mesh.h:198-204
    void rotate_y(float angle_deg);
    void rotate_z(float angle_deg);
    void rotate(const Vec3 &a) { rotate_x(a[0]); rotate_y(a[1]); rotate_z(a[2]); }
};

class Sphere : public Mesh {
public:
    Vec3 m_center;
    float m_radius = 0.f;
    Sphere() {}
    Sphere(Vec3 c, float r) : m_center(c), m_radius(r) {}
};

class Square : public Mesh {
public:
    Vec3 m_normal, m_bottom_left, m_right_vector, m_up_vector;
    Square() {}
    // Square.h:31-63 (uv corners are GL-only and dropped)
    void setQuad(const Vec3 &bottomLeft, const Vec3 &rightVector, const Vec3 &upVector,
                 float width = 1.f, float height = 1.f);
};

struct Light {
    Vec3 material;
    Vec3 pos;
    float radius = 0.f;
    float powerCorrection = 1.f;
    bool isInCamSpace = false;
};

}  // namespace hrt_host
