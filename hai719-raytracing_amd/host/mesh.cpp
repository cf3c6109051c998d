// Mesh / Square / PPM host code.  See geom.h for the reference lines mirrored.
#include "geom.h"

#include <fstream>
#include <iostream>
#include <sstream>

namespace hrt_host {

// ---------------------------------------------------------------------------
// OFF / COFF / per-face-colour reader (behaviour of Mesh.cpp:9-69):
//   header  : "OFF"|"COFF"  nV nT <ignored>
//   COFF    : x y z r g b <ignored>   colours /255
//   faces   : "<n> i j k [r g b]"; the FIRST face line decides whether the
//             file carries per-face colours (which then override COFF).
// ---------------------------------------------------------------------------
bool Mesh::loadOFF(const std::string &filename) {
    std::ifstream in(filename.c_str());
    if (!in) return false;
    std::string magic;
    unsigned nV = 0, nT = 0, ignored = 0;
    in >> magic >> nV >> nT >> ignored;
    if (!in) return false;
    vertices.assign(nV, MeshVertex());
    triangles.assign(nT, MeshTriangle());
    vertColors.clear();
    faceColors.clear();
    colorType = (magic == "COFF") ? ColorType_Vertex : ColorType_None;
    if (colorType == ColorType_Vertex) vertColors.resize(nV);
    for (unsigned i = 0; i < nV; ++i) {
        Vec3 &p = vertices[i].position;
        in >> p[0] >> p[1] >> p[2];
        if (colorType == ColorType_Vertex) {
            Vec3 &c = vertColors[i];
            in >> c[0] >> c[1] >> c[2] >> ignored;
            c /= 255.0;  // double literal -> float divide, as Mesh.cpp:30
        }
    }
    std::string line;
    std::getline(in, line);  // rest of the last vertex line
    for (unsigned t = 0; t < nT; ++t) {
        if (!std::getline(in, line)) break;
        std::istringstream ls(line);
        int arity = 0;
        ls >> arity;
        for (unsigned j = 0; j < 3; ++j) ls >> triangles[t].v[j];
        if (t == 0 && !(ls >> std::ws).eof()) {
            colorType = ColorType_Face;
            faceColors.resize(nT);
        }
        if (colorType == ColorType_Face) {
            Vec3 &c = faceColors[t];
            ls >> c[0] >> c[1] >> c[2];
            c /= 255.0f;
        }
        triangles[t].v[3] = t;
    }
    return true;
}

void Mesh::centerAndScaleToUnit() {
    if (vertices.empty()) return;
    Vec3 c(0.f, 0.f, 0.f);
    for (const MeshVertex &mv : vertices) c += mv.position;
    c /= (float)vertices.size();
    float maxD = (vertices[0].position - c).length();
    for (const MeshVertex &mv : vertices) {
        float d = (mv.position - c).length();
        if (d > maxD) maxD = d;
    }
    for (MeshVertex &mv : vertices) mv.position = (mv.position - c) / maxD;
}

void Mesh::computeAABB() {
    // Mesh.h:143-157: the max corner is seeded with FLT_MIN (smallest positive
    // float), not -FLT_MAX; kept because it is observable (N10).
    Vec3 lo(FLT_MAX), hi(FLT_MIN);
    for (const MeshVertex &mv : vertices)
        for (unsigned a = 0; a < 3; ++a) {
            if (mv.position[a] < lo[a]) lo[a] = mv.position[a];
            if (mv.position[a] > hi[a]) hi[a] = mv.position[a];
        }
    const float eps = (float)1e-5;  // Vec3(EPSILON): double literal narrowed to float
    lo -= Vec3(eps);
    hi += Vec3(eps);
    for (unsigned a = 0; a < 3; ++a) {  // AABB(a,b) orders per axis (AABB.h:29-39)
        if (lo[a] < hi[a]) { aabb.p0[a] = lo[a]; aabb.p1[a] = hi[a]; }
        else               { aabb.p1[a] = lo[a]; aabb.p0[a] = hi[a]; }
    }
}

void Mesh::translate(const Vec3 &t) {
    for (MeshVertex &mv : vertices) mv.position += t;
}

void Mesh::apply_transformation_matrix(const Mat3 &m) {
    for (MeshVertex &mv : vertices) mv.position = m * mv.position;
}

void Mesh::scale(const Vec3 &s) {
    apply_transformation_matrix(Mat3(s[0], 0.f, 0.f, 0.f, s[1], 0.f, 0.f, 0.f, s[2]));
}

// angle*M_PI/180. is evaluated in double and stored to a float; cos/sin are
// then taken of that float (the double overloads, result narrowed by Mat3).
static inline float deg2rad_f(float deg) { return (float)((double)deg * M_PI / 180.); }
static inline float cos_f(float a) { return (float)std::cos((double)a); }
static inline float sin_f(float a) { return (float)std::sin((double)a); }

void Mesh::rotate_x(float angle_deg) {
    float a = deg2rad_f(angle_deg);
    apply_transformation_matrix(Mat3(1.f, 0.f, 0.f, 0.f, cos_f(a), -sin_f(a), 0.f, sin_f(a), cos_f(a)));
}
void Mesh::rotate_y(float angle_deg) {
    float a = deg2rad_f(angle_deg);
    apply_transformation_matrix(Mat3(cos_f(a), 0.f, sin_f(a), 0.f, 1.f, 0.f, -sin_f(a), 0.f, cos_f(a)));
}
void Mesh::rotate_z(float angle_deg) {
    float a = deg2rad_f(angle_deg);
    apply_transformation_matrix(Mat3(cos_f(a), -sin_f(a), 0.f, sin_f(a), cos_f(a), 0.f, 0.f, 0.f, 1.f));
}

void Square::setQuad(const Vec3 &bottomLeft, const Vec3 &rightVector, const Vec3 &upVector,
                     float width, float height) {
    m_right_vector = rightVector;
    m_up_vector = upVector;
    m_normal = Vec3::cross(rightVector, upVector);
    m_bottom_left = bottomLeft;
    m_normal.normalize();
    m_right_vector.normalize();
    m_up_vector.normalize();
    m_right_vector = m_right_vector * width;
    m_up_vector = m_up_vector * height;
    vertices.assign(4, MeshVertex());
    vertices[0].position = bottomLeft;
    vertices[1].position = bottomLeft + m_right_vector;
    vertices[2].position = bottomLeft + m_right_vector + m_up_vector;
    vertices[3].position = bottomLeft + m_up_vector;
    triangles.assign(2, MeshTriangle());
    triangles[0][0] = 0; triangles[0][1] = 1; triangles[0][2] = 2;
    triangles[1][0] = 0; triangles[1][1] = 2; triangles[1][2] = 3;
}

namespace ppmLoader {

static void skip_blank_and_comment(std::ifstream &f) {
    int c;
    while ((c = f.peek()) == '\n' || c == '\r') f.get();
    if (c == '#') {
        std::string rest;
        std::getline(f, rest);
    }
}

bool load_ppm(ImageRGB &img, const std::string &name) {
    std::ifstream f(name.c_str(), std::ios::binary);
    if (f.fail()) {
        std::cout << "Could not open file: " << name << std::endl;
        return false;
    }
    skip_blank_and_comment(f);
    std::string magic;
    f >> magic;
    const bool ascii = (magic == "P3"), raw = (magic == "P6");
    int w = 0, h = 0, maxval = 0;
    skip_blank_and_comment(f); f >> w;
    skip_blank_and_comment(f); f >> h;
    skip_blank_and_comment(f); f >> maxval;
    if (!ascii && !raw) { std::cout << "Unsupported magic number" << std::endl; return false; }
    if (w < 1) { std::cout << "Unsupported width: " << w << std::endl; return false; }
    if (h < 1) { std::cout << "Unsupported height: " << h << std::endl; return false; }
    if (maxval < 1 || maxval > 255) { std::cout << "Unsupported number of bits: " << maxval << std::endl; return false; }
    img.w = w;
    img.h = h;
    img.data.assign((size_t)w * h, RGB{0, 0, 0});
    if (raw) {
        f.get();  // the single whitespace byte after maxval
        f.read(reinterpret_cast<char *>(img.data.data()), (std::streamsize)img.data.size() * 3);
    } else {
        for (RGB &px : img.data) {
            int r = 0, g = 0, b = 0;
            f >> r >> g >> b;
            px.r = (unsigned char)r; px.g = (unsigned char)g; px.b = (unsigned char)b;
        }
    }
    return true;
}

}  // namespace ppmLoader
}  // namespace hrt_host
