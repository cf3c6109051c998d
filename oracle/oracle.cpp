// ============================================================================
// oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A from-scratch CPU restatement of the reference's per-pixel x per-sample
// ray-trace loop (Kuuro-neko/HAI719-Raytracing), used only as the checker:
// by tests/, by __graft_entry__.smoke() and by bench.py's `cpu_baseline`
// leg.  Nothing under hai719-raytracing_amd/ may include, link or call it.
//
// It follows the reference function by function (file:line cited at each
// step, paths relative to /root/reference) including its float/double
// promotion rules and its quirks (SURVEY.md 8 parity notes N1-N13).
//
// PIN STATUS (see DESIGN.md "Oracle"):
//   pinned against the reference's own code compiled from /root/reference
//   (oracle/ref_parts.cpp -> oracle/_ref/): Triangle.h, AABB.h, Functions.cpp,
//   Vec3.h, Ray.h/Line.h, imageLoader.cpp and matrixUtilities.h (gluInvertMatrix +
//   screen_space_to_world_space_ray: camera rays bit for bit on 16 poses x 1024
//   (u, v))  (rows a2, a3, a11, a12, a16, a17, a18).
//   PARITY UNPINNED for Sphere.h, Square.h, Material.cpp, KDTree.cpp, Mesh.cpp,
//   Scene.h and main.cpp (rows a1, a4-a10, a13-a15): those translation
//   units include <GL/glut.h>, which this image lacks, the reference ships no
//   tests or golden vectors, and stand-in headers are not allowed; they are
//   restated from the source text only.
//
// RNG: the reference draws from a time-seeded, unsynchronised global mt19937
// (Functions.cpp:4-8) plus a random_device-seeded thread_local one
// (main.cpp:181), so no run of it is reproducible.  The renderer here uses a
// counter-based per-path stream keyed (seed, pixel, sample) and consumes it
// in the reference's call order (DESIGN.md "RNG stream").  The mt19937 form of
// random_float/random_unit_vector is kept for the known-answer test against
// oracle/_ref.
// ============================================================================
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "../include/hrt.h"

namespace {

// ---------------------------------------------------------------- Vec3.h:12-114
struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 v3(float a, float b, float c) { return V3{a, b, c}; }
inline V3 v3(const float *p) { return V3{p[0], p[1], p[2]}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
inline V3 operator*(V3 a, float s) { return v3(s * a.x, s * a.y, s * a.z); }
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }           // Vec3.h:48
inline V3 cross(V3 a, V3 b) {                                                         // Vec3.h:51
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline V3 comp_product(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }  // Vec3.h:81
inline float sq_length(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
// length() = sqrt(squareLength()) through the double overload, narrowed to
// float (Vec3.h:39); identical to a correctly rounded float sqrt.
inline float length(V3 a) { return (float)std::sqrt((double)sq_length(a)); }
inline V3 normalized(V3 a) {  // Vec3.h:46: three divides by the length, no guard
    float L = length(a);
    return v3(a.x / L, a.y / L, a.z / L);
}

const double EPS_D = HRT_EPSILON;         // Constants.h:18, a double literal
const float EPS_F = (float)HRT_EPSILON;   // where it is passed as a float

// Ray.h:4-9 / Line.h:13-16 -- the constructor normalises the direction.
struct Ray {
    V3 o, d;
    float time;
};
inline Ray make_ray(V3 o, V3 d, float time) { return Ray{o, normalized(d), time}; }

// ------------------------------------------------------------------- RNG
inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

struct Counters {  // per-sample work counters
    uint64_t closest_queries = 0, shadow_queries = 0, sphere_tests = 0, quad_tests = 0;
    uint64_t node_visits = 0, tri_tests = 0, shaded_hits = 0, texel_lookups = 0, rng_draws = 0;
    uint64_t samples = 0;
    void add(const Counters &o) {
        closest_queries += o.closest_queries; shadow_queries += o.shadow_queries;
        sphere_tests += o.sphere_tests; quad_tests += o.quad_tests; node_visits += o.node_visits;
        tri_tests += o.tri_tests; shaded_hits += o.shaded_hits; texel_lookups += o.texel_lookups;
        rng_draws += o.rng_draws; samples += o.samples;
    }
};

// Timing-only mode of oracle_render (flag ORACLE_FLAG_SHARED_RNG): every draw comes from ONE process-wide mt19937, as the
// reference's random_float() does (a static generator shared by all row threads, Functions.cpp:4-8).  The reference
// does not synchronise it (a data race); here a mutex keeps it defined.  Pixels are then not reproducible: this exists to
// time the reference's contention pattern beside the thread-local form, never for parity.
static bool g_shared_rng = false;
static std::mutex g_shared_rng_mutex;
static std::mt19937 g_shared_rng_gen(12345u);
static float shared_random_float() {
    std::lock_guard<std::mutex> lock(g_shared_rng_mutex);
    return std::uniform_real_distribution<float>(0.f, 1.f)(g_shared_rng_gen);
}

// Counter-based per-path stream: draw i of path (seed, pixel, sample).
struct PathRng {
    uint32_t k0, k1, i;
    Counters *cnt;
    PathRng(uint64_t seed, uint32_t pixel, uint32_t sample, Counters *c) : i(0), cnt(c) {
        k0 = mix32((uint32_t)seed ^ (pixel * 0x9E3779B1u + 0x7F4A7C15u));
        k1 = mix32((uint32_t)(seed >> 32) + sample * 0x85EBCA77u + 0xC2B2AE3Du);
    }
    float next() {  // stands in for random_float(), Functions.cpp:4-8
        if (g_shared_rng) return shared_random_float();
        uint32_t x = k0 + (i++) * 0x9E3779B9u;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x ^= k1; x *= 0x846ca68bu; x ^= x >> 16;
        if (cnt) cnt->rng_draws++;
        return (float)(x >> 8) * (1.0f / 16777216.0f);
    }
    float range(float lo, float hi) { return lo + (hi - lo) * next(); }  // Functions.cpp:10-12
    V3 unit_vector() {                                                   // Functions.cpp:14-18
        float a = range(-1.f, 1.f), b = range(-1.f, 1.f), c = range(-1.f, 1.f);  // x, y, z in draw order
        return normalized(v3(a, b, c));
    }
};

// ---------------------------------------------------- Functions.cpp:20-60
inline float fmin_ref(float a, float b) { return a < b ? a : b; }   // :20
inline float fmax_ref(float a, float b) { return a > b ? a : b; }   // :24
inline V3 reflect(V3 d, V3 n) { return d - 2 * dot(d, n) * n; }     // :38-40 (2 -> float)
inline V3 refract(V3 d, V3 n, float eta) {                          // :42-47
    float cos_theta = fmin_ref(dot(d, n), 1.0f);
    V3 perp = eta * (d + cos_theta * n);
    V3 par = (float)(-std::sqrt(std::fabs(1.0 - (double)sq_length(perp)))) * n;
    return perp + par;
}
inline float reflectance(float cosine, float ref_idx) {             // :49-54 Schlick
    float r0 = (1 - ref_idx) / (1 + ref_idx);
    r0 = r0 * r0;
    return (float)((double)r0 + (double)(1 - r0) * std::pow((double)(1 - cosine), 5.0));
}
inline float gamma_channel(float c) { return (float)std::pow((double)c, 1.0 / 2.2); }  // :56-60

// ----------------------------------------------------------- AABB.h:48-65
inline bool aabb_intersects(const float lo[3], const float hi[3], const Ray &r, float tmin = EPS_F,
                            float tmax = FLT_MAX) {
    for (int a = 0; a < 3; ++a) {
        const double adinv = 1.0 / (double)r.d[a];
        float t0 = (float)((double)(lo[a] - r.o[a]) * adinv);
        float t1 = (float)((double)(hi[a] - r.o[a]) * adinv);
        if (t0 < t1) {
            if (t0 > tmin) tmin = t0;
            if (t1 < tmax) tmax = t1;
        } else {
            if (t1 > tmin) tmin = t1;
            if (t0 < tmax) tmax = t0;
        }
        if (tmax <= tmin) return false;
    }
    return true;
}

// --------------------------------------------- Triangle.h:26-37, 62-126
struct TriHit {
    bool hit = false;
    float t = FLT_MAX, w0 = 0, w1 = 0, w2 = 0;
    uint32_t tIndex = 0;
    V3 p{0, 0, 0}, n{0, 0, 0};
};
inline TriHit triangle_intersect(V3 c0, V3 c1, V3 c2, const Ray &ray) {
    TriHit out;
    V3 nn = cross(c1 - c0, c2 - c0);        // updateAreaAndNormal :32-37
    float norm = length(nn);
    V3 n = nn / norm;
    float dotRN = dot(ray.d, n);
    if (dotRN == 0) return out;             // :80 parallel
    if (dotRN > 0) return out;              // :87 back face
    float D = dot(c0, n);
    float t = (D - dot(ray.o, n)) / dotRN;  // :95
    if (t < 0) return out;                  // :96
    V3 p = ray.o + t * ray.d;
    V3 v0 = c1 - c0, v1 = c2 - c0, v2 = p - c0;  // computeBarycentricCoordinates :62-75
    float d00 = dot(v0, v0), d01 = dot(v0, v1), d11 = dot(v1, v1), d20 = dot(v2, v0), d21 = dot(v2, v1);
    float denom = d00 * d11 - d01 * d01;
    float u1 = (d11 * d20 - d01 * d21) / denom;
    float u2 = (d00 * d21 - d01 * d20) / denom;
    float u0 = 1 - u1 - u2;
    if (u0 >= 0 && u0 <= 1 && u1 >= 0 && u1 <= 1 && u2 >= 0 && u2 <= 1) {
        out.hit = true; out.t = t; out.w0 = u0; out.w1 = u1; out.w2 = u2; out.p = p; out.n = n;
    }
    return out;
}

// ----------------------------------------------------- Sphere.h:91-132
struct SphereHit {
    bool hit = false;
    float t = FLT_MAX, theta = 0, phi = 0;
    V3 p{0, 0, 0}, n{0, 0, 0};
};
inline SphereHit sphere_intersect(V3 center, float radius, V3 motion, const Ray &ray) {
    SphereHit out;
    V3 tc = center + ray.time * motion;
    V3 o = ray.o, d = ray.d;
    float a = dot(d, d);
    float b = (float)(2. * (double)dot(d, o - tc));
    float c = dot(o - tc, o - tc) - radius * radius;
    float delta = b * b - 4 * a * c;
    if (delta < 0) return out;
    float sq = (float)std::sqrt((double)delta);
    float t = (-b - sq) / (2 * a);
    float t1 = (-b + sq) / (2 * a);
    if ((double)t1 > EPS_D && t1 < t) t = t1;  // :115 -- never true (N6), kept
    if ((double)t < -EPS_D) return out;        // :119
    out.p = o + t * d;
    out.n = normalized(out.p - tc);
    out.hit = true;
    out.t = t;
    out.theta = (float)std::acos((double)out.n.y * -1.);
    out.phi = (float)(std::atan2((double)out.n.z * -1., (double)out.n.x) + M_PI);
    return out;
}

// ------------------------------------------------------ Square.h:65-126
struct QuadHit {
    bool hit = false;
    float t = FLT_MAX, u = 0, v = 0;
    V3 p{0, 0, 0}, n{0, 0, 0};
};
inline QuadHit quad_intersect(V3 v0, V3 v1, V3 v3_, V3 motion, bool glass, const Ray &ray) {
    QuadHit out;
    V3 bl = v0 + ray.time * motion;
    V3 right = v1 - v0, up = v3_ - v0;
    V3 n = normalized(cross(right, up));
    float dotRN = dot(ray.d, n);
    if (dotRN == 0) return out;                   // :75 parallel
    if (dotRN > 0 && !glass) return out;          // :82 back-face cull except glass
    float D = dot(bl, n);
    float t = (D - dot(ray.o, n)) / dotRN;
    if ((double)t < -EPS_D) return out;           // :92
    if ((double)t >= EPS_D) {                     // :98
        V3 p = ray.o + t * ray.d;
        V3 q = p - bl;
        float proj1 = dot(q, right) / length(right);
        float proj2 = dot(q, up) / length(up);
        if ((proj1 <= length(right) && proj1 >= 0) && (proj2 <= length(up) && proj2 >= 0)) {
            out.hit = true; out.t = t;
            out.u = proj1 / length(right);
            out.v = proj2 / length(up);
            out.p = p; out.n = n;
        }
    }
    return out;
}

// ---------------------------------------------------------------- meshes
struct OMesh {
    const hrt_mesh *src;
    std::vector<V3> scaled;  // vertices * TRIANGLE_SCALING (KDTree.cpp:38-40)
    // the reference's pointer tree, restated (KDTree.cpp:4-151)
    struct Node {
        float lo[3], hi[3];
        int left = -1, right = -1;
        std::vector<uint32_t> tris;
        bool leaf() const { return left < 0 && right < 0; }
    };
    std::vector<Node> nodes;
    int root = -1;
};

struct TriBox { float lo[3], hi[3]; };
inline TriBox tri_box(const OMesh &m, uint32_t t) {  // Triangle::getAABB, Triangle.h:128-141 (UNscaled vertices)
    TriBox b;
    for (int a = 0; a < 3; ++a) { b.lo[a] = FLT_MAX; b.hi[a] = -FLT_MAX; }
    for (int k = 0; k < 3; ++k) {
        const float *p = m.src->positions + 3 * (size_t)m.src->indices[3 * (size_t)t + k];
        for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(b.lo[a], p[a]); b.hi[a] = std::max(b.hi[a], p[a]); }
    }
    return b;
}

// KDTree::buildTree, KDTree.cpp:100-151 (+ cut :87-98)
int build_ref_tree(OMesh &m, const std::vector<uint32_t> &tris, const float lo[3], const float hi[3], unsigned depth) {
    if (tris.empty() || depth > 100) return -1;   // KDTREE_MAX_DEPTH
    int self = (int)m.nodes.size();
    m.nodes.emplace_back();
    for (int a = 0; a < 3; ++a) { m.nodes[self].lo[a] = lo[a]; m.nodes[self].hi[a] = hi[a]; }
    if (tris.size() <= 40) {                      // KDTREE_TRIANGLES_PER_LEAF
        m.nodes[self].tris = tris;
        return self;
    }
    const int axis = depth % 3;
    std::vector<float> mins;
    mins.reserve(tris.size());
    for (uint32_t t : tris) mins.push_back(tri_box(m, t).lo[axis]);
    std::sort(mins.begin(), mins.end());
    const float pos = (float)((double)mins[mins.size() / 2] + EPS_D);
    std::vector<uint32_t> L, R;
    for (uint32_t t : tris) {
        TriBox b = tri_box(m, t);
        if ((double)b.hi[axis] <= (double)pos - EPS_D) L.push_back(t);
        else if ((double)b.lo[axis] >= (double)pos + EPS_D) R.push_back(t);
        else { L.push_back(t); R.push_back(t); }
    }
    if (L.size() == R.size()) {                   // :143-146
        m.nodes[self].tris = tris;
        return self;
    }
    float lhi[3] = {hi[0], hi[1], hi[2]}, rlo[3] = {lo[0], lo[1], lo[2]};
    lhi[axis] = pos;                              // AABB::split, AABB.h:67-73
    rlo[axis] = pos;
    int l = build_ref_tree(m, L, lo, lhi, depth + 1);
    int r = build_ref_tree(m, R, rlo, hi, depth + 1);
    m.nodes[self].left = l;
    m.nodes[self].right = r;
    return self;
}

inline TriHit leaf_triangle(const OMesh &m, uint32_t t, const Ray &ray, Counters *cnt) {
    if (cnt) cnt->tri_tests++;
    const uint32_t *ix = m.src->indices + 3 * (size_t)t;
    return triangle_intersect(m.scaled[ix[0]], m.scaled[ix[1]], m.scaled[ix[2]], ray);
}

// KDTree::Node::intersect, KDTree.cpp:31-69
TriHit ref_node_intersect(const OMesh &m, int ni, const Ray &ray, Counters *cnt) {
    const OMesh::Node &n = m.nodes[ni];
    if (cnt) cnt->node_visits++;
    if (!aabb_intersects(n.lo, n.hi, ray)) return TriHit();
    if (n.leaf()) {
        TriHit best;
        best.t = FLT_MAX;
        for (uint32_t t : n.tris) {
            TriHit h = leaf_triangle(m, t, ray, cnt);
            if (h.t < best.t) { best = h; best.tIndex = t; }
        }
        return best;
    }
    TriHit lh, rh;
    if (n.left >= 0) lh = ref_node_intersect(m, n.left, ray, cnt); else lh.t = FLT_MAX;
    if (n.right >= 0) rh = ref_node_intersect(m, n.right, ray, cnt); else rh.t = FLT_MAX;
    return (lh.t < rh.t) ? lh : rh;
}

inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// A CPU walk of the product's flattened rope tree (include/hrt.h, hrt_kdunit).
// Not reference behaviour: it exists so the traversal the kernel implements
// can be checked against the two forms above, and to count its work.
TriHit rope_tree_intersect(const OMesh &m, const Ray &ray, Counters *cnt) {
    TriHit best;
    const hrt_mesh &s = *m.src;
    for (uint32_t k = 0; k < s.n_exceptions;) {  // irregular triangles: through their reference leaves (include/hrt.h): the entries of one
        const hrt_tri_exception &e = s.exceptions[k];  // (triangle, group) are the boxes that must ALL be passed to reach that leaf
        bool reach = true;
        uint32_t k1 = k;
        for (; k1 < s.n_exceptions && s.exceptions[k1].triangle == e.triangle && s.exceptions[k1].group == e.group; ++k1)
            reach = reach && aabb_intersects(s.exceptions[k1].box_min, s.exceptions[k1].box_max, ray);
        k = k1;
        if (!reach) continue;
        TriHit h = leaf_triangle(m, e.triangle, ray, cnt);
        if (h.t < best.t) { best = h; best.tIndex = e.triangle; }
    }
    if (s.kd_root == HRT_KD_NIL || s.n_kd_units == 0) return best;
    uint32_t ref = s.kd_root;
    float t_entry = 0.f;
    const float *lo = s.kd_min, *hi = s.kd_max;  // root cell
    float t_exit_scene = FLT_MAX;
    for (int a = 0; a < 3; ++a) {
        float inv = 1.0f / ray.d[a];
        float t0 = (lo[a] - ray.o[a]) * inv, t1 = (hi[a] - ray.o[a]) * inv;
        if (t0 > t1) std::swap(t0, t1);
        if (t0 > t_entry) t_entry = t0;
        if (t1 < t_exit_scene) t_exit_scene = t1;
    }
    if (!(t_entry <= t_exit_scene)) return best;
    int guard = 0;
    while (ref != HRT_KD_NIL && guard++ < 4096) {
        V3 p = ray.o + t_entry * ray.d;
        while (!(ref & HRT_KD_LEAF)) {  // descend to the leaf containing p
            const hrt_kdunit *u = s.kd_units + ref;
            if (cnt) cnt->node_visits++;
            float split = u2f(u->w[0]);
            uint32_t axis = u->w[1];
            float pc = p[(int)axis];
            bool left = (pc < split) || (pc == split && ray.d[(int)axis] < 0.f);
            ref = left ? u->w[2] : u->w[3];
        }
        const hrt_kdunit *u = s.kd_units + (ref & ~HRT_KD_LEAF);
        if (cnt) cnt->node_visits++;
        float clo[3] = {u2f(u[0].w[0]), u2f(u[0].w[1]), u2f(u[0].w[2])};
        float chi[3] = {u2f(u[1].w[0]), u2f(u[1].w[1]), u2f(u[1].w[2])};
        uint32_t first = u[0].w[3], count = u[1].w[3];
        for (uint32_t k = 0; k < count; ++k) {
            uint32_t t = s.leaf_tris[first + k];
            TriHit h = leaf_triangle(m, t, ray, cnt);
            if (h.t < best.t) { best = h; best.tIndex = t; }
        }
        // exit face of this cell
        float t_exit = FLT_MAX;
        int face = -1;
        for (int a = 0; a < 3; ++a) {
            float da = ray.d[a];
            if (da == 0.f) continue;
            float plane = da > 0.f ? chi[a] : clo[a];
            float t = (plane - ray.o[a]) / da;
            if (t < t_exit) { t_exit = t; face = 2 * a + (da > 0.f ? 1 : 0); }
        }
        if (best.t <= t_exit || face < 0) break;
        if (t_exit > t_entry) t_entry = t_exit;
        ref = face < 4 ? u[2].w[face] : u[3].w[face - 4];
    }
    return best;
}

enum MeshMode { MESH_REF_TREE = 0, MESH_BRUTE = 1, MESH_ROPE_TREE = 2 };

// Mesh::intersect (Mesh.cpp:112-117) -> KDTree::intersect (KDTree.cpp:80-85)
TriHit mesh_intersect(const OMesh &m, const Ray &ray, int mode, Counters *cnt) {
    if (mode == MESH_BRUTE) {  // Mesh::intersectOld, Mesh.h:257-277
        TriHit best;
        best.t = FLT_MAX;
        if (cnt) cnt->node_visits++;
        if (!aabb_intersects(m.src->aabb_min, m.src->aabb_max, ray)) return best;
        for (uint32_t t = 0; t < m.src->n_triangles; ++t) {
            TriHit h = leaf_triangle(m, t, ray, cnt);
            if (h.hit && h.t < best.t) { best = h; best.tIndex = t; }
        }
        return best;
    }
    if (cnt) cnt->node_visits++;
    if (!aabb_intersects(m.src->aabb_min, m.src->aabb_max, ray)) return TriHit();  // KDTree.cpp:82
    if (mode == MESH_ROPE_TREE) return rope_tree_intersect(m, ray, cnt);
    if (m.root < 0) return TriHit();
    return ref_node_intersect(m, m.root, ray, cnt);
}

// ------------------------------------------------------------------ scene
struct OScene {
    const hrt_scene_desc *d;
    std::vector<OMesh> meshes;
    int mesh_mode = MESH_REF_TREE;
};

// Material::texture, Material.cpp:63-92
V3 material_texture(const OScene &S, const hrt_material &m, V3 color, float u, float v, Counters *cnt) {
    switch (m.texture_type) {
        case HRT_TEX_CHECKER:
            if ((int)(u * m.tex_scale_x) % 2 == (int)(v * m.tex_scale_y) % 2) color = v3(m.checker1);
            else color = v3(m.checker2);
            break;
        case HRT_TEX_IMAGE: {
            const hrt_image *img = (m.image >= 0 && (uint32_t)m.image < S.d->n_images) ? &S.d->images[m.image] : nullptr;
            if (!img || img->w < 1 || img->h < 1) {
                if ((int)((double)u * 8.) % 2 == (int)((double)v * 8.) % 2) color = v3(0.f, 0.f, 0.f);
                else color = v3(1.f, 0.f, 1.f);
                break;
            }
            u = (float)std::fmod((double)(u * m.tex_scale_x), 1.);
            v = (float)(1 - std::fmod((double)(v * m.tex_scale_y), 1.));
            int x = int(u * (img->w - 1));
            int y = int(v * (img->h - 1));
            int index = y * img->w + x;
            const uint8_t *px = img->rgb + 3 * (size_t)index;
            if (cnt) cnt->texel_lookups++;
            color = v3((float)(px[0] / 255.), (float)(px[1] / 255.), (float)(px[2] / 255.));
            break;
        }
        default: break;
    }
    return color;
}

// Material::get_normal, Material.cpp:114-130
V3 material_get_normal(const OScene &S, const hrt_material &m, V3 normal, float u, float v, V3 T, V3 B, Counters *cnt) {
    if (m.normal_map < 0) return normal;
    const hrt_image &img = S.d->images[m.normal_map];
    u = (float)std::fmod((double)(u * m.tex_scale_x), 1.);
    v = (float)(1 - std::fmod((double)(v * m.tex_scale_y), 1.));
    int x = int(u * (img.w - 1));
    int y = int(v * (img.h - 1));
    int index = y * img.w + x;
    const uint8_t *px = img.rgb + 3 * (size_t)index;
    if (cnt) cnt->texel_lookups++;
    V3 nm = v3((float)(px[0] / 127.5 - 1.), (float)(px[1] / 127.5 - 1.), (float)(px[2] / 127.5 - 1.));
    V3 n = nm.x * T + nm.y * B + nm.z * normal;
    return normalized(n);
}

// Material::emit, Material.cpp:13-24
V3 material_emit(const OScene &S, const hrt_material &m, float u, float v, Counters *cnt) {
    if (!m.emissive) return v3(0.f, 0.f, 0.f);
    V3 color = v3(0.f, 0.f, 0.f);  // `Vec3 emission;` is zero-initialised before emit() writes it
    if (m.texture_type == HRT_TEX_NONE) color = v3(m.light_color);
    else color = material_texture(S, m, color, u, v, cnt);
    return color * m.light_intensity;
}

// Material::scatter, Material.cpp:26-60
Ray material_scatter(const hrt_material &m, const Ray &in, V3 normal, V3 point, PathRng &rng) {
    V3 dir = v3(0.f, 0.f, 0.f);
    switch (m.type) {
        case HRT_MAT_GLASS: {
            float ri;
            if (dot(in.d, normal) > 0) ri = (float)(1. / (double)m.index_medium);
            else ri = m.index_medium;
            float cos_theta = fmin_ref(dot(in.d * -1.f, normal), 1.0f);
            float sin_theta = (float)std::sqrt(1. - (double)(cos_theta * cos_theta));
            bool cannot_refract = ((double)(ri * sin_theta) - 0.6) > 1.0;
            if (cannot_refract || reflectance(cos_theta, ri) > rng.next()) dir = reflect(in.d, normal);
            else dir = refract(in.d, normal, ri);
            break;
        }
        case HRT_MAT_DIFFUSE:
            dir = normal + rng.unit_vector();
            if ((double)length(dir) <= EPS_D) dir = normal;
            break;
        case HRT_MAT_MIRROR:
            dir = reflect(in.d, normal);
            break;
        default: break;
    }
    dir = normalized(dir);
    return make_ray(point + EPS_F * dir, dir, in.time);
}

struct SceneHit {
    int kind = 0;  // 0 none, 1 sphere, 2 square, 3 mesh (Scene.h:46)
    int index = -1;
    float t = FLT_MAX;
    SphereHit sph;
    QuadHit quad;
    TriHit tri;
};

// Debug instrument (oracle_trace_path): when set, every closest-hit query of this thread appends
// {o.xyz, d.xyz, time, kind, index, t, tIndex, 0} (12 floats) here.
thread_local float *g_trace = nullptr;
thread_local uint32_t g_trace_n = 0, g_trace_cap = 0;

SceneHit compute_intersection_impl(const OScene &S, const Ray &ray, Counters *cnt);
// Scene::computeIntersection, Scene.h:202-230
SceneHit compute_intersection(const OScene &S, const Ray &ray, Counters *cnt) {
    SceneHit res = compute_intersection_impl(S, ray, cnt);
    if (g_trace && g_trace_n < g_trace_cap) {
        float *o = g_trace + 12 * (size_t)g_trace_n++;
        o[0] = ray.o.x; o[1] = ray.o.y; o[2] = ray.o.z; o[3] = ray.d.x; o[4] = ray.d.y; o[5] = ray.d.z; o[6] = ray.time;
        o[7] = (float)res.kind; o[8] = (float)res.index; o[9] = res.kind ? res.t : 0.f;
        o[10] = res.kind == 3 ? (float)res.tri.tIndex : -1.f; o[11] = 0.f;
    }
    return res;
}
SceneHit compute_intersection_impl(const OScene &S, const Ray &ray, Counters *cnt) {
    SceneHit res;
    const hrt_scene_desc &d = *S.d;
    if (cnt) cnt->closest_queries++;
    for (uint32_t i = 0; i < d.n_spheres; ++i) {
        const hrt_sphere &s = d.spheres[i];
        if (cnt) cnt->sphere_tests++;
        SphereHit h = sphere_intersect(v3(s.center), s.radius, v3(d.materials[s.material].motion), ray);
        if (h.hit && h.t < res.t && (double)h.t >= EPS_D) { res.kind = 1; res.index = (int)i; res.t = h.t; res.sph = h; }
    }
    for (uint32_t i = 0; i < d.n_quads; ++i) {
        const hrt_quad &q = d.quads[i];
        const hrt_material &m = d.materials[q.material];
        if (cnt) cnt->quad_tests++;
        QuadHit h = quad_intersect(v3(q.v0), v3(q.v1), v3(q.v3), v3(m.motion), m.type == HRT_MAT_GLASS, ray);
        if (h.hit && h.t < res.t && (double)h.t >= EPS_D) { res.kind = 2; res.index = (int)i; res.t = h.t; res.quad = h; }
    }
    for (uint32_t i = 0; i < d.n_meshes; ++i) {
        TriHit h = mesh_intersect(S.meshes[i], ray, S.mesh_mode, cnt);
        if (h.hit && h.t < res.t && (double)h.t >= EPS_D) { res.kind = 3; res.index = (int)i; res.t = h.t; res.tri = h; }
    }
    return res;
}

// Scene::computeShadow, Scene.h:235-255
bool compute_shadow(const OScene &S, const Ray &ray, float tmax, PathRng &rng, Counters *cnt) {
    const hrt_scene_desc &d = *S.d;
    if (cnt) cnt->shadow_queries++;
    for (uint32_t i = 0; i < d.n_spheres; ++i) {
        const hrt_sphere &s = d.spheres[i];
        const hrt_material &m = d.materials[s.material];
        if (cnt) cnt->sphere_tests++;
        SphereHit h = sphere_intersect(v3(s.center), s.radius, v3(m.motion), ray);
        if (h.hit && h.t < tmax && (double)h.t >= EPS_D)
            if (rng.next() > m.transparency) return true;
    }
    for (uint32_t i = 0; i < d.n_quads; ++i) {
        const hrt_quad &q = d.quads[i];
        const hrt_material &m = d.materials[q.material];
        if (cnt) cnt->quad_tests++;
        QuadHit h = quad_intersect(v3(q.v0), v3(q.v1), v3(q.v3), v3(m.motion), m.type == HRT_MAT_GLASS, ray);
        if (h.hit && h.t < tmax && (double)h.t >= EPS_D)
            if (rng.next() > m.transparency) return true;
    }
    for (uint32_t i = 0; i < d.n_meshes; ++i) {
        const hrt_material &m = d.materials[d.meshes[i].material];
        TriHit h = mesh_intersect(S.meshes[i], ray, S.mesh_mode, cnt);
        if (h.hit && h.t < tmax && (double)h.t >= EPS_D)
            if (rng.next() > m.transparency) return true;
    }
    return false;
}

// Scene::skyboxTexture, Scene.h:149-161
V3 skybox_texture(const OScene &S, V3 dir, int remaining, Counters *cnt) {
    const hrt_scene_desc &d = *S.d;
    const hrt_image *sky = (d.skybox_image >= 0) ? &d.images[d.skybox_image] : nullptr;
    if (!sky || sky->w < 1 || sky->h < 1) {
        if (d.dark_sky) return v3(0.f, 0.f, 0.f);
        float a = (float)(0.5 * ((double)dir.y + 1.0));
        return (float)(1.0 - (double)a) * v3(1.f, 1.f, 1.f) + a * v3(0.5f, 0.7f, 1.0f) * (float)(remaining + 1);
    }
    float u = (float)(0.5 + std::atan2((double)dir.z, (double)dir.x) / (2 * M_PI));
    float v = (float)(0.5 - std::asin((double)dir.y) / M_PI);
    int x = (int)(u * sky->w);
    int y = (int)(v * sky->h);
    if (x >= sky->w) x = sky->w - 1;  // the reference reads out of bounds at u == 1; clamped here
    if (y >= sky->h) y = sky->h - 1;
    const uint8_t *px = sky->rgb + 3 * ((size_t)y * sky->w + x);
    if (cnt) cnt->texel_lookups++;
    return v3((float)(px[0] / 255.), (float)(px[1] / 255.), (float)(px[2] / 255.)) * (float)remaining;
}

struct Shading {  // what rayTraceRecursive derives from the closest hit, Scene.h:270-304
    V3 point, normal, albedo, emission;
    const hrt_material *mat;
};

Shading shade_hit(const OScene &S, const SceneHit &hit, Counters *cnt) {
    const hrt_scene_desc &d = *S.d;
    Shading sh;
    sh.emission = v3(0.f, 0.f, 0.f);
    if (cnt) cnt->shaded_hits++;
    if (hit.kind == 1) {
        const hrt_material &m = d.materials[d.spheres[hit.index].material];
        sh.mat = &m;
        sh.point = hit.sph.p;
        sh.normal = hit.sph.n;
        sh.albedo = v3(m.albedo);
        if (m.texture_type == HRT_TEX_CHECKER || m.texture_type == HRT_TEX_IMAGE)  // sphere_texture, Material.cpp:94-103
            sh.albedo = material_texture(S, m, sh.albedo, (float)((double)hit.sph.phi / (2 * M_PI)),
                                         (float)((double)hit.sph.theta / M_PI), cnt);
        sh.emission = material_emit(S, m, (float)((double)hit.sph.phi / (2 * M_PI)), (float)((double)hit.sph.theta / M_PI), cnt);
    } else if (hit.kind == 2) {
        const hrt_quad &q = d.quads[hit.index];
        const hrt_material &m = d.materials[q.material];
        sh.mat = &m;
        sh.point = hit.quad.p;
        sh.normal = hit.quad.n;
        sh.albedo = material_texture(S, m, v3(m.albedo), hit.quad.u, hit.quad.v, cnt);
        sh.normal = material_get_normal(S, m, sh.normal, hit.quad.u, hit.quad.v, v3(q.tangent), v3(q.bitangent), cnt);
        sh.emission = material_emit(S, m, hit.quad.u, hit.quad.v, cnt);
    } else {
        const hrt_mesh &me = d.meshes[hit.index];
        const hrt_material &m = d.materials[me.material];
        sh.mat = &m;
        sh.point = hit.tri.p;
        sh.normal = hit.tri.n;
        sh.albedo = v3(m.albedo);
        if (me.color_type == HRT_COLOR_VERTEX && me.vert_colors) {
            const uint32_t *ix = me.indices + 3 * (size_t)hit.tri.tIndex;
            V3 c0 = v3(me.vert_colors + 3 * (size_t)ix[0]), c1 = v3(me.vert_colors + 3 * (size_t)ix[1]),
               c2 = v3(me.vert_colors + 3 * (size_t)ix[2]);
            sh.albedo = hit.tri.w0 * c0 + hit.tri.w1 * c1 + hit.tri.w2 * c2;
        } else if (me.color_type == HRT_COLOR_FACE && me.face_colors) {
            sh.albedo = v3(me.face_colors + 3 * (size_t)hit.tri.tIndex);
        }
    }
    return sh;
}

// Scene::rayTraceRecursive, Scene.h:258-342
V3 ray_trace_recursive(const OScene &S, Ray ray, int remaining, PathRng &rng, Counters *cnt) {
    V3 color = v3(0.f, 0.f, 0.f);
    if (remaining == 0) return color;
    SceneHit hit = compute_intersection(S, ray, cnt);
    if (hit.kind == 0) return skybox_texture(S, ray.d, remaining, cnt);
    Shading sh = shade_hit(S, hit, cnt);
    const hrt_scene_desc &d = *S.d;
    for (uint32_t i = 0; i < d.n_lights; ++i) {
        V3 L = normalized(v3(d.lights[i].pos) - sh.point);
        float dotLN = dot(L, sh.normal);
        // always lights[0].material (Scene.h:311, N2)
        color = color + comp_product(v3(d.lights[0].color), sh.albedo) * fmax_ref(0.0f, dotLN) *
                            (float)(1. - (double)sh.mat->transparency);
        int blocked = 0;
        const float delta = (float)((double)d.lights[i].radius / 2.);
        for (int j = 0; j < HRT_NB_ECH; ++j) {
            V3 lp = v3(d.lights[i].pos) + rng.unit_vector() * delta;
            L = normalized(lp - sh.point);
            float tLight = length(lp - sh.point);
            if (compute_shadow(S, make_ray(sh.point + L * EPS_F, L, ray.time), tLight, rng, cnt)) blocked++;
        }
        float shadow = (float)(1. - (double)((float)blocked / (float)HRT_NB_ECH));
        color = color * shadow;  // multiplies the running sum (N3)
    }
    Ray next = material_scatter(*sh.mat, ray, sh.normal, sh.point, rng);
    next.time = ray.time;
    V3 child = ray_trace_recursive(S, next, remaining - 1, rng, cnt);
    child = comp_product(child, sh.albedo);
    return color + child + sh.emission;
}

// Scene::rayTrace, Scene.h:345-350
V3 ray_trace(const OScene &S, const Ray &start, PathRng &rng, Counters *cnt) {
    V3 c = v3(0.f, 0.f, 0.f) + ray_trace_recursive(S, start, HRT_MAXBOUNCES, rng, cnt);
    return c / (float)HRT_MAXBOUNCES;
}

// ----------------------------------------- matrixUtilities.h:53-74, 77-216
struct CameraMats {
    double mv_inv[16], p_inv[16];  // column-major like GL
};

// gluInvertMatrix, matrixUtilities.h:77-206: the adjugate over the determinant.  Entry e of the adjugate is a sum of six
// signed triple products, evaluated left to right -- ((s*m[a]) * m[b]) * m[c], terms added in the order written (a
// subtraction is the addition of the negated product, which rounds identically).  The table lists the terms of every
// entry in the reference's order, so each entry rounds exactly as the reference's; bit-exactness against the reference's
// own function is checked by tests/test_oracle_pins.py (live and through tests/golden/ref_kat.npz).
static const signed char kAdjugate[16][6][4] = {
    {{1, 5, 10, 15}, {-1, 5, 11, 14}, {-1, 9, 6, 15}, {1, 9, 7, 14}, {1, 13, 6, 11}, {-1, 13, 7, 10}},
    {{-1, 1, 10, 15}, {1, 1, 11, 14}, {1, 9, 2, 15}, {-1, 9, 3, 14}, {-1, 13, 2, 11}, {1, 13, 3, 10}},
    {{1, 1, 6, 15}, {-1, 1, 7, 14}, {-1, 5, 2, 15}, {1, 5, 3, 14}, {1, 13, 2, 7}, {-1, 13, 3, 6}},
    {{-1, 1, 6, 11}, {1, 1, 7, 10}, {1, 5, 2, 11}, {-1, 5, 3, 10}, {-1, 9, 2, 7}, {1, 9, 3, 6}},
    {{-1, 4, 10, 15}, {1, 4, 11, 14}, {1, 8, 6, 15}, {-1, 8, 7, 14}, {-1, 12, 6, 11}, {1, 12, 7, 10}},
    {{1, 0, 10, 15}, {-1, 0, 11, 14}, {-1, 8, 2, 15}, {1, 8, 3, 14}, {1, 12, 2, 11}, {-1, 12, 3, 10}},
    {{-1, 0, 6, 15}, {1, 0, 7, 14}, {1, 4, 2, 15}, {-1, 4, 3, 14}, {-1, 12, 2, 7}, {1, 12, 3, 6}},
    {{1, 0, 6, 11}, {-1, 0, 7, 10}, {-1, 4, 2, 11}, {1, 4, 3, 10}, {1, 8, 2, 7}, {-1, 8, 3, 6}},
    {{1, 4, 9, 15}, {-1, 4, 11, 13}, {-1, 8, 5, 15}, {1, 8, 7, 13}, {1, 12, 5, 11}, {-1, 12, 7, 9}},
    {{-1, 0, 9, 15}, {1, 0, 11, 13}, {1, 8, 1, 15}, {-1, 8, 3, 13}, {-1, 12, 1, 11}, {1, 12, 3, 9}},
    {{1, 0, 5, 15}, {-1, 0, 7, 13}, {-1, 4, 1, 15}, {1, 4, 3, 13}, {1, 12, 1, 7}, {-1, 12, 3, 5}},
    {{-1, 0, 5, 11}, {1, 0, 7, 9}, {1, 4, 1, 11}, {-1, 4, 3, 9}, {-1, 8, 1, 7}, {1, 8, 3, 5}},
    {{-1, 4, 9, 14}, {1, 4, 10, 13}, {1, 8, 5, 14}, {-1, 8, 6, 13}, {-1, 12, 5, 10}, {1, 12, 6, 9}},
    {{1, 0, 9, 14}, {-1, 0, 10, 13}, {-1, 8, 1, 14}, {1, 8, 2, 13}, {1, 12, 1, 10}, {-1, 12, 2, 9}},
    {{-1, 0, 5, 14}, {1, 0, 6, 13}, {1, 4, 1, 14}, {-1, 4, 2, 13}, {-1, 12, 1, 6}, {1, 12, 2, 5}},
    {{1, 0, 5, 10}, {-1, 0, 6, 9}, {-1, 4, 1, 10}, {1, 4, 2, 9}, {1, 8, 1, 6}, {-1, 8, 2, 5}}};

bool invert4(const double m[16], double out[16]) {
    double adj[16];
    for (int e = 0; e < 16; ++e) {
        double sum = 0.0;
        for (int k = 0; k < 6; ++k) {
            const signed char *t = kAdjugate[e][k];
            const double term = ((t[0] < 0 ? -m[t[1]] : m[t[1]]) * m[t[2]]) * m[t[3]];
            sum = k == 0 ? term : sum + term;
        }
        adj[e] = sum;
    }
    double det = m[0] * adj[0] + m[1] * adj[4] + m[2] * adj[8] + m[3] * adj[12];  // :195
    if (det == 0) return false;
    det = 1.0 / det;
    for (int e = 0; e < 16; ++e) out[e] = adj[e] * det;
    return true;
}

inline void mult4(const double m[16], double x, double y, double z, double w, double r[4]) {  // :210-216
    r[0] = m[0] * x + m[4] * y + m[8] * z + m[12] * w;
    r[1] = m[1] * x + m[5] * y + m[9] * z + m[13] * w;
    r[2] = m[2] * x + m[6] * y + m[10] * z + m[14] * w;
    r[3] = m[3] * x + m[7] * y + m[11] * z + m[15] * w;
}

// The GL matrices the reference would read back (matrixUtilities.h:33-46) for the pose an hrt_camera describes:
void camera_forward_matrices(const hrt_camera &c, double mv[16], double p[16]) {
    for (int k = 0; k < 16; ++k) { mv[k] = 0.0; p[k] = 0.0; }
    // modelview = [right; up; -forward] * translate(-eye)   (Camera.cpp:125-132 composes the same for the default pose)
    const double R[3][3] = {{c.right[0], c.right[1], c.right[2]},
                            {c.up[0], c.up[1], c.up[2]},
                            {-(double)c.forward[0], -(double)c.forward[1], -(double)c.forward[2]}};
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k) mv[k * 4 + r] = R[r][k];
        mv[12 + r] = -(R[r][0] * c.eye[0] + R[r][1] * c.eye[1] + R[r][2] * c.eye[2]);
    }
    mv[15] = 1.0;
    // gluPerspective(fovy, aspect, near, far)  (Camera.cpp:47)
    const double rad = (double)c.fovy_deg / 2.0 * M_PI / 180.0;
    const double cot = std::cos(rad) / std::sin(rad);
    const double dz = (double)c.zfar - (double)c.znear;
    p[0] = cot / (double)c.aspect;
    p[5] = cot;
    p[10] = -((double)c.zfar + (double)c.znear) / dz;
    p[11] = -1.0;
    p[14] = -2.0 * (double)c.znear * (double)c.zfar / dz;
}

CameraMats camera_matrices(const hrt_camera &c) {
    double mv[16], p[16];
    camera_forward_matrices(c, mv, p);
    CameraMats out;
    invert4(mv, out.mv_inv);
    invert4(p, out.p_inv);
    return out;
}

// screen_space_to_world_space_ray, matrixUtilities.h:70-74
void camera_ray(const CameraMats &cm, float u, float v, V3 &pos, V3 &dir) {
    double r[4];
    mult4(cm.mv_inv, 0.0, 0.0, 0.0, 1.0, r);                       // cameraSpaceToWorldSpace(0,0,0) :53-58
    pos = v3((float)(r[0] / r[3]), (float)(r[1] / r[3]), (float)(r[2] / r[3]));
    double ri[4];
    mult4(cm.p_inv, 2.0 * (double)u - 1.0, -(2.0 * (double)v - 1.0), 0.0 /* GL_DEPTH_RANGE[0] */, 1.0, ri);
    mult4(cm.mv_inv, ri[0], ri[1], ri[2], ri[3], r);
    V3 world = v3((float)(r[0] / r[3]), (float)(r[1] / r[3]), (float)(r[2] / r[3]));
    dir = normalized(world - pos);
}

struct Prepared {
    OScene scene;
};

void prepare(OScene &S, const hrt_scene_desc *d, int mesh_mode) {
    S.d = d;
    S.mesh_mode = mesh_mode;
    S.meshes.resize(d->n_meshes);
    for (uint32_t i = 0; i < d->n_meshes; ++i) {
        OMesh &m = S.meshes[i];
        m.src = &d->meshes[i];
        m.scaled.resize(m.src->n_vertices);
        for (uint32_t v = 0; v < m.src->n_vertices; ++v)
            m.scaled[v] = v3(m.src->positions + 3 * (size_t)v) * HRT_TRIANGLE_SCALING;
        if (mesh_mode == MESH_REF_TREE) {
            std::vector<uint32_t> all(m.src->n_triangles);
            for (uint32_t t = 0; t < m.src->n_triangles; ++t) all[t] = t;
            m.root = build_ref_tree(m, all, m.src->aabb_min, m.src->aabb_max, 0);
        }
    }
}

// trace_line, main.cpp:183-198, for one pixel
V3 render_pixel(const OScene &S, const CameraMats &cm, uint32_t x, uint32_t y, uint32_t w, uint32_t h,
                uint32_t spp, uint64_t seed, bool gamma, Counters *cnt) {
    V3 acc = v3(0.f, 0.f, 0.f);
    for (uint32_t s = 0; s < spp; ++s) {
        PathRng rng(seed, y * w + x, s, cnt);
        float u = ((float)x + rng.next()) / w;
        float v = ((float)y + rng.next()) / h;
        V3 pos, dir;
        camera_ray(cm, u, v, pos, dir);
        float time = rng.next();
        V3 c = ray_trace(S, make_ray(pos, dir, time), rng, cnt);
        acc = acc + c;
        if (cnt) cnt->samples++;
    }
    acc = acc / (float)spp;
    if (gamma) acc = v3(gamma_channel(acc.x), gamma_channel(acc.y), gamma_channel(acc.z));
    return acc;
}

}  // namespace

// ============================================================================
// C entry points (ctypes)
// ============================================================================
extern "C" {

struct oracle_scene {
    OScene S;
};

oracle_scene *oracle_scene_create(const hrt_scene_desc *desc, int mesh_mode) {
    oracle_scene *o = new oracle_scene();
    prepare(o->S, desc, mesh_mode);
    return o;
}
void oracle_scene_destroy(oracle_scene *o) { delete o; }

// reference-tree statistics: nodes, leaves, triangle refs, max leaf size
void oracle_ref_tree_stats(const oracle_scene *o, uint32_t mesh, uint32_t out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0;
    if (mesh >= o->S.meshes.size()) return;
    for (const OMesh::Node &n : o->S.meshes[mesh].nodes) {
        out[0]++;
        if (n.leaf()) { out[1]++; out[2] += (uint32_t)n.tris.size(); out[3] = std::max(out[3], (uint32_t)n.tris.size()); }
    }
}

// threads: >0 = pool of that many workers over rows; 0 = hardware_concurrency;
//          -1 = one std::thread per scanline, as main.cpp:232-238.
// counters (optional, 10 x u64): closest, shadow, sphere, quad, node, tri, shaded, texel, rng, samples.
int oracle_render(const oracle_scene *o, const hrt_camera *cam, uint32_t w, uint32_t h, uint32_t spp,
                  uint64_t seed, uint32_t flags, int threads, float *out_rgb, uint64_t *counters) {
    if (!o || !cam || !out_rgb || !w || !h || !spp) return -1;
    const OScene &S = o->S;
    const CameraMats cm = camera_matrices(*cam);
    const bool gamma = (flags & HRT_FLAG_GAMMA) != 0;
    g_shared_rng = (flags & (1u << 16)) != 0;  // ORACLE_FLAG_SHARED_RNG (timing only; see shared_random_float)
    struct Restore { ~Restore() { g_shared_rng = false; } } restore_shared_rng;
    Counters total;
    auto do_row = [&](uint32_t y, Counters *cnt) {
        for (uint32_t x = 0; x < w; ++x) {
            V3 c = render_pixel(S, cm, x, y, w, h, spp, seed, gamma, cnt);
            float *px = out_rgb + 3 * ((size_t)y * w + x);
            px[0] = c.x; px[1] = c.y; px[2] = c.z;
        }
    };
    if (threads == -1) {
        std::vector<std::thread> pool;
        std::vector<Counters> per(h);
        for (uint32_t y = 0; y < h; ++y) pool.emplace_back([&, y]() { do_row(y, counters ? &per[y] : nullptr); });
        for (auto &t : pool) t.join();
        for (auto &c : per) total.add(c);
    } else {
        unsigned n = threads > 0 ? (unsigned)threads : std::max(1u, std::thread::hardware_concurrency());
        if (n > h) n = h;
        std::atomic<uint32_t> next{0};
        std::vector<Counters> per(n);
        auto worker = [&](unsigned id) {
            for (;;) {
                uint32_t y = next.fetch_add(1);
                if (y >= h) break;
                do_row(y, counters ? &per[id] : nullptr);
            }
        };
        if (n == 1) worker(0);
        else {
            std::vector<std::thread> pool;
            for (unsigned i = 0; i < n; ++i) pool.emplace_back(worker, i);
            for (auto &t : pool) t.join();
        }
        for (auto &c : per) total.add(c);
    }
    if (counters) {
        counters[0] = total.closest_queries; counters[1] = total.shadow_queries; counters[2] = total.sphere_tests;
        counters[3] = total.quad_tests; counters[4] = total.node_visits; counters[5] = total.tri_tests;
        counters[6] = total.shaded_hits; counters[7] = total.texel_lookups; counters[8] = total.rng_draws;
        counters[9] = total.samples;
    }
    return 0;
}

// Deterministic first-hit AOVs through pixel centres (no RNG, time 0):
//   hit   : (t, kind, index)  index = sphere/quad id or triangle id, -1 on miss, t = 0 on miss
//   normal: shading normal (after normal mapping)    albedo: after textures    emission
int oracle_aov(const oracle_scene *o, const hrt_camera *cam, uint32_t w, uint32_t h, float *hit, float *normal,
               float *albedo, float *emission) {
    if (!o || !cam) return -1;
    const OScene &S = o->S;
    const CameraMats cm = camera_matrices(*cam);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            V3 pos, dir;
            camera_ray(cm, ((float)x + 0.5f) / w, ((float)y + 0.5f) / h, pos, dir);
            SceneHit sh = compute_intersection(S, make_ray(pos, dir, 0.f), nullptr);
            size_t i = 3 * ((size_t)y * w + x);
            V3 n = v3(0, 0, 0), a = v3(0, 0, 0), e = v3(0, 0, 0);
            float idx = -1.f, t = 0.f;
            if (sh.kind) {
                Shading s = shade_hit(S, sh, nullptr);
                n = s.normal; a = s.albedo; e = s.emission;
                idx = (sh.kind == 3) ? (float)sh.tri.tIndex : (float)sh.index;
                t = sh.t;
            }
            if (hit) { hit[i] = t; hit[i + 1] = (float)sh.kind; hit[i + 2] = idx; }
            if (normal) { normal[i] = n.x; normal[i + 1] = n.y; normal[i + 2] = n.z; }
            if (albedo) { albedo[i] = a.x; albedo[i + 1] = a.y; albedo[i + 2] = a.z; }
            if (emission) { emission[i] = e.x; emission[i + 1] = e.y; emission[i + 2] = e.z; }
        }
    return 0;
}

// Closest mesh hit for a batch of rays under a given mesh mode (tree cross-checks).
// rays: n x 7 (o, d, time); out: n x 3 (hit, t, tIndex)
int oracle_mesh_query(const hrt_scene_desc *desc, uint32_t mesh, int mesh_mode, const float *rays, uint32_t n, float *out) {
    if (!desc || mesh >= desc->n_meshes) return -1;
    OScene S;
    prepare(S, desc, mesh_mode);
    for (uint32_t i = 0; i < n; ++i) {
        const float *r = rays + 7 * (size_t)i;
        Ray ray = make_ray(v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), r[6]);
        TriHit h = mesh_intersect(S.meshes[mesh], ray, mesh_mode, nullptr);
        out[3 * (size_t)i] = h.hit ? 1.f : 0.f;
        out[3 * (size_t)i + 1] = h.hit ? h.t : 0.f;
        out[3 * (size_t)i + 2] = h.hit ? (float)h.tIndex : -1.f;
    }
    return 0;
}

// ---- per-primitive known-answer entry points (rays: n x 7 = o, d(unnormalised ok), time) ----
// triangle: tri = 9 floats; out n x 8: hit, t, w0, w1, w2, nx, ny, nz
void oracle_kat_triangle(const float *tri, const float *rays, uint32_t n, float *out) {
    for (uint32_t i = 0; i < n; ++i) {
        const float *r = rays + 7 * (size_t)i;
        TriHit h = triangle_intersect(v3(tri), v3(tri + 3), v3(tri + 6), make_ray(v3(r), v3(r + 3), r[6]));
        float *o = out + 8 * (size_t)i;
        o[0] = h.hit; o[1] = h.hit ? h.t : 0.f; o[2] = h.w0; o[3] = h.w1; o[4] = h.w2; o[5] = h.n.x; o[6] = h.n.y; o[7] = h.n.z;
    }
}
// aabb: box = lo(3), hi(3); out n: 0/1
void oracle_kat_aabb(const float *box, const float *rays, uint32_t n, float *out) {
    for (uint32_t i = 0; i < n; ++i) {
        const float *r = rays + 7 * (size_t)i;
        out[i] = aabb_intersects(box, box + 3, make_ray(v3(r), v3(r + 3), r[6])) ? 1.f : 0.f;
    }
}
// sphere: sph = center(3), radius, motion(3); out n x 9: hit, t, theta, phi, nx, ny, nz, px, py
void oracle_kat_sphere(const float *sph, const float *rays, uint32_t n, float *out) {
    for (uint32_t i = 0; i < n; ++i) {
        const float *r = rays + 7 * (size_t)i;
        SphereHit h = sphere_intersect(v3(sph), sph[3], v3(sph + 4), make_ray(v3(r), v3(r + 3), r[6]));
        float *o = out + 9 * (size_t)i;
        o[0] = h.hit; o[1] = h.hit ? h.t : 0.f; o[2] = h.theta; o[3] = h.phi; o[4] = h.n.x; o[5] = h.n.y; o[6] = h.n.z; o[7] = h.p.x; o[8] = h.p.y;
    }
}
// quad: q = v0(3), v1(3), v3(3), motion(3), glass flag; out n x 7: hit, t, u, v, nx, ny, nz
void oracle_kat_quad(const float *q, const float *rays, uint32_t n, float *out) {
    for (uint32_t i = 0; i < n; ++i) {
        const float *r = rays + 7 * (size_t)i;
        QuadHit h = quad_intersect(v3(q), v3(q + 3), v3(q + 6), v3(q + 9), q[12] != 0.f, make_ray(v3(r), v3(r + 3), r[6]));
        float *o = out + 7 * (size_t)i;
        o[0] = h.hit; o[1] = h.hit ? h.t : 0.f; o[2] = h.u; o[3] = h.v; o[4] = h.n.x; o[5] = h.n.y; o[6] = h.n.z;
    }
}
// optics: in n x 8 = d(3), n(3), eta, cosine ; out n x 8 = reflect(3), refract(3), reflectance, gamma(cosine clipped to >=0)
void oracle_kat_optics(const float *in, uint32_t n, float *out) {
    for (uint32_t i = 0; i < n; ++i) {
        const float *a = in + 8 * (size_t)i;
        V3 rf = reflect(v3(a), v3(a + 3));
        V3 rr = refract(v3(a), v3(a + 3), a[6]);
        float *o = out + 8 * (size_t)i;
        o[0] = rf.x; o[1] = rf.y; o[2] = rf.z; o[3] = rr.x; o[4] = rr.y; o[5] = rr.z;
        o[6] = reflectance(a[7], a[6]);
        o[7] = gamma_channel(std::fabs(a[7]));
    }
}
// Ray constructor normalisation (Line.h:13-16, Vec3.h:46): in n x 3 -> out n x 3
void oracle_kat_normalize(const float *in, uint32_t n, float *out) {
    for (uint32_t i = 0; i < n; ++i) {
        const Ray r = make_ray(v3(0.f, 0.f, 0.f), v3(in + 3 * (size_t)i), 0.f);
        out[3 * i] = r.d.x; out[3 * i + 1] = r.d.y; out[3 * i + 2] = r.d.z;
    }
}
// random_float / random_unit_vector as the reference defines them (Functions.cpp:4-18) on an
// mt19937 seeded with `seed` (the reference seeds with time(nullptr)); out n x 4 = float, unit vector.
// Argument evaluation order of Vec3(random_float(),random_float(),random_float()) is unspecified in
// C++; g++ evaluates the constructor arguments right to left, so z is drawn first.
void oracle_kat_random(uint32_t seed, uint32_t n, float *out) {
    std::uniform_real_distribution<float> distribution(0.0, 1.0);
    std::mt19937 generator(seed);
    auto rf = [&]() { return distribution(generator); };
    auto rfr = [&](float lo, float hi) { return lo + (hi - lo) * rf(); };
    for (uint32_t i = 0; i < n; ++i) {
        float *o = out + 4 * (size_t)i;
        o[0] = rf();
        float z = rfr(-1, 1), y = rfr(-1, 1), x = rfr(-1, 1);
        V3 p = normalized(v3(x, y, z));
        o[1] = p.x; o[2] = p.y; o[3] = p.z;
    }
}
// the renderer's counter-based stream, for the GPU-side stream test: out[n] = draws 0..n-1 of (seed,pixel,sample)
void oracle_path_stream(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float *out) {
    PathRng rng(seed, pixel, sample, nullptr);
    for (uint32_t i = 0; i < n; ++i) out[i] = rng.next();
}
// Debug instrument: the closest-hit queries of ONE path (pixel x,y, sample s) in order -- up to `cap` records of 12 floats
// {ray o, d, time, hit kind, object index, t, triangle id, 0}.  Returns the number of records.
uint32_t oracle_trace_path(const oracle_scene *o, const hrt_camera *cam, uint32_t w, uint32_t h, uint32_t x, uint32_t y,
                           uint32_t sample, uint64_t seed, float *out, uint32_t cap) {
    const CameraMats cm = camera_matrices(*cam);
    PathRng rng(seed, y * w + x, sample, nullptr);
    float u = ((float)x + rng.next()) / w;
    float v = ((float)y + rng.next()) / h;
    V3 pos, dir;
    camera_ray(cm, u, v, pos, dir);
    float time = rng.next();
    g_trace = out; g_trace_n = 0; g_trace_cap = cap;
    (void)ray_trace(o->S, make_ray(pos, dir, time), rng, nullptr);
    g_trace = nullptr;
    return g_trace_n;
}

// out: 64 doubles = modelview, projection (what the reference's GL read-back would hold), their inverses
void oracle_camera_matrices(const hrt_camera *cam, double *out) {
    camera_forward_matrices(*cam, out, out + 16);
    invert4(out, out + 32);
    invert4(out + 16, out + 48);
}

// camera rays through (u,v): uv n x 2 -> out n x 6 (pos, dir)
void oracle_camera_rays(const hrt_camera *cam, const float *uv, uint32_t n, float *out) {
    CameraMats cm = camera_matrices(*cam);
    for (uint32_t i = 0; i < n; ++i) {
        V3 p, d;
        camera_ray(cm, uv[2 * i], uv[2 * i + 1], p, d);
        const Ray r = make_ray(p, d, 0.f);  // main.cpp:192: the Ray constructor normalises once more (Line.h:13-16)
        float *o = out + 6 * (size_t)i;
        o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
    }
}

}  // extern "C"
