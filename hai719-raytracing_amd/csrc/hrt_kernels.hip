// Hand-written HIP kernels of the trace path for gfx950 (CDNA4, wave64).
//
// hrt_trace_kernel is the persistent-wavefront megakernel: every wave pulls
// 8x8-pixel tiles from a work-queue head, lane = pixel, and each lane runs the
// reference's per-sample loop (main.cpp:183-198) as a bounce state machine
// that regenerates its next camera sample the moment a path ends, so all 64
// lanes stay busy until the tile's last sample.  Per bounce:
//   ray generation            matrixUtilities.h:53-74 (fp64 mat-vec, as the reference)
//   closest hit               Scene.h:202-230  spheres -> squares -> meshes
//     sphere                  Sphere.h:91-132
//     square                  Square.h:65-126
//     mesh                    Mesh.cpp:112-117, KDTree.cpp:31-85 -- here a stackless walk of
//                             the flattened rope KD-tree, nodelets served from LDS
//     triangle                Triangle.h:62-126 on leaf-ordered rows with the per-triangle constants folded
//   shading                   Scene.h:270-334 (texture, normal map, emission, lights, soft shadows)
//   scatter                   Material.cpp:26-60
//
// NUMERICS.  This file is compiled with -ffp-contract=off and every geometric or
// branch-deciding expression is written in the reference's operation order, with
// its float/double promotions (the "fp64 islands" of SURVEY.md 7): given the same
// ray and the same random numbers a lane takes the same decisions and produces the
// same hit point as the reference arithmetic, bit for bit (IEEE fp32 add/mul/div/
// sqrt are correctly rounded on gfx950).  What is NOT op-identical: the KD-tree walk
// itself (a different tree; it only selects which triangles are tested), the radiance
// sum (throughput form instead of the recursion's inside-out order, ~1e-7 relative)
// and libm-level functions in fp64 (acos/atan2/asin/pow: different implementations,
// equal after rounding to fp32 except on rare ties).
#include "hrt_device.h"

namespace hrtk {

#define HRT_EPS 1e-5f
#define HRT_FLT_MAX 3.402823466e+38f
// (double)t >= 1e-5  <=>  t > 1e-5f   and   (double)t < -1e-5  <=>  t < -1e-5f   for fp32 t,
// because (float)1e-5 < 1e-5 < nextafterf((float)1e-5, 1).
#define HRT_T_ACCEPT(t) ((t) > HRT_EPS)

// u8 -> float tables built on the host in double: [0,256) = c/255., [256,512) = c/127.5 - 1.
__constant__ float c_u8_lut[512];

struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 mk(const float4 &v) { return f3{v.x, v.y, v.z}; }
__device__ __forceinline__ f3 mk(const float *p) { return f3{p[0], p[1], p[2]}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // Vec3.h:48 order
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float length(f3 a) { return sqrtf(dot(a, a)); }  // == (float)sqrt((double)x)
__device__ __forceinline__ f3 normalize(f3 a) {  // Vec3.h:46: divide by the length, no guard
    float L = length(a);
    return mk(a.x / L, a.y / L, a.z / L);
}
__device__ __forceinline__ float comp(f3 a, uint32_t axis) { return axis == 0 ? a.x : (axis == 1 ? a.y : a.z); }

// ------------------------------------------------------------------ RNG stream
// Counter-based per-path stream (DESIGN.md "RNG stream"): draw i of path (seed, pixel, sample).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
struct Rng {
    uint32_t k0, k1, i;
    __device__ __forceinline__ void start(uint32_t seed_lo, uint32_t seed_hi, uint32_t pixel, uint32_t sample) {
        k0 = mix32(seed_lo ^ (pixel * 0x9E3779B1u + 0x7F4A7C15u));
        k1 = mix32(seed_hi + sample * 0x85EBCA77u + 0xC2B2AE3Du);
        i = 0;
    }
    __device__ __forceinline__ float next() {
        uint32_t x = k0 + (i++) * 0x9E3779B9u;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x ^= k1; x *= 0x846ca68bu; x ^= x >> 16;
        return (float)(x >> 8) * (1.0f / 16777216.0f);
    }
    __device__ __forceinline__ f3 unit_vector() {  // Functions.cpp:10-18: min + (max-min)*u, normalised cube sample
        const float a = -1.f + 2.f * next(), b = -1.f + 2.f * next(), c = -1.f + 2.f * next();
        return normalize(mk(a, b, c));
    }
};

struct Ray {
    f3 o, d;
    float time;
};

struct Hit {
    uint32_t kind;   // 0 none, 1 sphere, 2 square, 3 mesh (Scene.h:46)
    uint32_t index;  // sphere / square / mesh index
    float t;
    uint32_t tri;    // mesh: soup slot of the triangle
    float a0, a1;    // square: (u,v); mesh: barycentric (w1,w2)
};

// ------------------------------------------------------------------ primitives
// Sphere.h:91-132; near root only (the far root is unreachable, N6).  2.*x is exact in fp32.
__device__ __forceinline__ bool sphere_t(const float4 r0, const float4 r1, const Ray &ray, float &t) {
    const f3 c = mk(r0) + ray.time * mk(r1);
    const f3 oc = ray.o - c;
    const float a = dot(ray.d, ray.d);
    const float b = 2.f * dot(ray.d, oc);
    const float cc = dot(oc, oc) - r0.w * r0.w;
    const float delta = b * b - 4 * a * cc;
    if (delta < 0) return false;
    const float sq = sqrtf(delta);
    t = (-b - sq) / (2 * a);
    return !(t < -HRT_EPS);
}

// Square.h:65-126 with the per-quad constants (n, |R|, |U|, D0) folded on the host in the same arithmetic.
__device__ __forceinline__ bool quad_t(const float4 *__restrict__ q, const Ray &ray, float tmax, float &t, float &u,
                                       float &v) {
    const float4 q0 = q[0], q1 = q[1];
    const uint32_t flags = __float_as_uint(q1.w);
    const f3 n = mk(q1);
    const float dotRN = dot(ray.d, n);
    if (dotRN == 0.f) return false;
    if (dotRN > 0.f && !(flags & HRT_QUAD_FLAG_GLASS)) return false;
    f3 p0 = mk(q0);
    float D = q0.w;
    if (flags & HRT_QUAD_FLAG_MOVING) {
        p0 = p0 + ray.time * mk(q[4]);
        D = dot(p0, n);
    }
    t = (D - dot(ray.o, n)) / dotRN;
    if (!HRT_T_ACCEPT(t) || !(t < tmax)) return false;
    const float4 q2 = q[2], q3 = q[3];
    const f3 qq = (ray.o + t * ray.d) - p0;
    const float proj1 = dot(qq, mk(q2)) / q2.w;
    const float proj2 = dot(qq, mk(q3)) / q3.w;
    if (!((proj1 <= q2.w && proj1 >= 0.f) && (proj2 <= q3.w && proj2 >= 0.f))) return false;
    u = proj1 / q2.w;
    v = proj2 / q3.w;
    return true;
}

// AABB.h:48-65 exactly: reciprocal in double, products narrowed to float.
__device__ __forceinline__ bool aabb_gate(const float *lo, const float *hi, const Ray &ray) {
    float tmin = HRT_EPS, tmax = HRT_FLT_MAX;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float da = a == 0 ? ray.d.x : (a == 1 ? ray.d.y : ray.d.z);
        const float oa = a == 0 ? ray.o.x : (a == 1 ? ray.o.y : ray.o.z);
        const double adinv = 1.0 / (double)da;
        const float t0 = (float)((double)(lo[a] - oa) * adinv);
        const float t1 = (float)((double)(hi[a] - oa) * adinv);
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

// Nodelet fetch: the leading `lds_n` units of the kd array are resident in LDS.
__device__ __forceinline__ uint4 kd_fetch(const uint4 *__restrict__ g, const uint4 *s, uint32_t lds_n, uint32_t i) {
    return (i < lds_n) ? s[i] : g[i];
}

// Closest triangle of one mesh with t >= 0 (KDTree.cpp:31-85 semantics: the caller applies
// `t >= EPSILON && t < best`).  Stackless: locate the leaf that holds the entry point, test its
// triangles, leave through the exit face's rope, repeat while no hit lies inside the visited cells.
// Triangle rows: 0 {c0, id} 1 {e1, d00} 2 {e2, d01} 3 {n, D} 4 {d11, denom, -, -}.
__device__ __forceinline__ bool mesh_closest(const DScene &S, const uint4 *s_units, const DMesh &M, const Ray &ray,
                                             float &best_t, uint32_t &best_tri, float &bu, float &bv) {
    if (!aabb_gate(M.aabb_lo, M.aabb_hi, ray)) return false;  // KDTree.cpp:82
    const f3 inv = mk(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
    float t_entry = 0.f, t_scene_exit = HRT_FLT_MAX;
    {
        float t0 = (M.kd_lo[0] - ray.o.x) * inv.x, t1 = (M.kd_hi[0] - ray.o.x) * inv.x;
        t_entry = fmaxf(t_entry, fminf(t0, t1)); t_scene_exit = fminf(t_scene_exit, fmaxf(t0, t1));
        t0 = (M.kd_lo[1] - ray.o.y) * inv.y; t1 = (M.kd_hi[1] - ray.o.y) * inv.y;
        t_entry = fmaxf(t_entry, fminf(t0, t1)); t_scene_exit = fminf(t_scene_exit, fmaxf(t0, t1));
        t0 = (M.kd_lo[2] - ray.o.z) * inv.z; t1 = (M.kd_hi[2] - ray.o.z) * inv.z;
        t_entry = fmaxf(t_entry, fminf(t0, t1)); t_scene_exit = fminf(t_scene_exit, fmaxf(t0, t1));
    }
    if (!(t_entry <= t_scene_exit)) return false;
    best_t = HRT_FLT_MAX;
    bool found = false;
    uint32_t ref = M.root;
    const uint4 *__restrict__ g_units = S.kd_units;
    const float4 *__restrict__ tris = S.tris;
    for (int guard = 0; guard < 2048 && ref != HRT_KD_NIL; ++guard) {  // every wave reaches the bound
        const f3 p = ray.o + t_entry * ray.d;
        while (!(ref & HRT_KD_LEAF)) {
            const uint4 nd = kd_fetch(g_units, s_units, S.lds_units, ref);
            const float split = __uint_as_float(nd.x);
            const float pc = comp(p, nd.y), dc = comp(ray.d, nd.y);
            const bool left = (pc < split) || (pc == split && dc < 0.f);
            ref = left ? nd.z : nd.w;
        }
        const uint32_t lu = ref & ~HRT_KD_LEAF;
        const uint4 l0 = kd_fetch(g_units, s_units, S.lds_units, lu);
        const uint4 l1 = kd_fetch(g_units, s_units, S.lds_units, lu + 1);
        const uint32_t first = M.tri_base + l0.w, count = l1.w;
        for (uint32_t k = 0; k < count; ++k) {
            const float4 *__restrict__ tr = tris + HRT_TRI_ROWS * (first + k);
            const float4 r3 = tr[3];
            const f3 n = mk(r3);
            const float dotRN = dot(ray.d, n);
            if (!(dotRN < 0.f)) continue;                       // Triangle.h:80-91: parallel or back-facing (NaN: no hit)
            const float t = (r3.w - dot(ray.o, n)) / dotRN;     // :95
            if (t < 0.f || !(t < best_t)) continue;             // :96, then the leaf's strict `<` (KDTree.cpp:44)
            const float4 r0 = tr[0], r1 = tr[1], r2 = tr[2], r4 = tr[4];
            const f3 v2 = (ray.o + t * ray.d) - mk(r0);
            const float d20 = dot(v2, mk(r1)), d21 = dot(v2, mk(r2));
            const float u1 = (r4.x * d20 - r2.w * d21) / r4.y;  // Triangle.h:72-74
            const float u2 = (r1.w * d21 - r2.w * d20) / r4.y;
            const float u0 = 1 - u1 - u2;
            if (u0 >= 0 && u0 <= 1 && u1 >= 0 && u1 <= 1 && u2 >= 0 && u2 <= 1) {
                best_t = t; best_tri = first + k; bu = u1; bv = u2; found = true;
            }
        }
        // exit face of this cell
        const float ex = ((ray.d.x > 0.f ? __uint_as_float(l1.x) : __uint_as_float(l0.x)) - ray.o.x) * inv.x;
        const float ey = ((ray.d.y > 0.f ? __uint_as_float(l1.y) : __uint_as_float(l0.y)) - ray.o.y) * inv.y;
        const float ez = ((ray.d.z > 0.f ? __uint_as_float(l1.z) : __uint_as_float(l0.z)) - ray.o.z) * inv.z;
        float t_exit = HRT_FLT_MAX;
        uint32_t face = 6;
        if (ray.d.x != 0.f && ex < t_exit) { t_exit = ex; face = ray.d.x > 0.f ? 1u : 0u; }
        if (ray.d.y != 0.f && ey < t_exit) { t_exit = ey; face = ray.d.y > 0.f ? 3u : 2u; }
        if (ray.d.z != 0.f && ez < t_exit) { t_exit = ez; face = ray.d.z > 0.f ? 5u : 4u; }
        if (best_t <= t_exit || face == 6) break;
        t_entry = fmaxf(t_entry, t_exit);
        const uint4 rp = kd_fetch(g_units, s_units, S.lds_units, lu + 2 + (face >> 2));
        const uint32_t sel = face & 3u;
        ref = sel == 0 ? rp.x : (sel == 1 ? rp.y : (sel == 2 ? rp.z : rp.w));
    }
    return found;
}

// Scene::computeIntersection, Scene.h:202-230.
__device__ __forceinline__ Hit closest_hit(const DScene &S, const uint4 *s_units, const Ray &ray) {
    Hit h;
    h.kind = 0; h.index = 0; h.t = HRT_FLT_MAX; h.tri = 0; h.a0 = 0.f; h.a1 = 0.f;
    const float4 *__restrict__ sph = S.spheres;
    for (uint32_t i = 0; i < S.n_spheres; ++i) {
        float t;
        if (sphere_t(sph[2 * i], sph[2 * i + 1], ray, t) && t < h.t && HRT_T_ACCEPT(t)) { h.kind = 1; h.index = i; h.t = t; }
    }
    const float4 *__restrict__ qd = S.quads;
    for (uint32_t i = 0; i < S.n_quads; ++i) {
        float t, u, v;
        if (quad_t(qd + HRT_QUAD_ROWS * i, ray, h.t, t, u, v)) { h.kind = 2; h.index = i; h.t = t; h.a0 = u; h.a1 = v; }
    }
    for (uint32_t i = 0; i < S.n_meshes; ++i) {
        float t, u, v;
        uint32_t tri;
        if (mesh_closest(S, s_units, S.meshes[i], ray, t, tri, u, v) && t < h.t && HRT_T_ACCEPT(t)) {
            h.kind = 3; h.index = i; h.t = t; h.tri = tri; h.a0 = u; h.a1 = v;
        }
    }
    return h;
}

// Scene::computeShadow, Scene.h:235-255: candidates in object order, each lets the ray
// through with probability `transparency` (one draw per candidate).
__device__ __forceinline__ bool shadow_blocked(const DScene &S, const uint4 *s_units, const Ray &ray, float tmax, Rng &rng) {
    const float4 *__restrict__ sph = S.spheres;
    const float4 *__restrict__ mats = S.materials;
    for (uint32_t i = 0; i < S.n_spheres; ++i) {
        float t;
        const float4 r1 = sph[2 * i + 1];
        if (sphere_t(sph[2 * i], r1, ray, t) && t < tmax && HRT_T_ACCEPT(t)) {
            const float transparency = mats[HRT_MAT_ROWS * __float_as_uint(r1.w)].w;
            if (rng.next() > transparency) return true;
        }
    }
    const float4 *__restrict__ qd = S.quads;
    for (uint32_t i = 0; i < S.n_quads; ++i) {
        float t, u, v;
        if (quad_t(qd + HRT_QUAD_ROWS * i, ray, tmax, t, u, v)) {
            const float transparency = mats[HRT_MAT_ROWS * __float_as_uint(qd[HRT_QUAD_ROWS * i + 4].w)].w;
            if (rng.next() > transparency) return true;
        }
    }
    for (uint32_t i = 0; i < S.n_meshes; ++i) {
        float t, u, v;
        uint32_t tri;
        if (mesh_closest(S, s_units, S.meshes[i], ray, t, tri, u, v) && t < tmax && HRT_T_ACCEPT(t)) {
            const float transparency = mats[HRT_MAT_ROWS * S.meshes[i].material].w;
            if (rng.next() > transparency) return true;
        }
    }
    return false;
}

// ------------------------------------------------------------------ materials
// rows: 0 {albedo.xyz, transparency} 1 {index_medium, type, texture_type, emissive}
//       2 {checker1.xyz, scale_x} 3 {checker2.xyz, scale_y} 4 {light_color.xyz, intensity}
//       5 {image, normal_map, -, -}
__device__ __forceinline__ uint32_t texel(const DScene &S, int img, float u, float v, float sx, float sy) {
    const DImage im = S.images[img];
    float uu = u * sx, vv = v * sy;
    uu = uu - truncf(uu);            // (float)fmod((double)(u*sx), 1.): exact
    vv = 1.f - (vv - truncf(vv));    // (float)(1 - fmod(...)): one correctly rounded subtraction either way
    const int x = (int)(uu * (float)(im.w - 1));
    const int y = (int)(vv * (float)(im.h - 1));
    return S.texels[im.offset + (uint32_t)(y * im.w + x)];
}
__device__ __forceinline__ f3 unit_rgb(uint32_t px) {  // c/255. in double, narrowed (Material.cpp:87)
    return mk(c_u8_lut[px & 255u], c_u8_lut[(px >> 8) & 255u], c_u8_lut[(px >> 16) & 255u]);
}

// Material::texture, Material.cpp:63-92
__device__ __forceinline__ f3 mat_texture(const DScene &S, const float4 *__restrict__ m, uint32_t tex_type, f3 color, float u, float v) {
    if (tex_type == 1u) {
        const float4 c1 = m[2], c2 = m[3];
        color = ((int)(u * c1.w) % 2 == (int)(v * c2.w) % 2) ? mk(c1) : mk(c2);
    } else if (tex_type == 2u) {
        const int img = (int)__float_as_uint(m[5].x);
        if (img < 0 || S.images[img].w < 1 || S.images[img].h < 1) {
            color = ((int)((double)u * 8.) % 2 == (int)((double)v * 8.) % 2) ? mk(0.f, 0.f, 0.f) : mk(1.f, 0.f, 1.f);
        } else {
            color = unit_rgb(texel(S, img, u, v, m[2].w, m[3].w));
        }
    }
    return color;
}

// Material::emit, Material.cpp:13-24
__device__ __forceinline__ f3 mat_emit(const DScene &S, const float4 *__restrict__ m, uint32_t tex_type, bool emissive, float u, float v) {
    if (!emissive) return mk(0.f, 0.f, 0.f);
    const float4 lc = m[4];
    f3 c = mk(0.f, 0.f, 0.f);
    if (tex_type == 0u) c = mk(lc);
    else c = mat_texture(S, m, tex_type, c, u, v);
    return c * lc.w;
}

struct Surface {
    f3 p, n, albedo, emission;
    float transparency, index_medium;
    uint32_t type;
};

// The hit-dependent part of Scene::rayTraceRecursive, Scene.h:270-300.
__device__ __forceinline__ Surface shade(const DScene &S, const Ray &ray, const Hit &h) {
    Surface sf;
    const float4 *__restrict__ mats = S.materials;
    const f3 p = ray.o + h.t * ray.d;
    sf.p = p;
    sf.emission = mk(0.f, 0.f, 0.f);
    uint32_t mat_id;
    if (h.kind == 1u) {
        const float4 r0 = S.spheres[2 * h.index], r1 = S.spheres[2 * h.index + 1];
        mat_id = __float_as_uint(r1.w);
        const float4 *m = mats + HRT_MAT_ROWS * mat_id;
        const float4 m0 = m[0], m1 = m[1];
        const uint32_t tex_type = __float_as_uint(m1.z);
        const bool emissive = __float_as_uint(m1.w) != 0u;
        const f3 c = mk(r0) + ray.time * mk(r1);
        sf.n = normalize(p - c);
        sf.albedo = mk(m0);
        if (tex_type != 0u || emissive) {  // Sphere.h:129-130, Scene.h:275-277 (fp64 libm as the reference)
            const float theta = (float)acos((double)sf.n.y * -1.);
            const float phi = (float)(atan2((double)sf.n.z * -1., (double)sf.n.x) + 3.14159265358979323846);
            const float u = (float)((double)phi / (2 * 3.14159265358979323846));
            const float v = (float)((double)theta / 3.14159265358979323846);
            sf.albedo = mat_texture(S, m, tex_type, sf.albedo, u, v);
            sf.emission = mat_emit(S, m, tex_type, emissive, u, v);
        }
    } else if (h.kind == 2u) {
        const float4 *q = S.quads + HRT_QUAD_ROWS * h.index;
        mat_id = __float_as_uint(q[4].w);
        const float4 *m = mats + HRT_MAT_ROWS * mat_id;
        const float4 m0 = m[0], m1 = m[1];
        const uint32_t tex_type = __float_as_uint(m1.z);
        const bool emissive = __float_as_uint(m1.w) != 0u;
        sf.n = mk(q[1]);
        sf.albedo = mat_texture(S, m, tex_type, mk(m0), h.a0, h.a1);
        const int nmap = (int)__float_as_uint(m[5].y);
        if (nmap >= 0) {  // Material::get_normal, Material.cpp:114-130
            const uint32_t px = texel(S, nmap, h.a0, h.a1, m[2].w, m[3].w);
            const float nx = c_u8_lut[256u + (px & 255u)], ny = c_u8_lut[256u + ((px >> 8) & 255u)],
                        nz = c_u8_lut[256u + ((px >> 16) & 255u)];
            sf.n = normalize(nx * mk(q[5]) + ny * mk(q[6]) + nz * sf.n);
        }
        sf.emission = mat_emit(S, m, tex_type, emissive, h.a0, h.a1);
    } else {
        const DMesh &M = S.meshes[h.index];
        mat_id = M.material;
        const float4 m0 = mats[HRT_MAT_ROWS * mat_id];
        const float4 r0 = S.tris[HRT_TRI_ROWS * h.tri], r3 = S.tris[HRT_TRI_ROWS * h.tri + 3];
        sf.n = mk(r3);  // Triangle.h:32-37 flat normal, folded on the host
        sf.albedo = mk(m0);
        const uint32_t tid = __float_as_uint(r0.w);
        if (M.color_type == 1) {
            sf.albedo = mk(S.colors[M.color_base + tid]);
        } else if (M.color_type == 0) {
            const uint4 vi = S.tri_vids[M.color_base + tid];
            const float w1 = h.a0, w2 = h.a1, w0 = 1 - w1 - w2;
            sf.albedo = w0 * mk(S.colors[M.vcolor_base + vi.x]) + w1 * mk(S.colors[M.vcolor_base + vi.y]) +
                        w2 * mk(S.colors[M.vcolor_base + vi.z]);
        }
    }
    const float4 *m = mats + HRT_MAT_ROWS * mat_id;
    const float4 m0 = m[0], m1 = m[1];
    sf.transparency = m0.w;
    sf.index_medium = m1.x;
    sf.type = __float_as_uint(m1.y);
    return sf;
}

// Functions.cpp:38-54
__device__ __forceinline__ f3 reflect(f3 d, f3 n) { return d - (2 * dot(d, n)) * n; }
__device__ __forceinline__ f3 refract(f3 d, f3 n, float eta) {
    const float cos_theta = fminf(dot(d, n), 1.0f);
    const f3 perp = eta * (d + cos_theta * n);
    const f3 par = (float)(-sqrt(fabs(1.0 - (double)dot(perp, perp)))) * n;
    return perp + par;
}
__device__ __forceinline__ float reflectance(float cosine, float ref_idx) {
    float r0 = (1 - ref_idx) / (1 + ref_idx);
    r0 = r0 * r0;
    const double m = (double)(1 - cosine);
    const double m2 = m * m;
    return (float)((double)r0 + (double)(1 - r0) * (m2 * m2 * m));  // pow(x, 5) in double
}

// Material::scatter, Material.cpp:26-60
__device__ __forceinline__ void scatter(const Surface &sf, Ray &ray, Rng &rng) {
    f3 dir;
    if (sf.type == 1u) {  // glass (the reference's inverted convention, N7)
        const float ri = (dot(ray.d, sf.n) > 0) ? (float)(1. / (double)sf.index_medium) : sf.index_medium;
        const float cos_theta = fminf(dot(ray.d * -1.f, sf.n), 1.0f);
        const float sin_theta = (float)sqrt(1. - (double)(cos_theta * cos_theta));
        const bool cannot_refract = ((double)(ri * sin_theta) - 0.6) > 1.0;
        if (cannot_refract || reflectance(cos_theta, ri) > rng.next()) dir = reflect(ray.d, sf.n);
        else dir = refract(ray.d, sf.n, ri);
    } else if (sf.type == 0u) {  // diffuse
        dir = sf.n + rng.unit_vector();
        if (!(length(dir) > HRT_EPS)) dir = sf.n;  // (double)len <= 1e-5
    } else {  // mirror
        dir = reflect(ray.d, sf.n);
    }
    dir = normalize(dir);
    ray.o = sf.p + HRT_EPS * dir;
    ray.d = normalize(dir);  // Ray's constructor normalises again (Line.h:15)
}

// Scene::skyboxTexture, Scene.h:149-161
__device__ __forceinline__ f3 sky(const DScene &S, f3 dir, int remaining) {
    if (S.skybox_image < 0) {
        if (S.dark_sky) return mk(0.f, 0.f, 0.f);
        const float a = (float)(0.5 * ((double)dir.y + 1.0));
        return (float)(1.0 - (double)a) * mk(1.f, 1.f, 1.f) + (a * mk(0.5f, 0.7f, 1.0f)) * (float)(remaining + 1);
    }
    const DImage im = S.images[S.skybox_image];
    const float u = (float)(0.5 + atan2((double)dir.z, (double)dir.x) / (2 * 3.14159265358979323846));
    const float v = (float)(0.5 - asin((double)dir.y) / 3.14159265358979323846);
    int x = (int)(u * (float)im.w), y = (int)(v * (float)im.h);
    x = min(x, im.w - 1);  // the reference reads out of bounds at u == 1; clamped (as the oracle)
    y = min(y, im.h - 1);
    return unit_rgb(S.texels[im.offset + (uint32_t)(y * im.w + x)]) * (float)remaining;
}

// Direct light with soft shadows, Scene.h:305-334.
__device__ __forceinline__ f3 direct_light(const DScene &S, const uint4 *s_units, const Surface &sf, const Ray &ray, Rng &rng) {
    f3 color = mk(0.f, 0.f, 0.f);
    const float4 *__restrict__ L = S.lights;
    for (uint32_t i = 0; i < S.n_lights; ++i) {
        const float4 l0 = L[2 * i];
        const f3 lpos = mk(l0);
        const f3 Ld = normalize(lpos - sf.p);
        const float dotLN = dot(Ld, sf.n);
        color = color + ((mk(L[1]) * sf.albedo) * fmaxf(0.0f, dotLN)) * (float)(1. - (double)sf.transparency);  // lights[0] (N2)
        int blocked = 0;
        const float delta = l0.w / 2.f;
        for (int j = 0; j < 10; ++j) {  // NB_ECH
            const f3 lp = lpos + rng.unit_vector() * delta;
            const f3 to = lp - sf.p;
            const f3 Ls = normalize(to);
            const float tLight = length(to);
            Ray sr;
            sr.o = sf.p + Ls * HRT_EPS;
            sr.d = normalize(Ls);
            sr.time = ray.time;
            if (shadow_blocked(S, s_units, sr, tLight, rng)) blocked++;
        }
        const float shadow = (float)(1. - (double)((float)blocked / 10.f));
        color = color * shadow;  // the running sum, earlier lights included (N3)
    }
    return color;
}

// matrixUtilities.h:53-74 with the two inverse matrices supplied by the host (fp64, column-major).
__device__ __forceinline__ void mult4(const double *__restrict__ m, double x, double y, double z, double w, double *r) {
    r[0] = m[0] * x + m[4] * y + m[8] * z + m[12] * w;
    r[1] = m[1] * x + m[5] * y + m[9] * z + m[13] * w;
    r[2] = m[2] * x + m[6] * y + m[10] * z + m[14] * w;
    r[3] = m[3] * x + m[7] * y + m[11] * z + m[15] * w;
}
__device__ __forceinline__ Ray camera_ray(const DCamera &C, float u, float v, float time) {
    double ri[4], r[4];
    mult4(C.p_inv, 2.0 * (double)u - 1.0, -(2.0 * (double)v - 1.0), 0.0, 1.0, ri);
    mult4(C.mv_inv, ri[0], ri[1], ri[2], ri[3], r);
    const f3 world = mk((float)(r[0] / r[3]), (float)(r[1] / r[3]), (float)(r[2] / r[3]));
    Ray out;
    out.o = mk(C.eye);
    // normalised twice, as the reference does: once in screen_space_to_world_space_ray
    // (matrixUtilities.h:73) and again by the Ray constructor (main.cpp:192, Line.h:15)
    out.d = normalize(normalize(world - out.o));
    out.time = time;
    return out;
}

}  // namespace hrtk

using namespace hrtk;

// ---------------------------------------------------------------------------
// The megakernel.  256 threads = 4 waves per workgroup; grid = resident workgroups only.
// ---------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(256) hrt_trace_kernel(const DRender R) {
    extern __shared__ uint4 s_units[];
    const DScene &S = R.scene;
    for (uint32_t i = threadIdx.x; i < S.lds_units; i += blockDim.x) s_units[i] = S.kd_units[i];
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    for (;;) {
        uint32_t j = 0;
        if (lane == 0) j = atomicAdd(R.tile_counter, 1u);
        j = __builtin_amdgcn_readfirstlane(j);
        if (j >= R.tiles_owned) break;  // the queue is finite: every wave gets here
        const uint32_t tile = R.rank + j * R.world;
        const uint32_t px = (tile % R.tiles_x) * 8u + (lane & 7u);
        const uint32_t py = (tile / R.tiles_x) * 8u + (lane >> 3);
        const bool inside = px < R.w && py < R.h;
        const uint32_t pixel = py * R.w + px;

        f3 sum = mk(0.f, 0.f, 0.f);
        uint32_t s = 0;          // next sample of this lane's pixel
        int remaining = 0;       // bounces left on the current path; 0 = needs a new path
        Ray ray;
        ray.o = mk(0.f, 0.f, 0.f); ray.d = mk(0.f, 0.f, 1.f); ray.time = 0.f;
        f3 thr = mk(1.f, 1.f, 1.f), rad = mk(0.f, 0.f, 0.f);
        Rng rng;
        rng.k0 = rng.k1 = rng.i = 0;
        bool live = inside && R.spp > 0;

        while (__ballot(live) != 0ull) {
            if (live) {
                if (remaining == 0) {  // regenerate: next camera sample of this pixel (main.cpp:188-192)
                    rng.start(R.seed_lo, R.seed_hi, pixel, s);
                    const float u = ((float)px + rng.next()) / (float)R.w;
                    const float v = ((float)py + rng.next()) / (float)R.h;
                    const float tm = rng.next();
                    ray = camera_ray(R.cam, u, v, tm);
                    thr = mk(1.f, 1.f, 1.f);
                    rad = mk(0.f, 0.f, 0.f);
                    remaining = 6;  // MAXBOUNCES
                }
                const Hit h = closest_hit(S, s_units, ray);
                bool ended;
                if (h.kind == 0u) {
                    rad = rad + thr * sky(S, ray.d, remaining);
                    ended = true;
                } else {
                    const Surface sf = shade(S, ray, h);
                    f3 direct = mk(0.f, 0.f, 0.f);
                    if (S.n_lights) direct = direct_light(S, s_units, sf, ray, rng);
                    rad = rad + thr * (direct + sf.emission);
                    thr = thr * sf.albedo;
                    scatter(sf, ray, rng);
                    --remaining;
                    ended = (remaining == 0);
                }
                if (ended) {
                    sum = sum + mk(rad.x / 6.f, rad.y / 6.f, rad.z / 6.f);  // Scene.h:348
                    remaining = 0;
                    ++s;
                    live = s < R.spp;
                }
            }
        }
        float *o = R.out_tiles + ((size_t)j * 64u + lane) * 3u;
        f3 c = mk(0.f, 0.f, 0.f);
        if (inside) {
            const float nspp = (float)R.spp;
            c = mk(sum.x / nspp, sum.y / nspp, sum.z / nspp);  // main.cpp:195
            if (R.flags & 1u)  // gamma_correct, Functions.cpp:56-60: pow in double
                c = mk((float)pow((double)c.x, 1.0 / 2.2), (float)pow((double)c.y, 1.0 / 2.2), (float)pow((double)c.z, 1.0 / 2.2));
        }
        o[0] = c.x; o[1] = c.y; o[2] = c.z;
    }
}

// Tile-major per-rank blocks -> row-major frame (rank 0, after the gather).
extern "C" __global__ void hrt_assemble_kernel(const float *__restrict__ gathered, uint32_t tiles_per_rank_padded,
                                               uint32_t w, uint32_t h, uint32_t world, float *__restrict__ frame) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= w * h) return;
    const uint32_t x = idx % w, y = idx / w;
    const uint32_t tiles_x = (w + 7u) / 8u;
    const uint32_t tile = (y / 8u) * tiles_x + (x / 8u);
    const uint32_t rank = tile % world, slot = tile / world;
    const uint32_t lane = (y & 7u) * 8u + (x & 7u);
    const float *src = gathered + (((size_t)rank * tiles_per_rank_padded + slot) * 64u + lane) * 3u;
    float *dst = frame + (size_t)idx * 3u;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

// Deterministic first-hit AOVs through pixel centres (no RNG, time 0); parity instrument.
// which: 0 hit (t, kind, index), 1 shading normal, 2 albedo, 3 emission.
extern "C" __global__ void hrt_aov_kernel(const DRender R, uint32_t which, float *__restrict__ out) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R.w * R.h) return;
    const uint32_t x = idx % R.w, y = idx / R.w;
    DScene S = R.scene;
    S.lds_units = 0;  // every nodelet from global memory here
    const Ray ray = camera_ray(R.cam, ((float)x + 0.5f) / (float)R.w, ((float)y + 0.5f) / (float)R.h, 0.f);
    const Hit h = closest_hit(S, nullptr, ray);
    f3 o = mk(0.f, 0.f, 0.f);
    if (which == 0u) {
        float id = -1.f;
        if (h.kind == 3u) id = (float)__float_as_uint(S.tris[HRT_TRI_ROWS * h.tri].w);
        else if (h.kind) id = (float)h.index;
        o = mk(h.kind ? h.t : 0.f, (float)h.kind, id);
    } else if (h.kind) {
        const Surface sf = shade(S, ray, h);
        o = which == 1u ? sf.n : (which == 2u ? sf.albedo : sf.emission);
    }
    out[3 * (size_t)idx] = o.x; out[3 * (size_t)idx + 1] = o.y; out[3 * (size_t)idx + 2] = o.z;
}

// The path stream on the device, for the RNG parity test: out[i] = draw i of (seed, pixel, sample).
extern "C" __global__ void hrt_stream_kernel(uint32_t seed_lo, uint32_t seed_hi, uint32_t pixel, uint32_t sample,
                                             uint32_t n, float *__restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Rng rng;
    rng.start(seed_lo, seed_hi, pixel, sample);
    for (uint32_t i = 0; i < n; ++i) out[i] = rng.next();
}
