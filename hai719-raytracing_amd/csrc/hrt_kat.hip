// Known-answer instrument (hrt_debug_kat, include/hrt.h): the DEVICE functions of the trace path on caller vectors,
// one lane per vector.  Nothing here is on the render path; the kernels only call the functions the trace kernels
// call (camera_ray, tri_test, aabb_gate_exact / mesh_gate_box, sphere_t, quad_t, reflect / refract / reflectance,
// normalize), so a test can compare the device arithmetic bit for bit with vectors produced by the reference's own code
// (tests/golden/ref_kat.npz) instead of inferring it from rendered pixels.
#include "hrt_device.h"

namespace hrtk {

// rays: n x 7 = origin, direction, time.  The Ray constructor normalises the direction (Line.h:13-16).
__device__ __forceinline__ Ray kat_ray(const float *__restrict__ rays, uint32_t i) {
    const float *r = rays + 7 * (size_t)i;
    Ray ray;
    ray.o = mk(r[0], r[1], r[2]);
    ray.d = normalize(mk(r[3], r[4], r[5]));
    ray.time = r[6];
    return ray;
}

}  // namespace hrtk

// out n x 12: {origin, direction} of the shipped camera_ray, then of the exact-division build
extern "C" __global__ void hrt_kat_camera_kernel(const DCamera *cam, const float *__restrict__ uv, uint32_t n, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Ray a = camera_ray<false>((ccam)cam, uv[2 * i], uv[2 * i + 1], 0.f);
    const Ray b = camera_ray<true>((ccam)cam, uv[2 * i], uv[2 * i + 1], 0.f);
    float *o = out + 12 * (size_t)i;
    o[0] = a.o.x; o[1] = a.o.y; o[2] = a.o.z; o[3] = a.d.x; o[4] = a.d.y; o[5] = a.d.z;
    o[6] = b.o.x; o[7] = b.o.y; o[8] = b.o.z; o[9] = b.d.x; o[10] = b.d.y; o[11] = b.d.z;
}

// rows: the folded rows of one triangle, plane {n, D} first, then the HRT_TRI_ROWS others.  out n x 8: hit, t, w0, w1, w2, normal (Triangle.h:110-118)
extern "C" __global__ void hrt_kat_triangle_kernel(const float4 *rows, const float *__restrict__ rays, uint32_t n, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Ray ray = kat_ray(rays, i);
    float t = HRT_FLT_MAX, u1 = 0.f, u2 = 0.f;
    const float4 r3 = ld((gf4)rows, 0);
    const bool hit = tri_test_plane((gf4)rows + 1, r3, ray, t, u1, u2);
    float *o = out + 8 * (size_t)i;
    o[0] = hit ? 1.f : 0.f; o[1] = hit ? t : 0.f;
    o[2] = hit ? 1 - u1 - u2 : 0.f; o[3] = hit ? u1 : 0.f; o[4] = hit ? u2 : 0.f;
    o[5] = hit ? r3.x : 0.f; o[6] = hit ? r3.y : 0.f; o[7] = hit ? r3.z : 0.f;
}

// box: lo(3), hi(3).  out n x 2: AABB::intersects in its fp64 form, and the shipped gate (fp32 filter in front of it)
extern "C" __global__ void hrt_kat_aabb_kernel(const float *__restrict__ box, const float *__restrict__ rays, uint32_t n, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Ray ray = kat_ray(rays, i);
    GateBox b;
    for (int a = 0; a < 3; ++a) { b.l[a] = box[a]; b.h[a] = box[3 + a]; }
    out[2 * i] = aabb_gate_exact(b, ray) ? 1.f : 0.f;
    out[2 * i + 1] = mesh_gate_box<false>(b, ray, ray_inv<false>(ray)) ? 1.f : 0.f;
}

// rows: the 2 rows of one sphere.  out n x 9: hit, t, theta, phi, normal, p.x, p.y (Sphere.h:91-132; the layout of oracle_kat_sphere)
extern "C" __global__ void hrt_kat_sphere_kernel(const float4 *rows, const float *__restrict__ rays, uint32_t n, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Ray ray = kat_ray(rays, i);
    const float4 r0 = ld((gf4)rows, 0), r1 = ld((gf4)rows, 1);
    float t = 0.f;
    const bool hit = sphere_t(r0, r1, ray, t);
    float *o = out + 9 * (size_t)i;
    for (int k = 0; k < 9; ++k) o[k] = 0.f;
    if (!hit) return;
    const f3 p = ray.o + t * ray.d;
    const f3 nn = normalize(p - (mk(r0) + ray.time * mk(r1)));
    float theta, phi;
    sphere_angles(nn, theta, phi);
    o[0] = 1.f; o[1] = t; o[2] = theta; o[3] = phi; o[4] = nn.x; o[5] = nn.y; o[6] = nn.z; o[7] = p.x; o[8] = p.y;
}

// rows: the 11 rows of one square.  out n x 8: hit, t, u, v, normal (Square.h:65-126; oracle_kat_quad), then whether the
// shipped no-division filter lets the square through (it must whenever hit is set)
extern "C" __global__ void hrt_kat_quad_kernel(const float4 *rows, const DScene *filter_scene, const float *__restrict__ rays, uint32_t n, float err_abs,
                                               float *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Ray ray = kat_ray(rays, i);
    float t = 0.f, u = 0.f, v = 0.f;
    const bool hit = hrtk::quad_t((gf4)rows, ray, HRT_FLT_MAX, t, u, v);
    const float4 q1 = ld((gf4)rows, 1);
    Ctx cx;
    cx.S = (cscene)filter_scene; cx.tq = cx.tm = cx.ts = cx.texc = (gf4) nullptr; cx.tmesh = (gmesh) nullptr; cx.lut = (gf1) nullptr; cx.lds = (lu4) nullptr; cx.lds_n = 0; cx.err_abs = err_abs; cx.flags = 0; cx.st = nullptr;
    const uint32_t cand = quad_filter<uint32_t>(cx, ray, HRT_FLT_MAX);  // filter_scene: a DScene whose only content is this square's filter rows
    float *o = out + 8 * (size_t)i;
    o[0] = hit ? 1.f : 0.f; o[1] = hit ? t : 0.f; o[2] = hit ? u : 0.f; o[3] = hit ? v : 0.f;
    o[4] = hit ? q1.x : 0.f; o[5] = hit ? q1.y : 0.f; o[6] = hit ? q1.z : 0.f;
    o[7] = (cand & 1u) ? 1.f : 0.f;
}

// in n x 8 = d(3), n(3), eta, cosine; out n x 8 = reflect(3), refract(3), reflectance, gamma(|cosine|)  (Functions.cpp:38-60)
extern "C" __global__ void hrt_kat_optics_kernel(const float *__restrict__ in, uint32_t n, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *a = in + 8 * (size_t)i;
    const f3 d = mk(a[0], a[1], a[2]), nn = mk(a[3], a[4], a[5]);
    const f3 rf = reflect(d, nn), rr = refract(d, nn, a[6]);
    float *o = out + 8 * (size_t)i;
    o[0] = rf.x; o[1] = rf.y; o[2] = rf.z; o[3] = rr.x; o[4] = rr.y; o[5] = rr.z;
    o[6] = reflectance(a[7], a[6]);
    o[7] = (float)pow((double)fabsf(a[7]), 1.0 / 2.2);  // the expression of hrt_gamma_kernel / hrt_finalize_kernel
}

// in n x 3 -> out n x 3: Vec3::normalize as the Ray constructor applies it (Vec3.h:46, Line.h:13-16)
extern "C" __global__ void hrt_kat_normalize_kernel(const float *__restrict__ in, uint32_t n, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 v = normalize(mk(in[3 * i], in[3 * i + 1], in[3 * i + 2]));
    out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z;
}
