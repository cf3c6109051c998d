// ============================================================================
// ref_parts.cpp -- TEST INFRASTRUCTURE.  C entry points over the parts of the
// reference that compile from their own sources with g++ alone:
//     src/Vec3.h  src/Ray.h  src/Line.h  src/AABB.h  src/Triangle.h
//     src/Functions.{h,cpp}  src/imageLoader.{h,cpp}
//     src/matrixUtilities.h  (needs <GL/gl.h> + libGL, both in the image; no GL call is made at run time because
//                             the *Updated flags stay false, so no context is needed)
// The sources are compiled WHERE THEY LIE under /root/reference (never copied);
// the output goes to oracle/_ref/ (git-ignored).  Sphere.h, Square.h, Mesh.*,
// Material.*, KDTree.*, Scene.h and main.cpp include <GL/glut.h>, which this
// image does not have, so they are unbuildable here (DESIGN.md "Oracle").
// Only tests/ and the golden-vector generator load the result.
// ============================================================================
#include <cstdint>
#include <ctime>

#include "src/Vec3.h"
#include "src/Ray.h"
#include "src/AABB.h"
#include "src/Triangle.h"
#include "src/Functions.h"
#include "src/imageLoader.h"
#include "src/matrixUtilities.h"

// random_float() seeds its static mt19937 with time(nullptr) (Functions.cpp:6).
// Interposing time() fixes that seed without touching the reference source.
static time_t g_fixed_time = 12345;
extern "C" time_t time(time_t *t) {
    if (t) *t = g_fixed_time;
    return g_fixed_time;
}

extern "C" {

uint32_t ref_fixed_seed(void) { return (uint32_t)g_fixed_time; }

// rays: n x 7 = origin, direction (normalised by Ray's constructor), time
void ref_kat_triangle(const float *tri, const float *rays, uint32_t n, float *out) {
    Triangle T(Vec3(tri[0], tri[1], tri[2]), Vec3(tri[3], tri[4], tri[5]), Vec3(tri[6], tri[7], tri[8]));
    for (uint32_t i = 0; i < n; ++i) {
        const float *r = rays + 7 * (size_t)i;
        Ray ray(Vec3(r[0], r[1], r[2]), Vec3(r[3], r[4], r[5]), r[6]);
        RayTriangleIntersection h = T.getIntersection(ray);
        float *o = out + 8 * (size_t)i;
        o[0] = h.intersectionExists ? 1.f : 0.f;
        o[1] = h.intersectionExists ? h.t : 0.f;
        o[2] = h.intersectionExists ? h.w0 : 0.f;
        o[3] = h.intersectionExists ? h.w1 : 0.f;
        o[4] = h.intersectionExists ? h.w2 : 0.f;
        o[5] = h.intersectionExists ? h.normal[0] : 0.f;
        o[6] = h.intersectionExists ? h.normal[1] : 0.f;
        o[7] = h.intersectionExists ? h.normal[2] : 0.f;
    }
}

void ref_kat_aabb(const float *box, const float *rays, uint32_t n, float *out) {
    AABB b(Vec3(box[0], box[1], box[2]), Vec3(box[3], box[4], box[5]));
    for (uint32_t i = 0; i < n; ++i) {
        const float *r = rays + 7 * (size_t)i;
        Ray ray(Vec3(r[0], r[1], r[2]), Vec3(r[3], r[4], r[5]), r[6]);
        out[i] = b.intersects(ray) ? 1.f : 0.f;
    }
}

// in n x 8 = d(3), n(3), eta, cosine ; out n x 8 = reflect(3), refract(3), reflectance, gamma(|cosine|)
void ref_kat_optics(const float *in, uint32_t n, float *out) {
    for (uint32_t i = 0; i < n; ++i) {
        const float *a = in + 8 * (size_t)i;
        Vec3 d(a[0], a[1], a[2]), nn(a[3], a[4], a[5]);
        Vec3 rf = reflect(d, nn);
        Vec3 rr = refract(d, nn, a[6]);
        float *o = out + 8 * (size_t)i;
        o[0] = rf[0]; o[1] = rf[1]; o[2] = rf[2]; o[3] = rr[0]; o[4] = rr[1]; o[5] = rr[2];
        o[6] = reflectance(a[7], a[6]);
        Vec3 g(fabs(a[7]), 0.f, 0.f);
        gamma_correct(g);
        o[7] = g[0];
    }
}

// out n x 4 = random_float(), random_unit_vector(); first call fixes the seed through time()
void ref_kat_random(uint32_t n, float *out) {
    for (uint32_t i = 0; i < n; ++i) {
        float *o = out + 4 * (size_t)i;
        o[0] = random_float();
        Vec3 p = random_unit_vector();
        o[1] = p[0]; o[2] = p[1]; o[3] = p[2];
    }
}

// Ray constructor normalisation: in n x 3 -> out n x 3
void ref_kat_normalize(const float *in, uint32_t n, float *out) {
    for (uint32_t i = 0; i < n; ++i) {
        Ray r(Vec3(0, 0, 0), Vec3(in[3 * i], in[3 * i + 1], in[3 * i + 2]), 0.f);
        out[3 * i] = r.direction()[0]; out[3 * i + 1] = r.direction()[1]; out[3 * i + 2] = r.direction()[2];
    }
}

// Camera rays of main.cpp:189-192 from given GL matrices (column-major doubles, what glGetDoublev would have
// returned): the reference's own gluInvertMatrix (matrixUtilities.h:77-206), screen_space_to_world_space_ray
// (:53-74) and the Ray constructor's second normalisation (Line.h:13-16).  GL_DEPTH_RANGE reads {0, 1} by default.
// uv: n x 2 -> out: n x 6 (Ray origin, Ray direction).  Returns 0 when a matrix is singular.
int ref_camera_rays(const double modelview[16], const double projection[16], const float *uv, uint32_t n, float *out) {
    MatrixUtilities mu;  // the constructor's updateMatrices() does nothing: all three flags are false
    for (int k = 0; k < 16; ++k) { mu.modelview[k] = modelview[k]; mu.projection[k] = projection[k]; }
    if (!gluInvertMatrix(mu.modelview, mu.modelviewInverse)) return 0;   // what updateMatrices() does after the read-back (:36, :42)
    if (!gluInvertMatrix(mu.projection, mu.projectionInverse)) return 0;
    mu.nearAndFarPlanes[0] = 0.0;
    mu.nearAndFarPlanes[1] = 1.0;
    for (uint32_t i = 0; i < n; ++i) {
        Vec3 pos, dir;
        mu.screen_space_to_world_space_ray(uv[2 * i], uv[2 * i + 1], pos, dir);
        Ray ray(pos, dir, 0.f);
        float *o = out + 6 * (size_t)i;
        o[0] = ray.origin()[0]; o[1] = ray.origin()[1]; o[2] = ray.origin()[2];
        o[3] = ray.direction()[0]; o[4] = ray.direction()[1]; o[5] = ray.direction()[2];
    }
    return 1;
}
// the two inverses themselves: out 32 doubles
int ref_camera_inverses(const double modelview[16], const double projection[16], double *out) {
    return (gluInvertMatrix(modelview, out) && gluInvertMatrix(projection, out + 16)) ? 1 : 0;
}

// PPM loader: returns w*h (0 on failure) and a checksum of the bytes
uint64_t ref_ppm_info(const char *path, int32_t *w, int32_t *h) {
    ppmLoader::ImageRGB img;
    img.w = 0; img.h = 0;
    ppmLoader::load_ppm(img, path);
    *w = img.w; *h = img.h;
    uint64_t sum = 1469598103934665603ull;
    for (size_t i = 0; i < img.data.size(); ++i) {
        sum = (sum ^ img.data[i].r) * 1099511628211ull;
        sum = (sum ^ img.data[i].g) * 1099511628211ull;
        sum = (sum ^ img.data[i].b) * 1099511628211ull;
    }
    return sum;
}

}  // extern "C"
