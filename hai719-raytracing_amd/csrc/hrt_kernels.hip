// Hand-written HIP kernels of the trace path for gfx950 (CDNA4, wave64): the exact-arithmetic device functions every
// kernel form shares (intersectors, KD walk, shading, scattering, camera) and the lane-per-pixel kernel.
// hrt_stream.hip (workgroup-streaming kernel, the default for scenes with meshes or lights) and hrt_dual.hip (two
// pixel streams per lane) schedule the same functions differently.
//
// hrt_trace_kernel is the persistent-wavefront megakernel: every wave pulls 8x8-pixel tiles from a
// work-queue head, lane = pixel, and each lane runs the reference's per-sample loop
// (main.cpp:183-198) as a bounce state machine that regenerates its next camera sample the moment a
// path ends, so lanes stay busy until the tile's last sample.  Per bounce a lane goes through
//   stage A  ray generation     matrixUtilities.h:53-74 (fp64 mat-vec, as the reference)
//            spheres, squares   Scene.h:202-222, Sphere.h:91-132, Square.h:65-126
//            mesh gates         KDTree.cpp:82 / AABB.h:48-65
//   stage B  mesh traversal     Mesh.cpp:112-117, KDTree.cpp:31-85 -- a stackless walk of the flattened
//                               rope KD-tree (nodelets from LDS), triangles Triangle.h:62-126
//   stage C  shading + scatter  Scene.h:270-342, Material.cpp:13-130
// Only a few lanes of a wave have a ray that enters a mesh's box on a given bounce, so stage B is
// DEFERRED: such lanes park (their ray and best-so-far hit stay in registers) while the others go on
// with A and C; the wave runs B when __popcll(__ballot(parked)) reaches HRT_MESH_BATCH or nobody else
// can make progress.  This is the per-wavefront compaction of divergent secondary rays: it is done in
// time (ballot / popcount vote on which stage the wave executes next) rather than by moving rays
// between lanes, so per-pixel sample order -- and therefore every pixel -- stays deterministic.
//
// NUMERICS.  Compiled with -ffp-contract=off; every geometric or branch-deciding expression is written
// in the reference's operation order with its float/double promotions (SURVEY.md 7 "fp64 islands"):
// given the same ray and random numbers a lane takes the same decisions and reaches the same hit point
// as the reference arithmetic, bit for bit (IEEE fp32 add/mul/div/sqrt are correctly rounded on
// gfx950).  Cheap no-division FILTERS (approximate t, conservative margins) only choose which
// primitives go through that exact arithmetic; they never decide a hit.  NOT op-identical: the KD-tree
// walk itself (a different tree; it only selects which triangles are tested), the radiance sum
// (throughput form instead of the recursion's inside-out order, ~1e-7 relative) and fp64 libm calls
// (acos/atan2/asin/pow: other implementations, equal after rounding to fp32 except on rare ties).
#include "hrt_device.h"

#ifndef HRT_MESH_BATCH
#define HRT_MESH_BATCH 16  // parked lanes that trigger a mesh stage (A/B on MI355X: 1 -> 45.4 ms, 16 -> 43.3, 32 -> 53.4)
#endif
#ifndef HRT_WG
#define HRT_WG 256        // threads per workgroup of the trace kernels; the workgroup shares one LDS copy of the nodelets
#endif
#ifndef HRT_MIN_WAVES
#define HRT_MIN_WAVES 4    // waves per SIMD the register allocator must leave room for
#endif
#ifndef HRT_MIN_WAVES_SINGLE
#define HRT_MIN_WAVES_SINGLE 5         // hrt_trace_kernel: 96 VGPRs; A/B on MI355X, Cornell box 1080p@32: 4 -> 20.1 ms, 5 -> 18.4, 6 -> 19.2, 8 -> 20.9
#endif
#ifndef HRT_MIN_WAVES_SINGLE_LIGHTS
#define HRT_MIN_WAVES_SINGLE_LIGHTS 5  // hrt_trace_kernel_lights: random_spheres 1080p@8: 4 -> 8.02 ms, 5 -> 7.70, 6 -> 8.09, 8 -> 9.33
#endif

namespace hrtk {

#define HRT_EPS 1e-5f
#define HRT_FLT_MAX 3.402823466e+38f
// (double)t >= 1e-5  <=>  t > 1e-5f   and   (double)t < -1e-5  <=>  t < -1e-5f   for fp32 t,
// because (float)1e-5 < 1e-5 < nextafterf((float)1e-5, 1).
#define HRT_T_ACCEPT(t) ((t) > HRT_EPS)

// Diagnostic build only (-DHRT_STAMPS): cycles per stage, accumulated per wave in LDS by lane 0 and
// added to DRender::stamps at the end.  The shipped kernel executes no stamp.
#ifdef HRT_STAMPS
// wave-uniform accumulators in registers: cx.st[0..15] = cycles per stage, cx.st[16] = previous stamp
#define STAMP(k)                                                      \
    do {                                                              \
        __builtin_amdgcn_sched_barrier(0);                            \
        const unsigned long long t_ = __builtin_readcyclecounter();  \
        __builtin_amdgcn_sched_barrier(0);                            \
        cx.st[k] += t_ - cx.st[16];                                   \
        cx.st[16] = t_;                                               \
    } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

// u8 -> float tables built on the host in double: [0,256) = c/255., [256,512) = c/127.5 - 1.
__constant__ float c_u8_lut[512];

// ---- address spaces -------------------------------------------------------------------------
// constant (4): wave-uniform records -> scalar loads (s_load_dwordx4 into SGPRs).  The kernel also
//               stores to out_tiles, so plain pointers would not be provably unclobbered.
// global   (1): per-lane indexed rows -> global_load (not flat_load).
// local    (3): nodelets staged in LDS -> ds_read_b128.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef const v4f __attribute__((address_space(4))) *cf4;
typedef const v4f __attribute__((address_space(1))) *gf4;
typedef const v4u __attribute__((address_space(1))) *gu4;
typedef const uint32_t __attribute__((address_space(1))) *gu1;
typedef const v4u __attribute__((address_space(3))) *lu4;
typedef const DMesh __attribute__((address_space(4))) *cmesh;
typedef const DScene __attribute__((address_space(4))) *cscene;
typedef const DCamera __attribute__((address_space(4))) *ccam;
typedef const DImage __attribute__((address_space(1))) *gimg;
typedef const v4f __attribute__((address_space(3))) *lf4;         // scene tables staged in LDS (streaming kernel)
typedef const DMesh __attribute__((address_space(1))) *gmesh;     // per-lane mesh records, global
typedef const DMesh __attribute__((address_space(3))) *lmesh;     // per-lane mesh records, LDS
__device__ __forceinline__ float4 ld(cf4 p, uint32_t i) { const v4f v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float4 ld(gf4 p, uint32_t i) { const v4f v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint4 ld(gu4 p, uint32_t i) { const v4u v = p[i]; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint4 ld(lu4 p, uint32_t i) { const v4u v = p[i]; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float4 ld(lf4 p, uint32_t i) { const v4f v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }

struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 mk(const float4 &v) { return f3{v.x, v.y, v.z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // Vec3.h:48 order
__device__ __forceinline__ float length(f3 a) { return sqrtf(dot(a, a)); }  // == (float)sqrt((double)x)
__device__ __forceinline__ f3 normalize(f3 a) {  // Vec3.h:46: divide by the length, no guard
    const float L = length(a);
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

// Everything a device function needs to reach the scene.  EXACT selects, at compile time, the proof build of every
// kernel form (HRT_FLAG_EXACT_ONLY): no filter and no v_rcp_f32 anywhere in front of the reference arithmetic -- every
// square goes through quad_t in index order, every mesh gate through aabb_gate_exact, shadow rays test every sphere,
// the camera quotient and the walk's reciprocals are IEEE divisions.  The shipped (EXACT = false) kernels carry none
// of that code; tests compare the two bit for bit at full frame size.
// LTAB: where a lane finds the small per-object TABLES when it needs a row of ITS OWN object (the square it refines, the
// material of its hit, its mesh record).  The streaming kernel stages them in LDS (true): a shade there is a chain of
// dependent row fetches -- object row -> material rows -> image geometry -> texel -- and with the path pool streaming
// through the CU's 32 KB L1 every hop went to L2 or beyond (measured: 17 k of the 42 k clocks of a square-hit chunk).
// From LDS only the texel itself is a memory access.  The lane-per-pixel kernels read the same rows from global memory.
typedef const float __attribute__((address_space(1))) *gf1;
typedef const float __attribute__((address_space(3))) *lf1;
template <bool LTAB> struct TabPtr { typedef gf4 f4; typedef gmesh mesh; typedef gf1 f1; };
template <> struct TabPtr<true> { typedef lf4 f4; typedef lmesh mesh; typedef lf1 f1; };

// SPHF: the kernel carries the spheres' pair filter (sphere_filter; scenes with HRT_SPHERE_FILTER_MIN..128 spheres run such a
// build).  It is a build of its own because its code in prims_hit and shadow_blocked costs the register allocator room that
// the scenes without a crowd of spheres -- every wall-and-mesh scene -- then pay for in spills (measured: Cornell+mesh -17 %).
template <bool EXACT, bool LTAB = false, bool SPHF = false>
struct CtxT {
    static constexpr bool exact = EXACT;
    static constexpr bool sphf = SPHF && !EXACT;
    typedef typename TabPtr<LTAB>::f4 tab4;
    typedef typename TabPtr<LTAB>::mesh tabmesh;
    typename TabPtr<LTAB>::f1 lut;  // the u8 -> float tables (c_u8_lut; staged in LDS beside the tables when LTAB)
    tab4 texc;         // exception lists of the meshes when they are short enough to travel with the tables (DScene::exc_in_tabs)
    tab4 tq, tm, ts;   // per-lane rows of squares (HRT_QUAD_ROWS each), materials (HRT_MAT_ROWS), spheres (HRT_SPHERE_ROWS)
    tab4 tsf;          // per-lane rows of the spheres' pair filter (4 per pair; sphere_filter reads the same rows wave-uniformly)
    tabmesh tmesh;     // per-lane mesh records
    // the tables live in one array (DScene::tabs) in the order squares, materials, spheres, meshes
    __device__ __forceinline__ void set_tables(tab4 base, typename TabPtr<LTAB>::f1 lut_, cscene S_) {
        lut = lut_;
        tq = base + S_->tab_quads; tm = base + S_->tab_mats; ts = base + S_->tab_spheres; texc = base + S_->tab_exc; tsf = base + S_->tab_sfilter;
        tmesh = (tabmesh)(base + S_->tab_meshes);
    }
    cscene S;
    lu4 lds;         // nodelets staged in LDS
    uint32_t lds_n;  // how many
    float err_abs;   // margin scale of the filters
    uint32_t flags;  // HRT_FLAG_* of this launch
    unsigned long long *st;  // diagnostic stamps (HRT_STAMPS builds), else unused
};
typedef CtxT<false> Ctx;

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
// Q = cf4 (wave-uniform quad, scalar rows) or gf4 (per-lane quad, vector rows).
template <class Q>
__device__ __forceinline__ bool quad_t(Q q, const Ray &ray, float tmax, float &t, float &u, float &v) {
    const float4 q0 = ld(q, 0), q1 = ld(q, 1);
    const uint32_t flags = __float_as_uint(q1.w);
    const f3 n = mk(q1);
    const float dotRN = dot(ray.d, n);
    if (dotRN == 0.f) return false;
    if (dotRN > 0.f && !(flags & HRT_QUAD_FLAG_GLASS)) return false;
    f3 p0 = mk(q0);
    float D = q0.w;
    if (flags & HRT_QUAD_FLAG_MOVING) {
        p0 = p0 + ray.time * mk(ld(q, 4));
        D = dot(p0, n);
    }
    t = (D - dot(ray.o, n)) / dotRN;
    if (!HRT_T_ACCEPT(t) || !(t < tmax)) return false;
    const float4 q2 = ld(q, 2), q3 = ld(q, 3);
    const f3 qq = (ray.o + t * ray.d) - p0;
    const float proj1 = dot(qq, mk(q2)) / q2.w;
    const float proj2 = dot(qq, mk(q3)) / q3.w;
    if (!((proj1 <= q2.w && proj1 >= 0.f) && (proj2 <= q3.w && proj2 >= 0.f))) return false;
    u = proj1 / q2.w;
    v = proj2 / q3.w;
    return true;
}

// The gate box of a mesh (Mesh::computeAABB), either read from the mesh record at the point of use
// (scalar loads) or held in registers by the caller so that no load sits on the critical path.
struct MeshBox {
    cmesh M;
    __device__ __forceinline__ float lo(int a) const { return M->aabb_lo[a]; }
    __device__ __forceinline__ float hi(int a) const { return M->aabb_hi[a]; }
};
struct GateBox {
    float l[3], h[3];
    __device__ __forceinline__ float lo(int a) const { return l[a]; }
    __device__ __forceinline__ float hi(int a) const { return h[a]; }
};
__device__ __forceinline__ GateBox gate_box_of(cmesh M) {
    GateBox b;
    b.l[0] = M->aabb_lo[0]; b.l[1] = M->aabb_lo[1]; b.l[2] = M->aabb_lo[2];
    b.h[0] = M->aabb_hi[0]; b.h[1] = M->aabb_hi[1]; b.h[2] = M->aabb_hi[2];
    return b;
}

// AABB.h:48-65 exactly: reciprocal in double, products narrowed to float.
template <class B>
__device__ __forceinline__ bool aabb_gate_exact(const B &box, const Ray &ray) {
    float tmin = HRT_EPS, tmax = HRT_FLT_MAX;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float da = a == 0 ? ray.d.x : (a == 1 ? ray.d.y : ray.d.z);
        const float oa = a == 0 ? ray.o.x : (a == 1 ? ray.o.y : ray.o.z);
        const double adinv = 1.0 / (double)da;
        const float t0 = (float)((double)(box.lo(a) - oa) * adinv);
        const float t1 = (float)((double)(box.hi(a) - oa) * adinv);
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

// KDTree.cpp:82 gate.  An fp32 slab test with a margin settles the clear cases (each slab distance
// differs from the reference's by <= 3e-7 |t|); only a ray that grazes the box within the margin, or
// has a zero direction component, pays for the exact fp64 form.
// The fp32 slab test in front of AABB::intersects: -1 the ray certainly misses the box, +1 it certainly passes, 0 too close
// to call (within the margin, or a zero direction component).
template <class B>
__device__ __forceinline__ int gate_filter(const B &box, const Ray &ray, f3 inv) {
    float g0 = HRT_EPS, g1 = HRT_FLT_MAX, big = 0.f;
    float t0 = (box.lo(0) - ray.o.x) * inv.x, t1 = (box.hi(0) - ray.o.x) * inv.x;
    g0 = fmaxf(g0, fminf(t0, t1)); g1 = fminf(g1, fmaxf(t0, t1)); big = fmaxf(big, fmaxf(fabsf(t0), fabsf(t1)));
    t0 = (box.lo(1) - ray.o.y) * inv.y; t1 = (box.hi(1) - ray.o.y) * inv.y;
    g0 = fmaxf(g0, fminf(t0, t1)); g1 = fminf(g1, fmaxf(t0, t1)); big = fmaxf(big, fmaxf(fabsf(t0), fabsf(t1)));
    t0 = (box.lo(2) - ray.o.z) * inv.z; t1 = (box.hi(2) - ray.o.z) * inv.z;
    g0 = fmaxf(g0, fminf(t0, t1)); g1 = fminf(g1, fmaxf(t0, t1)); big = fmaxf(big, fmaxf(fabsf(t0), fabsf(t1)));
    const float mg = fmaxf(fabsf(g0), fabsf(g1)) * 2e-6f + 1e-30f;
    const bool finite = big < 1e30f;  // a zero direction component makes a slab distance infinite
    if (finite && g1 < g0 - mg) return -1;
    if (finite && g1 > g0 + mg) return 1;
    return 0;
}
template <bool EXACT, class B>
__device__ __forceinline__ bool mesh_gate_box(const B &box, const Ray &ray, f3 inv) {
    if (EXACT) return aabb_gate_exact(box, ray);
    const int f = gate_filter(box, ray, inv);
    if (f != 0) return f > 0;
#ifdef HRT_ABL_NO_EXACT_GATE  // ablation only: timing experiment, not parity-safe
    return true;
#else
    return aabb_gate_exact(box, ray);
#endif
}
template <bool EXACT>
__device__ __forceinline__ bool mesh_gate(cmesh M, const Ray &ray, f3 inv) { return mesh_gate_box<EXACT>(MeshBox{M}, ray, inv); }

// Nodelet fetch: the leading `lds_n` units of the kd array (a multiple of 4: treelets and leaves never straddle) are resident in
// LDS.  A wave's lanes can sit on both sides, so both paths run, and the second one writes the registers of the first: hipcc
// drains the first path's loads before it issues the second's.  Hence ALL units of a treelet / leaf go under ONE pair of
// branches -- their global loads are in flight together and the drain costs one round trip.  (With one if / else per unit a
// leaf's four units cost four memory round trips in a row: 1080p@64 pool 113.9 -> 109.2 ms, mesh_in_box 63.8 -> 61.8.  Forcing
// the LDS path first with a compiler barrier, so that only the short LDS latency is drained, measured no better.)
template <class CX>
__device__ __forceinline__ uint4 kd_fetch(gu4 g, const CX &cx, uint32_t i) {
    uint4 r = make_uint4(0u, 0u, 0u, 0u);
    if (i < cx.lds_n) r = ld(cx.lds, i);
    if (i >= cx.lds_n) r = ld(g, i);
    return r;
}
template <class CX>
__device__ __forceinline__ void kd_fetch2(gu4 g, const CX &cx, uint32_t i, uint4 &a, uint4 &b) {
    a = b = make_uint4(0u, 0u, 0u, 0u);
    if (i < cx.lds_n) { a = ld(cx.lds, i); b = ld(cx.lds, i + 1u); }
    if (i >= cx.lds_n) { a = ld(g, i); b = ld(g, i + 1u); }
}
template <class CX>
__device__ __forceinline__ void kd_fetch4(gu4 g, const CX &cx, uint32_t i, uint4 &a, uint4 &b, uint4 &c, uint4 &d) {
    a = b = c = d = make_uint4(0u, 0u, 0u, 0u);
    if (i < cx.lds_n) { a = ld(cx.lds, i); b = ld(cx.lds, i + 1u); c = ld(cx.lds, i + 2u); d = ld(cx.lds, i + 3u); }
    if (i >= cx.lds_n) { a = ld(g, i); b = ld(g, i + 1u); c = ld(g, i + 2u); d = ld(g, i + 3u); }
}

// Inner nodes of the device's tree are TREELETS of two levels in 32 bytes (hrt_api.hip scene_create_impl builds them from the
// caller's 16-byte nodelets): {split, split of the left child, split of the right child, axes} {the four grandchildren's refs},
// axes = 2 bits per node, 3 = "no such node: the child is a leaf, take the first exit of its pair".  One round trip (two
// 16-byte loads of one half line) descends two levels; the per-node rule is KDTree.cpp's: left when the point lies below the
// split, or on it with the ray heading down.
template <class CX>
__device__ __forceinline__ uint32_t kd_descend(gu4 g, const CX &cx, uint32_t ref, f3 p, f3 d) {
    uint4 a, b;
    kd_fetch2(g, cx, ref, a, b);
    const uint32_t ax0 = a.w & 3u;
    const float s0 = __uint_as_float(a.x);
    const float pc0 = comp(p, ax0), dc0 = comp(d, ax0);
    const bool left0 = (pc0 < s0) || (pc0 == s0 && dc0 < 0.f);
    const uint32_t ax1 = (left0 ? a.w >> 2 : a.w >> 4) & 3u;
    const float s1 = __uint_as_float(left0 ? a.y : a.z);
    const float pc1 = comp(p, ax1), dc1 = comp(d, ax1);
    const bool left1 = ax1 == 3u || (pc1 < s1) || (pc1 == s1 && dc1 < 0.f);
    return left0 ? (left1 ? b.x : b.y) : (left1 ? b.z : b.w);
}

// One triangle of the soup against the ray: Triangle::getIntersection (Triangle.h:77-126) with the constructor's and
// computeBarycentricCoordinates' constants folded on the host, then the leaf's strict `<` against the best so far
// (KDTree.cpp:44).  True when this triangle became the best.  The soup is two arrays indexed by slot:
//   planes  {n, D}: 16 bytes, so the planes of a leaf's (<= 4) triangles share one cache line -- the plane test rejects most
//   rows    {c0, d11} {e1, d00} {e2, d01} {id}: one aligned 64-byte line; the first three fetched only by a triangle whose plane is hit in
//           front (the denominator d00 d11 - d01^2 is recomputed, in the constructor's roundings), the id only when the triangle is shaded
#ifndef HRT_LEAF_BATCH
#define HRT_LEAF_BATCH 2   // triangles of a leaf tested per trip of a walk, their planes requested together
#endif
struct Soup {
    gf4 planes, rows;
};
__device__ __forceinline__ Soup soup_of(cscene S) { return Soup{(gf4)S->tri_planes, (gf4)S->tris}; }
// In two halves, so that a batch can request the rows of all its candidates together: tri_plane_t is :80-96 (false: the ray does
// not reach the triangle's plane in front of it and closer than the best so far), tri_inside the rest.
__device__ __forceinline__ bool tri_plane_t(const float4 pl, const Ray &ray, float best_t, float &t) {
    const f3 n = mk(pl);
    const float dotRN = dot(ray.d, n);
    if (!(dotRN < 0.f)) return false;                     // :80-91 parallel / back-facing (NaN: no hit)
    t = (pl.w - dot(ray.o, n)) / dotRN;                   // :95
    return !(t < 0.f || !(t < best_t));                   // :96, then KDTree.cpp:44
}
__device__ __forceinline__ bool tri_inside(const float4 r0, const float4 r1, const float4 r2, const Ray &ray, float t, float &u1, float &u2) {
    const f3 v2 = (ray.o + t * ray.d) - mk(r0);
    const float d20 = dot(v2, mk(r1)), d21 = dot(v2, mk(r2));
    const float d00 = r1.w, d01 = r2.w, d11 = r0.w;
    const float denom = d00 * d11 - d01 * d01;            // :66-70, the constructor's own two roundings: a row less to fetch per candidate
    u1 = (d11 * d20 - d01 * d21) / denom;                 // :72-74
    u2 = (d00 * d21 - d01 * d20) / denom;
    const float u0 = 1 - u1 - u2;
    return u0 >= 0 && u0 <= 1 && u1 >= 0 && u1 <= 1 && u2 >= 0 && u2 <= 1;
}
__device__ __forceinline__ bool tri_test_plane(gf4 tr, const float4 pl, const Ray &ray, float &best_t, float &bu, float &bv) {
    float t, u1, u2;
    if (!tri_plane_t(pl, ray, best_t, t)) return false;
    if (!tri_inside(ld(tr, 0), ld(tr, 1), ld(tr, 2), ray, t, u1, u2)) return false;
    best_t = t; bu = u1; bv = u2;
    return true;
}
__device__ __forceinline__ bool tri_test(const Soup &sp, uint32_t slot, const Ray &ray, float &best_t, float &bu, float &bv) {
    return tri_test_plane(sp.rows + HRT_TRI_ROWS * slot, ld(sp.planes, slot), ray, best_t, bu, bv);
}
// Triangles [k, min(k + HRT_LEAF_BATCH, cnt)) of the run that starts at soup slot `first`, in order; returns the new k.
// Requires k < cnt.  (Requesting the rows of every candidate of the batch together, before finishing them in order, saves a
// round trip when two planes are reached -- and cost 22 more spilled VGPRs: 1-2 % slower on MI355X, dropped.)
__device__ __forceinline__ uint32_t tri_test_run(const Soup &sp, uint32_t first, uint32_t cnt, uint32_t k, const Ray &ray, float &best_t,
                                                 uint32_t &best_tri, float &bu, float &bv, bool &found) {
    float4 pl[HRT_LEAF_BATCH];
#pragma unroll
    for (uint32_t j = 0; j < HRT_LEAF_BATCH; ++j) pl[j] = ld(sp.planes, first + min(k + j, cnt - 1u));  // clamped: stays inside the run
#pragma unroll
    for (uint32_t j = 0; j < HRT_LEAF_BATCH; ++j)
        if (k + j < cnt && tri_test_plane(sp.rows + HRT_TRI_ROWS * (first + k + j), pl[j], ray, best_t, bu, bv)) { best_tri = first + k + j; found = true; }
    return min(k + (uint32_t)HRT_LEAF_BATCH, cnt);
}

// The IRREGULAR triangles of a mesh (include/hrt.h hrt_tri_exception; host/ref_tree.h): triangles the reference's builder
// drops in part of its tree, and slivers whose barycentric test accepts phantom points.  They are not in the rope tree;
// each counts exactly when the reference would test it -- when the ray passes the box of a reference leaf that holds
// it (KDTree.cpp:32-46 with AABB.h:48-65; EXACT: the fp64 form only, else the fp32 filter in front of it).  The triangle
// test comes FIRST (its outcome does not depend on the box) and the boxes are consulted only for a hit closer than the best.
template <bool EXACT, class EP, class MP>
__device__ __forceinline__ bool mesh_exceptions_walk(cscene S, EP ex_all, MP M, const Ray &ray, f3 inv, float &best_t, uint32_t &best_tri, float &bu, float &bv) {
    const uint32_t n = M->n_exc;
    bool found = false;
    const EP ex = ex_all + 2u * M->exc_base;
    const Soup sp = soup_of(S);
    // entries (hrt_api.hip scene_create_impl): {lo', HRT_EXC_INNER} {hi', skip} bounds of a subtree; {cull lo, soup slot} {cull hi, nb}
    // one irregular triangle, followed by its nb reference leaf boxes {lo, 0} {hi, 0}
    for (uint32_t i = 0; i < n;) {
        const float4 lo = ld(ex, 2u * i), hi = ld(ex, 2u * i + 1u);
        const uint32_t first = __float_as_uint(lo.w), cnt = __float_as_uint(hi.w);
        GateBox b;
        b.l[0] = lo.x; b.l[1] = lo.y; b.l[2] = lo.z; b.h[0] = hi.x; b.h[1] = hi.y; b.h[2] = hi.z;
        ++i;
        if (first == HRT_EXC_INNER) {  // bounds of a subtree: only culls (never in the proof builds)
            if (!EXACT && gate_filter(b, ray, inv) < 0) i = cnt;
            continue;
        }
        const uint32_t boxes = i;
        i += cnt;
        if (!EXACT && gate_filter(b, ray, inv) < 0) continue;  // the cull box (padded; never in the proof builds)
        float t = best_t, u = 0.f, v = 0.f;
        if (!tri_test(sp, first, ray, t, u, v)) continue;      // Triangle.h:77-126 and the leaf's strict `<` (KDTree.cpp:44)
        // would the reference have tested it: does the ray reach a reference leaf that holds it -- pass that leaf's box and the boxes
        // of the ancestors that stick into it (AABB.h:48-65 each; the boxes of one leaf follow each other, .w = 1 on the last)
        bool tested = false, reach = true;
        for (uint32_t j = 0; j < cnt && !tested; ++j) {
            const float4 bl = ld(ex, 2u * (boxes + j)), bh = ld(ex, 2u * (boxes + j) + 1u);
            GateBox rb;
            rb.l[0] = bl.x; rb.l[1] = bl.y; rb.l[2] = bl.z; rb.h[0] = bh.x; rb.h[1] = bh.y; rb.h[2] = bh.z;
            if (reach) reach = mesh_gate_box<EXACT>(rb, ray, inv);
            if (__float_as_uint(bl.w) != 0u) { tested = reach; reach = true; }
        }
        if (tested) { best_t = t; bu = u; bv = v; best_tri = first; found = true; }
    }
    return found;
}
// Short lists travel with the per-object tables (staged in LDS by the streaming kernel: the walk through the list is a
// chain of dependent row fetches); long ones (a mesh the reference's builder mangles badly) stay in global memory.
template <bool EXACT, class CX, class MP>
__device__ __forceinline__ bool mesh_exceptions(const CX &cx, MP M, const Ray &ray, f3 inv, float &best_t, uint32_t &best_tri, float &bu, float &bv) {
    if (M->n_exc == 0u) return false;
    if (cx.S->exc_in_tabs) return mesh_exceptions_walk<EXACT>(cx.S, cx.texc, M, ray, inv, best_t, best_tri, bu, bv);
    return mesh_exceptions_walk<EXACT>(cx.S, (gf4)cx.S->exceptions, M, ray, inv, best_t, best_tri, bu, bv);
}

// HRT_FLAG_MESH_BRUTE (exact builds only): every triangle of the mesh's leaf-ordered soup, no tree -- the device-side
// statement of Mesh::intersectOld (Mesh.h:257-277) with the leaf's strict `<` (KDTree.cpp:44).  A straddling triangle
// sits in the soup once per leaf: the repeats give an equal t and lose to the first.  Tests compare this with the rope
// walk at frame scale: the walk only chooses WHICH triangles are tested, so both must select the same closest hit.
template <class CX>
__device__ __forceinline__ bool mesh_brute(const CX &cx, cmesh M, const Ray &ray, float &best_t, uint32_t &best_tri, float &bu, float &bv) {
    const Soup sp = soup_of(cx.S);
    const uint32_t first = M->tri_base, cnt = M->n_soup;
    best_t = HRT_FLT_MAX;
    bool found = mesh_exceptions<true>(cx, M, ray, mk(0.f, 0.f, 0.f), best_t, best_tri, bu, bv);
    for (uint32_t k = 0; k < cnt;) k = tri_test_run(sp, first, cnt, k, ray, best_t, best_tri, bu, bv, found);
    return found;
}

// Closest triangle of one mesh with t >= 0 (KDTree.cpp:31-85 semantics: the caller applies
// `t >= EPSILON && t < best`); the ray has already passed mesh_gate.  Stackless: locate the leaf that
// holds the entry point, test its triangles, leave through the exit face's rope, repeat while no hit lies
// inside the visited cells.  Triangle rows: 0 {c0, id} 1 {e1, d00} 2 {e2, d01} 3 {n, D} 4 {d11, denom}.
template <class CX>
__device__ __forceinline__ bool mesh_traverse(const CX &cx, cmesh M, const Ray &ray, f3 inv, float &best_t,
                                              uint32_t &best_tri, float &bu, float &bv) {
    if (CX::exact && (cx.flags & HRT_FLAG_MESH_BRUTE)) return mesh_brute(cx, M, ray, best_t, best_tri, bu, bv);
    float t_entry = 0.f, t_scene_exit = HRT_FLT_MAX;
    {
        float t0 = (M->kd_lo[0] - ray.o.x) * inv.x, t1 = (M->kd_hi[0] - ray.o.x) * inv.x;
        t_entry = fmaxf(t_entry, fminf(t0, t1)); t_scene_exit = fminf(t_scene_exit, fmaxf(t0, t1));
        t0 = (M->kd_lo[1] - ray.o.y) * inv.y; t1 = (M->kd_hi[1] - ray.o.y) * inv.y;
        t_entry = fmaxf(t_entry, fminf(t0, t1)); t_scene_exit = fminf(t_scene_exit, fmaxf(t0, t1));
        t0 = (M->kd_lo[2] - ray.o.z) * inv.z; t1 = (M->kd_hi[2] - ray.o.z) * inv.z;
        t_entry = fmaxf(t_entry, fminf(t0, t1)); t_scene_exit = fminf(t_scene_exit, fmaxf(t0, t1));
    }
    best_t = HRT_FLT_MAX;
    bool found = mesh_exceptions<CX::exact>(cx, M, ray, inv, best_t, best_tri, bu, bv);
    if (!(t_entry <= t_scene_exit)) return found;
    gu4 g_units = (gu4)cx.S->kd_units;
    const Soup sp = soup_of(cx.S);
    const uint32_t tri_base = M->tri_base;
    // One flat loop, one small unit of work per lane per trip (descend <= 2 levels, then enter the leaf /
    // test up to HRT_LEAF_BATCH of its triangles / leave through a rope): a trip costs about the same for all
    // lanes and the trip count is the largest number of units any lane needs.
    uint32_t ref = M->root;
    uint32_t k = 0, cnt = ~0u, first = 0;  // triangle cursor of the current leaf; cnt == ~0: leaf not entered yet
    f3 p = ray.o + t_entry * ray.d;
#ifndef HRT_WALK_CELLS
#define HRT_WALK_CELLS 8192  // cells one walk may cross: far beyond any real walk (a ray crosses O(depth * n^(1/3)) cells); it only
#endif                       // stops a walk that rounding sends back and forth between two cells.  Triangle tests are NOT counted:
                             // a leaf of any size is tested to its end (smaller values: ablation only, not parity-safe)
    for (uint32_t cells = 0; cells < HRT_WALK_CELLS && ref != HRT_KD_NIL;) {  // bounded: every wave leaves
        if (!(ref & HRT_KD_LEAF)) ref = kd_descend(g_units, cx, ref, p, ray.d);  // two levels
        if (ref & HRT_KD_LEAF) {
            const uint32_t lu = ref & ~HRT_KD_LEAF;
            uint4 l0, l1, rp0, rp1;
            kd_fetch4(g_units, cx, lu, l0, l1, rp0, rp1);  // the ropes with the header: one round trip less per cell
            if (cnt == ~0u) { first = tri_base + l0.w; cnt = l1.w; k = 0; }
#ifdef HRT_ABL_NO_TRI  // ablation only
            k = cnt;
#endif
            if (k < cnt) k = tri_test_run(sp, first, cnt, k, ray, best_t, best_tri, bu, bv, found);
            if (k >= cnt) {  // leave the cell through its exit face
                const float ex = ((ray.d.x > 0.f ? __uint_as_float(l1.x) : __uint_as_float(l0.x)) - ray.o.x) * inv.x;
                const float ey = ((ray.d.y > 0.f ? __uint_as_float(l1.y) : __uint_as_float(l0.y)) - ray.o.y) * inv.y;
                const float ez = ((ray.d.z > 0.f ? __uint_as_float(l1.z) : __uint_as_float(l0.z)) - ray.o.z) * inv.z;
                float t_exit = HRT_FLT_MAX;
                uint32_t face = 6;
                if (ray.d.x != 0.f && ex < t_exit) { t_exit = ex; face = ray.d.x > 0.f ? 1u : 0u; }
                if (ray.d.y != 0.f && ey < t_exit) { t_exit = ey; face = ray.d.y > 0.f ? 3u : 2u; }
                if (ray.d.z != 0.f && ez < t_exit) { t_exit = ez; face = ray.d.z > 0.f ? 5u : 4u; }
                if (best_t <= t_exit || face == 6) {
                    ref = HRT_KD_NIL;  // the closest hit lies inside the cells already visited
                } else {
                    t_entry = fmaxf(t_entry, t_exit);
                    p = ray.o + t_entry * ray.d;
                    const uint4 rp = (face >> 2) ? rp1 : rp0;
                    const uint32_t sel = face & 3u;
                    ref = sel == 0 ? rp.x : (sel == 1 ? rp.y : (sel == 2 ? rp.z : rp.w));
                    cnt = ~0u;
                    ++cells;
                }
            }
        }
    }
    return found;
}

template <bool EXACT>
__device__ __forceinline__ f3 ray_inv(const Ray &ray) {
    if (EXACT) return mk(1.f / ray.d.x, 1.f / ray.d.y, 1.f / ray.d.z);
    return mk(__builtin_amdgcn_rcpf(ray.d.x), __builtin_amdgcn_rcpf(ray.d.y), __builtin_amdgcn_rcpf(ray.d.z));
}

template <int A>
__device__ __forceinline__ float cget(f3 v) { return A == 0 ? v.x : (A == 1 ? v.y : v.z); }

// FILTER, squares that lie (nearly) in an axis plane -- every wall of the reference's scenes: setQuad builds them axis-
// aligned and the set-up code's rotate_x / rotate_y by multiples of 90 degrees leave residues of a few 1e-8 in the other
// components (cos(pi/2) in fp32).  The host recognises a static square whose folded normal is within eps_n of +-e_K and
// whose edges run within the same tolerance along the other two axes (hrt_api.hip build_quad_filter) and stores
//   r0 {sgn * D, centre_I, centre_J, half_I}   r1 {half_J, bits, par, cq}     bits: 1 glass, 2 normal along -K, square index << 8
// The reference's t = (D - o.n) / (d.n) then differs from ta = (sgn * D - o_K) / d_K by at most
//   |t - ta| <= 1.15 (eps_n + 3e-7) (ext + |ta|) / |d_K|        (numerator and denominator each off by <= eps_n (ext | 1);
//                                                                 3e-7: the reference's own rounding; valid for |d_K| >= 8 eps_n)
// and a hit has |ta| <= 3 ext (ext = scene + camera extent = err_abs / 2e-6), so with cq = 2.5e6 (eps_n + 3e-7) the
// margin  E = cq |1/d_K| err_abs + 5e-6 |ta| + err_abs  bounds |t - ta| and the distance between the filter's hit point
// and the reference's.  The rectangle bounds along I, J are those of the four corners, widened on the host by the same
// kind of term.  Rays with |d_K| < par (= 8 (eps_n + 4e-7)) are too parallel to judge: they go to the exact arithmetic.
// 21 VALU per square instead of the 40 of the general form below; the reciprocal is taken once per ray and axis.
template <class M, int K, class CX>
__device__ __forceinline__ void quad_filter_axis(const CX &cx, const Ray &ray, f3 inv, cf4 rows, uint32_t n, float tsure_up, M &cand) {
    constexpr int I = (K + 1) % 3, J = (K + 2) % 3;
    const float ok = cget<K>(ray.o), dk = cget<K>(ray.d), ik = cget<K>(inv);
    const float oi = cget<I>(ray.o), di = cget<I>(ray.d), oj = cget<J>(ray.o), dj = cget<J>(ray.d);
    const float iek = fabsf(ik) * cx.err_abs;
    auto test = [&](const float4 &r0, const float4 &r1) {
        const uint32_t bits = __float_as_uint(r1.y);
        const float ta = (r0.x - ok) * ik;
        const float xi = __builtin_fmaf(ta, di, oi) - r0.y, xj = __builtin_fmaf(ta, dj, oj) - r0.z;
        const float E = __builtin_fmaf(r1.w, iek, __builtin_fmaf(fabsf(ta), 5e-6f, cx.err_abs));
        const float dn = __uint_as_float(__float_as_uint(dk) ^ ((bits & 2u) << 30));  // sgn * d_K ~ d . n
        const bool glass = (bits & 1u) != 0u;
        const bool front = glass | (dn < r1.z);
        const bool unsure = fabsf(dk) < r1.z;
        const bool loose = front & (ta + E >= 9e-6f) & (ta - E <= tsure_up) & (fabsf(xi) <= r0.w + E) & (fabsf(xj) <= r1.x + E);
        if (loose | unsure) cand |= (M)1 << (bits >> 8);
    };
    float4 a0 = make_float4(0, 0, 0, 0), a1 = a0, b0 = a0, b1 = a0;  // two row sets in ping-pong (see quad_filter)
    if (n > 0u) { a0 = ld(rows, 0); a1 = ld(rows, 1); }
    if (n > 1u) { b0 = ld(rows, 2); b1 = ld(rows, 3); }
    for (uint32_t i = 0; i < n; i += 2u) {
        test(a0, a1);
        if (i + 2u < n) { a0 = ld(rows, 2u * (i + 2u)); a1 = ld(rows, 2u * (i + 2u) + 1u); }
        if (i + 1u < n) {
            test(b0, b1);
            if (i + 3u < n) { b0 = ld(rows, 2u * (i + 3u)); b1 = ld(rows, 2u * (i + 3u) + 1u); }
        }
    }
}

// The no-division FILTER over the squares (wave-uniform loops, scalar rows): bit i of the result = square i can possibly
// be the closest accepted hit (see prims_hit).  M = uint32_t for up to 32 squares (the per-lane mask costs half the VALU
// work of a 64-bit one), uint64_t for up to 64.  The filter rows live in DScene::qfilter in four sections: squares in an
// axis plane by normal axis (x, y, z; 2 rows each, quad_filter_axis), then all others (4 rows each, below).
template <class M, class CX>
__device__ __forceinline__ M quad_filter(const CX &cx, const Ray &ray, float tsure) {
    M cand = 0;
    cscene S = cx.S;
    cf4 qf = (cf4)S->qfilter;
    const float tsure_up = tsure * (1.f + 2e-6f);  // the approximate t may exceed the exact one by a few ulp
    const uint32_t n0 = S->qf_n[0], n1 = S->qf_n[1], n2 = S->qf_n[2], n3 = S->qf_n[3];
    if (n0 + n1 + n2 != 0u) {
        const f3 inv = ray_inv<false>(ray);
        quad_filter_axis<M, 0>(cx, ray, inv, qf, n0, tsure_up, cand);
        quad_filter_axis<M, 1>(cx, ray, inv, qf + 2u * n0, n1, tsure_up, cand);
        quad_filter_axis<M, 2>(cx, ray, inv, qf + 2u * (n0 + n1), n2, tsure_up, cand);
    }
    // General squares.  Two row sets in ping-pong: while quad i is evaluated from one set, the rows of quad i + 1 are
    // already in flight into the other, and quad i + 2 is requested into the first as soon as quad i is done -- the scalar
    // load latency overlaps the ~40 VALU instructions of a quad and no row is ever copied (a rolling single-set
    // prefetch costs 16 s_mov per quad, as many issue slots as a third of the filter).
    // f0 {p0.xyz, D0}  f1 {n.y, n.z, n.x, flags | index << 8}  f2 {R.x, U.x, R.y, U.y}  f3 {R.z, U.z, |R|, |U|}: the (R, U)
    // and (n.y, n.z) pairs sit in aligned SGPR pairs (a layout chosen for v_pk_mul/fma_f32; the device code is now compiled
    // WITHOUT packed fp32 -- see the Makefile -- and the scalar instructions read the same registers).
    auto filter = [&](const float4 &f0, const float4 &f1, const float4 &f2, const float4 &f3) {
        const uint32_t flags = __float_as_uint(f1.w), i = flags >> 8;
        if (flags & HRT_QUAD_FLAG_MOVING) { cand |= (M)1 << i; return; }  // uniform branch; the exact path decides
        const float dotRN = ray.d.x * f1.z + ray.d.y * f1.x + ray.d.z * f1.y;  // exact, Vec3.h:48 order: the sign tests are the reference's
        const bool glass = (flags & HRT_QUAD_FLAG_GLASS) != 0u;
        const bool front = (dotRN < 0.f) | (glass & (dotRN > 0.f));  // `|`, `&`: lane masks combined, no short-circuit branches
        const float num = f0.w - (ray.o.x * f1.z + ray.o.y * f1.x + ray.o.z * f1.y);  // exact numerator
        const float ta = num * __builtin_amdgcn_rcpf(dotRN);  // |ta - fl(num/dotRN)| <= 4e-7 |t|
        const float ax = __builtin_fmaf(ta, ray.d.x, ray.o.x) - f0.x, ay = __builtin_fmaf(ta, ray.d.y, ray.o.y) - f0.y,
                    az = __builtin_fmaf(ta, ray.d.z, ray.o.z) - f0.z;
        const float x1 = __builtin_fmaf(az, f3.x, __builtin_fmaf(ay, f2.z, ax * f2.x));
        const float x2 = __builtin_fmaf(az, f3.y, __builtin_fmaf(ay, f2.w, ax * f2.y));
        const float e = __builtin_fmaf(fabsf(ta), 4e-6f, cx.err_abs);  // bound on |p' - p|, generous
        const float m1 = f3.z * e, m2 = f3.w * e, s1 = f3.z * f3.z, s2 = f3.w * f3.w;
        const bool loose = front & (ta >= 9e-6f) & (ta <= tsure_up) & (x1 >= -m1) & (x1 <= s1 + m1) & (x2 >= -m2) & (x2 <= s2 + m2);
        if (loose) cand |= (M)1 << i;
        // (A second, strict test used to shrink `tsure` to the nearest square that is certainly hit, so that squares
        // behind it were not refined.  It cost 11 VALU per square and saved a second refinement on ~5 % of the rays:
        // dropping it is 1-4 % faster on every scene.)
    };
    float4 a0 = make_float4(0, 0, 0, 0), a1 = a0, a2 = a0, a3 = a0, b0 = a0, b1 = a0, b2 = a0, b3 = a0;
    cf4 fr = qf + 2u * (n0 + n1 + n2);
    if (n3 > 0u) { a0 = ld(fr, 0); a1 = ld(fr, 1); a2 = ld(fr, 2); a3 = ld(fr, 3); }
    if (n3 > 1u) { b0 = ld(fr, 4); b1 = ld(fr, 5); b2 = ld(fr, 6); b3 = ld(fr, 7); }
    for (uint32_t i = 0; i < n3; i += 2u) {
        filter(a0, a1, a2, a3);
        if (i + 2u < n3) { a0 = ld(fr, 4u * (i + 2u)); a1 = ld(fr, 4u * (i + 2u) + 1u); a2 = ld(fr, 4u * (i + 2u) + 2u); a3 = ld(fr, 4u * (i + 2u) + 3u); }
        if (i + 1u < n3) {
            filter(b0, b1, b2, b3);
            if (i + 3u < n3) { b0 = ld(fr, 4u * (i + 3u)); b1 = ld(fr, 4u * (i + 3u) + 1u); b2 = ld(fr, 4u * (i + 3u) + 2u); b3 = ld(fr, 4u * (i + 3u) + 3u); }
        }
    }
    return cand;
}

// FILTER over the spheres of a scene that has many (no square root, no division): which spheres can possibly give
// Sphere::intersect an accepted hit; the exact arithmetic (sphere_t) then runs only on those, per lane, in index order.  Two
// spheres A, B go through the arithmetic together: the rows hold (A, B) pairs and the arithmetic is written on two-element
// vectors.  (Written for gfx950's v_pk_add / v_pk_mul / v_pk_fma_f32; on MI355X a packed instruction occupies the SIMD as long as
// the two scalar ones it stands for and wants aligned register pairs on top -- random_spheres 1080p @ 64: 4 008 Msamples/s
// packed, 4 252 with this function alone scalar, 4 401 with no packed fp32 anywhere -- so the Makefile compiles the device code
// without them and each vector operation below becomes two scalar ones.  What the filter saves is the square root and the
// division of the exact test, not issue slots through packing.)  For the ray
// o + t d and the sphere centre c = c0 + time * motion the exact test forms  oc = o - c,  b = 2 d.oc,  cc = oc.oc - r^2,
// delta = b^2 - 4 a cc  and has NO accepted hit when delta < 0 or when b >= 0 (then t = (-b - sqrt(delta)) / 2a <= 0 fails
// `t >= EPSILON`).  The filter evaluates the same quantities with fused multiply-adds,
//     bh = d.oc     S = oc.oc     E1 = bh^2 - a (S - r^2) + m     E3 = mb - bh |bh|
// and keeps the sphere when E1 >= 0 and E3 >= 0, where the margins bound everything that can differ between the two
// evaluations.  Both compute oc to within 0.35 e, e = err_abs = 2e-6 ext (ext = extent of scene and camera; three roundings of
// a coordinate are 1.8e-7 ext per component).  delta / 4 = bh^2 - a cc has gradient <= 4 sqrt(S) in oc, so the two values
// differ by <= 4 sqrt(S) e <= 2 e (S / L + L) for any L > 0; with L = ext / 4 that is 1.6e-5 S + 1e-6 ext^2.  The roundings
// of the products and sums add <= 1e-6 (S + r^2) on either side.  Hence
//     m = 2.6e-5 (S + r^2) + 2.5e5 e^2 + 1e-30          mb = 2 e^2 + 1e-12 S  >=  (e + 2e-7 sqrt(S))^2  >=  |bh - b/2|^2.
// (A ray that starts inside a sphere -- cc < 0 -- is not rejected here; the exact test sends it away.)
// Rows, 4 per pair (DScene::tab_sfilter): {c.x A, c.x B, c.y A, c.y B} {c.z A, c.z B, r^2 A, r^2 B} {motion.x A, B, motion.y A, B}
// {motion.z A, B, -, -}; an odd last sphere is paired with itself.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk(float x) { v2f r; r.x = x; r.y = x; return r; }
__device__ __forceinline__ v2f pkfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
#ifndef HRT_SPHERE_FILTER_MIN
#define HRT_SPHERE_FILTER_MIN 8u  // fewer spheres go through the exact test directly (the two of the Cornell box)
#endif
struct SphereFilterRay {  // what the filter needs of one ray, as (x, x) pairs
    v2f ox, oy, oz, dx, dy, dz, ntime, a_lo, a_hi, m_abs, k12;
    float mb0;
    __device__ __forceinline__ void set(const Ray &ray, float e) {
        const float a = dot(ray.d, ray.d), km = 2.6e-5f;
        ox = pk(ray.o.x); oy = pk(ray.o.y); oz = pk(ray.o.z); dx = pk(ray.d.x); dy = pk(ray.d.y); dz = pk(ray.d.z);
        ntime = pk(-ray.time); a_lo = pk(-(a - km)); a_hi = pk(a + km); m_abs = pk(2.5e5f * e * e + 1e-30f); k12 = pk(1e-12f);
        mb0 = -(2.f * e * e);
    }
    // min(E1, E3) of the pair: sphere A may be hit iff .x >= 0, sphere B iff .y >= 0
    __device__ __forceinline__ v2f pair(const float4 &r0, const float4 &r1, const float4 &r2, const float4 &r3) const {
        v2f cx_, cy_, cz_, r2_, mx_, my_, mz_;
        cx_.x = r0.x; cx_.y = r0.y; cy_.x = r0.z; cy_.y = r0.w; cz_.x = r1.x; cz_.y = r1.y; r2_.x = r1.z; r2_.y = r1.w;
        mx_.x = r2.x; mx_.y = r2.y; my_.x = r2.z; my_.y = r2.w; mz_.x = r3.x; mz_.y = r3.y;
        // (always: a static sphere has motion 0 and x - 0 * time == x; a uniform branch here became six selects)
        const v2f x = pkfma(mx_, ntime, ox - cx_), y = pkfma(my_, ntime, oy - cy_), z = pkfma(mz_, ntime, oz - cz_);
        const v2f bh = pkfma(dz, z, pkfma(dy, y, dx * x));
        const v2f S2 = pkfma(z, z, pkfma(y, y, x * x));
        const v2f e1 = pkfma(a_hi, r2_, pkfma(a_lo, S2, pkfma(bh, bh, m_abs)));
        v2f w;
        w.x = __builtin_fmaf(bh.x, fabsf(bh.x), mb0); w.y = __builtin_fmaf(bh.y, fabsf(bh.y), mb0);
        const v2f e3 = pkfma(S2, k12, -w);
        v2f r;
        r.x = fminf(e1.x, e3.x); r.y = fminf(e1.y, e3.y);
        return r;
    }
};
// The filter over ALL spheres (wave-uniform loop, scalar rows): bit i of word i / 32 = sphere i may be hit.  Up to 128 spheres.
template <class CX>
__device__ __forceinline__ void sphere_filter(const CX &cx, const Ray &ray, uint32_t (&mask)[4]) {
    cscene S = cx.S;
    cf4 rows = (cf4)S->tabs + S->tab_sfilter;
    const uint32_t pairs = S->sf_pairs;
    SphereFilterRay fr;
    fr.set(ray, cx.err_abs);
    // Two row sets in ping-pong, as in quad_filter: the scalar loads of the next pair overlap the arithmetic of this one.
    const float4 z4 = make_float4(0, 0, 0, 0);
    float4 a0 = z4, a1 = z4, a2 = z4, a3 = z4, b0 = z4, b1 = z4, b2 = z4, b3 = z4;
    auto fetch = [&](uint32_t p, float4 &q0, float4 &q1, float4 &q2, float4 &q3) {
        q0 = ld(rows, 4u * p); q1 = ld(rows, 4u * p + 1u); q2 = ld(rows, 4u * p + 2u); q3 = ld(rows, 4u * p + 3u);
    };
    if (pairs > 0u) fetch(0u, a0, a1, a2, a3);
    if (pairs > 1u) fetch(1u, b0, b1, b2, b3);
#pragma unroll
    for (uint32_t wd = 0; wd < 4u; ++wd) {  // 16 pairs = 32 spheres per mask word
        uint32_t acc = 0u;
        const uint32_t p1 = min(16u * wd + 16u, pairs);
        for (uint32_t p = 16u * wd; p < p1; p += 2u) {
            const uint32_t bit = 1u << ((2u * p) & 31u);
            v2f k = fr.pair(a0, a1, a2, a3);
            acc |= (k.x >= 0.f ? bit : 0u) | (k.y >= 0.f ? bit << 1 : 0u);
            if (p + 2u < pairs) fetch(p + 2u, a0, a1, a2, a3);
            if (p + 1u < p1) {
                k = fr.pair(b0, b1, b2, b3);
                acc |= (k.x >= 0.f ? bit << 2 : 0u) | (k.y >= 0.f ? bit << 3 : 0u);
                if (p + 3u < pairs) fetch(p + 3u, b0, b1, b2, b3);
            }
        }
        mask[wd] = acc;
    }
}

// Spheres then squares of Scene::computeIntersection (Scene.h:207-221).
template <class CX>
__device__ __forceinline__ Hit prims_hit(const CX &cx, const Ray &ray) {
    cscene S = cx.S;
    Hit h;
    h.kind = 0; h.index = 0; h.t = HRT_FLT_MAX; h.tri = 0; h.a0 = 0.f; h.a1 = 0.f;
    cf4 sph = (cf4)S->spheres;
    const uint32_t ns = S->n_spheres;
    if (CX::sphf && ns >= HRT_SPHERE_FILTER_MIN && ns <= 128u) {
        // FILTER (sphere_filter, above), then REFINE: every lane takes ITS candidate spheres through the exact arithmetic, in
        // index order with the reference's strict `<` (Scene.h:207-213), rows fetched per lane.
        uint32_t mask[4];
        sphere_filter(cx, ray, mask);
        const typename CX::tab4 rows = cx.ts;
#pragma unroll
        for (uint32_t wd = 0; wd < 4u; ++wd) {
            uint32_t cand = mask[wd];
            while (cand) {
                const uint32_t i = 32u * wd + (uint32_t)__builtin_ctz(cand);
                cand &= cand - 1u;
                float t;
                if (i < ns && sphere_t(ld(rows, 2 * i), ld(rows, 2 * i + 1), ray, t) && t < h.t && HRT_T_ACCEPT(t)) { h.kind = 1; h.index = i; h.t = t; }
            }
        }
    } else {
        for (uint32_t i = 0; i < ns; ++i) {
            float t;
            if (sphere_t(ld(sph, 2 * i), ld(sph, 2 * i + 1), ray, t) && t < h.t && HRT_T_ACCEPT(t)) { h.kind = 1; h.index = i; h.t = t; }
        }
    }
    STAMP(1);
    cf4 qd = (cf4)S->quads;
    const uint32_t nq = S->n_quads;
    if (!CX::exact && nq <= 64u) {
        // FILTER (wave-uniform loop, scalar rows, no division): an approximate t and inside test with
        // conservative error margins decide which quads can possibly be the closest accepted hit.
        // REFINE: only those (usually one per lane) go through the exact Square::intersect arithmetic,
        // in index order with the reference's strict `<`, so the selected hit is the reference's.
        const float tsure = h.t;  // upper bound on the exact t of a hit that certainly exists
        uint64_t cand = nq <= 32u ? (uint64_t)quad_filter<uint32_t>(cx, ray, tsure) : quad_filter<uint64_t>(cx, ray, tsure);
        STAMP(2);
        const typename CX::tab4 gq = cx.tq;
        while (cand) {
            const uint32_t i = (uint32_t)__builtin_ctzll(cand);
            cand &= cand - 1ull;
            float t, u, v;
            if (quad_t(gq + HRT_QUAD_ROWS * i, ray, h.t, t, u, v)) { h.kind = 2; h.index = i; h.t = t; h.a0 = u; h.a1 = v; }
        }
    } else {
        for (uint32_t i = 0; i < nq; ++i) {
            float t, u, v;
            if (quad_t(qd + HRT_QUAD_ROWS * i, ray, h.t, t, u, v)) { h.kind = 2; h.index = i; h.t = t; h.a0 = u; h.a1 = v; }
        }
    }
    return h;
}

// Which meshes' boxes the ray enters (bit i = mesh i; the first 32 meshes).
template <class CX>
__device__ __forceinline__ uint32_t mesh_gates(const CX &cx, const Ray &ray) {
    const uint32_t nm = min(cx.S->n_meshes, 32u);
    if (nm == 0) return 0u;
#ifdef HRT_ABL_NO_GATES  // ablation only (timing experiments, not parity-safe)
    return 0u;
#endif
    const f3 inv = ray_inv<CX::exact>(ray);
    uint32_t m = 0;
    for (uint32_t i = 0; i < nm; ++i)
        if (mesh_gate<CX::exact>((cmesh)cx.S->meshes + i, ray, inv)) m |= 1u << i;
    return m;
}

// mesh_gates with the count and the first mesh's box already in registers (loaded once per wave): for the
// common one-mesh scene no scalar load -- three dependent ones otherwise -- sits between the primitives and
// the vote of every bounce.
template <class CX>
__device__ __forceinline__ uint32_t mesh_gates_pre(const CX &cx, const Ray &ray, uint32_t nm, const GateBox &box0) {
    if (nm == 0) return 0u;
#ifdef HRT_ABL_NO_GATES  // ablation only
    return 0u;
#endif
    const f3 inv = ray_inv<CX::exact>(ray);
    uint32_t m = mesh_gate_box<CX::exact>(box0, ray, inv) ? 1u : 0u;
    for (uint32_t i = 1; i < nm; ++i)
        if (mesh_gate<CX::exact>((cmesh)cx.S->meshes + i, ray, inv)) m |= 1u << i;
    return m;
}

// The mesh loop of Scene::computeIntersection (Scene.h:222-228) over the meshes in `mask`.
template <class CX>
__device__ __forceinline__ void meshes_hit(const CX &cx, const Ray &ray, uint32_t mask, Hit &h) {
    const f3 inv = ray_inv<CX::exact>(ray);
    const uint32_t nm = min(cx.S->n_meshes, 32u);
    for (uint32_t i = 0; i < nm; ++i) {  // wave-uniform loop: scalar mesh records
        if (mask & (1u << i)) {
            float t, u, v;
            uint32_t tri;
#ifdef HRT_ABL_NO_WALK  // ablation only
            if (false) {
#else
            if (mesh_traverse(cx, (cmesh)cx.S->meshes + i, ray, inv, t, tri, u, v) && t < h.t && HRT_T_ACCEPT(t)) {
#endif
                h.kind = 3; h.index = i; h.t = t; h.tri = tri; h.a0 = u; h.a1 = v;
            }
        }
    }
}

// Scene::computeIntersection, Scene.h:202-230 (immediate form: AOV kernel).
template <class CX>
__device__ __forceinline__ Hit closest_hit(const CX &cx, const Ray &ray) {
    Hit h = prims_hit(cx, ray);
    const uint32_t m = mesh_gates(cx, ray);
    if (m) meshes_hit(cx, ray, m, h);
    return h;
}

// Every shadow ray of one shading point runs from p (plus 1e-5 along the ray) to a point within `reach` of the
// light centre, so it stays inside the capsule of radius `reach` around the segment [p, lpos].  A sphere whose
// centre is farther than radius + reach from that segment cannot give any of those rays a root, and the
// reference's loop would skip it without a draw.  Bit g of the result = some sphere of group g (DScene::sf_psize
// consecutive PAIRS of spheres) may be touched.  The test is a FILTER: its margin covers (a) its own fp32 error
// (perpendicular-vector form, no cancellation) and (b) the error of the exact test's discriminant, which
// cancels |oc|^2-sized terms (<= ~1e-6 |oc|^2 absolute; the margin allows 1e-5 |oc|^2).  shadow_sphere_groups takes two
// spheres at a time from the pair rows of sphere_filter (two-element vectors, compiled to scalar fp32 like sphere_filter).
__device__ __forceinline__ uint64_t shadow_sphere_groups_scalar(cscene S, f3 p, f3 lpos, float reach, float time, uint32_t gsize) {
    // one sphere at a time, groups of gsize consecutive SPHERES: the kernels without the pair filter (few spheres)
    cf4 sph = (cf4)S->spheres;
    const uint32_t ns = S->n_spheres;
    uint32_t g = 0, in_group = 0;
    const f3 ax = lpos - p;
    const float len2 = dot(ax, ax);
    const float inv_len2 = len2 > 0.f ? 1.f / len2 : 0.f;
    uint64_t groups = 0ull;
    for (uint32_t i = 0; i < ns; ++i) {  // wave-uniform: scalar rows
        const float4 r0 = ld(sph, 2 * i), r1 = ld(sph, 2 * i + 1);
        const f3 v = (mk(r0) + time * mk(r1)) - p;
        const float s = fminf(fmaxf(dot(v, ax) * inv_len2, 0.f), 1.f);
        const f3 w = v - s * ax;
        const float R = fabsf(r0.w) + reach;
        const float lim = R * R * 1.01f + 1e-5f * (1.f + dot(v, v) + len2);
        if (dot(w, w) <= lim) groups |= 1ull << g;
        if (++in_group == gsize) { in_group = 0; ++g; }
    }
    return groups;
}
__device__ __forceinline__ uint64_t shadow_sphere_groups(cscene S, f3 p, f3 lpos, float reach, float time) {
    cf4 rows = (cf4)S->tabs + S->tab_sfilter;
    const uint32_t pairs = S->sf_pairs, psize = S->sf_psize;
    const f3 ax = lpos - p;
    const float len2 = dot(ax, ax);
    const float inv_len2 = len2 > 0.f ? 1.f / len2 : 0.f;
    const v2f px = pk(p.x), py = pk(p.y), pz = pk(p.z), tm = pk(time), axx = pk(ax.x), axy = pk(ax.y), axz = pk(ax.z);
    const v2f nax = pk(-ax.x), nay = pk(-ax.y), naz = pk(-ax.z), il2 = pk(inv_len2), rch = pk(reach), c1 = pk(1e-5f * (1.f + len2));
    uint64_t groups = 0ull;
    for (uint32_t q = 0; q < pairs; ++q) {  // wave-uniform: scalar rows
        const float4 r0 = ld(rows, 4u * q), r1 = ld(rows, 4u * q + 1u), r2 = ld(rows, 4u * q + 2u), r3 = ld(rows, 4u * q + 3u);
        v2f cx_, cy_, cz_, mx_, my_, mz_, rr;
        cx_.x = r0.x; cx_.y = r0.y; cy_.x = r0.z; cy_.y = r0.w; cz_.x = r1.x; cz_.y = r1.y;
        mx_.x = r2.x; mx_.y = r2.y; my_.x = r2.z; my_.y = r2.w; mz_.x = r3.x; mz_.y = r3.y; rr.x = r3.z; rr.y = r3.w;
        const v2f vx = pkfma(mx_, tm, cx_ - px), vy = pkfma(my_, tm, cy_ - py), vz = pkfma(mz_, tm, cz_ - pz);
        v2f sp = pkfma(vz, axz, pkfma(vy, axy, vx * axx)) * il2;
        sp.x = fminf(fmaxf(sp.x, 0.f), 1.f); sp.y = fminf(fmaxf(sp.y, 0.f), 1.f);
        const v2f wx = pkfma(nax, sp, vx), wy = pkfma(nay, sp, vy), wz = pkfma(naz, sp, vz);
        const v2f ww = pkfma(wz, wz, pkfma(wy, wy, wx * wx)), vv = pkfma(vz, vz, pkfma(vy, vy, vx * vx));
        const v2f R = rr + rch;
        const v2f lim = pkfma(R * R, pk(1.01f), pkfma(vv, pk(1e-5f), c1));
        if ((ww.x <= lim.x) | (ww.y <= lim.y)) groups |= 1ull << (q / psize);
    }
    return groups;
}

// Scene::computeShadow, Scene.h:235-255: candidates in object order, each lets the ray
// through with probability `transparency` (one draw per candidate).
// Only the sphere groups in `groups` are tested (shadow_sphere_groups: no other sphere can be reached), in ascending
// index order, so the draws fall exactly where the reference's full loop puts them; `filter`: the pairs of a group first go
// through the pair filter of sphere_filter (rows per lane) and only a sphere it keeps through the exact arithmetic.
template <class CX>
__device__ __forceinline__ bool shadow_blocked(const CX &cx, const Ray &ray, float tmax, Rng &rng, uint64_t groups, bool filter) {
    cscene S = cx.S;
    const typename CX::tab4 sph = cx.ts, mats = cx.tm, frows = cx.tsf;
    const uint32_t ns = S->n_spheres, psize = S->sf_psize, pairs = S->sf_pairs;
    if (!CX::sphf) {  // groups of gsize consecutive spheres, every one through the exact arithmetic
        const uint32_t gsize = (ns + 63u) / 64u;
        while (groups) {
            const uint32_t i0 = (uint32_t)__builtin_ctzll(groups) * gsize, i1 = min(i0 + gsize, ns);
            groups &= groups - 1ull;
            for (uint32_t i = i0; i < i1; ++i) {
                float t;
                const float4 r1 = ld(sph, 2 * i + 1);
                if (sphere_t(ld(sph, 2 * i), r1, ray, t) && t < tmax && HRT_T_ACCEPT(t)) {
                    const float transparency = ld(mats, HRT_MAT_ROWS * __float_as_uint(r1.w)).w;
                    if (rng.next() > transparency) return true;
                }
            }
        }
    }
    SphereFilterRay fr;
    if (CX::sphf && filter) fr.set(ray, cx.err_abs);
    while (CX::sphf && groups) {
        const uint32_t q0 = (uint32_t)__builtin_ctzll(groups) * psize, q1 = min(q0 + psize, pairs);
        groups &= groups - 1ull;
        for (uint32_t q = q0; q < q1; ++q) {
            bool keep_a = true, keep_b = true;
            if (CX::sphf && filter) {
                const v2f k = fr.pair(ld(frows, 4u * q), ld(frows, 4u * q + 1u), ld(frows, 4u * q + 2u), ld(frows, 4u * q + 3u));
                keep_a = k.x >= 0.f; keep_b = k.y >= 0.f;
            }
#pragma unroll
            for (uint32_t half = 0; half < 2u; ++half) {
                const uint32_t i = 2u * q + half;
                if (!(half ? keep_b : keep_a) || i >= ns) continue;
                float t;
                const float4 r1 = ld(sph, 2 * i + 1);
                if (sphere_t(ld(sph, 2 * i), r1, ray, t) && t < tmax && HRT_T_ACCEPT(t)) {
                    const float transparency = ld(mats, HRT_MAT_ROWS * __float_as_uint(r1.w)).w;
                    if (rng.next() > transparency) return true;
                }
            }
        }
    }
    cf4 qd = (cf4)S->quads;
    const uint32_t nq = S->n_quads;
    for (uint32_t i = 0; i < nq; ++i) {
        float t, u, v;
        if (quad_t(qd + HRT_QUAD_ROWS * i, ray, tmax, t, u, v)) {
            const float transparency = ld(mats, HRT_MAT_ROWS * __float_as_uint(ld(qd, HRT_QUAD_ROWS * i + 4).w)).w;
            if (rng.next() > transparency) return true;
        }
    }
#ifdef HRT_ABL_NO_SHADOW_MESH  // ablation only
    const uint32_t nm = 0;
#else
    const uint32_t nm = min(S->n_meshes, 32u);
#endif
    if (nm) {
        const f3 inv = ray_inv<CX::exact>(ray);
        for (uint32_t i = 0; i < nm; ++i) {
            cmesh M = (cmesh)S->meshes + i;
            float t, u, v;
            uint32_t tri;
            if (mesh_gate<CX::exact>(M, ray, inv) && mesh_traverse(cx, M, ray, inv, t, tri, u, v) && t < tmax && HRT_T_ACCEPT(t)) {
                const float transparency = ld(mats, HRT_MAT_ROWS * M->material).w;
                if (rng.next() > transparency) return true;
            }
        }
    }
    return false;
}

// ------------------------------------------------------------------ materials
// rows: 0 {albedo.xyz, transparency} 1 {index_medium, type, texture_type, emissive}
//       2 {checker1.xyz, scale_x} 3 {checker2.xyz, scale_y} 4 {light_color.xyz, intensity}
//       5 {image, normal_map, -, -}
// geo: row 6 (texture) or 7 (normal map) of the material = {texel offset, w, h, -} as integers, copied there by the host so
// that no lane has to chase the image table
__device__ __forceinline__ uint32_t texel_index(const float4 geo, float u, float v, float sx, float sy) {
    const int iw = (int)__float_as_uint(geo.y), ih = (int)__float_as_uint(geo.z);
    float uu = u * sx, vv = v * sy;
    uu = uu - truncf(uu);            // (float)fmod((double)(u*sx), 1.): exact
    vv = 1.f - (vv - truncf(vv));    // (float)(1 - fmod(...)): one correctly rounded subtraction either way
    const int x = (int)(uu * (float)(iw - 1));
    const int y = (int)(vv * (float)(ih - 1));
    return __float_as_uint(geo.x) + (uint32_t)(y * iw + x);
}
__device__ __forceinline__ uint32_t texel(cscene S, const float4 geo, float u, float v, float sx, float sy) {
    return ((gu1)S->texels)[texel_index(geo, u, v, sx, sy)];
}
template <class LP>
__device__ __forceinline__ f3 unit_rgb(LP lut, uint32_t px) {  // c/255. in double, narrowed (Material.cpp:87)
    return mk(lut[px & 255u], lut[(px >> 8) & 255u], lut[(px >> 16) & 255u]);
}

// Material::texture, Material.cpp:63-92
template <class LP, class MP>
__device__ __forceinline__ f3 mat_texture(cscene S, LP lut, MP m, uint32_t tex_type, f3 color, float u, float v) {
    if (tex_type == 1u) {
        const float4 c1 = ld(m, 2), c2 = ld(m, 3);
        color = ((int)(u * c1.w) % 2 == (int)(v * c2.w) % 2) ? mk(c1) : mk(c2);
    } else if (tex_type == 2u) {
        const float4 geo = ld(m, 6);
        const bool empty = (int)__float_as_uint(ld(m, 5).x) < 0 || (int)__float_as_uint(geo.y) < 1 || (int)__float_as_uint(geo.z) < 1;
        if (empty) {
            color = ((int)((double)u * 8.) % 2 == (int)((double)v * 8.) % 2) ? mk(0.f, 0.f, 0.f) : mk(1.f, 0.f, 1.f);
        } else {
            color = unit_rgb(lut, texel(S, geo, u, v, ld(m, 2).w, ld(m, 3).w));
        }
    }
    return color;
}

// Material::emit, Material.cpp:13-24
template <class LP, class MP>
__device__ __forceinline__ f3 mat_emit(cscene S, LP lut, MP m, uint32_t tex_type, bool emissive, float u, float v) {
    if (!emissive) return mk(0.f, 0.f, 0.f);
    const float4 lc = ld(m, 4);
    f3 c = mk(0.f, 0.f, 0.f);
    if (tex_type == 0u) c = mk(lc);
    else c = mat_texture(S, lut, m, tex_type, c, u, v);
    return c * lc.w;
}

// Sphere.h:129-130 (fp64 libm as the reference)
__device__ __forceinline__ void sphere_angles(f3 n, float &theta, float &phi) {
    theta = (float)acos((double)n.y * -1.);
    phi = (float)(atan2((double)n.z * -1., (double)n.x) + 3.14159265358979323846);
}

struct Surface {
    f3 p, n, albedo, emission;
    float transparency, index_medium;
    uint32_t type;
};

// The hit-dependent part of Scene::rayTraceRecursive, Scene.h:270-300.
template <class CX>
__device__ __forceinline__ Surface shade(const CX &cx, const Ray &ray, const Hit &h) {
    typedef typename CX::tab4 tab4;
    cscene S = cx.S;
    Surface sf;
    const tab4 mats = cx.tm;
    const f3 p = ray.o + h.t * ray.d;
    sf.p = p;
    sf.emission = mk(0.f, 0.f, 0.f);
    uint32_t mat_id;
    if (h.kind == 1u) {
        const tab4 sp = cx.ts;
        const float4 r0 = ld(sp, 2 * h.index), r1 = ld(sp, 2 * h.index + 1);
        mat_id = __float_as_uint(r1.w);
        const tab4 m = mats + HRT_MAT_ROWS * mat_id;
        const float4 m0 = ld(m, 0), m1 = ld(m, 1);
        const uint32_t tex_type = __float_as_uint(m1.z);
        const bool emissive = __float_as_uint(m1.w) != 0u;
        const f3 c = mk(r0) + ray.time * mk(r1);
        sf.n = normalize(p - c);
        sf.albedo = mk(m0);
        if (tex_type != 0u || emissive) {  // Sphere.h:129-130, Scene.h:275-277 (fp64 libm as the reference)
            float theta, phi;
            sphere_angles(sf.n, theta, phi);
            const float u = (float)((double)phi / (2 * 3.14159265358979323846));
            const float v = (float)((double)theta / 3.14159265358979323846);
            sf.albedo = mat_texture(S, cx.lut, m, tex_type, sf.albedo, u, v);
            sf.emission = mat_emit(S, cx.lut, m, tex_type, emissive, u, v);
        }
    } else if (h.kind == 2u) {
        const tab4 q = cx.tq + HRT_QUAD_ROWS * h.index;
        mat_id = __float_as_uint(ld(q, 4).w);
        const tab4 m = mats + HRT_MAT_ROWS * mat_id;
        const float4 m0 = ld(m, 0), m1 = ld(m, 1);
        const uint32_t tex_type = __float_as_uint(m1.z);
        const bool emissive = __float_as_uint(m1.w) != 0u;
        sf.n = mk(ld(q, 1));
        // The colour texel and the normal-map texel are REQUESTED TOGETHER (two masked loads, nothing between them that needs
        // either), then used: mat_texture() followed by the normal-map branch made them two memory round trips in a row on
        // every textured, normal-mapped wall.  Same expressions, same values.
        const float4 m2 = ld(m, 2), m3 = ld(m, 3), m5 = ld(m, 5);
        const int nmap = (int)__float_as_uint(m5.y);
        bool image = false;
        uint32_t px_c = 0u, px_n = 0u;
        if (tex_type == 2u) {
            const float4 geo = ld(m, 6);
            image = !((int)__float_as_uint(m5.x) < 0 || (int)__float_as_uint(geo.y) < 1 || (int)__float_as_uint(geo.z) < 1);
            if (image) px_c = texel(S, geo, h.a0, h.a1, m2.w, m3.w);
        }
        if (nmap >= 0) px_n = texel(S, ld(m, 7), h.a0, h.a1, m2.w, m3.w);
        f3 tex = mk(m0);  // Material::texture, Material.cpp:62-112 (mat_texture above, with the texel already on its way)
        if (tex_type == 1u) tex = ((int)(h.a0 * m2.w) % 2 == (int)(h.a1 * m3.w) % 2) ? mk(m2) : mk(m3);
        else if (tex_type == 2u)
            tex = image ? unit_rgb(cx.lut, px_c) : (((int)((double)h.a0 * 8.) % 2 == (int)((double)h.a1 * 8.) % 2) ? mk(0.f, 0.f, 0.f) : mk(1.f, 0.f, 1.f));
        sf.albedo = tex;
        if (nmap >= 0) {  // Material::get_normal, Material.cpp:114-130
            const float nx = cx.lut[256u + (px_n & 255u)], ny = cx.lut[256u + ((px_n >> 8) & 255u)], nz = cx.lut[256u + ((px_n >> 16) & 255u)];
            sf.n = normalize(nx * mk(ld(q, 5)) + ny * mk(ld(q, 6)) + nz * sf.n);
        }
        if (emissive) {  // mat_emit: the light colour, or the texture value at the same (u, v) -- the texel fetched above -- times the intensity
            const float4 lc = ld(m, 4);
            sf.emission = (tex_type == 0u ? mk(lc) : (tex_type <= 2u ? tex : mk(0.f, 0.f, 0.f))) * lc.w;
        }
    } else {
        const typename CX::tabmesh M = cx.tmesh + h.index;  // per-lane mesh record
        mat_id = M->material;
        const float4 m0 = ld(mats, HRT_MAT_ROWS * mat_id);
        const float4 r0 = ld((gf4)S->tris, HRT_TRI_ROWS * h.tri + 3u), r3 = ld((gf4)S->tri_planes, h.tri);  // r0.x: the triangle's id
        sf.n = mk(r3);  // Triangle.h:32-37 flat normal, folded on the host
        sf.albedo = mk(m0);
        const uint32_t tid = __float_as_uint(r0.x);
        const int ct = M->color_type;
        gf4 colors = (gf4)S->colors;
        if (ct == 1) {
            sf.albedo = mk(ld(colors, M->color_base + tid));
        } else if (ct == 0) {
            const uint4 vi = ld((gu4)S->tri_vids, M->color_base + tid);
            const uint32_t vb = M->vcolor_base;
            const float w1 = h.a0, w2 = h.a1, w0 = 1 - w1 - w2;
            sf.albedo = w0 * mk(ld(colors, vb + vi.x)) + w1 * mk(ld(colors, vb + vi.y)) + w2 * mk(ld(colors, vb + vi.z));
        }
    }
    const tab4 m = mats + HRT_MAT_ROWS * mat_id;
    const float4 m0 = ld(m, 0), m1 = ld(m, 1);
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
template <class CX>
__device__ __forceinline__ f3 sky(const CX &cx, f3 dir, int remaining) {
    cscene S = cx.S;
    const int sb = S->skybox_image;
    if (sb < 0) {
        if (S->dark_sky) return mk(0.f, 0.f, 0.f);
        const float a = (float)(0.5 * ((double)dir.y + 1.0));
        return (float)(1.0 - (double)a) * mk(1.f, 1.f, 1.f) + (a * mk(0.5f, 0.7f, 1.0f)) * (float)(remaining + 1);
    }
    gimg im = (gimg)S->images + sb;
    const int iw = im->w, ih = im->h;
    const float u = (float)(0.5 + atan2((double)dir.z, (double)dir.x) / (2 * 3.14159265358979323846));
    const float v = (float)(0.5 - asin((double)dir.y) / 3.14159265358979323846);
    int x = (int)(u * (float)iw), y = (int)(v * (float)ih);
    x = min(x, iw - 1);  // the reference reads out of bounds at u == 1; clamped (as the oracle)
    y = min(y, ih - 1);
    return unit_rgb(cx.lut, ((gu1)S->texels)[im->offset + (uint32_t)(y * iw + x)]) * (float)remaining;
}

// Direct light with soft shadows, Scene.h:305-334.
template <class CX>
__device__ __forceinline__ f3 direct_light(const CX &cx, const Surface &sf, const Ray &ray, Rng &rng) {
    f3 color = mk(0.f, 0.f, 0.f);
    cf4 L = (cf4)cx.S->lights;
    const uint32_t nl = cx.S->n_lights;
    for (uint32_t i = 0; i < nl; ++i) {
        const float4 l0 = ld(L, 2 * i);
        const f3 lpos = mk(l0);
        const f3 Ld = normalize(lpos - sf.p);
        const float dotLN = dot(Ld, sf.n);
        color = color + ((mk(ld(L, 1)) * sf.albedo) * fmaxf(0.0f, dotLN)) * (float)(1. - (double)sf.transparency);  // lights[0] (N2)
        int blocked = 0;
        const float delta = l0.w / 2.f;
        const bool cull = !(CX::exact || (cx.flags & HRT_FLAG_NO_SHADOW_CULL));
        const uint64_t groups = !cull ? ~0ull  // every group, every sphere: the reference's full loop (tests compare the two bit for bit)
                                      : (CX::sphf ? shadow_sphere_groups(cx.S, sf.p, lpos, fabsf(delta) * 1.0001f + 1e-6f, ray.time)
                                                  : shadow_sphere_groups_scalar(cx.S, sf.p, lpos, fabsf(delta) * 1.0001f + 1e-6f, ray.time, (cx.S->n_spheres + 63u) / 64u));
        for (int j = 0; j < 10; ++j) {  // NB_ECH
            const f3 lp = lpos + rng.unit_vector() * delta;
            const f3 to = lp - sf.p;
            const f3 Ls = normalize(to);
            const float tLight = length(to);
            Ray sr;
            sr.o = sf.p + Ls * HRT_EPS;
            sr.d = normalize(Ls);
            sr.time = ray.time;
            if (shadow_blocked(cx, sr, tLight, rng, groups, cull)) blocked++;
        }
        const float shadow = (float)(1. - (double)((float)blocked / 10.f));
        color = color * shadow;  // the running sum, earlier lights included (N3)
    }
    return color;
}

// matrixUtilities.h:53-74 with the two inverse matrices supplied by the host (fp64, column-major).
template <class M>
__device__ __forceinline__ void mult4(M m, double x, double y, double z, double w, double *r) {
    r[0] = m[0] * x + m[4] * y + m[8] * z + m[12] * w;
    r[1] = m[1] * x + m[5] * y + m[9] * z + m[13] * w;
    r[2] = m[2] * x + m[6] * y + m[10] * z + m[14] * w;
    r[3] = m[3] * x + m[7] * y + m[11] * z + m[15] * w;
}
// (float)(r / pi15) without the fp64 division: q = r * (1/pi15) is within 3.4e-16 |q| of the correctly
// rounded quotient; if the ends of that interval narrow to the same float, so does the quotient
// (rounding is monotonic).  Otherwise (about 1e-8 of the calls) the exact division is done.
__device__ __forceinline__ float div_narrow(double r, double d, double inv_d) {
    const double q = r * inv_d;
    const float lo = (float)(q * (1.0 - 8.881784197001252e-16)), hi = (float)(q * (1.0 + 8.881784197001252e-16));  // 1 -+ 2^-50
    return (lo == hi) ? lo : (float)(r / d);
}
template <bool EXACT = false>
__device__ __forceinline__ Ray camera_ray(ccam C, float u, float v, float time) {
    const double ri0 = C->pi0 * (2.0 * (double)u - 1.0);
    const double ri1 = C->pi5 * -(2.0 * (double)v - 1.0);
    const double d = C->pi15, inv_d = C->inv15;
    const double r0 = ((C->mx[0] * ri0 + C->my[0] * ri1) + C->c1[0]) + C->c2[0];
    const double r1 = ((C->mx[1] * ri0 + C->my[1] * ri1) + C->c1[1]) + C->c2[1];
    const double r2 = ((C->mx[2] * ri0 + C->my[2] * ri1) + C->c1[2]) + C->c2[2];
    const f3 world = EXACT ? mk((float)(r0 / d), (float)(r1 / d), (float)(r2 / d))  // matrixUtilities.h:66-68 as written
                           : mk(div_narrow(r0, d, inv_d), div_narrow(r1, d, inv_d), div_narrow(r2, d, inv_d));
    Ray out;
    out.o = mk(C->eye[0], C->eye[1], C->eye[2]);
    // normalised twice, as the reference does: once in screen_space_to_world_space_ray
    // (matrixUtilities.h:73) and again by the Ray constructor (main.cpp:192, Line.h:15)
    out.d = normalize(normalize(world - out.o));
    out.time = time;
    return out;
}

// ---------------------------------------------------------------------------
// The megakernel body.  LIGHTS = the scene has point lights (soft-shadow fan-out, Scene.h:305-334,
// compiled in); scenes without lights run the variant that carries none of that code or its registers.
// ---------------------------------------------------------------------------
template <bool LIGHTS, bool EXACT = false>
__device__ __forceinline__ void trace_body(const DRender &R) {
    extern __shared__ uint4 s_units[];
    CtxT<EXACT> cx;
    cx.S = (cscene)R.scene;
    cx.set_tables((gf4)cx.S->tabs, (gf1)c_u8_lut, cx.S);
    cx.lds = (lu4)s_units;
    cx.lds_n = R.lds_units;
    cx.err_abs = R.err_abs;
    cx.flags = R.flags;
    ccam cam = (ccam)R.cam;
    {
        gu4 g_units = (gu4)cx.S->kd_units;
        for (uint32_t i = threadIdx.x; i < cx.lds_n; i += blockDim.x) s_units[i] = ld(g_units, i);
    }
    unsigned long long stamps_local[17];
    cx.st = stamps_local;
#ifdef HRT_STAMPS
    for (int k = 0; k < 16; ++k) stamps_local[k] = 0ull;
    stamps_local[16] = __builtin_readcyclecounter();
#endif
    __syncthreads();
    const bool has_mesh = cx.S->n_meshes != 0u;
    // exact path pruning, as in the streaming kernel (hrt_stream.hip HRT_SP_PRUNE; never in the proof builds)
    const bool prune = !EXACT && cx.S->prune_ok != 0u;
    const bool sky_is_zero = cx.S->skybox_image < 0 && cx.S->dark_sky != 0;

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
        if (R.accumulate && inside) {  // progressive mode: continue the running sum of samples [0, s0)
            const float *a = R.out_tiles + ((size_t)j * 64u + lane) * 3u;
            sum = mk(a[0], a[1], a[2]);
        }
        uint32_t s = 0;          // next sample of this lane's pixel
        int remaining = 0;       // bounces left on the current path; 0 = needs a new path
        Ray ray;
        ray.o = mk(0.f, 0.f, 0.f); ray.d = mk(0.f, 0.f, 1.f); ray.time = 0.f;
        f3 thr = mk(1.f, 1.f, 1.f), rad = mk(0.f, 0.f, 0.f);
        Rng rng;
        rng.k0 = rng.k1 = rng.i = 0;
        Hit h;
        h.kind = 0; h.index = 0; h.t = HRT_FLT_MAX; h.tri = 0; h.a0 = 0.f; h.a1 = 0.f;
        uint32_t parked = 0;     // meshes whose box this lane's ray enters and that are still to be walked
        uint32_t stage = 0;      // 0: needs stage A   1: parked for stage B   2: ready for stage C
        bool live = inside && R.spp > 0;

        while (__ballot(live) != 0ull) {
            // ---- stage A: (re)generate, spheres + squares, mesh gates
            if (live && stage == 0u) {
                if (remaining == 0) {  // next camera sample of this pixel (main.cpp:188-192)
                    rng.start(R.seed_lo, R.seed_hi, pixel, R.s0 + s);
                    const float u = ((float)px + rng.next()) / (float)R.w;
                    const float v = ((float)py + rng.next()) / (float)R.h;
                    const float tm = rng.next();
                    ray = camera_ray<EXACT>(cam, u, v, tm);
                    thr = mk(1.f, 1.f, 1.f);
                    rad = mk(0.f, 0.f, 0.f);
                    remaining = 6;  // MAXBOUNCES
                }
                STAMP(0);
                h = prims_hit(cx, ray);
                STAMP(3);
                parked = has_mesh ? mesh_gates(cx, ray) : 0u;
#ifdef HRT_ABL_GATES_IGNORED  // ablation only: pay for the gates, never park
                asm volatile("" : "+v"(parked));
                parked &= 0x80000000u;
#endif
                stage = parked ? 1u : 2u;
                if (!LIGHTS && prune && remaining == 1) {  // the path's last segment: only an emitting closest hit can still reach the sample
                    bool dead;
                    if (h.kind == 0u) {
                        dead = sky_is_zero;
                    } else {
                        const uint32_t mat = h.kind == 1u ? __float_as_uint(ld(cx.ts, HRT_SPHERE_ROWS * h.index + 1u).w)
                                                          : __float_as_uint(ld(cx.tq, HRT_QUAD_ROWS * h.index + 4u).w);
                        dead = __float_as_uint(ld(cx.tm, HRT_MAT_ROWS * mat + 1u).w) == 0u;
                    }
                    if (dead) { h.kind = 0u; parked = 0u; stage = 3u; }  // ends below with the radiance it has
                }
            }
            STAMP(4);
            // ---- stage B: walk the meshes for the parked lanes once enough of them have gathered
            if (has_mesh) {
                const uint64_t waiting = __ballot(live && stage == 1u);
                const uint64_t ready = __ballot(live && stage >= 2u);
                if (waiting != 0ull && (__popcll(waiting) >= HRT_MESH_BATCH || ready == 0ull)) {
                    if (live && stage == 1u) {
                        meshes_hit(cx, ray, parked, h);
                        stage = 2u;
                    }
                }
            }
            STAMP(5);
            // ---- stage C: shade, scatter, end of path
            if (live && stage >= 2u) {
                bool ended;
                if (stage == 3u) {
                    ended = true;  // pruned on its last segment (stage A): nothing is added
                } else if (h.kind == 0u) {
                    rad = rad + thr * sky(cx, ray.d, remaining);
                    ended = true;
                } else {
                    const Surface sf = shade(cx, ray, h);
                    STAMP(6);
                    f3 direct = mk(0.f, 0.f, 0.f);
                    if (LIGHTS) direct = direct_light(cx, sf, ray, rng);
                    STAMP(7);
                    rad = rad + thr * (direct + sf.emission);
                    thr = thr * sf.albedo;
                    scatter(sf, ray, rng);
                    STAMP(8);
                    --remaining;
                    ended = (remaining == 0) || (prune && thr.x == 0.f && thr.y == 0.f && thr.z == 0.f);
                }
                if (ended) {
                    sum = sum + mk(rad.x / 6.f, rad.y / 6.f, rad.z / 6.f);  // Scene.h:348
                    remaining = 0;
                    ++s;
                    live = s < R.spp;
                }
                stage = 0u;
            }
            STAMP(9);
        }
        float *o = R.out_tiles + ((size_t)j * 64u + lane) * 3u;
        f3 c = mk(0.f, 0.f, 0.f);
        if (inside) {
            const float nspp = R.accumulate ? 1.f : (float)R.spp;  // progressive mode stores the raw sum (x / 1.f is exact)
            c = mk(sum.x / nspp, sum.y / nspp, sum.z / nspp);  // main.cpp:195
        }
        o[0] = c.x; o[1] = c.y; o[2] = c.z;
        STAMP(10);
    }
#ifdef HRT_STAMPS
    if ((threadIdx.x & 63u) == 0u && R.stamps)
        for (int k = 0; k < 16; ++k) atomicAdd(R.stamps + k, stamps_local[k]);
#endif
}

}  // namespace hrtk

using namespace hrtk;

extern "C" __global__ void __launch_bounds__(HRT_WG, HRT_MIN_WAVES_SINGLE) hrt_trace_kernel(const DRender R) { trace_body<false>(R); }
extern "C" __global__ void __launch_bounds__(HRT_WG, HRT_MIN_WAVES_SINGLE_LIGHTS) hrt_trace_kernel_lights(const DRender R) { trace_body<true>(R); }
// HRT_FLAG_EXACT_ONLY: the proof builds (no filters, no v_rcp_f32; see CtxT).  Not tuned: 2 waves per SIMD.
extern "C" __global__ void __launch_bounds__(HRT_WG, 2) hrt_trace_kernel_exact(const DRender R) { trace_body<false, true>(R); }
extern "C" __global__ void __launch_bounds__(HRT_WG, 2) hrt_trace_kernel_lights_exact(const DRender R) { trace_body<true, true>(R); }

// gamma_correct (Functions.cpp:56-60): pow(c, 1/2.2) in double, over this rank's tile buffer.  Kept out of
// the megakernel: fp64 pow is register-hungry and runs once per pixel.
extern "C" __global__ void hrt_gamma_kernel(float *__restrict__ v, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (float)pow((double)v[i], 1.0 / 2.2);
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
    Ctx cx;
    cx.S = (cscene)R.scene;
    cx.set_tables((gf4)cx.S->tabs, (gf1)c_u8_lut, cx.S);
    cx.lds = (lu4) nullptr;
    cx.lds_n = 0;  // every nodelet from global memory here
    cx.err_abs = R.err_abs;
    cx.flags = R.flags;
    unsigned long long stamps_local[17] = {0};
    cx.st = stamps_local;
    const Ray ray = camera_ray((ccam)R.cam, ((float)x + 0.5f) / (float)R.w, ((float)y + 0.5f) / (float)R.h, 0.f);
    const Hit h = closest_hit(cx, ray);
    f3 o = mk(0.f, 0.f, 0.f);
    if (which == 0u) {
        float id = -1.f;
        if (h.kind == 3u) id = (float)__float_as_uint(ld((gf4)cx.S->tris, HRT_TRI_ROWS * h.tri + 3u).x);
        else if (h.kind) id = (float)h.index;
        o = mk(h.kind ? h.t : 0.f, (float)h.kind, id);
    } else if (h.kind) {
        const Surface sf = shade(cx, ray, h);
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
