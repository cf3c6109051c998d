// libhrt.so -- C ABI (include/hrt.h) over the HIP kernels in hrt_kernels.hip.
// gfx950 only.  No CPU fallback: every entry point that needs the GPU fails
// with HRT_ERR_DEVICE when HIP cannot provide one.
#include "hrt_kernels.hip"
#include "hrt_dual.hip"
#include "hrt_stream.hip"
#include "hrt_output.hip"
#include "hrt_kat.hip"

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl.so is opened with dlopen by hrt_multi_create (hrt_multi.hip)

#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_error;
int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(HRT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

struct Runtime {
    bool ready = false;
    int device = -1;
    int cus = 0;
    int blocks_per_cu = 0;
    uint32_t lds_budget = 0;  // bytes of dynamic LDS per workgroup for nodelets
    bool use_dual = true;     // two pixel streams per lane (hrt_dual.hip) for scenes with meshes; HRT_KERNEL=single turns it off
    int use_stream = -1;      // workgroup-streaming kernel (hrt_stream.hip): -1 where it pays (default), 1 always (HRT_KERNEL=stream), 0 never
    hipFuncAttributes attr{};
    std::vector<int> dev_cus;  // per ordinal: CUs of the devices hrt_init has prepared (0 = not prepared); one process may drive several
} g_rt;

// Makes `ordinal` (a device hrt_init has prepared) the current one for this thread and for the launch geometry.
int use_device(int ordinal) {
    if (ordinal < 0 || ordinal >= (int)g_rt.dev_cus.size() || g_rt.dev_cus[ordinal] == 0)
        return fail(HRT_ERR_STATE, "device " + std::to_string(ordinal) + " has not been initialised (hrt_init)");
    HIP_TRY(hipSetDevice(ordinal));
    g_rt.device = ordinal;
    g_rt.cus = g_rt.dev_cus[ordinal];
    return HRT_OK;
}

float as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

struct H3 { float x, y, z; };
// host fp32 helpers in the reference's evaluation order (no contraction): these fold the
// per-quad constants exactly as Square::intersect would compute them per call.
H3 h_sub(H3 a, H3 b) {
#pragma clang fp contract(off)
    return H3{a.x - b.x, a.y - b.y, a.z - b.z};
}
H3 h_cross(H3 a, H3 b) {
#pragma clang fp contract(off)
    return H3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
float h_dot(H3 a, H3 b) {
#pragma clang fp contract(off)
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
float h_msub(float a, float b, float c, float d) {  // a*b - c*d, each product rounded first
#pragma clang fp contract(off)
    return a * b - c * d;
}
double h_mul64(double a, double b) {
#pragma clang fp contract(off)
    return a * b;
}
double h_neg_dot3(const double r[3], const float e[3]) {  // -(r . e), left to right, no contraction (as oracle/oracle.cpp builds it)
#pragma clang fp contract(off)
    return -(r[0] * (double)e[0] + r[1] * (double)e[1] + r[2] * (double)e[2]);
}
// gluInvertMatrix (matrixUtilities.h:77-206): adjugate over determinant.  Each adjugate entry is six signed triple
// products ((s*m[a]) * m[b]) * m[c] summed in the order listed, which is the order the reference writes them in, so
// every entry rounds as the reference's.
static const signed char k_adjugate[16][6][4] = {
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
bool host_invert4(const double m[16], double out[16]) {
#pragma clang fp contract(off)
    double adj[16];
    for (int e = 0; e < 16; ++e) {
        double sum = 0.0;
        for (int k = 0; k < 6; ++k) {
            const signed char *t = k_adjugate[e][k];
            const double term = ((t[0] < 0 ? -m[t[1]] : m[t[1]]) * m[t[2]]) * m[t[3]];
            sum = k == 0 ? term : sum + term;
        }
        adj[e] = sum;
    }
    double det = m[0] * adj[0] + m[1] * adj[4] + m[2] * adj[8] + m[3] * adj[12];
    if (det == 0) return false;
    det = 1.0 / det;
    for (int e = 0; e < 16; ++e) out[e] = adj[e] * det;
    return true;
}
float h_len(H3 a) { return (float)std::sqrt((double)h_dot(a, a)); }
H3 h_normalize(H3 a) {
    float L = h_len(a);
    return H3{a.x / L, a.y / L, a.z / L};
}

// Square::intersect's per-call constants (Square.h:66-72: edges, normal, |R|, |U|, D0), computed once in the same fp32
// arithmetic -> the 11 float4 rows of hrt_device.h.
void fold_quad(const hrt_quad &q, const hrt_material &m, std::vector<float4> &quads) {
    const H3 v0{q.v0[0], q.v0[1], q.v0[2]}, v1{q.v1[0], q.v1[1], q.v1[2]}, v3{q.v3[0], q.v3[1], q.v3[2]};
    const H3 R = h_sub(v1, v0), U = h_sub(v3, v0);
    const H3 n = h_normalize(h_cross(R, U));
    uint32_t flags = 0;
    if (m.type == HRT_MAT_GLASS) flags |= HRT_QUAD_FLAG_GLASS;
    if (m.motion[0] != 0.f || m.motion[1] != 0.f || m.motion[2] != 0.f) flags |= HRT_QUAD_FLAG_MOVING;
    quads.push_back(make_float4(v0.x, v0.y, v0.z, h_dot(v0, n)));
    quads.push_back(make_float4(n.x, n.y, n.z, as_float(flags)));
    quads.push_back(make_float4(R.x, R.y, R.z, h_len(R)));
    quads.push_back(make_float4(U.x, U.y, U.z, h_len(U)));
    quads.push_back(make_float4(m.motion[0], m.motion[1], m.motion[2], as_float((uint32_t)q.material)));
    quads.push_back(make_float4(q.tangent[0], q.tangent[1], q.tangent[2], 0.f));
    quads.push_back(make_float4(q.bitangent[0], q.bitangent[1], q.bitangent[2], 0.f));
}

// The rows of the squares' no-division filter (hrt_device.h DScene::qfilter; hrt_kernels.hip quad_filter), from the folded
// rows: sections for static squares lying (nearly) in an axis plane, by normal axis, then all others.  The constants of
// the axis form and the error analysis behind them are stated at quad_filter_axis.
void build_quad_filter(const std::vector<float4> &quads, uint32_t nq, std::vector<float4> &qf, uint32_t count[4]) {
    std::vector<float4> sec[4];
    for (int k = 0; k < 4; ++k) count[k] = 0;
    for (uint32_t i = 0; i < nq; ++i) {
        const float4 *q = &quads[(size_t)HRT_QUAD_ROWS * i];
        const double p0[3] = {q[0].x, q[0].y, q[0].z}, n[3] = {q[1].x, q[1].y, q[1].z}, R[3] = {q[2].x, q[2].y, q[2].z}, U[3] = {q[3].x, q[3].y, q[3].z};
        const double lenR = q[2].w, lenU = q[3].w;
        uint32_t flags;
        std::memcpy(&flags, &q[1].w, 4);
        int K = -1, ra = -1;   // normal axis; axis R runs along
        double eps_n = 0.0, eps_e = 0.0;
        if (!(flags & HRT_QUAD_FLAG_MOVING) && lenR > 0.0 && lenU > 0.0) {
            int k = 0;
            for (int c = 1; c < 3; ++c) if (std::fabs(n[c]) > std::fabs(n[k])) k = c;
            const int a = (k + 1) % 3, b = (k + 2) % 3;
            eps_n = std::fabs(n[a]) + std::fabs(n[b]) + std::fabs(1.0 - std::fabs(n[k]));
            const double dev_ab = (std::fabs(R[k]) + std::fabs(R[b])) / lenR + (std::fabs(U[k]) + std::fabs(U[a])) / lenU;  // R along a, U along b
            const double dev_ba = (std::fabs(R[k]) + std::fabs(R[a])) / lenR + (std::fabs(U[k]) + std::fabs(U[b])) / lenU;  // or the other way round
            eps_e = std::min(dev_ab, dev_ba);
            if (eps_n <= 1e-4 && eps_e <= 1e-4 && std::isfinite(eps_n) && std::isfinite(eps_e)) { K = k; ra = dev_ab <= dev_ba ? a : b; }
        }
        if (K >= 0) {
            const int a = (K + 1) % 3, b = (K + 2) % 3;
            (void)ra;
            double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int c = 0; c < 4; ++c)      // the four corners p0, p0 + R, p0 + U, p0 + R + U
                for (int x = 0; x < 3; ++x) {
                    const double v = p0[x] + ((c & 1) ? R[x] : 0.0) + ((c & 2) ? U[x] : 0.0);
                    lo[x] = std::min(lo[x], v); hi[x] = std::max(hi[x], v);
                }
            const double widen = (2.0 * eps_e + 2e-6) * (lenR + lenU);  // proj = qq.R / |R| against qq_a, and its own rounding
            auto centre = [&](int x) { return (float)(0.5 * (lo[x] + hi[x])); };
            auto half = [&](int x) { return std::nextafter((float)(0.5 * (hi[x] - lo[x]) + widen + 2.4e-7 * (std::fabs(lo[x]) + std::fabs(hi[x]))), INFINITY); };
            const float sgn = n[K] < 0.0 ? -1.f : 1.f;
            const float par = (float)(8.0 * (eps_n + 4e-7));
            const float cq = (float)(2.5e6 * (eps_n + 3e-7));
            const uint32_t bits = ((flags & HRT_QUAD_FLAG_GLASS) ? 1u : 0u) | (sgn < 0.f ? 2u : 0u) | (i << 8);
            sec[K].push_back(make_float4(sgn * q[0].w, centre(a), centre(b), half(a)));
            sec[K].push_back(make_float4(half(b), as_float(bits), par, cq));
            ++count[K];
        } else {
            sec[3].push_back(make_float4(q[0].x, q[0].y, q[0].z, q[0].w));
            sec[3].push_back(make_float4(q[1].y, q[1].z, q[1].x, as_float(flags | (i << 8))));
            sec[3].push_back(make_float4(q[2].x, q[3].x, q[2].y, q[3].y));
            sec[3].push_back(make_float4(q[2].z, q[3].z, q[2].w, q[3].w));
            ++count[3];
        }
    }
    qf.clear();
    for (int k = 0; k < 4; ++k) qf.insert(qf.end(), sec[k].begin(), sec[k].end());
}

// Triangle(c0,c1,c2) + computeBarycentricCoordinates constants (Triangle.h:26-37, 62-70) -> the 5 rows of hrt_device.h.
void fold_triangle(const H3 c[3], uint32_t id, std::vector<float4> &tris, std::vector<float4> &planes) {
    const H3 e1 = h_sub(c[1], c[0]), e2 = h_sub(c[2], c[0]);
    const H3 nn = h_cross(e1, e2);
    const float norm = h_len(nn);
    const H3 n{nn.x / norm, nn.y / norm, nn.z / norm};
    const float d00 = h_dot(e1, e1), d01 = h_dot(e1, e2), d11 = h_dot(e2, e2);
    const float denom = h_msub(d00, d11, d01, d01);
    planes.push_back(make_float4(n.x, n.y, n.z, h_dot(c[0], n)));
    (void)denom;  // = fl(fl(d00 * d11) - fl(d01 * d01)): the kernels recompute it from the rows, in the same two roundings (tri_inside)
    tris.push_back(make_float4(c[0].x, c[0].y, c[0].z, d11));
    tris.push_back(make_float4(e1.x, e1.y, e1.z, d00));
    tris.push_back(make_float4(e2.x, e2.y, e2.z, d01));
    tris.push_back(make_float4(as_float(id), 0.f, 0.f, 0.f));  // read only when the triangle is shaded
}

template <class T>
int upload(const std::vector<T> &v, T **dptr) {
    *dptr = nullptr;
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc((void **)dptr, bytes));
    if (!v.empty()) HIP_TRY(hipMemcpy(*dptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return HRT_OK;
}

}  // namespace

struct hrt_scene {
    DScene d{};                  // host copy of the device scene header
    DScene *d_scene = nullptr;   // the header in HBM (read by the kernels through a constant-space pointer)
    DCamera *d_cam = nullptr;    // camera block in HBM, re-uploaded only when the camera changes
    DCamera h_cam{};
    bool cam_valid = false;
    uint32_t lds_units = 0;
    uint32_t max_leaf = 0;       // most triangles in one KD leaf (the resumable walk keeps a 16-bit leaf cursor)
    std::vector<void *> allocations;
    uint32_t *tile_counter = nullptr;
    unsigned long long *stamps = nullptr;  // diagnostic cycle counters (HRT_STAMPS builds)
    float *sp_scratch = nullptr;           // per-workgroup sample scratch of the streaming kernel
    size_t sp_scratch_cap = 0;
    uint32_t *sp_pool = nullptr;           // path records of the streaming kernel when they live in global memory
    size_t sp_pool_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    float bound = 0.f;  // largest distance of any scene point from the origin (filter margins)
    // scratch of hrt_render (whole frame on one GPU)
    float *d_tiles = nullptr, *d_frame = nullptr;
    size_t tiles_cap = 0, frame_cap = 0;
    uint32_t last_grid = 0, last_waves = 0, last_lds = 0;
    hipStream_t last_stream = nullptr;  // stream of the previous launch on this scene
    int device = 0;                     // the device that holds this scene (current when it was created)
};

#include "hrt_kdbuild.hip"

extern "C" {

const char *hrt_last_error(void) { return g_error.c_str(); }

int hrt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int hrt_init(int device_ordinal) {
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (n <= 0) return fail(HRT_ERR_DEVICE, "hrt_init: no HIP device");
    if (device_ordinal < 0 || device_ordinal >= n) return fail(HRT_ERR_INVALID, "hrt_init: device ordinal out of range");
    HIP_TRY(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal));
    g_rt.device = device_ordinal;
    g_rt.cus = prop.multiProcessorCount;
    if ((int)g_rt.dev_cus.size() < n) g_rt.dev_cus.resize((size_t)n, 0);
    if (g_rt.dev_cus[device_ordinal] != 0) { g_rt.ready = true; return HRT_OK; }  // this device is prepared already: it is current again
    HIP_TRY(hipFuncGetAttributes(&g_rt.attr, (const void *)hrt_trace_kernel));
    // LDS for nodelets per 256-thread workgroup.  Default 32 KiB (4 workgroups/CU keep 128 of the
    // CU's 160 KiB); HRT_LDS_KB overrides for tuning.
    uint32_t kb = 36u * (HRT_WG / 256u);  // 4 x 36 KiB (HRT_WG 256) or 1 x 144 KiB (HRT_WG 1024) of the CU's 160 KiB
    if (const char *e = std::getenv("HRT_LDS_KB")) kb = (uint32_t)std::max(0, atoi(e));
    if (kb > 160) kb = 160;
    g_rt.lds_budget = kb * 1024u;
    if (g_rt.lds_budget > 64u * 1024u) {
        HIP_TRY(hipFuncSetAttribute((const void *)hrt_trace_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)g_rt.lds_budget));
        HIP_TRY(hipFuncSetAttribute((const void *)hrt_trace_kernel_lights, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)g_rt.lds_budget));
        HIP_TRY(hipFuncSetAttribute((const void *)hrt_trace_kernel_exact, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_rt.lds_budget));
        HIP_TRY(hipFuncSetAttribute((const void *)hrt_trace_kernel_lights_exact, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)g_rt.lds_budget));
    }
    if (HRT_WG > 256) {  // one big workgroup per CU: backed-up streams + nodelets go past the 64 KiB default
        HIP_TRY(hipFuncSetAttribute((const void *)hrt_trace2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void *)hrt_trace2_kernel_lights, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    {   // u8 -> float tables in double, as the reference evaluates c/255. and c/127.5 - 1. (Material.cpp:87,124)
        float lut[512];
        for (int c = 0; c < 256; ++c) { lut[c] = (float)(c / 255.); lut[256 + c] = (float)(c / 127.5 - 1.); }
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_u8_lut), lut, sizeof(lut)));
    }
    {   // the streaming kernel keeps its path pool in LDS: 113 KiB + nodelets, one 1024-thread workgroup per CU
        const int max_lds = 160 * 1024;
        HIP_TRY(hipFuncSetAttribute((const void *)hrt_wgstream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        HIP_TRY(hipFuncSetAttribute((const void *)hrt_wgstream_kernel_lights, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        HIP_TRY(hipFuncSetAttribute((const void *)hrt_wgstream_kernel_exact, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        HIP_TRY(hipFuncSetAttribute((const void *)hrt_wgstream_kernel_lights_exact, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        HIP_TRY(hipFuncSetAttribute((const void *)hrt_wgstream_kernel_sph, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        HIP_TRY(hipFuncSetAttribute((const void *)hrt_wgstream_kernel_lights_sph, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
        const char *k = std::getenv("HRT_KERNEL");
        const std::string ks = k ? k : "";
        g_rt.use_dual = ks != "single";
        g_rt.use_stream = ks == "stream" ? 1 : ((ks == "single" || ks == "dual") ? 0 : -1);
    }
    g_rt.dev_cus[device_ordinal] = prop.multiProcessorCount;
    g_rt.ready = true;
    return HRT_OK;
}

void hrt_shutdown(void) { g_rt.ready = false; g_rt.dev_cus.clear(); }

void hrt_scene_destroy(hrt_scene *s) {
    if (!s) return;
    if (g_rt.ready) (void)use_device(s->device);
    for (void *p : s->allocations) (void)hipFree(p);
    if (s->tile_counter) (void)hipFree(s->tile_counter);
    if (s->stamps) (void)hipFree(s->stamps);
    if (s->sp_scratch) (void)hipFree(s->sp_scratch);
    if (s->sp_pool) (void)hipFree(s->sp_pool);
    if (s->d_scene) (void)hipFree(s->d_scene);
    if (s->d_cam) (void)hipFree(s->d_cam);
    if (s->d_tiles) (void)hipFree(s->d_tiles);
    if (s->d_frame) (void)hipFree(s->d_frame);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
}

static int scene_create_impl(const hrt_scene_desc *desc, hrt_scene *s) {
    const hrt_scene_desc &D = *desc;
    // ---- validation: every index the kernel will follow is checked here, on the host
    for (uint32_t i = 0; i < D.n_materials; ++i) {
        const hrt_material &m = D.materials[i];
        if (m.image >= (int32_t)D.n_images || m.normal_map >= (int32_t)D.n_images)
            return fail(HRT_ERR_INVALID, "material references an image that does not exist");
        if (m.type < 0 || m.type > 2 || m.texture_type < 0 || m.texture_type > 2)
            return fail(HRT_ERR_INVALID, "material type / texture type out of range");
        if (m.normal_map >= 0 && (D.images[m.normal_map].w < 1 || D.images[m.normal_map].h < 1))
            return fail(HRT_ERR_INVALID, "normal map image is empty");
    }
    for (uint32_t i = 0; i < D.n_spheres; ++i)
        if (D.spheres[i].material < 0 || (uint32_t)D.spheres[i].material >= D.n_materials)
            return fail(HRT_ERR_INVALID, "sphere material out of range");
    for (uint32_t i = 0; i < D.n_quads; ++i)
        if (D.quads[i].material < 0 || (uint32_t)D.quads[i].material >= D.n_materials)
            return fail(HRT_ERR_INVALID, "quad material out of range");
    if (D.skybox_image >= (int32_t)D.n_images) return fail(HRT_ERR_INVALID, "skybox image out of range");
    if (D.n_lights && !D.lights) return fail(HRT_ERR_INVALID, "lights missing");
    if (D.n_meshes > 32u) return fail(HRT_ERR_INVALID, "more than 32 meshes in one scene (the parked-mesh mask is 32 bits)");

    // ---- scene extent: an upper bound on |point| for every point a ray can start from or hit
    {
        double b = 0.0;
        auto grow = [&](double x, double y, double z, double extra) { b = std::max(b, std::sqrt(x * x + y * y + z * z) + extra); };
        auto mlen = [&](const hrt_material &m) { return std::sqrt((double)m.motion[0] * m.motion[0] + (double)m.motion[1] * m.motion[1] + (double)m.motion[2] * m.motion[2]); };
        for (uint32_t i = 0; i < D.n_spheres; ++i) {
            const hrt_sphere &sp = D.spheres[i];
            grow(sp.center[0], sp.center[1], sp.center[2], std::fabs((double)sp.radius) + mlen(D.materials[sp.material]));
        }
        for (uint32_t i = 0; i < D.n_quads; ++i) {
            const hrt_quad &q = D.quads[i];
            const double ml = mlen(D.materials[q.material]);
            grow(q.v0[0], q.v0[1], q.v0[2], ml);
            grow(q.v1[0], q.v1[1], q.v1[2], ml);
            grow(q.v3[0], q.v3[1], q.v3[2], ml);
            grow((double)q.v1[0] + q.v3[0] - q.v0[0], (double)q.v1[1] + q.v3[1] - q.v0[1], (double)q.v1[2] + q.v3[2] - q.v0[2], ml);
        }
        for (uint32_t mi = 0; mi < D.n_meshes; ++mi)
            for (uint32_t v = 0; v < D.meshes[mi].n_vertices; ++v) {
                const float *p = D.meshes[mi].positions + 3 * (size_t)v;
                grow(p[0] * 1.00001, p[1] * 1.00001, p[2] * 1.00001, 0.0);
            }
        for (uint32_t i = 0; i < D.n_lights; ++i) grow(D.lights[i].pos[0], D.lights[i].pos[1], D.lights[i].pos[2], std::fabs((double)D.lights[i].radius));
        s->bound = (float)b;
    }

    // ---- spheres / quads / lights / materials
    std::vector<float4> spheres, quads, mats, lights;
    for (uint32_t i = 0; i < D.n_spheres; ++i) {
        const hrt_sphere &sp = D.spheres[i];
        const hrt_material &m = D.materials[sp.material];
        spheres.push_back(make_float4(sp.center[0], sp.center[1], sp.center[2], sp.radius));
        spheres.push_back(make_float4(m.motion[0], m.motion[1], m.motion[2], as_float((uint32_t)sp.material)));
    }
    for (uint32_t i = 0; i < D.n_quads; ++i) {
        const hrt_quad &q = D.quads[i];
        const hrt_material &m = D.materials[q.material];
        fold_quad(q, m, quads);
    }
    std::vector<float4> qfilter;
    build_quad_filter(quads, D.n_quads, qfilter, s->d.qf_n);
    std::vector<float4> sfilter;  // hrt_device.h DScene::sfilter; hrt_kernels.hip sphere_filter
    {
        const uint32_t ns = D.n_spheres, pairs = (ns + 1u) / 2u;
        for (uint32_t p = 0; p < pairs; ++p) {
            const uint32_t a = 2u * p, b = std::min(2u * p + 1u, ns - 1u);
            const float4 a0 = spheres[2u * a], a1 = spheres[2u * a + 1u], b0 = spheres[2u * b], b1 = spheres[2u * b + 1u];
            sfilter.push_back(make_float4(a0.x, b0.x, a0.y, b0.y));
            sfilter.push_back(make_float4(a0.z, b0.z, a0.w * a0.w, b0.w * b0.w));
            sfilter.push_back(make_float4(a1.x, b1.x, a1.y, b1.y));
            sfilter.push_back(make_float4(a1.z, b1.z, std::fabs(a0.w), std::fabs(b0.w)));
        }
        s->d.sf_pairs = pairs;
        s->d.sf_psize = std::max(1u, (pairs + 63u) / 64u);
    }
    for (uint32_t i = 0; i < D.n_lights; ++i) {
        const hrt_light &l = D.lights[i];
        lights.push_back(make_float4(l.pos[0], l.pos[1], l.pos[2], l.radius));
        lights.push_back(make_float4(l.color[0], l.color[1], l.color[2], 0.f));
    }
    for (uint32_t i = 0; i < D.n_materials; ++i) {
        const hrt_material &m = D.materials[i];
        mats.push_back(make_float4(m.albedo[0], m.albedo[1], m.albedo[2], m.transparency));
        mats.push_back(make_float4(m.index_medium, as_float((uint32_t)m.type), as_float((uint32_t)m.texture_type),
                                   as_float((uint32_t)(m.emissive ? 1 : 0))));
        mats.push_back(make_float4(m.checker1[0], m.checker1[1], m.checker1[2], m.tex_scale_x));
        mats.push_back(make_float4(m.checker2[0], m.checker2[1], m.checker2[2], m.tex_scale_y));
        mats.push_back(make_float4(m.light_color[0], m.light_color[1], m.light_color[2], m.light_intensity));
        mats.push_back(make_float4(as_float((uint32_t)m.image), as_float((uint32_t)m.normal_map), 0.f, 0.f));
        mats.push_back(make_float4(0.f, 0.f, 0.f, 0.f));  // rows 6, 7: geometry of the texture / the normal map, filled in below
        mats.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
    }

    // ---- images -> RGBA8 words
    std::vector<DImage> images;
    std::vector<uint32_t> texels;
    for (uint32_t i = 0; i < D.n_images; ++i) {
        const hrt_image &im = D.images[i];
        DImage di;
        di.offset = (uint32_t)texels.size();
        di.w = im.w; di.h = im.h; di.pad = 0;
        if (im.w >= 1 && im.h >= 1) {
            if (!im.rgb) return fail(HRT_ERR_INVALID, "image without pixels");
            const size_t n = (size_t)im.w * im.h;
            for (size_t p = 0; p < n; ++p)
                texels.push_back((uint32_t)im.rgb[3 * p] | ((uint32_t)im.rgb[3 * p + 1] << 8) | ((uint32_t)im.rgb[3 * p + 2] << 16));
        }
        images.push_back(di);
    }

    for (uint32_t i = 0; i < D.n_materials; ++i) {  // {texel offset, w, h} of each material's images, so that a lane need not chase the image table
        const hrt_material &m = D.materials[i];
        if (m.image >= 0) mats[(size_t)HRT_MAT_ROWS * i + 6] = make_float4(as_float(images[m.image].offset), as_float((uint32_t)images[m.image].w), as_float((uint32_t)images[m.image].h), 0.f);
        if (m.normal_map >= 0) mats[(size_t)HRT_MAT_ROWS * i + 7] = make_float4(as_float(images[m.normal_map].offset), as_float((uint32_t)images[m.normal_map].w), as_float((uint32_t)images[m.normal_map].h), 0.f);
    }

    // ---- meshes: nodelets (refs rebased), leaf-ordered triangle soup, colours
    std::vector<DMesh> meshes;
    std::vector<uint4> units;
    std::vector<float4> tris, planes, colors, exceptions;
    std::vector<uint4> vids;
    for (uint32_t mi = 0; mi < D.n_meshes; ++mi) {
        const hrt_mesh &M = D.meshes[mi];
        if (M.material < 0 || (uint32_t)M.material >= D.n_materials) return fail(HRT_ERR_INVALID, "mesh material out of range");
        for (uint32_t k = 0; k < 3 * M.n_triangles; ++k)
            if (M.indices[k] >= M.n_vertices) return fail(HRT_ERR_INVALID, "mesh vertex index out of range");
        if (M.n_leaf_tris && (M.kd_root == HRT_KD_NIL || !M.kd_units || !M.n_kd_units))
            return fail(HRT_ERR_INVALID, "mesh has triangles but no flattened KD-tree");
        if (M.n_exceptions && !M.exceptions) return fail(HRT_ERR_INVALID, "mesh exceptions missing");
        DMesh dm;
        std::memset(&dm, 0, sizeof(dm));
        for (int a = 0; a < 3; ++a) {
            dm.aabb_lo[a] = M.aabb_min[a]; dm.aabb_hi[a] = M.aabb_max[a];
            dm.kd_lo[a] = M.kd_min[a]; dm.kd_hi[a] = M.kd_max[a];
        }
        units.resize((units.size() + 3u) & ~(size_t)3u, make_uint4(0, 0, 0, 0));  // every mesh's nodelets start on a 64-byte line (the host aligns clusters and leaves)
        const uint32_t unit_base = (uint32_t)units.size();
        const uint32_t tri_base = (uint32_t)(tris.size() / HRT_TRI_ROWS);
        // The caller's tree (include/hrt.h: 16-byte inner nodelets, 64-byte leaves, any numbering) is re-laid for the walk:
        //   inner nodes become TREELETS of two levels in 32 bytes  {split, left child's split, right child's split, axes}
        //   {refs of the four grandchildren}  (axes: 2 bits per node; 3 = the child is a leaf, both exits of its pair hold its ref),
        //   one for the root, one for every grandchild that is an inner node and one for every inner node a rope points at --
        //   a walk then descends two levels per round trip (csrc/hrt_kernels.hip kd_descend);
        //   leaves keep their four units {lo, first} {hi, count} {ropes -x +x -y +y} {ropes -z +z}, refs translated, on 64-byte lines.
        // Numbering is breadth-first from the root, so a prefix of the array is the top of the tree (what the kernels stage in LDS).
        // Only well-formed nodelets the ROOT reaches through child links are accepted, as what they are: a rope, too, may only name
        // such a nodelet, with its own kind (an inner unit named as a leaf would be read as four units from a two-unit slot).
        auto in_range = [&](uint32_t ref) -> bool {
            if (ref == HRT_KD_NIL) return true;
            const uint32_t idx = ref & ~HRT_KD_LEAF;
            return (uint64_t)idx + ((ref & HRT_KD_LEAF) ? 4u : 1u) <= M.n_kd_units;
        };
        if (M.n_leaf_tris) {
            if (M.kd_root == HRT_KD_NIL || !in_range(M.kd_root)) return fail(HRT_ERR_INVALID, "malformed flattened KD-tree");
            std::vector<uint8_t> reached(M.n_kd_units, 0);  // 1: an inner nodelet of the tree, 2: a leaf of the tree
            {   // the child links must form a TREE: a nodelet reached twice (a shared subtree, or a cycle -- on which a walk would
                // descend forever) is refused
                std::vector<uint32_t> stack{M.kd_root};
                reached[M.kd_root & ~HRT_KD_LEAF] = (M.kd_root & HRT_KD_LEAF) ? 2 : 1;
                while (!stack.empty()) {
                    const uint32_t ref = stack.back();
                    stack.pop_back();
                    if (ref & HRT_KD_LEAF) continue;
                    const hrt_kdunit &u = M.kd_units[ref];
                    for (int c = 2; c < 4; ++c) {
                        const uint32_t child = u.w[c];
                        if (child == HRT_KD_NIL || !in_range(child) || reached[child & ~HRT_KD_LEAF]) return fail(HRT_ERR_INVALID, "malformed flattened KD-tree (a nodelet is reached twice through child links)");
                        reached[child & ~HRT_KD_LEAF] = (child & HRT_KD_LEAF) ? 2 : 1;
                        stack.push_back(child);
                    }
                }
            }
            std::vector<uint32_t> new_of(M.n_kd_units, 0xFFFFFFFFu);  // caller's unit index -> unit index in this mesh's new list
            std::vector<uint32_t> order;                               // caller's refs in the order they are laid out
            uint32_t cur = 0;
            bool ok = true;
            auto want = [&](uint32_t ref) {
                if (ref == HRT_KD_NIL) return;
                if (!in_range(ref)) { ok = false; return; }
                const uint32_t idx = ref & ~HRT_KD_LEAF;
                if (reached[idx] != ((ref & HRT_KD_LEAF) ? 2 : 1)) { ok = false; return; }  // (ropes: child links were checked above)
                if (new_of[idx] != 0xFFFFFFFFu) return;
                const uint32_t size = (ref & HRT_KD_LEAF) ? 4u : 2u;
                cur = (cur + size - 1u) & ~(size - 1u);
                new_of[idx] = cur;
                cur += size;
                order.push_back(ref);
            };
            auto inner_ok = [&](const hrt_kdunit &u) { return u.w[1] <= 2u && u.w[2] != HRT_KD_NIL && u.w[3] != HRT_KD_NIL && in_range(u.w[2]) && in_range(u.w[3]); };
            want(M.kd_root);
            for (size_t q = 0; ok && q < order.size(); ++q) {
                const uint32_t ref = order[q], idx = ref & ~HRT_KD_LEAF;
                const hrt_kdunit *u = M.kd_units + idx;
                if (ref & HRT_KD_LEAF) {
                    if ((uint64_t)u[0].w[3] + u[1].w[3] > M.n_leaf_tris) { ok = false; break; }
                    if (u[1].w[3] >= 0xFFFFu) return fail(HRT_ERR_INVALID, "KD leaf with 65535 or more triangles (the resumable walk keeps a 16-bit leaf cursor): build the tree with a smaller leaf_max");
                    s->max_leaf = std::max(s->max_leaf, u[1].w[3]);
                    for (int f = 0; f < 4; ++f) want(u[2].w[f]);
                    want(u[3].w[0]);
                    want(u[3].w[1]);
                } else {
                    if (!inner_ok(*u)) { ok = false; break; }
                    for (int c = 0; c < 2 && ok; ++c) {
                        const uint32_t child = u->w[2 + c];
                        if (child & HRT_KD_LEAF) { want(child); continue; }
                        const hrt_kdunit &y = M.kd_units[child];
                        if (!inner_ok(y)) { ok = false; break; }
                        want(y.w[2]);
                        want(y.w[3]);
                    }
                }
            }
            if (!ok) return fail(HRT_ERR_INVALID, "malformed flattened KD-tree (a link or rope names a unit that is not a nodelet of this tree, or not of that kind)");
            auto tr = [&](uint32_t ref) -> uint32_t { return ref == HRT_KD_NIL ? ref : ((new_of[ref & ~HRT_KD_LEAF] + unit_base) | (ref & HRT_KD_LEAF)); };
            units.resize(unit_base + ((cur + 3u) & ~3u), make_uint4(0, 0, 0, 0));
            for (uint32_t ref : order) {
                const uint32_t idx = ref & ~HRT_KD_LEAF;
                const hrt_kdunit *u = M.kd_units + idx;
                uint4 *o = &units[unit_base + new_of[idx]];
                if (ref & HRT_KD_LEAF) {
                    o[0] = make_uint4(u[0].w[0], u[0].w[1], u[0].w[2], u[0].w[3]);
                    o[1] = make_uint4(u[1].w[0], u[1].w[1], u[1].w[2], u[1].w[3]);
                    o[2] = make_uint4(tr(u[2].w[0]), tr(u[2].w[1]), tr(u[2].w[2]), tr(u[2].w[3]));
                    o[3] = make_uint4(tr(u[3].w[0]), tr(u[3].w[1]), 0, 0);
                } else {
                    uint32_t split[2] = {0, 0}, axis[2] = {3, 3}, exits[4];
                    for (int c = 0; c < 2; ++c) {
                        const uint32_t child = u->w[2 + c];
                        if (child & HRT_KD_LEAF) {
                            exits[2 * c] = exits[2 * c + 1] = tr(child);
                        } else {
                            const hrt_kdunit &y = M.kd_units[child];
                            split[c] = y.w[0]; axis[c] = y.w[1];
                            exits[2 * c] = tr(y.w[2]); exits[2 * c + 1] = tr(y.w[3]);
                        }
                    }
                    o[0] = make_uint4(u->w[0], split[0], split[1], u->w[1] | (axis[0] << 2) | (axis[1] << 4));
                    o[1] = make_uint4(exits[0], exits[1], exits[2], exits[3]);
                }
            }
            dm.root = tr(M.kd_root);
        } else {
            dm.root = HRT_KD_NIL;
        }
        // triangle soup in leaf order
        auto push_triangle = [&](uint32_t t) {
            H3 c[3];
            for (int j = 0; j < 3; ++j) {
                const float *p = M.positions + 3 * (size_t)M.indices[3 * (size_t)t + j];
                c[j] = H3{p[0] * HRT_TRIANGLE_SCALING, p[1] * HRT_TRIANGLE_SCALING, p[2] * HRT_TRIANGLE_SCALING};
            }
            fold_triangle(c, t, tris, planes);
        };
        for (uint32_t k = 0; k < M.n_leaf_tris; ++k) {
            const uint32_t t = M.leaf_tris[k];
            if (t >= M.n_triangles) return fail(HRT_ERR_INVALID, "leaf triangle id out of range");
            push_triangle(t);
        }
        dm.tri_base = tri_base;
        dm.n_soup = M.n_leaf_tris;
        // Irregular triangles (include/hrt.h hrt_tri_exception), grouped by TRIANGLE.  The reference tests such a triangle when
        // the ray passes a leaf box that holds it (KDTree.cpp:32-46), and the outcome of the triangle test does not depend on
        // which box that was: so each irregular triangle is folded once (its rows sit behind the mesh's leaf-ordered soup) and
        // tested at most once per ray, and only a ray that HITS it closer than the best so far goes through the list of its
        // reference boxes (exact AABB.h:48-65 arithmetic) to learn whether the reference would have tested it at all.
        // Entries, 2 rows each, threaded depth-first:
        //   inner   {lo', HRT_EXC_INNER} {hi', skip}     padded bounds of a subtree: only culls
        //   leaf    {cull lo, soup slot} {cull hi, nb}   then nb box entries {box lo, last} {box hi, 0} the walk jumps over; the boxes of one
        //           reference leaf follow each other (`last` = 1 on the final one) and must ALL be passed for that leaf to count
        // The cull box of a well-conditioned triangle is its own padded bounds (an accepted hit point lies in the triangle up to
        // the rounding of the barycentric solve, ~1e-7 / sin^2); a sliver's barycentric test accepts points anywhere in its
        // plane, so its cull box is the padded union of its reference boxes (a ray that passes none of them is not tested).
        dm.exc_base = (uint32_t)(exceptions.size() / 2);
        dm.n_exc = 0;
        if (M.n_exceptions) {
            struct Group { float lo[3], hi[3]; uint32_t tri; std::vector<std::array<float, 7>> boxes; };  // box: lo, hi, 1.f on the last box of its leaf
            std::vector<Group> groups;
            {
                std::vector<uint32_t> order(M.n_exceptions);
                for (uint32_t k = 0; k < M.n_exceptions; ++k) {
                    if (M.exceptions[k].triangle >= M.n_triangles) return fail(HRT_ERR_INVALID, "exception triangle id out of range");
                    order[k] = k;
                }
                static_assert(offsetof(hrt_tri_exception, box_max) == offsetof(hrt_tri_exception, box_min) + 12, "box_min and box_max are contiguous");
                auto box_cmp = [&](uint32_t x, uint32_t y) { return std::memcmp(M.exceptions[x].box_min, M.exceptions[y].box_min, 24); };
                std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {   // by triangle, then by reference leaf (the caller's order inside a leaf)
                    const hrt_tri_exception &ex = M.exceptions[x], &ey = M.exceptions[y];
                    return ex.triangle < ey.triangle || (ex.triangle == ey.triangle && ex.group < ey.group);
                });
                for (size_t q = 0; q < order.size(); ++q) {
                    const hrt_tri_exception &e = M.exceptions[order[q]];
                    const bool same_tri = !groups.empty() && groups.back().tri == e.triangle;
                    const bool same_leaf = same_tri && q > 0 && M.exceptions[order[q - 1]].group == e.group;
                    if (!same_tri) {
                        groups.emplace_back();
                        groups.back().tri = e.triangle;
                    }
                    if (same_leaf && box_cmp(order[q - 1], order[q]) == 0) continue;  // the same box twice
                    if (!same_leaf && !groups.back().boxes.empty()) groups.back().boxes.back()[6] = 1.f;  // the previous leaf's boxes end here
                    std::array<float, 7> bx;
                    std::memcpy(bx.data(), e.box_min, 24);
                    bx[6] = 0.f;
                    groups.back().boxes.push_back(bx);
                }
                for (Group &g : groups) g.boxes.back()[6] = 1.f;
                for (Group &g : groups) {
                    double c[3][3];
                    for (int j = 0; j < 3; ++j) {
                        const float *pp = M.positions + 3 * (size_t)M.indices[3 * (size_t)g.tri + j];
                        for (int a = 0; a < 3; ++a) c[j][a] = (double)(pp[a] * HRT_TRIANGLE_SCALING);
                    }
                    double e1[3], e2[3], cr[3];
                    for (int a = 0; a < 3; ++a) { e1[a] = c[1][a] - c[0][a]; e2[a] = c[2][a] - c[0][a]; }
                    cr[0] = e1[1] * e2[2] - e1[2] * e2[1]; cr[1] = e1[2] * e2[0] - e1[0] * e2[2]; cr[2] = e1[0] * e2[1] - e1[1] * e2[0];
                    const double l1 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2], l2 = e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2];
                    const double sin2 = (cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]) / (l1 * l2);
                    const bool own_bounds = std::isfinite(sin2) && sin2 >= 1e-2;  // NaN (zero edge): the union of the boxes
                    for (int a = 0; a < 3; ++a) {
                        double lo = INFINITY, hi = -INFINITY;
                        if (own_bounds) {
                            for (int j = 0; j < 3; ++j) { lo = std::min(lo, c[j][a]); hi = std::max(hi, c[j][a]); }
                        } else {
                            for (const auto &bx : g.boxes) {  // (AABB::intersects reads a box's faces in either order: a cut outside its node leaves one)
                                lo = std::min(lo, (double)std::min(bx[a], bx[3 + a])); hi = std::max(hi, (double)std::max(bx[a], bx[3 + a]));
                            }
                        }
                        double ext = 0.0;
                        for (int x = 0; x < 3; ++x) ext = std::max(ext, std::max(std::fabs(e1[x]), std::fabs(e2[x])));
                        const double pad = (own_bounds ? 1e-3 * ext : 0.0) + 1e-4 * std::max(1.0, std::max(std::fabs(lo), std::fabs(hi)));
                        g.lo[a] = (float)(lo - pad); g.hi[a] = (float)(hi + pad);
                    }
                }
            }
            const size_t first_entry = exceptions.size() / 2;
            struct Emit {
                std::vector<Group> &g;
                std::vector<float4> &out;
                std::vector<float4> &tris;
                const std::function<void(uint32_t)> &push_triangle;
                size_t first_entry;
                void run(size_t lo, size_t hi) {
                    if (hi - lo == 1) {
                        const Group &b = g[lo];
                        const uint32_t slot = (uint32_t)(tris.size() / HRT_TRI_ROWS);
                        push_triangle(b.tri);
                        out.push_back(make_float4(b.lo[0], b.lo[1], b.lo[2], as_float(slot)));
                        out.push_back(make_float4(b.hi[0], b.hi[1], b.hi[2], as_float((uint32_t)b.boxes.size())));
                        for (const auto &bx : b.boxes) {
                            out.push_back(make_float4(bx[0], bx[1], bx[2], as_float(bx[6] != 0.f ? 1u : 0u)));  // .w: 1 = the last box of its reference leaf
                            out.push_back(make_float4(bx[3], bx[4], bx[5], 0.f));
                        }
                        return;
                    }
                    float bmin[3] = {INFINITY, INFINITY, INFINITY}, bmax[3] = {-INFINITY, -INFINITY, -INFINITY};
                    float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
                    for (size_t i = lo; i < hi; ++i)
                        for (int a = 0; a < 3; ++a) {
                            bmin[a] = std::min(bmin[a], g[i].lo[a]); bmax[a] = std::max(bmax[a], g[i].hi[a]);
                            const float c = 0.5f * (g[i].lo[a] + g[i].hi[a]);
                            cmin[a] = std::min(cmin[a], c); cmax[a] = std::max(cmax[a], c);
                        }
                    int axis = 0;
                    for (int a = 1; a < 3; ++a) if (cmax[a] - cmin[a] > cmax[axis] - cmin[axis]) axis = a;
                    const size_t mid = lo + (hi - lo) / 2;
                    std::nth_element(g.begin() + lo, g.begin() + mid, g.begin() + hi, [axis](const Group &x, const Group &y) {
                        const float cx = x.lo[axis] + x.hi[axis], cy = y.lo[axis] + y.hi[axis];
                        return cx < cy || (cx == cy && x.tri < y.tri);
                    });
                    const size_t self = out.size();
                    float pmin[3], pmax[3];
                    for (int a = 0; a < 3; ++a) {
                        const float pad = 1e-4f * std::max(1.f, std::max(std::fabs(bmin[a]), std::fabs(bmax[a])));
                        pmin[a] = bmin[a] - pad; pmax[a] = bmax[a] + pad;
                    }
                    out.push_back(make_float4(pmin[0], pmin[1], pmin[2], as_float(HRT_EXC_INNER)));
                    out.push_back(make_float4(pmax[0], pmax[1], pmax[2], 0.f));
                    run(lo, mid);
                    run(mid, hi);
                    out[self + 1].w = as_float((uint32_t)(out.size() / 2 - first_entry));  // skip: first entry behind this subtree, relative to the mesh's list
                }
            };
            const std::function<void(uint32_t)> pt = push_triangle;
            Emit em{groups, exceptions, tris, pt, first_entry};
            em.run(0, groups.size());
            dm.n_exc = (uint32_t)(exceptions.size() / 2 - first_entry);
        }
        dm.material = (uint32_t)M.material;
        dm.color_type = HRT_COLOR_NONE;
        if (M.color_type == HRT_COLOR_FACE && M.face_colors) {
            dm.color_type = HRT_COLOR_FACE;
            dm.color_base = (uint32_t)colors.size();
            for (uint32_t t = 0; t < M.n_triangles; ++t)
                colors.push_back(make_float4(M.face_colors[3 * t], M.face_colors[3 * t + 1], M.face_colors[3 * t + 2], 0.f));
        } else if (M.color_type == HRT_COLOR_VERTEX && M.vert_colors) {
            dm.color_type = HRT_COLOR_VERTEX;
            dm.vcolor_base = (uint32_t)colors.size();
            for (uint32_t v = 0; v < M.n_vertices; ++v)
                colors.push_back(make_float4(M.vert_colors[3 * v], M.vert_colors[3 * v + 1], M.vert_colors[3 * v + 2], 0.f));
            dm.color_base = (uint32_t)vids.size();
            for (uint32_t t = 0; t < M.n_triangles; ++t)
                vids.push_back(make_uint4(M.indices[3 * t], M.indices[3 * t + 1], M.indices[3 * t + 2], 0));
        }
        meshes.push_back(dm);
    }

    // ---- upload
    DScene &d = s->d;
    float4 *p4 = nullptr;
    int rc;
#define UP(vec, field, type)                                     \
    {                                                            \
        type *ptr = nullptr;                                     \
        if ((rc = upload(vec, &ptr)) != HRT_OK) return rc;       \
        s->allocations.push_back(ptr);                           \
        d.field = ptr;                                           \
    }
    (void)p4;
    {   // squares | materials | spheres | mesh records in one array: `quads`, `materials`, `spheres`, `meshes` point into it
        static_assert(sizeof(DMesh) % sizeof(float4) == 0, "mesh records are whole rows");
        std::vector<float4> tabs;
        d.tab_quads = 0;
        tabs.insert(tabs.end(), quads.begin(), quads.end());
        d.tab_mats = (uint32_t)tabs.size();
        tabs.insert(tabs.end(), mats.begin(), mats.end());
        d.tab_spheres = (uint32_t)tabs.size();
        tabs.insert(tabs.end(), spheres.begin(), spheres.end());
        d.tab_meshes = (uint32_t)tabs.size();
        tabs.resize(tabs.size() + meshes.size() * (sizeof(DMesh) / sizeof(float4)));
        if (!meshes.empty()) std::memcpy(&tabs[d.tab_meshes], meshes.data(), meshes.size() * sizeof(DMesh));
        d.tab_sfilter = (uint32_t)tabs.size();
        tabs.insert(tabs.end(), sfilter.begin(), sfilter.end());
        d.tab_exc = (uint32_t)tabs.size();
        d.exc_in_tabs = exceptions.size() <= 1536u ? 1u : 0u;  // short exception lists ride along (24 KB at most)
        if (d.exc_in_tabs) tabs.insert(tabs.end(), exceptions.begin(), exceptions.end());
        d.tab_rows = (uint32_t)tabs.size();
        UP(tabs, tabs, float4)
        d.quads = d.tabs + d.tab_quads;
        d.materials = d.tabs + d.tab_mats;
        d.spheres = d.tabs + d.tab_spheres;
        d.meshes = reinterpret_cast<const DMesh *>(d.tabs + d.tab_meshes);
    }
    UP(qfilter, qfilter, float4)
    UP(units, kd_units, uint4)
    UP(tris, tris, float4)
    UP(planes, tri_planes, float4)
    UP(colors, colors, float4)
    UP(vids, tri_vids, uint4)
    UP(images, images, DImage)
    UP(texels, texels, uint32_t)
    UP(lights, lights, float4)
    UP(exceptions, exceptions, float4)
#undef UP
    d.n_spheres = D.n_spheres; d.n_quads = D.n_quads; d.n_meshes = D.n_meshes; d.n_lights = D.n_lights;
    d.n_images = D.n_images;
    d.n_kd_units = (uint32_t)units.size();
    s->lds_units = std::min<uint32_t>(d.n_kd_units, g_rt.lds_budget / 16u) & ~3u;  // whole 64-byte lines: no treelet or leaf straddles
    d.dark_sky = D.dark_sky;
    {   // exact path pruning needs 0 x value == 0: no colour of the scene may be infinite or NaN (hrt_device.h DScene::prune_ok)
        bool finite = true;
        auto fin3 = [&](const float *v) { finite = finite && std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); };
        for (uint32_t i = 0; i < D.n_materials; ++i) {
            const hrt_material &m = D.materials[i];
            fin3(m.albedo); fin3(m.checker1); fin3(m.checker2); fin3(m.light_color);
            finite = finite && std::isfinite(m.light_intensity);
        }
        for (uint32_t i = 0; i < D.n_lights; ++i) fin3(D.lights[i].color);
        for (const float4 &c : colors) finite = finite && std::isfinite(c.x) && std::isfinite(c.y) && std::isfinite(c.z);
        d.prune_ok = finite ? 1u : 0u;
        if (const char *e = std::getenv("HRT_PRUNE")) if (e[0] == '0') d.prune_ok = 0u;  // measurement aid: the same kernels without the pruning (bench.py reports both rates)
    }
    d.any_motion = 0u;  // (time x 0 == 0 whatever the time: with no motion anywhere a ray's time is never looked at)
    for (uint32_t i = 0; i < D.n_materials; ++i)
        if (!(D.materials[i].motion[0] == 0.f && D.materials[i].motion[1] == 0.f && D.materials[i].motion[2] == 0.f)) d.any_motion = 1u;
    d.skybox_image = (D.skybox_image >= 0 && D.images[D.skybox_image].w >= 1 && D.images[D.skybox_image].h >= 1) ? D.skybox_image : -1;
    HIP_TRY(hipMalloc((void **)&s->tile_counter, sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&s->stamps, 16 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(s->stamps, 0, 16 * sizeof(unsigned long long)));
    HIP_TRY(hipEventCreate(&s->ev0));
    HIP_TRY(hipEventCreate(&s->ev1));
    HIP_TRY(hipMalloc((void **)&s->d_scene, sizeof(DScene)));
    HIP_TRY(hipMemcpy(s->d_scene, &s->d, sizeof(DScene), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void **)&s->d_cam, sizeof(DCamera)));
    return HRT_OK;
}

int hrt_scene_create(const hrt_scene_desc *desc, hrt_scene **out) {
    if (!desc || !out) return fail(HRT_ERR_INVALID, "hrt_scene_create: NULL argument");
    if (!g_rt.ready) return fail(HRT_ERR_STATE, "hrt_scene_create: call hrt_init first");
    hrt_scene *s = nullptr;
    try {
        s = new hrt_scene();
        s->device = g_rt.device;
        int rc = scene_create_impl(desc, s);
        if (rc != HRT_OK) {
            std::string keep = g_error;
            hrt_scene_destroy(s);
            g_error = keep;
            return rc;
        }
    } catch (const std::exception &e) {
        if (s) hrt_scene_destroy(s);
        return fail(HRT_ERR_STATE, e.what());
    }
    *out = s;
    return HRT_OK;
}

uint32_t hrt_tiles_total(uint32_t w, uint32_t h) { return ((w + HRT_TILE - 1) / HRT_TILE) * ((h + HRT_TILE - 1) / HRT_TILE); }
uint32_t hrt_tiles_owned(uint32_t w, uint32_t h, uint32_t rank, uint32_t world) {
    const uint32_t t = hrt_tiles_total(w, h);
    if (!world || rank >= t) return 0;
    return (t - rank + world - 1) / world;
}

// hrt_camera -> the constants camera_ray() reads (hrt_device.h DCamera).
static int make_camera(const hrt_camera *cam, DCamera &C) {
    std::memset(&C, 0, sizeof(C));
    // The GL matrices the reference reads back (matrixUtilities.h:33-46) for this pose, fp64, column-major:
    // modelview = [right; up; -forward] * translate(-eye) (Camera.cpp:125-132), projection = gluPerspective(fovy,
    // aspect, znear, zfar) (Camera.cpp:41-50) -- then inverted by the reference's own method, the adjugate over the
    // determinant term by term (matrixUtilities.h:77-206; host_invert4 above), so that every constant below carries
    // the reference's rounding.  hrt_debug_kat(HRT_KAT_CAMERA) exposes the resulting rays; tests compare them bit for
    // bit with rays produced by the reference's gluInvertMatrix + screen_space_to_world_space_ray.
    {
        double mv[16], pr[16], mi[16], pi[16];
        for (int k = 0; k < 16; ++k) { mv[k] = 0.0; pr[k] = 0.0; }
        const double Rm[3][3] = {{cam->right[0], cam->right[1], cam->right[2]},
                                 {cam->up[0], cam->up[1], cam->up[2]},
                                 {-(double)cam->forward[0], -(double)cam->forward[1], -(double)cam->forward[2]}};
        for (int r = 0; r < 3; ++r) {
            for (int k = 0; k < 3; ++k) mv[k * 4 + r] = Rm[r][k];
            mv[12 + r] = h_neg_dot3(Rm[r], cam->eye);
        }
        mv[15] = 1.0;
        const double rad = (double)cam->fovy_deg / 2.0 * M_PI / 180.0;
        const double cot = std::cos(rad) / std::sin(rad);
        const double dz = (double)cam->zfar - (double)cam->znear;
        pr[0] = cot / (double)cam->aspect;
        pr[5] = cot;
        pr[10] = -((double)cam->zfar + (double)cam->znear) / dz;
        pr[11] = -1.0;
        pr[14] = h_mul64(h_mul64(-2.0, (double)cam->znear), (double)cam->zfar) / dz;
        if (!host_invert4(mv, mi) || !host_invert4(pr, pi)) return fail(HRT_ERR_INVALID, "render: singular camera matrix");
        // The kernel evaluates the two mat-vecs of matrixUtilities.h:60-68 with their structural zeros removed, which is
        // exact only for this sparsity (a zero coefficient contributes a signed zero, and x + (+-0) == x):
        //   P^-1 = [pi0 . . .; . pi5 . .; . . . pi14; . . pi11 pi15]      MV^-1 = [* * * *; * * * *; * * * *; 0 0 0 m15]
        static const int p_zero[] = {1, 2, 3, 4, 6, 7, 8, 9, 10, 12, 13}, m_zero[] = {3, 7, 11};
        for (int k : p_zero) if (pi[k] != 0.0) return fail(HRT_ERR_INVALID, "render: projection inverse is not of the gluPerspective form");
        for (int k : m_zero) if (mi[k] != 0.0) return fail(HRT_ERR_INVALID, "render: modelview inverse is not affine");
        bool finite = true;
        for (int k = 0; k < 16; ++k) finite = finite && std::isfinite(mi[k]) && std::isfinite(pi[k]);
        if (!finite) return fail(HRT_ERR_INVALID, "render: camera is not finite");
        // resInt = P^-1 (x, y, 0, 1):  resInt0 = pi0*x, resInt1 = pi5*y, resInt2 = pi14, resInt3 = pi15   (z = GL_DEPTH_RANGE[0] = 0)
        // res_k  = ((m[k]*resInt0 + m[4+k]*resInt1) + m[8+k]*resInt2) + m[12+k]*resInt3,   res_3 = m[15]*resInt3
        const double ri2 = pi[14], ri3 = pi[15];
        for (int a = 0; a < 3; ++a) C.eye[a] = (float)(mi[12 + a] / mi[15]);  // cameraSpaceToWorldSpace(0,0,0), :53-58
        C.pi0 = pi[0]; C.pi5 = pi[5];
        C.pi15 = h_mul64(mi[15], ri3);   // res_3, the divisor of :66-68
        C.inv15 = 1.0 / C.pi15;
        for (int k = 0; k < 3; ++k) {
            C.mx[k] = mi[k];
            C.my[k] = mi[4 + k];
            C.c1[k] = h_mul64(mi[8 + k], ri2);
            C.c2[k] = h_mul64(mi[12 + k], ri3);
        }
    }
    return HRT_OK;
}

static int fill_render(hrt_scene *s, const hrt_camera *cam, uint32_t w, uint32_t h, uint32_t spp, uint64_t seed,
                       uint32_t flags, uint32_t rank, uint32_t world, DRender &R, hipStream_t stream) {
    if (!s || !cam) return fail(HRT_ERR_INVALID, "render: NULL argument");
    if (!g_rt.ready) return fail(HRT_ERR_STATE, "render: call hrt_init first");
    { const int drc = use_device(s->device); if (drc != HRT_OK) return drc; }  // a scene lives on its device.  Unconditional: HIP's current
                                                                               // device is per thread, g_rt's copy of it per process
    if (!w || !h || !spp) return fail(HRT_ERR_INVALID, "render: w, h and spp must be positive");
    if ((uint64_t)w * h > 0x7fffffffull) return fail(HRT_ERR_INVALID, "render: image too large");
    if (w > 65535u || h > 65535u) return fail(HRT_ERR_INVALID, "render: w and h must be below 65536 (tile origins are packed in 16 + 16 bits)");
    if (!world || rank >= world) return fail(HRT_ERR_INVALID, "render: bad rank/world");
    R.scene = s->d_scene;
    R.cam = s->d_cam;
    R.lds_units = (flags & HRT_FLAG_NO_LDS_TREE) ? 0u : s->lds_units;
    R.err_abs = 2e-6f * (s->bound + std::sqrt(cam->eye[0] * cam->eye[0] + cam->eye[1] * cam->eye[1] + cam->eye[2] * cam->eye[2]) + 1.f);
    DCamera C;
    {
        const int crc = make_camera(cam, C);
        if (crc != HRT_OK) return crc;
    }
    if (!s->cam_valid || std::memcmp(&C, &s->h_cam, sizeof(C)) != 0) {
        // the previous launch may still be reading the old block: stream order makes the copy wait for it
        s->h_cam = C;
        HIP_TRY(hipMemcpyAsync(s->d_cam, &s->h_cam, sizeof(C), hipMemcpyHostToDevice, stream));
        s->cam_valid = true;
    }
    R.w = w; R.h = h; R.spp = spp;
    R.seed_lo = (uint32_t)seed; R.seed_hi = (uint32_t)(seed >> 32);
    R.flags = flags;
    R.rank = rank; R.world = world;
    R.tiles_x = (w + HRT_TILE - 1) / HRT_TILE;
    R.tiles_total = hrt_tiles_total(w, h);
    R.tiles_owned = hrt_tiles_owned(w, h, rank, world);
    R.tile_counter = s->tile_counter;
    R.stamps = s->stamps;
    return HRT_OK;
}

// One launch of the trace kernel over this rank's tiles: samples [s0, s0 + spp) of every pixel.
// accumulate = false: d_tiles receives the pixel means (s0 must be 0).
// accumulate = true : d_tiles holds the running sums of samples [0, s0) and receives the sums of [0, s0 + spp).
static int launch_trace(hrt_scene *s, const hrt_camera *cam, uint32_t w, uint32_t h, uint32_t s0, uint32_t spp, uint64_t seed,
                        uint32_t flags, uint32_t rank, uint32_t world, float *d_tiles, void *stream_, bool accumulate) {
    DRender R;
    int rc = fill_render(s, cam, w, h, spp, seed, flags, rank, world, R, (hipStream_t)stream_);
    if (rc != HRT_OK) return rc;
    if (!d_tiles) return fail(HRT_ERR_INVALID, "render: NULL tile buffer");
    if ((uint64_t)s0 + spp > 0xffffffffull) return fail(HRT_ERR_INVALID, "render: sample index overflows 32 bits");
    R.out_tiles = d_tiles;
    R.s0 = s0;
    R.accumulate = accumulate ? 1u : 0u;
    if (accumulate) flags &= ~(uint32_t)HRT_FLAG_GAMMA;  // hrt_finalize_tiles applies the gamma
    hipStream_t stream = (hipStream_t)stream_;
    if (R.tiles_owned == 0) { s->timed = false; return HRT_OK; }
    // Which schedule of the same arithmetic (all give identical pixels).  Measured on MI355X at 1080p: the
    // workgroup-streaming kernel wins where bounces diverge -- meshes (+3..10 %) and lit open scenes (random_spheres
    // +30 %) -- and loses on a closed box of squares (-45 %), where the lane-per-pixel kernel keeps its lanes busy anyway.
    // The streaming kernel pays where bounces diverge (meshes, lights) -- and on SMALL frames with many samples per pixel: the
    // lane-per-pixel kernel gives a wave 64 pixels and walks their samples one after the other, so below ~5 k tiles (20 waves
    // on each of 256 CUs) its time is spp x one sample's latency, while the streaming kernel spreads samples over the whole pool
    // (Cornell box 256 x 256 on MI355X: 4 spp 0.22 vs 0.29 ms, 64 spp 3.22 vs 1.52 ms; 1080p @ 16: 8.6 vs 11.1 ms).
    const bool small_and_deep = R.tiles_owned <= 5120u && spp >= 8u;
    const bool stream_pays = s->d.n_meshes > 0u || s->d.n_lights > 0u || small_and_deep;
    // the streaming kernel stages the per-object tables (squares, materials, spheres, mesh records) in LDS beside its queues
    const bool stream_fits = (size_t)s->d.tab_rows * 16u <= 48u * 1024u;
    if ((flags & HRT_FLAG_STREAM_KERNEL) && !stream_fits)
        return fail(HRT_ERR_INVALID, "render: the scene's object tables exceed the 48 KiB the streaming kernel keeps in LDS; use another kernel form");
    const bool stream_kernel = stream_fits && !(flags & (HRT_FLAG_WAVE_KERNEL | HRT_FLAG_DUAL_KERNEL)) &&
                               (g_rt.use_stream == 1 || (flags & HRT_FLAG_STREAM_KERNEL) || (g_rt.use_stream < 0 && stream_pays));
    const bool exact = (flags & HRT_FLAG_EXACT_ONLY) != 0u;  // proof builds exist for the lane-per-pixel and streaming forms
    if ((flags & HRT_FLAG_MESH_BRUTE) && !exact) return fail(HRT_ERR_INVALID, "render: HRT_FLAG_MESH_BRUTE needs HRT_FLAG_EXACT_ONLY");
    if (exact && (flags & HRT_FLAG_DUAL_KERNEL)) return fail(HRT_ERR_INVALID, "render: no exact-only build of the two-stream kernel");
    const bool dual_kernel = !exact && !stream_kernel && (g_rt.use_dual || (flags & HRT_FLAG_DUAL_KERNEL)) && s->d.n_meshes > 0u &&
                             !(flags & HRT_FLAG_WAVE_KERNEL);
    uint32_t grid, lds_bytes;
    if (stream_kernel) {
        const uint32_t fixed = (uint32_t)((HRT_SP_GLOBAL ? 0 : SP_FIELDS * HRT_SP_POOL * 4) + HRT_SP_NQ * HRT_SP_POOL * 2 + HRT_SP_STREAMS * sizeof(SpCtl) + sizeof(SpShared) + HRT_SP_UNITS * sizeof(SpUnit) + HRT_SP_UNITS * HRT_SP_MAXG * 4) +
                               2048u + s->d.tab_rows * 16u  // + the scene's per-object tables (stream_tables_fit)
#ifdef HRT_WALK_SEG
                               + 2048u  // diagnostic build: 16 accumulators per wave
#endif
                               ;
        uint32_t per_cu = (64u * 4u * HRT_SP_MINW) / HRT_SP_WG;  // workgroups resident per CU (HRT_SP_MINW waves per SIMD in all) ...
        while (per_cu > 1u && 160u * 1024u / per_cu < fixed + 16u * 1024u) --per_cu;  // ... as far as the LDS pools allow
        const uint32_t room = (160u * 1024u / per_cu - fixed) / 16u;
        if (!(flags & HRT_FLAG_NO_LDS_TREE)) R.lds_units = std::min<uint32_t>(s->d.n_kd_units, room) & ~3u;  // whole 64-byte lines: no treelet or leaf straddles
        lds_bytes = fixed + R.lds_units * 16u;
        grid = std::min<uint32_t>((uint32_t)g_rt.cus * per_cu, R.tiles_owned);
        {   // tiles per work unit: as many as keep one unit (tiles x 64 pixels x samples per fold) within HRT_SP_UNIT paths
            const uint32_t per_tile = 64u * std::min<uint32_t>(spp, HRT_SP_SCHUNK);
            uint32_t glog = 0;
            while ((2u << glog) <= HRT_SP_MAXG && (per_tile << (glog + 1u)) <= HRT_SP_UNIT) ++glog;
            R.sp_group_log2 = glog;
            // Few, heavy tiles (a rank's share of a frame at thousands of samples per pixel): the launch ends when the last
            // workgroup finishes its last item, and an item is a whole tile's samples -- in order, so a tile cannot be split
            // across workgroups by samples.  Split it by ROWS instead: bands of 4, 2 rows until a workgroup has ~64 items.
            uint32_t band = 0;
            while (glog == 0u && band < 2u && ((uint64_t)R.tiles_owned << band) < 64ull * grid && (64u >> (band + 1u)) * (uint64_t)std::min<uint32_t>(spp, HRT_SP_SCHUNK) >= 4096u) ++band;
            R.sp_band_log2 = band;
        }
        const size_t need_floats = (size_t)grid * HRT_SP_UNITS * HRT_SP_UNIT * 3u;  // HRT_SP_UNITS units in flight per workgroup
        if (s->sp_scratch_cap < need_floats) {
            if (s->sp_scratch) (void)hipFree(s->sp_scratch);
            s->sp_scratch = nullptr; s->sp_scratch_cap = 0;
            HIP_TRY(hipMalloc((void **)&s->sp_scratch, need_floats * sizeof(float)));
            s->sp_scratch_cap = need_floats;
        }
        R.sp_scratch = s->sp_scratch;
        if (HRT_SP_GLOBAL) {
            const size_t need_words = (size_t)grid * SP_FIELDS * HRT_SP_POOL;
            if (s->sp_pool_cap < need_words) {
                if (s->sp_pool) (void)hipFree(s->sp_pool);
                s->sp_pool = nullptr; s->sp_pool_cap = 0;
                HIP_TRY(hipMalloc((void **)&s->sp_pool, need_words * sizeof(uint32_t)));
                s->sp_pool_cap = need_words;
            }
        }
        R.sp_pool = s->sp_pool;
    } else {
        const void *kfn = exact ? (s->d.n_lights ? (const void *)hrt_trace_kernel_lights_exact : (const void *)hrt_trace_kernel_exact)
                                : (s->d.n_lights ? (const void *)hrt_trace_kernel_lights : (const void *)hrt_trace_kernel);
        lds_bytes = R.lds_units * 16u;
        if (dual_kernel) {
            // 4 workgroups per CU: 160 KiB = 4 x (27 KiB of backed-up streams + 12 KiB of nodelets)
            const uint32_t wgs = 1024u / HRT_WG, backup = HRT_DS_FIELDS * HRT_WG * 4u;
            const uint32_t room = (156u * 1024u / wgs - backup) / 16u;
            if (R.lds_units > room) R.lds_units = room & ~3u;
            lds_bytes = R.lds_units * 16u + backup;
            kfn = s->d.n_lights ? (const void *)hrt_trace2_kernel_lights : (const void *)hrt_trace2_kernel;
        }
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, HRT_WG, lds_bytes));
        if (per_cu < 1) per_cu = 1;
        grid = (uint32_t)(per_cu * g_rt.cus);
        // 4 waves per workgroup, one tile (two in the dual-stream kernel) per wave at a time
        const uint32_t per_wg = (HRT_WG / 64u) * (dual_kernel ? 2u : 1u);
        const uint32_t need = (R.tiles_owned + per_wg - 1u) / per_wg;
        if (grid > need) grid = need;
        R.sp_scratch = nullptr;
        R.sp_pool = nullptr;
    }
    s->last_grid = grid;
    s->last_lds = lds_bytes;
    s->last_waves = stream_kernel ? grid * (HRT_SP_WG / 64) : grid * (HRT_WG / 64u);
    // One hrt_scene carries ONE launch at a time (work-queue head, stamps, path pool, camera block).  Launches on one
    // stream are ordered by the stream; a launch on another stream first waits for the previous one.
    if (s->timed && s->last_stream != stream) HIP_TRY(hipStreamWaitEvent(stream, s->ev1, 0));
    s->last_stream = stream;
    HIP_TRY(hipMemsetAsync(s->tile_counter, 0, sizeof(uint32_t), stream));
    HIP_TRY(hipMemsetAsync(s->stamps, 0, 16 * sizeof(unsigned long long), stream));  // [15] = give-up code of the streaming kernel
    HIP_TRY(hipEventRecord(s->ev0, stream));
    if (stream_kernel && exact) {
        if (s->d.n_lights) hipLaunchKernelGGL(hrt_wgstream_kernel_lights_exact, dim3(grid), dim3(HRT_SP_WG), lds_bytes, stream, R);
        else hipLaunchKernelGGL(hrt_wgstream_kernel_exact, dim3(grid), dim3(HRT_SP_WG), lds_bytes, stream, R);
    } else if (stream_kernel && s->d.n_spheres >= HRT_SPHERE_FILTER_MIN && s->d.n_spheres <= 128u) {  // a crowd of spheres: the builds with the pair filter
        if (s->d.n_lights) hipLaunchKernelGGL(hrt_wgstream_kernel_lights_sph, dim3(grid), dim3(HRT_SP_WG), lds_bytes, stream, R);
        else hipLaunchKernelGGL(hrt_wgstream_kernel_sph, dim3(grid), dim3(HRT_SP_WG), lds_bytes, stream, R);
    } else if (stream_kernel) {
        if (s->d.n_lights) hipLaunchKernelGGL(hrt_wgstream_kernel_lights, dim3(grid), dim3(HRT_SP_WG), lds_bytes, stream, R);
        else hipLaunchKernelGGL(hrt_wgstream_kernel, dim3(grid), dim3(HRT_SP_WG), lds_bytes, stream, R);
    } else if (exact) {
        if (s->d.n_lights) hipLaunchKernelGGL(hrt_trace_kernel_lights_exact, dim3(grid), dim3(HRT_WG), lds_bytes, stream, R);
        else hipLaunchKernelGGL(hrt_trace_kernel_exact, dim3(grid), dim3(HRT_WG), lds_bytes, stream, R);
    } else {
        if (dual_kernel) {
            if (s->d.n_lights) hipLaunchKernelGGL(hrt_trace2_kernel_lights, dim3(grid), dim3(HRT_WG), lds_bytes, stream, R);
            else hipLaunchKernelGGL(hrt_trace2_kernel, dim3(grid), dim3(HRT_WG), lds_bytes, stream, R);
        } else if (s->d.n_lights) hipLaunchKernelGGL(hrt_trace_kernel_lights, dim3(grid), dim3(HRT_WG), lds_bytes, stream, R);
        else hipLaunchKernelGGL(hrt_trace_kernel, dim3(grid), dim3(HRT_WG), lds_bytes, stream, R);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev1, stream));
    if (flags & HRT_FLAG_GAMMA) {
        const uint32_t n = R.tiles_owned * 64u * 3u;
        hipLaunchKernelGGL(hrt_gamma_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_tiles, n);
        HIP_TRY(hipGetLastError());
    }
    s->timed = true;
    return HRT_OK;
}

int hrt_render_tiles(hrt_scene *s, const hrt_camera *cam, uint32_t w, uint32_t h, uint32_t spp, uint64_t seed,
                     uint32_t flags, uint32_t rank, uint32_t world, float *d_tiles, void *stream) {
    return launch_trace(s, cam, w, h, 0u, spp, seed, flags, rank, world, d_tiles, stream, false);
}

int hrt_render_accumulate(hrt_scene *s, const hrt_camera *cam, uint32_t w, uint32_t h, uint32_t first_sample, uint32_t n_samples,
                          uint64_t seed, uint32_t flags, uint32_t rank, uint32_t world, float *d_sum_tiles, void *stream) {
    return launch_trace(s, cam, w, h, first_sample, n_samples, seed, flags, rank, world, d_sum_tiles, stream, true);
}

int hrt_finalize_tiles(const float *d_sum_tiles, uint32_t n_tiles, uint32_t total_samples, uint32_t flags, float *d_tiles,
                       void *stream_) {
    if (!d_sum_tiles || !d_tiles || !total_samples) return fail(HRT_ERR_INVALID, "hrt_finalize_tiles: bad argument");
    if (!g_rt.ready) return fail(HRT_ERR_STATE, "hrt_finalize_tiles: call hrt_init first");
    if (!n_tiles) return HRT_OK;
    const uint32_t n = n_tiles * 64u * 3u;
    hipLaunchKernelGGL(hrt_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream_, d_sum_tiles, d_tiles, n,
                       total_samples, (flags & HRT_FLAG_GAMMA) ? 1u : 0u);
    HIP_TRY(hipGetLastError());
    return HRT_OK;
}

int hrt_encode_ppm(const float *d_frame, uint32_t w, uint32_t h, int format, unsigned char *d_out, size_t capacity,
                   size_t *bytes, void *stream_) {
    if (!d_frame || !d_out || !bytes || !w || !h) return fail(HRT_ERR_INVALID, "hrt_encode_ppm: bad argument");
    if (format != 3 && format != 6) return fail(HRT_ERR_INVALID, "hrt_encode_ppm: format must be 3 (ASCII) or 6 (binary)");
    if (!g_rt.ready) return fail(HRT_ERR_STATE, "hrt_encode_ppm: call hrt_init first");
    if ((uint64_t)w * h > 0x7fffffffull / 16u) return fail(HRT_ERR_INVALID, "hrt_encode_ppm: image too large");
    hipStream_t stream = (hipStream_t)stream_;
    char head[64];
    // main.cpp:258: "P3" endl w " " h endl 255 endl
    const int hl = std::snprintf(head, sizeof(head), "P%d\n%u %u\n255\n", format, w, h);
    const uint32_t npix = w * h;
    if (format == 6) {
        const size_t total = (size_t)hl + (size_t)npix * 3u;
        if (capacity < total) return fail(HRT_ERR_INVALID, "hrt_encode_ppm: output buffer too small (need " + std::to_string(total) + ")");
        HIP_TRY(hipMemcpyAsync(d_out, head, (size_t)hl, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(hrt_ppm6_kernel, dim3((npix * 3u + 255u) / 256u), dim3(256), 0, stream, d_frame, npix * 3u, d_out + hl);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));  // `head` is a stack buffer
        *bytes = total;
        return HRT_OK;
    }
    // P3, byte for byte what the reference's ofstream writes (main.cpp:259-261): per pixel "r g b " with each
    // value printed as a decimal int; pass 1 measures every pixel's text, a block scan turns lengths into
    // offsets, pass 2 writes the digits.  Blocks of 1024 pixels; block totals are scanned on the host side of
    // the launch (n/1024 words) to keep the device code to two simple kernels.
    const uint32_t nblocks = (npix + 1023u) / 1024u;
    uint32_t *d_len = nullptr;
    HIP_TRY(hipMalloc((void **)&d_len, ((size_t)npix + nblocks) * sizeof(uint32_t)));
    uint32_t *d_block = d_len + npix;
    hipLaunchKernelGGL(hrt_ppm3_measure_kernel, dim3(nblocks), dim3(256), 0, stream, d_frame, npix, d_len, d_block);
    std::vector<uint32_t> block(nblocks);
    hipError_t e = hipMemcpyAsync(block.data(), d_block, nblocks * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { (void)hipFree(d_len); return fail(HRT_ERR_DEVICE, std::string("hrt_encode_ppm: ") + hipGetErrorString(e)); }
    std::vector<unsigned long long> base(nblocks);
    unsigned long long run = (unsigned long long)hl;
    for (uint32_t b = 0; b < nblocks; ++b) { base[b] = run; run += block[b]; }
    const size_t total = (size_t)run + 1u;  // the closing endl
    if (capacity < total) { (void)hipFree(d_len); return fail(HRT_ERR_INVALID, "hrt_encode_ppm: output buffer too small (need " + std::to_string(total) + ")"); }
    unsigned long long *d_base = nullptr;
    e = hipMalloc((void **)&d_base, nblocks * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemcpyAsync(d_base, base.data(), nblocks * sizeof(unsigned long long), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_out, head, (size_t)hl, hipMemcpyHostToDevice, stream);
    const unsigned char nl = '\n';
    if (e == hipSuccess) e = hipMemcpyAsync(d_out + run, &nl, 1, hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(hrt_ppm3_write_kernel, dim3(nblocks), dim3(256), 0, stream, d_frame, npix, d_len, d_base, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_len);
    if (d_base) (void)hipFree(d_base);
    if (e != hipSuccess) return fail(HRT_ERR_DEVICE, std::string("hrt_encode_ppm: ") + hipGetErrorString(e));
    *bytes = total;
    return HRT_OK;
}

int hrt_check_last_launch(hrt_scene *s) {
    if (!s) return fail(HRT_ERR_INVALID, "hrt_check_last_launch: NULL scene");
    if (!s->timed) return HRT_OK;
    { const int drc = use_device(s->device); if (drc != HRT_OK) return drc; }
    HIP_TRY(hipEventSynchronize(s->ev1));
    unsigned long long gave_up = 0;
    HIP_TRY(hipMemcpy(&gave_up, s->stamps + 15, sizeof(gave_up), hipMemcpyDeviceToHost));
    if (gave_up) return fail(HRT_ERR_DEVICE, "trace kernel gave up (scheduler cycle bound exceeded): the frame of the last launch is incomplete");
    return HRT_OK;
}

int hrt_last_kernel_ms(hrt_scene *s, double *ms) {
    if (!s || !ms) return fail(HRT_ERR_INVALID, "hrt_last_kernel_ms: NULL argument");
    if (!s->timed) { *ms = 0.0; return HRT_OK; }
    {
        const int rc = hrt_check_last_launch(s);
        if (rc != HRT_OK) return rc;
    }
    float f = 0.f;
    HIP_TRY(hipEventElapsedTime(&f, s->ev0, s->ev1));
    *ms = (double)f;
    return HRT_OK;
}

int hrt_assemble_frame(const float *d_gathered, uint32_t tiles_per_rank_padded, uint32_t w, uint32_t h, uint32_t world,
                       float *d_frame, void *stream_) {
    if (!d_gathered || !d_frame || !w || !h || !world) return fail(HRT_ERR_INVALID, "hrt_assemble_frame: bad argument");
    if (tiles_per_rank_padded < hrt_tiles_owned(w, h, 0, world))
        return fail(HRT_ERR_INVALID, "hrt_assemble_frame: tiles_per_rank_padded too small");
    const uint32_t n = w * h;
    hipLaunchKernelGGL(hrt_assemble_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream_, d_gathered,
                       tiles_per_rank_padded, w, h, world, d_frame);
    HIP_TRY(hipGetLastError());
    return HRT_OK;
}

int hrt_debug_read_stamps(hrt_scene *s, uint64_t out[16]) {
    if (!s || !out) return fail(HRT_ERR_INVALID, "hrt_debug_read_stamps: NULL argument");
    { const int drc = use_device(s->device); if (drc != HRT_OK) return drc; }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, s->stamps, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return HRT_OK;
}

int hrt_kernel_info(hrt_stats *out) {
    if (!out) return fail(HRT_ERR_INVALID, "hrt_kernel_info: NULL argument");
    if (!g_rt.ready) return fail(HRT_ERR_STATE, "hrt_kernel_info: call hrt_init first");
    std::memset(out, 0, sizeof(*out));
    out->vgprs = (uint32_t)g_rt.attr.numRegs;
    out->lds_bytes = g_rt.lds_budget;
    return HRT_OK;
}

int hrt_render(hrt_scene *s, const hrt_camera *cam, uint32_t w, uint32_t h, uint32_t spp, uint64_t seed, uint32_t flags,
               float *out_rgb, hrt_stats *stats) {
    if (!out_rgb) return fail(HRT_ERR_INVALID, "hrt_render: NULL output");
    if (!s) return fail(HRT_ERR_INVALID, "hrt_render: NULL scene");
    if (!g_rt.ready) return fail(HRT_ERR_STATE, "render: call hrt_init first");
    { const int drc = use_device(s->device); if (drc != HRT_OK) return drc; }  // the frame and tile buffers below belong on the scene's device
    const auto t0 = std::chrono::steady_clock::now();
    const size_t tiles = hrt_tiles_total(w, h);
    const size_t tile_floats = tiles * 64 * 3, frame_floats = (size_t)w * h * 3;
    if (s->tiles_cap < tile_floats) {
        if (s->d_tiles) (void)hipFree(s->d_tiles);
        s->d_tiles = nullptr; s->tiles_cap = 0;
        HIP_TRY(hipMalloc((void **)&s->d_tiles, tile_floats * sizeof(float)));
        s->tiles_cap = tile_floats;
    }
    if (s->frame_cap < frame_floats) {
        if (s->d_frame) (void)hipFree(s->d_frame);
        s->d_frame = nullptr; s->frame_cap = 0;
        HIP_TRY(hipMalloc((void **)&s->d_frame, frame_floats * sizeof(float)));
        s->frame_cap = frame_floats;
    }
    int rc = hrt_render_tiles(s, cam, w, h, spp, seed, flags, 0, 1, s->d_tiles, nullptr);
    if (rc != HRT_OK) return rc;
    rc = hrt_assemble_frame(s->d_tiles, (uint32_t)tiles, w, h, 1, s->d_frame, nullptr);
    if (rc != HRT_OK) return rc;
    HIP_TRY(hipMemcpy(out_rgb, s->d_frame, frame_floats * sizeof(float), hipMemcpyDeviceToHost));
    rc = hrt_check_last_launch(s);  // never hand back a frame the kernel did not finish
    if (rc != HRT_OK) return rc;
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        double ms = 0.0;
        rc = hrt_last_kernel_ms(s, &ms);
        if (rc != HRT_OK) return rc;
        stats->kernel_ms = ms;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        stats->samples = (uint64_t)w * h * spp;
        stats->vgprs = (uint32_t)g_rt.attr.numRegs;
        stats->lds_bytes = s->last_lds;
        stats->waves_launched = s->last_waves;
    }
    return HRT_OK;
}

int hrt_render_aov(hrt_scene *s, const hrt_camera *cam, uint32_t w, uint32_t h, uint32_t which, float *out_rgb) {
    if (!out_rgb || which > 3u) return fail(HRT_ERR_INVALID, "hrt_render_aov: bad argument");
    DRender R;
    int rc = fill_render(s, cam, w, h, 1, 0, 0, 0, 1, R, nullptr);
    if (rc != HRT_OK) return rc;
    float *d = nullptr;
    const size_t bytes = (size_t)w * h * 3 * sizeof(float);
    HIP_TRY(hipMalloc((void **)&d, bytes));
    hipLaunchKernelGGL(hrt_aov_kernel, dim3((w * h + 255) / 256), dim3(256), 0, 0, R, which, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out_rgb, d, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(HRT_ERR_DEVICE, std::string("hrt_render_aov: ") + hipGetErrorString(e));
    return HRT_OK;
}

int hrt_debug_path_stream(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, float *out) {
    if (!out || !n) return fail(HRT_ERR_INVALID, "hrt_debug_path_stream: bad argument");
    if (!g_rt.ready) return fail(HRT_ERR_STATE, "hrt_debug_path_stream: call hrt_init first");
    float *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, n * sizeof(float)));
    hipLaunchKernelGGL(hrt_stream_kernel, dim3(1), dim3(64), 0, 0, (uint32_t)seed, (uint32_t)(seed >> 32), pixel, sample, n, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out, d, n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(HRT_ERR_DEVICE, std::string("hrt_debug_path_stream: ") + hipGetErrorString(e));
    return HRT_OK;
}

// Known-answer instrument: device functions on caller vectors (include/hrt.h).
int hrt_debug_kat(uint32_t which, const hrt_camera *cam, const float *prim, const float *in, uint32_t n, float *out) {
    if (!g_rt.ready) return fail(HRT_ERR_STATE, "hrt_debug_kat: call hrt_init first");
    if (!in || !out || !n) return fail(HRT_ERR_INVALID, "hrt_debug_kat: bad argument");
    static const uint32_t in_w[] = {2, 7, 7, 7, 7, 8, 3}, out_w[] = {12, 8, 2, 9, 8, 8, 3};
    if (which > HRT_KAT_NORMALIZE) return fail(HRT_ERR_INVALID, "hrt_debug_kat: unknown instrument");
    if ((which == HRT_KAT_CAMERA) != (cam != nullptr) || ((which >= HRT_KAT_TRIANGLE && which <= HRT_KAT_QUAD) != (prim != nullptr)))
        return fail(HRT_ERR_INVALID, "hrt_debug_kat: cam is for HRT_KAT_CAMERA, prim for the primitive instruments");
    std::vector<float4> rows, kat_qf;
    std::vector<float> box;
    float err_abs = 0.f;
    DCamera C;
    DScene kat_scene{};   // HRT_KAT_QUAD: carries nothing but the square's filter rows
    float4 *d_qf = nullptr;
    DScene *d_scene = nullptr;
    if (which == HRT_KAT_CAMERA) {
        const int rc = make_camera(cam, C);
        if (rc != HRT_OK) return rc;
    } else if (which == HRT_KAT_TRIANGLE) {  // prim: c0, c1, c2 as handed to the Triangle constructor
        const H3 c[3] = {{prim[0], prim[1], prim[2]}, {prim[3], prim[4], prim[5]}, {prim[6], prim[7], prim[8]}};
        std::vector<float4> rest;
        fold_triangle(c, 0u, rest, rows);           // plane first, then the HRT_TRI_ROWS others (hrt_kat_triangle_kernel)
        rows.insert(rows.end(), rest.begin(), rest.end());
    } else if (which == HRT_KAT_AABB) {
        box.assign(prim, prim + 6);
    } else if (which == HRT_KAT_SPHERE) {    // prim: centre, radius, motion
        rows.push_back(make_float4(prim[0], prim[1], prim[2], prim[3]));
        rows.push_back(make_float4(prim[4], prim[5], prim[6], as_float(0u)));
    } else if (which == HRT_KAT_QUAD) {      // prim: v0, v1, v3, motion, glass flag
        hrt_quad q;
        std::memset(&q, 0, sizeof(q));
        hrt_material m;
        std::memset(&m, 0, sizeof(m));
        for (int k = 0; k < 3; ++k) { q.v0[k] = prim[k]; q.v1[k] = prim[3 + k]; q.v3[k] = prim[6 + k]; m.motion[k] = prim[9 + k]; }
        m.type = prim[12] != 0.f ? HRT_MAT_GLASS : HRT_MAT_DIFFUSE;
        fold_quad(q, m, rows);
        build_quad_filter(rows, 1u, kat_qf, kat_scene.qf_n);
        double b = 0.0;
        for (int k = 0; k < 13; ++k) b = std::max(b, (double)std::fabs(prim[k]));
        for (size_t k = 0; k < (size_t)n * 7; ++k) if (k % 7 < 3) b = std::max(b, (double)std::fabs(in[k]));
        err_abs = 2e-6f * ((float)(b * 4.0) + 1.f);  // the margin scale fill_render derives from the scene extent
    }
    void *d_prim = nullptr;
    float *d_in = nullptr, *d_out = nullptr;
    hipError_t e = hipSuccess;
    const size_t in_bytes = (size_t)n * in_w[which] * sizeof(float), out_bytes = (size_t)n * out_w[which] * sizeof(float);
    const void *h_prim = which == HRT_KAT_CAMERA ? (const void *)&C : (which == HRT_KAT_AABB ? (const void *)box.data() : (const void *)rows.data());
    const size_t prim_bytes = which == HRT_KAT_CAMERA ? sizeof(C) : (which == HRT_KAT_AABB ? 6 * sizeof(float) : rows.size() * sizeof(float4));
    if (prim_bytes) e = hipMalloc(&d_prim, prim_bytes);
    if (e == hipSuccess && prim_bytes) e = hipMemcpy(d_prim, h_prim, prim_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&d_in, in_bytes);
    if (e == hipSuccess) e = hipMemcpy(d_in, in, in_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, out_bytes);
    if (e == hipSuccess) {
        const dim3 grid((n + 255u) / 256u), block(256);
        switch (which) {
            case HRT_KAT_CAMERA: hipLaunchKernelGGL(hrt_kat_camera_kernel, grid, block, 0, 0, (const DCamera *)d_prim, d_in, n, d_out); break;
            case HRT_KAT_TRIANGLE: hipLaunchKernelGGL(hrt_kat_triangle_kernel, grid, block, 0, 0, (const float4 *)d_prim, d_in, n, d_out); break;
            case HRT_KAT_AABB: hipLaunchKernelGGL(hrt_kat_aabb_kernel, grid, block, 0, 0, (const float *)d_prim, d_in, n, d_out); break;
            case HRT_KAT_SPHERE: hipLaunchKernelGGL(hrt_kat_sphere_kernel, grid, block, 0, 0, (const float4 *)d_prim, d_in, n, d_out); break;
            case HRT_KAT_QUAD:
                e = hipMalloc((void **)&d_qf, kat_qf.size() * sizeof(float4));
                if (e == hipSuccess) e = hipMemcpy(d_qf, kat_qf.data(), kat_qf.size() * sizeof(float4), hipMemcpyHostToDevice);
                kat_scene.qfilter = d_qf;
                if (e == hipSuccess) e = hipMalloc((void **)&d_scene, sizeof(DScene));
                if (e == hipSuccess) e = hipMemcpy(d_scene, &kat_scene, sizeof(DScene), hipMemcpyHostToDevice);
                if (e == hipSuccess) hipLaunchKernelGGL(hrt_kat_quad_kernel, grid, block, 0, 0, (const float4 *)d_prim, (const DScene *)d_scene, d_in, n, err_abs, d_out);
                break;
            case HRT_KAT_OPTICS: hipLaunchKernelGGL(hrt_kat_optics_kernel, grid, block, 0, 0, d_in, n, d_out); break;
            default: hipLaunchKernelGGL(hrt_kat_normalize_kernel, grid, block, 0, 0, d_in, n, d_out); break;
        }
        if (e == hipSuccess) e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, d_out, out_bytes, hipMemcpyDeviceToHost);
    if (d_qf) (void)hipFree(d_qf);
    if (d_scene) (void)hipFree(d_scene);
    if (d_prim) (void)hipFree(d_prim);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return fail(HRT_ERR_DEVICE, std::string("hrt_debug_kat: ") + hipGetErrorString(e));
    return HRT_OK;
}

// main.cpp:252-262: "P3", one line of "(int)(255*min(1,c))" triples.
int hrt_write_ppm(const char *path, const float *rgb, uint32_t w, uint32_t h) {
    if (!path || !rgb) return fail(HRT_ERR_INVALID, "hrt_write_ppm: NULL argument");
    FILE *f = std::fopen(path, "wb");
    if (!f) return fail(HRT_ERR_IO, std::string("Could not open file: ") + path);
    std::fprintf(f, "P3\n%u %u\n255\n", w, h);
    const size_t n = (size_t)w * h;
    for (size_t i = 0; i < n; ++i) {
        int c[3];
        for (int k = 0; k < 3; ++k) c[k] = (int)(255.f * std::min<float>(1.f, rgb[3 * i + k]));
        std::fprintf(f, "%d %d %d ", c[0], c[1], c[2]);
    }
    std::fprintf(f, "\n");
    std::fclose(f);
    return HRT_OK;
}

#include "hrt_multi.hip"

int hrt_kd_build_gpu(const hrt_kd_build_input *in, hrt_kd_build_output *out, void *user) {
    (void)user;
    if (!in || !out || (in->n_refs && (!in->ids || !in->lo || !in->hi))) return fail(HRT_ERR_INVALID, "hrt_kd_build_gpu: bad argument");
    if (!g_rt.ready) return fail(HRT_ERR_STATE, "hrt_kd_build_gpu: call hrt_init first");
    std::memset(out, 0, sizeof(*out));
    {
        const int drc = use_device(g_rt.device);
        if (drc != HRT_OK) return drc;
    }
    try {
        return kd_build_gpu_impl(in, out);
    } catch (const std::exception &e) {
        return fail(HRT_ERR_STATE, std::string("hrt_kd_build_gpu: ") + e.what());
    }
}

}  // extern "C"
