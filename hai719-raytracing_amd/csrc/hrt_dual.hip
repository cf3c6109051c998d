// hrt_trace2_kernel -- the lane-per-pixel megakernel with TWO pixel streams per lane (mesh scenes).
// Included by hrt_api.hip after hrt_kernels.hip; every arithmetic function is the one the single-stream
// kernel uses, so results are bit-identical to it.
//
// WHY.  In hrt_trace_kernel the KD walk (stage B) only runs for the lanes whose ray entered a mesh box:
// about 30 % of the rays on Cornell+mesh, so the most expensive stage runs at 15-25 % lane occupancy,
// and the lanes parked for it idle through the other lanes' stage A/C iterations.  Here a wave owns two
// 8x8 tiles at a time and each lane two pixels, one per tile: the state of one pixel stream lives in
// registers ("active"), the other in a lane-private LDS slot ("backup", 27 dwords, SoA, conflict free,
// no atomics and no barriers: nobody else ever touches it).  A lane whose active stream is parked for
// stage B swaps to its other stream and keeps tracing; stage B fires when most lanes have SOME stream
// parked (HRT_DS_ANY), or too many lanes have nothing else left to do (HRT_DS_BLOCKED), and then walks
// up to two rays per lane.  Stage B therefore runs with most lanes occupied, and A/C lose lanes only
// when both of a lane's streams wait.
//
// Each stream is one pixel with its samples in order, exactly as in the single-stream kernel, so the
// schedule cannot change random numbers, arithmetic, or the order of the pixel sum (main.cpp:193).
#include "hrt_device.h"

#ifndef HRT_DS_ANY
#define HRT_DS_ANY 48      // lanes with a parked stream that trigger stage B
#endif
#ifndef HRT_DS_BLOCKED
#define HRT_DS_BLOCKED 16  // lanes with nothing runnable that trigger stage B
#endif
#ifndef HRT_DS_SECOND
#define HRT_DS_SECOND 8    // lanes with a second parked stream that make a second pass worth while
#endif
#ifndef HRT_DS_TRIPS
#define HRT_DS_TRIPS 8     // KD-walk trips per stage B visit; an unfinished walk resumes at the next visit
#endif
#ifndef HRT_WALK_LEVELS
#define HRT_WALK_LEVELS 2  // treelets (two levels each) descended per trip; with single 16-byte nodes, levels per trip on Cornell+mesh / mesh_in_box: 1 -> 48.3 / 59.8 ms, 2 -> 44.2 / 54.6, 3 -> 43.5 / 53.0, 4 -> 43.6 / 52.7
#endif
#ifndef HRT_WALK_ROPES
#define HRT_WALK_ROPES 1   // 1: request the rope nodelets together with the leaf header (one round trip less per cell; 1080p@64 ms Cornell+mesh / mesh_in_box / pool: 0 -> 54.4 / 61.8 / 109.2, 1 -> 53.9 / 61.0 / 107.2)
#endif
#define HRT_DS_FIELDS 34   // dwords of one backed-up stream
#define HRT_DS_NONE 0xFFFFFFFFu

namespace hrtk {

// A KD walk that can stop after any trip and resume later (same trips, same order, same arithmetic
// as mesh_traverse: only where the loop is cut changes).
struct Walk {
    uint32_t ref;      // next nodelet, HRT_KD_NIL = no walk in progress
    float t_entry;     // entry distance of the current cell
    uint32_t kk;       // cells crossed so far << 16 | triangle cursor of the current leaf (0xFFFF: leaf not entered yet)
    float best_t;      // closest triangle of THIS mesh so far (KDTree.cpp:44), merged into the hit when the walk ends
    uint32_t best_tri;
    float bu, bv;
};

struct PathState {
    Ray ray;
    f3 thr, rad, sum;
    Rng rng;
    Hit h;
    Walk w;
    uint32_t parked;   // meshes still to be walked for the current ray
    uint32_t stage;    // 0: needs stage A   1: parked for stage B   2: ready for stage C
    uint32_t s;        // next sample of the pixel
    int remaining;     // bounces left on the current path; 0 = needs a new path
    bool live;         // the pixel still has samples to trace
};

__device__ __forceinline__ uint32_t ds_flags(const PathState &p) { return p.stage | ((uint32_t)p.remaining << 2) | (p.live ? 0x100u : 0u); }
__device__ __forceinline__ bool fl_live(uint32_t fl) { return (fl & 0x100u) != 0u; }
__device__ __forceinline__ bool fl_waiting(uint32_t fl) { return (fl & 0x103u) == 0x101u; }
__device__ __forceinline__ bool fl_runnable(uint32_t fl) { return fl_live(fl) && (fl & 3u) != 1u; }

// the float and integer fields of a stream in their LDS order
#define HRT_DS_FLOATS(F)                                                                              \
    F(0, p.ray.o.x) F(1, p.ray.o.y) F(2, p.ray.o.z) F(3, p.ray.d.x) F(4, p.ray.d.y) F(5, p.ray.d.z)   \
    F(6, p.ray.time) F(7, p.thr.x) F(8, p.thr.y) F(9, p.thr.z) F(10, p.rad.x) F(11, p.rad.y)          \
    F(12, p.rad.z) F(13, p.sum.x) F(14, p.sum.y) F(15, p.sum.z) F(16, p.h.t) F(17, p.h.a0) F(18, p.h.a1) \
    F(27, p.w.t_entry) F(28, p.w.best_t) F(29, p.w.bu) F(30, p.w.bv)
#define HRT_DS_UINTS(U) U(19, p.rng.k0) U(20, p.rng.k1) U(21, p.rng.i) U(22, p.h.tri) U(23, p.parked) U(24, p.s) \
    U(31, p.w.ref) U(32, p.w.kk) U(33, p.w.best_tri)
// 25: hit kind | index   26: flags

__device__ __forceinline__ void ds_store(uint32_t *sb, const PathState &p) {
    const uint32_t i = threadIdx.x;
#define F(k, v) sb[(k) * (uint32_t)HRT_WG + i] = __float_as_uint(v);
#define U(k, v) sb[(k) * (uint32_t)HRT_WG + i] = (v);
    HRT_DS_FLOATS(F) HRT_DS_UINTS(U)
#undef F
#undef U
    sb[25u * (uint32_t)HRT_WG + i] = (p.h.kind << 28) | p.h.index;
    sb[26u * (uint32_t)HRT_WG + i] = ds_flags(p);
}

__device__ __forceinline__ void ds_load(const uint32_t *sb, PathState &p) {
    const uint32_t i = threadIdx.x;
#define F(k, v) v = __uint_as_float(sb[(k) * (uint32_t)HRT_WG + i]);
#define U(k, v) v = sb[(k) * (uint32_t)HRT_WG + i];
    HRT_DS_FLOATS(F) HRT_DS_UINTS(U)
#undef F
#undef U
    const uint32_t hid = sb[25u * (uint32_t)HRT_WG + i], fl = sb[26u * (uint32_t)HRT_WG + i];
    p.h.kind = hid >> 28; p.h.index = hid & 0x0FFFFFFFu;
    p.stage = fl & 3u; p.remaining = (int)((fl >> 2) & 63u); p.live = fl_live(fl);
}

// registers <-> this lane's LDS slot; returns the flags of the stream that went to LDS
__device__ __forceinline__ uint32_t ds_swap(uint32_t *sb, PathState &p) {
    PathState q;
    ds_load(sb, q);
    const uint32_t fl = ds_flags(p);
    ds_store(sb, p);
    p = q;
    return fl;
}

__device__ __forceinline__ void ds_fresh(PathState &p, bool live) {
    p.ray.o = mk(0.f, 0.f, 0.f); p.ray.d = mk(0.f, 0.f, 1.f); p.ray.time = 0.f;
    p.thr = mk(1.f, 1.f, 1.f); p.rad = mk(0.f, 0.f, 0.f); p.sum = mk(0.f, 0.f, 0.f);
    p.rng.k0 = p.rng.k1 = p.rng.i = 0;
    p.h.kind = 0; p.h.index = 0; p.h.t = HRT_FLT_MAX; p.h.tri = 0; p.h.a0 = 0.f; p.h.a1 = 0.f;
    p.w.ref = HRT_KD_NIL; p.w.t_entry = 0.f; p.w.kk = 0xFFFFu; p.w.best_t = HRT_FLT_MAX; p.w.best_tri = 0; p.w.bu = 0.f; p.w.bv = 0.f;
    p.parked = 0; p.stage = 0; p.s = 0; p.remaining = 0; p.live = live;
}

// Up to `trips` trips of the walk of mesh M (mesh_traverse's loop body, KDTree.cpp:31-85 semantics);
// true when the walk is complete: w.best_* then hold the mesh's closest triangle with t >= 0, if any.
#ifdef HRT_WALK_SEG  // diagnostic: every stamp drains the wave's memory counters first, so the segments are serialised.  cx.st points at
                     // 16 u64 accumulators of THIS wave in LDS (hrt_stream.hip); the first active lane adds the wave's clocks
typedef unsigned long long __attribute__((address_space(3))) *lu64;
#define WSEG_ADD(k, v) do { const unsigned long long v_ = (unsigned long long)(v); if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__ballot(true))) ((lu64)(uint32_t)(uintptr_t)cx.st)[k] += v_; } while (0)
#define WSEG_START() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); unsigned long long wseg_last = __builtin_readcyclecounter()
#define WSEG(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_readcyclecounter(); \
                     WSEG_ADD(k, t_ - wseg_last); wseg_last = t_; } while (0)
#define WCOUNT(k, v) WSEG_ADD(k, v)
#else
#define WSEG_START() do { } while (0)
#define WSEG(k) do { } while (0)
#define WCOUNT(k, v) do { } while (0)
#endif
// Start of a walk: the mesh's irregular triangles (not in the tree) and the clip of the ray to the root cell.  False: the ray
// misses the tree, w.best_* is final.
template <class CX, class MP>  // cmesh: wave-uniform mesh, scalar record loads; gmesh: every lane its own mesh, vector loads
__device__ __forceinline__ bool mesh_walk_start(const CX &cx, MP M, const Ray &ray, f3 inv, Walk &w) {
    float t_entry = 0.f, t_scene_exit = HRT_FLT_MAX;
    float t0 = (M->kd_lo[0] - ray.o.x) * inv.x, t1 = (M->kd_hi[0] - ray.o.x) * inv.x;
    t_entry = fmaxf(t_entry, fminf(t0, t1)); t_scene_exit = fminf(t_scene_exit, fmaxf(t0, t1));
    t0 = (M->kd_lo[1] - ray.o.y) * inv.y; t1 = (M->kd_hi[1] - ray.o.y) * inv.y;
    t_entry = fmaxf(t_entry, fminf(t0, t1)); t_scene_exit = fminf(t_scene_exit, fmaxf(t0, t1));
    t0 = (M->kd_lo[2] - ray.o.z) * inv.z; t1 = (M->kd_hi[2] - ray.o.z) * inv.z;
    t_entry = fmaxf(t_entry, fminf(t0, t1)); t_scene_exit = fminf(t_scene_exit, fmaxf(t0, t1));
    w.best_t = HRT_FLT_MAX; w.best_tri = 0; w.bu = 0.f; w.bv = 0.f;
    (void)mesh_exceptions<CX::exact>(cx, M, ray, inv, w.best_t, w.best_tri, w.bu, w.bv);  // irregular triangles first
    const uint32_t root = M->root;
    if (!(t_entry <= t_scene_exit) || root == HRT_KD_NIL) return false;  // (a mesh whose triangles are all irregular or dead has no tree)
    w.ref = root; w.t_entry = t_entry; w.kk = 0xFFFFu;
    return true;
}
// The state of a walk between trips, unpacked
struct WalkCursor {
    uint32_t ref, k, count;
    float t_entry;
    f3 p;
    __device__ __forceinline__ void open(const Walk &w, const Ray &ray) {
        ref = w.ref; k = w.kk & 0xFFFFu; count = w.kk >> 16; t_entry = w.t_entry;
        p = ray.o + t_entry * ray.d;
    }
    __device__ __forceinline__ void close(Walk &w) const { w.ref = ref; w.t_entry = t_entry; w.kk = (count << 16) | k; }
};
// One trip: descend <= HRT_WALK_LEVELS treelets, then enter the leaf / test up to HRT_LEAF_BATCH of its triangles / leave it
// through a rope (mesh_traverse's loop body, KDTree.cpp:31-85 semantics).  c.ref == HRT_KD_NIL afterwards: the walk is complete.
template <class CX>
__device__ __forceinline__ void kd_trip(const CX &cx, gu4 g_units, const Soup &sp, uint32_t tri_base, const Ray &ray, f3 inv, WalkCursor &c, Walk &w) {
    WSEG_START();
    WCOUNT(8, 1); WCOUNT(11, __popcll(__ballot(true)));
#pragma unroll
    for (int lvl = 0; lvl < HRT_WALK_LEVELS; ++lvl)
        if (!(c.ref & HRT_KD_LEAF)) c.ref = kd_descend(g_units, cx, c.ref, c.p, ray.d);  // two levels each
    WSEG(0);  // descent
    if (c.ref & HRT_KD_LEAF) {
        WCOUNT(9, 1); WCOUNT(5, __popcll(__ballot(true)));
        const uint32_t lu = c.ref & ~HRT_KD_LEAF;
#if HRT_WALK_ROPES
        uint4 l0, l1, rp0, rp1;
        kd_fetch4(g_units, cx, lu, l0, l1, rp0, rp1);
#else
        uint4 l0, l1;
        kd_fetch2(g_units, cx, lu, l0, l1);
#endif
        const uint32_t first = tri_base + l0.w, cnt = l1.w;
        WSEG(1);  // leaf nodelets
        if (c.k == 0xFFFFu) c.k = 0;
        if (c.k < cnt) {
            WCOUNT(10, 1); WCOUNT(6, __popcll(__ballot(true)));
            bool found_ = false;
            c.k = tri_test_run(sp, first, cnt, c.k, ray, w.best_t, w.best_tri, w.bu, w.bv, found_);
        }
        WSEG(2);  // triangles
        if (c.k >= cnt) {  // leave the cell through its exit face
            const float ex = ((ray.d.x > 0.f ? __uint_as_float(l1.x) : __uint_as_float(l0.x)) - ray.o.x) * inv.x;
            const float ey = ((ray.d.y > 0.f ? __uint_as_float(l1.y) : __uint_as_float(l0.y)) - ray.o.y) * inv.y;
            const float ez = ((ray.d.z > 0.f ? __uint_as_float(l1.z) : __uint_as_float(l0.z)) - ray.o.z) * inv.z;
            float t_exit = HRT_FLT_MAX;
            uint32_t face = 6;
            if (ray.d.x != 0.f && ex < t_exit) { t_exit = ex; face = ray.d.x > 0.f ? 1u : 0u; }
            if (ray.d.y != 0.f && ey < t_exit) { t_exit = ey; face = ray.d.y > 0.f ? 3u : 2u; }
            if (ray.d.z != 0.f && ez < t_exit) { t_exit = ez; face = ray.d.z > 0.f ? 5u : 4u; }
            if (w.best_t <= t_exit || face == 6) {
                c.ref = HRT_KD_NIL;  // the closest hit lies inside the cells already visited
            } else {
                c.t_entry = fmaxf(c.t_entry, t_exit);
                c.p = ray.o + c.t_entry * ray.d;
#if HRT_WALK_ROPES
                const uint4 rp = (face >> 2) ? rp1 : rp0;
#else
                const uint4 rp = kd_fetch(g_units, cx, lu + 2 + (face >> 2));
#endif
                const uint32_t sel = face & 3u;
                c.ref = sel == 0 ? rp.x : (sel == 1 ? rp.y : (sel == 2 ? rp.z : rp.w));
                c.k = 0xFFFFu;
                if (++c.count >= HRT_WALK_CELLS) c.ref = HRT_KD_NIL;  // mesh_traverse's bound on the cells of one walk
            }
        }
        WSEG(3);  // exit face, rope
    }
}
// Up to `trips` trips of the walk of mesh M; true when the walk is complete: w.best_* then hold the mesh's closest triangle
// with t >= 0, if any.
template <class CX, class MP>
__device__ __forceinline__ bool mesh_walk(const CX &cx, MP M, const Ray &ray, f3 inv, Walk &w, int trips) {
    if (w.ref == HRT_KD_NIL && !mesh_walk_start(cx, M, ray, inv, w)) return true;
    gu4 g_units = (gu4)cx.S->kd_units;
    const Soup sp = soup_of(cx.S);
    const uint32_t tri_base = M->tri_base;
    WalkCursor c;
    c.open(w, ray);
    for (int trip = 0; trip < trips && c.ref != HRT_KD_NIL; ++trip) kd_trip(cx, g_units, sp, tri_base, ray, inv, c, w);
    c.close(w);
    return c.ref == HRT_KD_NIL;
}

// Up to `trips` trips on each mesh still to be walked, in mesh order (Scene.h:222-228); a finished mesh is merged
// into the closest hit with the caller's `t >= EPSILON && t < best` and leaves `parked`.  True when none is left.
template <class CX>
__device__ __forceinline__ bool walk_some(const CX &cx, const Ray &ray, uint32_t &parked, Walk &w, Hit &h, int trips) {
    if (CX::exact && (cx.flags & HRT_FLAG_MESH_BRUTE)) {  // proof build: no tree, every triangle of every gated mesh, at once
        meshes_hit(cx, ray, parked, h);
        parked = 0u;
        return true;
    }
    const f3 inv = ray_inv<CX::exact>(ray);
    const uint32_t nm = min(cx.S->n_meshes, 32u);
    for (uint32_t i = 0; i < nm; ++i) {  // wave-uniform loop: scalar mesh records
        if ((parked & (0u - parked)) == (1u << i)) {  // mesh i is this lane's next one
            if (mesh_walk(cx, (cmesh)cx.S->meshes + i, ray, inv, w, trips)) {
                const float t = w.best_t;
                if (t < HRT_FLT_MAX && t < h.t && HRT_T_ACCEPT(t)) {
                    h.kind = 3; h.index = i; h.t = t; h.tri = w.best_tri; h.a0 = w.bu; h.a1 = w.bv;
                }
                parked &= ~(1u << i);
            }
        }
    }
    return parked == 0u;
}

// The same with every lane on ITS next mesh at once (per-lane mesh records), in ONE flat loop of `trips` steps: a step is
// either the start of the lane's next mesh (irregular triangles, root clip) or one trip of the walk in progress; a finished
// mesh is merged and the lane moves on to its next one in the following step.  (Giving every mesh its own `trips` trips made a
// visit as long as its slowest lane's meshes together: 20 of 64 lanes active per trip on the three-mesh pool scene.)  Results
// cannot differ: a walk depends only on its ray and its mesh, and a lane's meshes are still merged in mesh order.
template <class CX>
__device__ __forceinline__ bool walk_some_per_lane(const CX &cx, const Ray &ray, uint32_t &parked, Walk &w, Hit &h, int trips) {
    if (CX::exact && (cx.flags & HRT_FLAG_MESH_BRUTE)) return walk_some(cx, ray, parked, w, h, trips);
    const f3 inv = ray_inv<CX::exact>(ray);
    const typename CX::tabmesh meshes = cx.tmesh;
    gu4 g_units = (gu4)cx.S->kd_units;
    const Soup sp = soup_of(cx.S);
    WalkCursor c;
    c.open(w, ray);
    for (int step = 0; step < trips && parked != 0u; ++step) {
        const uint32_t i = (uint32_t)__builtin_ctz(parked);
        const typename CX::tabmesh M = meshes + i;
        bool finished;
        if (c.ref == HRT_KD_NIL) {  // no walk in progress: start this lane's next mesh
            finished = !mesh_walk_start(cx, M, ray, inv, w);
            if (!finished) c.open(w, ray);
        } else {
            kd_trip(cx, g_units, sp, M->tri_base, ray, inv, c, w);
            finished = c.ref == HRT_KD_NIL;
        }
        if (finished) {
            const float t = w.best_t;
            if (t < HRT_FLT_MAX && t < h.t && HRT_T_ACCEPT(t)) {
                h.kind = 3; h.index = i; h.t = t; h.tri = w.best_tri; h.a0 = w.bu; h.a1 = w.bv;
            }
            parked &= parked - 1u;
            c.ref = HRT_KD_NIL;
        }
    }
    c.close(w);
    return parked == 0u;
}

// The T visit of the streaming kernel: the same walks, the same transitions of their state (mesh_walk_start, kd_descend, the
// leaf's triangles in batches of HRT_LEAF_BATCH, the exit face and its rope, the merge of a finished mesh in mesh order), but
// SCHEDULED BY VOTE.  A trip of kd_trip runs descent, leaf fetch, triangle batch and exit one after the other, each under the
// mask of the lanes that happen to need it: with 64 walks in different places the triangle batch -- half of a trip's
// instructions -- ran with 14-17 of 64 lanes (rocprofv3 on the pool scene: 0.83 of the SIMD cycles issue vector instructions at
// 0.59 lane utilisation: the walk is bound by instruction issue, not by memory).  Here every lane is in one of three states and
// each step executes the ONE block most lanes wait for:
//     S  no walk in progress: start the lane's next mesh (irregular triangles, root clip)
//     B  inside a leaf with triangles left: test the next HRT_LEAF_BATCH of them
//     M  anything else: leave a finished leaf through its exit face (or end the walk there), descend one treelet, fetch the
//        leaf the lane has arrived at
// so a block runs with at least a third -- in practice most -- of the lanes that still walk.  Which lane advances when cannot
// change a result: a walk depends on its ray and its mesh only.  `steps` blocks per visit; an unfinished walk keeps the same
// seven dwords of state as before (a lane inside a leaf re-fetches it at the next visit).
template <class CX>
__device__ __forceinline__ bool walk_vote(const CX &cx, const Ray &ray, uint32_t &parked, Walk &w, Hit &h, int steps) {
    if (CX::exact && (cx.flags & HRT_FLAG_MESH_BRUTE)) return walk_some(cx, ray, parked, w, h, steps);
    const f3 inv = ray_inv<CX::exact>(ray);
    const typename CX::tabmesh meshes = cx.tmesh;
    gu4 g_units = (gu4)cx.S->kd_units;
    const Soup sp = soup_of(cx.S);
    WalkCursor c;
    c.open(w, ray);
    bool in_leaf = false;  // the four units of the leaf c.ref names are in l0, l1, rp0, rp1
    uint4 l0 = make_uint4(0u, 0u, 0u, 0u), l1 = l0, rp0 = l0, rp1 = l0;
    uint32_t first = 0u, cnt = 0u;
    uint32_t tri_base = parked ? (meshes + (uint32_t)__builtin_ctz(parked))->tri_base : 0u;
    auto finish_mesh = [&]() {  // the walk of the lane's current mesh is complete: Scene.h:222-228's `t >= EPSILON && t < best`
        const float t = w.best_t;
        if (t < HRT_FLT_MAX && t < h.t && HRT_T_ACCEPT(t)) {
            h.kind = 3; h.index = (uint32_t)__builtin_ctz(parked); h.t = t; h.tri = w.best_tri; h.a0 = w.bu; h.a1 = w.bv;
        }
        parked &= parked - 1u;
        c.ref = HRT_KD_NIL;
        in_leaf = false;
    };
    for (int step = 0; step < steps; ++step) {
        const bool alive = parked != 0u;
        const bool want_s = alive && c.ref == HRT_KD_NIL;
        const bool want_b = alive && !want_s && in_leaf && c.k < cnt;
        const bool want_m = alive && !want_s && !want_b;
        const uint32_t n_s = (uint32_t)__popcll(__ballot(want_s)), n_b = (uint32_t)__popcll(__ballot(want_b)), n_m = (uint32_t)__popcll(__ballot(want_m));
        if ((n_s | n_b | n_m) == 0u) break;
        if (n_b >= n_m && n_b >= n_s) {
            if (want_b) {
                bool found_ = false;
                c.k = tri_test_run(sp, first, cnt, c.k, ray, w.best_t, w.best_tri, w.bu, w.bv, found_);
            }
        } else if (n_m >= n_s) {
            if (want_m) {
                if (in_leaf) {  // every triangle of the leaf has been tested: leave the cell through its exit face (kd_trip)
                    const float ex = ((ray.d.x > 0.f ? __uint_as_float(l1.x) : __uint_as_float(l0.x)) - ray.o.x) * inv.x;
                    const float ey = ((ray.d.y > 0.f ? __uint_as_float(l1.y) : __uint_as_float(l0.y)) - ray.o.y) * inv.y;
                    const float ez = ((ray.d.z > 0.f ? __uint_as_float(l1.z) : __uint_as_float(l0.z)) - ray.o.z) * inv.z;
                    float t_exit = HRT_FLT_MAX;
                    uint32_t face = 6;
                    if (ray.d.x != 0.f && ex < t_exit) { t_exit = ex; face = ray.d.x > 0.f ? 1u : 0u; }
                    if (ray.d.y != 0.f && ey < t_exit) { t_exit = ey; face = ray.d.y > 0.f ? 3u : 2u; }
                    if (ray.d.z != 0.f && ez < t_exit) { t_exit = ez; face = ray.d.z > 0.f ? 5u : 4u; }
                    in_leaf = false;
                    if (w.best_t <= t_exit || face == 6) {
                        finish_mesh();  // the closest hit lies inside the cells already visited
                    } else {
                        c.t_entry = fmaxf(c.t_entry, t_exit);
                        c.p = ray.o + c.t_entry * ray.d;
                        const uint4 rp = (face >> 2) ? rp1 : rp0;
                        const uint32_t sel = face & 3u;
                        c.ref = sel == 0 ? rp.x : (sel == 1 ? rp.y : (sel == 2 ? rp.z : rp.w));
                        c.k = 0xFFFFu;
                        if (++c.count >= HRT_WALK_CELLS) c.ref = HRT_KD_NIL;  // mesh_traverse's bound on the cells of one walk
                        if (c.ref == HRT_KD_NIL) finish_mesh();              // left the tree (or hit the bound): complete
                    }
                }
                const bool moving = parked != 0u && c.ref != HRT_KD_NIL;  // (a lane whose mesh has just ended starts its next one in an S step)
                if (moving && !(c.ref & HRT_KD_LEAF)) c.ref = kd_descend(g_units, cx, c.ref, c.p, ray.d);  // two levels
                if (moving && (c.ref & HRT_KD_LEAF)) {
                    kd_fetch4(g_units, cx, c.ref & ~HRT_KD_LEAF, l0, l1, rp0, rp1);
                    first = tri_base + l0.w; cnt = l1.w;
                    if (c.k == 0xFFFFu) c.k = 0;
                    in_leaf = true;
                }
            }
        } else {
            if (want_s) {
                const typename CX::tabmesh M = meshes + (uint32_t)__builtin_ctz(parked);
                tri_base = M->tri_base;
                if (mesh_walk_start(cx, M, ray, inv, w)) { c.open(w, ray); in_leaf = false; }
                else finish_mesh();
            }
        }
    }
    c.close(w);
    return parked == 0u;
}

// One stage B visit for a parked stream.
template <class CX>
__device__ __forceinline__ void walk_visit(const CX &cx, PathState &p) {
    if (walk_some(cx, p.ray, p.parked, p.w, p.h, HRT_DS_TRIPS)) p.stage = 2u;
}

template <bool LIGHTS>
__device__ __forceinline__ void trace_body_dual(const DRender &R) {
    extern __shared__ uint4 s_units[];
    Ctx cx;
    cx.S = (cscene)R.scene;
    cx.set_tables((gf4)cx.S->tabs, (gf1)c_u8_lut, cx.S);
    cx.lds = (lu4)s_units;
    cx.lds_n = R.lds_units;
    cx.err_abs = R.err_abs;
    cx.flags = R.flags;
    ccam cam = (ccam)R.cam;
    uint32_t *sb = reinterpret_cast<uint32_t *>(s_units + R.lds_units);  // HRT_DS_FIELDS x HRT_WG dwords
    {
        gu4 g_units = (gu4)cx.S->kd_units;
        for (uint32_t i = threadIdx.x; i < cx.lds_n; i += blockDim.x) s_units[i] = ld(g_units, i);
    }
    unsigned long long stamps_local[17];
    cx.st = stamps_local;
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_meshes = min(cx.S->n_meshes, 32u);  // >= 1 in this kernel
    const GateBox box0 = gate_box_of((cmesh)cx.S->meshes);
    // the two tile slots of the wave (wave-uniform), and this lane's pixel in each
    uint32_t tj[2], pxy[2];
    uint32_t cur = 0;     // which stream is in registers
    uint32_t bak_fl = 0;  // flags of the stream in LDS (mirror, so votes never read LDS)
    PathState p;

    auto pull = [&]() -> uint32_t {
        uint32_t j = 0;
        if (lane == 0) j = atomicAdd(R.tile_counter, 1u);
        j = __builtin_amdgcn_readfirstlane(j);
        return j < R.tiles_owned ? j : HRT_DS_NONE;  // the queue is finite: every wave runs dry
    };
    auto locate = [&](uint32_t j, uint32_t &xy) -> bool {  // pixel of this lane in tile slot j; false = outside the image
        if (j == HRT_DS_NONE) { xy = 0; return false; }
        const uint32_t tile = R.rank + j * R.world;
        const uint32_t px = (tile % R.tiles_x) * 8u + (lane & 7u), py = (tile / R.tiles_x) * 8u + (lane >> 3);
        xy = px | (py << 16);
        return px < R.w && py < R.h && R.spp > 0u;
    };

    auto fresh = [&](PathState &q, uint32_t j, bool in) {  // a stream at the start of tile slot j
        ds_fresh(q, in);
        if (R.accumulate && in) {  // progressive mode: continue the running sum of samples [0, s0)
            const float *a = R.out_tiles + ((size_t)j * 64u + lane) * 3u;
            q.sum = mk(a[0], a[1], a[2]);
        }
    };

    tj[0] = pull();
    tj[1] = pull();
    {
        const bool in1 = locate(tj[1], pxy[1]);
        fresh(p, tj[1], in1);
        ds_store(sb, p);
        bak_fl = ds_flags(p);
        const bool in0 = locate(tj[0], pxy[0]);
        fresh(p, tj[0], in0);
    }

#ifdef HRT_DS_PROF  // diagnostic build: per-wave cycle and lane-occupancy sums -> DRender::stamps (tools/variants.sh)
    unsigned long long prof[16];
    for (int k = 0; k < 16; ++k) prof[k] = 0ull;
    unsigned long long prof_last = __builtin_readcyclecounter();
#define DSP_T(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); \
                      __builtin_amdgcn_sched_barrier(0); prof[k] += t_ - prof_last; prof_last = t_; } while (0)
#define DSP_N(k, n) do { prof[k] += (unsigned long long)(n); } while (0)
#else
#define DSP_T(k) do { } while (0)
#define DSP_N(k, n) do { } while (0)
#endif
    for (;;) {
        // ---- tile turnover: a slot whose 64 pixels are finished is written out and refilled
#pragma unroll
        for (uint32_t X = 0; X < 2u; ++X) {
            if (tj[X] == HRT_DS_NONE) continue;
            const bool mine = cur == X;  // the stream of slot X is in this lane's registers
            if (__ballot(mine ? p.live : fl_live(bak_fl)) != 0ull) continue;
            f3 sum = p.sum;
            if (!mine) {
                const uint32_t i = threadIdx.x;
                sum = mk(__uint_as_float(sb[13u * (uint32_t)HRT_WG + i]), __uint_as_float(sb[14u * (uint32_t)HRT_WG + i]), __uint_as_float(sb[15u * (uint32_t)HRT_WG + i]));
            }
            const uint32_t px = pxy[X] & 0xFFFFu, py = pxy[X] >> 16;
            f3 c = mk(0.f, 0.f, 0.f);
            if (px < R.w && py < R.h) {
                const float nspp = R.accumulate ? 1.f : (float)R.spp;  // progressive mode stores the raw sum
                c = mk(sum.x / nspp, sum.y / nspp, sum.z / nspp);  // main.cpp:195
            }
            float *o = R.out_tiles + ((size_t)tj[X] * 64u + lane) * 3u;
            o[0] = c.x; o[1] = c.y; o[2] = c.z;
            tj[X] = pull();
            const bool in = locate(tj[X], pxy[X]);
            if (mine) {
                fresh(p, tj[X], in);
            } else {
                PathState q;
                fresh(q, tj[X], in);
                ds_store(sb, q);
                bak_fl = ds_flags(q);
            }
        }
        if (tj[0] == HRT_DS_NONE && tj[1] == HRT_DS_NONE) break;
        DSP_T(0); DSP_N(8, 1);

        // ---- the active stream cannot run and the other one can: swap
        if (!fl_runnable(ds_flags(p)) && fl_runnable(bak_fl)) { bak_fl = ds_swap(sb, p); cur ^= 1u; }
        DSP_T(1);
        DSP_N(9, __popcll(__ballot(p.live && p.stage == 0u)));

        // ---- stage A: (re)generate, spheres + squares, mesh gates
        if (p.live && p.stage == 0u) {
            if (p.remaining == 0) {  // next camera sample of this pixel (main.cpp:188-192)
                const uint32_t xy = cur ? pxy[1] : pxy[0];
                const uint32_t px = xy & 0xFFFFu, py = xy >> 16;
                p.rng.start(R.seed_lo, R.seed_hi, py * R.w + px, R.s0 + p.s);
                const float u = ((float)px + p.rng.next()) / (float)R.w;
                const float v = ((float)py + p.rng.next()) / (float)R.h;
                const float tm = p.rng.next();
                p.ray = camera_ray(cam, u, v, tm);
                p.thr = mk(1.f, 1.f, 1.f);
                p.rad = mk(0.f, 0.f, 0.f);
                p.remaining = 6;  // MAXBOUNCES
            }
            DSP_T(7);
            p.h = prims_hit(cx, p.ray);
            DSP_T(13);
            p.parked = mesh_gates_pre(cx, p.ray, n_meshes, box0);
            p.w.ref = HRT_KD_NIL;
            p.stage = p.parked ? 1u : 2u;
        }
        DSP_T(2);

        // ---- stage B: walk the meshes once enough lanes hold a parked stream
        {
            const bool aw = p.live && p.stage == 1u, bw = fl_waiting(bak_fl);
            const uint64_t any = __ballot(aw || bw);
            // a lane is blocked when its active stream waits and the other one cannot run either
            const uint64_t blocked = __ballot(aw && !fl_runnable(bak_fl));
            const uint64_t runnable = __ballot((p.live && p.stage != 1u) || fl_runnable(bak_fl));
            if (any != 0ull && (__popcll(any) >= HRT_DS_ANY || __popcll(blocked) >= HRT_DS_BLOCKED || runnable == 0ull)) {
#pragma nounroll
                for (int pass = 0; pass < 2; ++pass) {
                    const bool a2 = p.live && p.stage == 1u, b2 = fl_waiting(bak_fl);
                    const uint64_t want = __ballot(a2 || b2);
                    if (want == 0ull || (pass == 1 && __popcll(want) < HRT_DS_SECOND)) break;
                    if (!a2 && b2) { bak_fl = ds_swap(sb, p); cur ^= 1u; }  // bring the parked stream in
                    DSP_T(3); DSP_N(10, 1); DSP_N(11, __popcll(want));
                    if (p.live && p.stage == 1u) walk_visit(cx, p);
                    DSP_T(4);
                }
            } else if (aw && fl_runnable(bak_fl)) {  // not yet: trace the other stream meanwhile
                bak_fl = ds_swap(sb, p);
                cur ^= 1u;
            }
        }

        DSP_T(5);
        DSP_N(12, __popcll(__ballot(p.live && p.stage == 2u)));
        // ---- stage C: shade, scatter, end of path
        if (p.live && p.stage == 2u) {
            bool ended;
            if (p.h.kind == 0u) {
                p.rad = p.rad + p.thr * sky(cx, p.ray.d, p.remaining);
                ended = true;
            } else {
                DSP_T(6);
                const Surface sf = shade(cx, p.ray, p.h);
                DSP_T(14);
                f3 direct = mk(0.f, 0.f, 0.f);
                if (LIGHTS) direct = direct_light(cx, sf, p.ray, p.rng);
                p.rad = p.rad + p.thr * (direct + sf.emission);
                p.thr = p.thr * sf.albedo;
                scatter(sf, p.ray, p.rng);
                --p.remaining;
                ended = (p.remaining == 0);
            }
            if (ended) {
                p.sum = p.sum + mk(p.rad.x / 6.f, p.rad.y / 6.f, p.rad.z / 6.f);  // Scene.h:348
                p.remaining = 0;
                ++p.s;
                p.live = p.s < R.spp;
            }
            p.stage = 0u;
        }
        DSP_T(6);
    }
#ifdef HRT_DS_PROF
    if (lane == 0 && R.stamps)
        for (int k = 0; k < 15; ++k) atomicAdd(R.stamps + k, prof[k]);
#endif
}

}  // namespace hrtk

extern "C" __global__ void __launch_bounds__(HRT_WG, HRT_MIN_WAVES) hrt_trace2_kernel(const DRender R) { hrtk::trace_body_dual<false>(R); }
extern "C" __global__ void __launch_bounds__(HRT_WG, HRT_MIN_WAVES) hrt_trace2_kernel_lights(const DRender R) { hrtk::trace_body_dual<true>(R); }
