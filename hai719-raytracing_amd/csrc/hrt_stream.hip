// hrt_wgstream_kernel -- the workgroup-streaming form of the trace megakernel (included by hrt_api.hip after
// hrt_kernels.hip and hrt_dual.hip, whose exact-arithmetic device functions it reuses unchanged).
//
// WHY.  In hrt_trace_kernel a lane owns a pixel and walks all stages of every bounce itself; the PMC profile
// (profiles/r01_pmc.json) shows 40 % VALU lane utilisation on Cornell+mesh because on any given bounce the 64
// lanes of a wave want different things (new camera ray / mesh walk / shade a sphere, a square, a triangle).
// Here the divergent rays are COMPACTED.  A 1024-thread workgroup (one per CU) keeps a pool of HRT_SP_POOL
// paths as 128-byte records in global memory (sized to stay in the Infinity Cache) and queues of 16-bit slot
// ids in LDS.  One cycle runs 64-entry chunks of
//     G   free slot          -> camera ray (main.cpp:188-192), spheres + squares, mesh box gates
//     T   ray + best hit     -> up to HRT_SP_TRIPS trips of the rope KD walk; an unfinished walk keeps its state
//                               in the record and comes back next cycle, so T chunks stay full
//     H_k closest hit, by kind k = miss / sphere / square / mesh: sky | shade + direct light + scatter (path end
//                               -> free list), then spheres + squares + gates for the scattered ray
// G and H hand the path to T when its ray enters a mesh box, else to the hit queue of its kind; T does the same
// when the walk is complete.  Sorting by kind makes a chunk run ONE branch of shade().  Queues are double-
// buffered: a cycle consumes the "in" buffers, frozen at its start, and appends to the "out" buffers, so all
// stages run in the SAME cycle; the 16 waves pull chunks from one LDS cursor (T first: the longest).  Appends
// are wave-aggregated: one __ballot, one LDS atomicAdd by the leader lane, positions by popcount of the lower
// lanes (__shfl of the base).
//
// The pool is split into HRT_SP_STREAMS = 2 streams, each with its own slots, queues and cycle counter, and there is
// no barrier in the scheduler: a wave that finds no chunk left in a stream's cycle ARRIVES (an LDS counter) and goes on
// with the other stream; the wave whose arrival completes the cycle runs the serial section (new path numbers, buffer
// swap, unused free slots) and publishes the next cycle (see the comment at the scheduler, below).  Every control
// value a loop condition depends on is read back through readfirstlane, so those loops are provably wave-uniform
// for the compiler.  (With per-thread copies of the same values hipcc if-converted the loop exits into exec masks
// and waves left the rendezvous sequence at different points: wrong pixels, hangs.)  A cycle bound (1 << 16) and
// bounded spins end a run that a scheduling bug would otherwise spin; the host then reports HRT_ERR_DEVICE.
//
// DETERMINISM.  A path is keyed (pixel, sample) as before, so the schedule cannot change its random numbers or
// its arithmetic.  A work unit is up to 16 tiles x 64 pixels x the samples of one fold (<= HRT_SP_UNIT paths);
// HRT_SP_UNITS units are in flight, so that new paths come from the next unit while the long paths of the last one
// are still bouncing.  Finished samples go to the unit's part of a per-workgroup scratch, indexed by sample-major
// path number, and are added to the pixel sums in sample order once the unit's last path has finished: the fold is
// the reference's `image += color` order (main.cpp:193), bit for bit, whatever order paths finished in.
#include "hrt_device.h"

#ifndef HRT_SP_POOL
#define HRT_SP_POOL 4096   // paths resident per workgroup, 128-byte records.  A/B on MI355X, 1080p@256, ms Cornell+mesh / mesh_in_box / pool:
                           // 2048 -> 306 / 354 / 775, 4096 -> 278 / 320 / 673 (unit 32768), 8192 -> 360 / 391 / 730: pool + scratch of all
                           // 256 workgroups should stay within the 256 MB Infinity Cache
#endif
#ifndef HRT_SP_GLOBAL
#define HRT_SP_GLOBAL 1    // 1: the path pool lives in global memory (one 128-byte record per path, L2 / Infinity Cache
#endif                     //    resident), which lets HRT_SP_POOL grow past what LDS holds; 0: SoA arrays in LDS
#ifndef HRT_SP_WG
#define HRT_SP_WG 1024     // threads per workgroup (16 waves = 4 per SIMD, one workgroup per CU)
#endif
#ifndef HRT_SP_MINW
#define HRT_SP_MINW 4      // waves per SIMD the register allocator leaves room for (A/B builds: 5 with two 640-thread workgroups per CU)
#endif
#define HRT_SP_SCHUNK 1024 // most samples per pixel traced between two ordered folds
#ifndef HRT_SP_UNIT
#define HRT_SP_UNIT 8192   // most paths of one work unit = tiles of the group x 64 pixels x samples per fold (scratch: 12 B each).  1080p, ms
                           // Cornell+mesh @256 / pool @256 / Cornell+mesh @16 with units x size: 2 x 16384 -> 186.1 / 348.6 / 12.7, 4 x 8192 -> 186.7 /
                           // 349.8 / 12.3, 3 x 16384 -> 195.0 / 347.5 / 12.9, 2 x 32768 -> 201.6 / 352.2 / 12.8 (the scratch of all workgroups competes
                           // with the path pool for the Infinity Cache)
#endif
#ifndef HRT_SP_UNITS
#define HRT_SP_UNITS 4     // work units in flight per workgroup (1..4)
#endif
#define HRT_SP_MAXG 16     // most tiles per unit; a power of two
#ifndef HRT_SP_TRIPS
#define HRT_SP_TRIPS 6     // KD-walk trips per T visit (A/B 1080p@256, ms Cornell+mesh / mesh_in_box / pool: 6 -> 264 / 301 / 649, 8 -> 264 / 305 / 659, 12 -> 265 / 311 / 686);
                           // an unfinished walk goes back to the T queue with its state
#endif
#ifndef HRT_SP_TPRIO
#define HRT_SP_TPRIO 3     // s_setprio level of a wave while it runs a T chunk (0: none).  1080p@64, ms Cornell+mesh / mesh_in_box / pool:
                           // 0 -> 40.92 / 47.19 / 75.39, 1 -> 40.79 / 46.38 / 74.76, 3 -> 40.76 / 46.38 / 74.28
#endif
#ifndef HRT_SP_VOTE
#define HRT_SP_VOTE 0      // 1: T visits schedule the blocks of the walk by vote (hrt_dual.hip walk_vote); 0: HRT_SP_TRIPS trips of kd_trip
#endif
#ifndef HRT_SP_STEPS
#define HRT_SP_STEPS 16    // blocks per T visit of walk_vote
#endif
#ifndef HRT_SP_CYCLE_BOUND
#define HRT_SP_CYCLE_BOUND (1u << 16)  // consecutive scheduler cycles (of either stream) that may run NOTHING -- no chunk, no new path, no
#endif                                 // reduction -- before the workgroup gives up.  A cycle that runs any chunk is progress: every chunk
                                       // consumes something finite (a path has <= 6 hit visits; a walk crosses <= HRT_WALK_CELLS cells of a
                                       // tree hrt_scene_create has checked, and a leaf's cursor only advances), so the bound does not depend
                                       // on how long a legal walk is -- a leaf of 65 534 triangles is ~5 500 T visits of one path, all progress
#ifndef HRT_SP_WAIT_SECONDS
#define HRT_SP_WAIT_SECONDS 900ull     // how long a wave waits for the other waves of a cycle before it declares the launch dead (a watchdog)
#endif
#ifndef HRT_SP_GIVEUP_AFTER
#define HRT_SP_GIVEUP_AFTER 0u         // test build only (Makefile, libhrt_var_bound.so: 3): give up after that many serial sections whatever
#endif                                 // they ran, to exercise the path on which a launch reports HRT_ERR_DEVICE
#ifndef HRT_SP_STREAMS
#define HRT_SP_STREAMS 2   // independent path streams per workgroup (each with its own slots, queues and cycle counter).  With 2, a wave
#endif                     // that runs out of chunks in one stream's cycle does not idle at a barrier: it arrives (an LDS counter) and goes
                           // on with the other stream's cycle; the last wave to arrive prepares the stream's next cycle.  1 = one stream,
                           // the same code with nothing to overlap (equivalent to a barrier per cycle)
#ifndef HRT_SP_NOG5
#define HRT_SP_NOG5 1      // 1: hit visits do not load group 5 of the record (time, RNG keys, path number): recomputed from the number, kept in
                           // g7.w.  1080p@64, Msamples/s Cornell+mesh / mesh_in_box / pool / random_spheres: 0 -> 3 294 / 2 831 / 1 824 / 4 419, 1 -> 3 380 / 2 951 / 1 843 / 4 541
#endif
#ifndef HRT_SP_PM4
#define HRT_SP_PM4 1       // 1: the meshes still to walk and the walk ref travel in g4 and a T visit loads four groups (not g2); the sender's one store
                           // of g4 replaces the invariant "walk ref NIL outside T" and its two dword stores.  1080p@64, Msamples/s Cornell+mesh /
                           // mesh_in_box / pool: 0 -> 3 373 / 2 945 / 1 843, 1 -> 3 437 / 2 970 / 1 855
#endif
#ifndef HRT_SP_PMQ
#define HRT_SP_PMQ 1       // 1 (needs HRT_SP_PM4): in scenes of <= 4 meshes the meshes to walk ride in the T-queue entry beside the slot id; the sender stores
                           // no g4.  1080p@64, Msamples/s Cornell+mesh / mesh_in_box / pool: 0 -> 3 443 / 2 974 / 1 875, 1 -> 3 475 / 3 021 / 1 890
#endif
#ifndef HRT_SP_THALF
#define HRT_SP_THALF 0     // experiment: T chunks of 32 paths (half the lanes idle) -- latency- or throughput-bound?
#endif
#ifndef HRT_SP_DEFER
#define HRT_SP_DEFER 1     // 1: only whole chunks run while the fold still has paths to start (see the serial section)
#endif
#ifndef HRT_SP_PRUNE
#define HRT_SP_PRUNE 3     // exact path pruning (never in the proof builds; only when DScene::prune_ok): bit 0 -- a path whose throughput
#endif                     // has become exactly (0, 0, 0) ends (every later term is throughput x a finite value = 0); bit 1 -- on the LAST
                           // segment of a path in a scene without point lights only the emission of the closest hit can still reach the
                           // sample: when spheres + squares already give a closest hit that does not emit (a mesh in front of it emits
                           // nothing either, Scene.h:288-299), or nothing and the sky is dark, the path ends there -- no mesh walk, no sixth
                           // hit visit.  Same pixels bit for bit: rad + throughput x 0 == rad
#define HRT_SP_QCAP (HRT_SP_POOL / HRT_SP_STREAMS)  // slots, and entries per queue, of one stream
#define HRT_SP_NQ 8        // queues: T0 T1 A0 A1 B0 B1 F0 F1 (A: sphere hits from the front, quad hits from the back;
                           // B: misses from the front, mesh hits from the back -- a path sits in exactly one place)

namespace hrtk {

// One path = one 128-byte record.  Dwords 0-19 are what a T visit touches (ray, best hit, meshes to walk, walk
// state: five aligned 16-byte groups), 0-11 and 20-31 what a hit / new-path visit touches (it writes back 0-11 and 24-31).
enum { SP_OX = 0, SP_OY, SP_OZ, SP_DX, SP_DY, SP_DZ, SP_HT, SP_HID, SP_HA0, SP_HA1, SP_HTRI, SP_PM,
       SP_WREF, SP_WTE, SP_WKK, SP_WBT, SP_WTRI, SP_WBU, SP_WBV, SP_PAD,
       SP_TM, SP_K0, SP_K1, SP_N,                            // written once, when the path starts
       SP_TR, SP_TG, SP_TB, SP_RR, SP_RG, SP_RB, SP_RI, SP_REM,  // rewritten by every hit visit: two aligned 16-byte stores
       SP_FIELDS };  // 32 dwords: one 128-byte record

struct SpCtl {           // control block of ONE stream in LDS (28 dwords)
    uint32_t cQ[6][2];       // queue fills, [queue][parity]: 0 = T, 1..4 = the closest-hit queues by hit kind (0 miss, 1 sphere, 2 square,
                             // 3 mesh), 5 = free slots.  Contiguous: sp_push_all addresses them by queue number
    uint32_t cursor;         // chunk cursor of the running cycle
    uint32_t ngen, gen_n0;   // new paths of the running cycle and the number (within its unit's fold) of the first one
    uint32_t gen_slot, gen_s0;  // ... the unit slot they belong to and the first sample of that unit's fold
    uint32_t red_slot;       // unit slot whose finished fold this cycle reduces into the pixel sums (HRT_SP_UNITS: none), with its
    uint32_t red_j, red_s0, red_ns;  // first tile, first sample and samples per pixel
    uint32_t more;           // paths will still be started later (partial chunks may wait, see the serial section)
    uint32_t done, parity;
    uint32_t arrive;         // waves that have finished their part of a cycle of this stream, ever (monotonic)
    uint32_t ready;          // number of the cycle whose control values above are valid (monotonic; waves wait for it)
    uint32_t pad[2];
};
struct SpUnit {          // a WORK UNIT in flight: G tiles x 64 pixels x the samples of one fold (8 dwords); a workgroup keeps two
    uint32_t state;          // SP_U_*
    uint32_t j;              // first tile slot (of this rank's tiles)
    uint32_t s0, ns;         // the fold: samples [s0, s0 + ns) of every pixel
    uint32_t gen_next, gen_total;  // paths of the fold handed out / in all
    uint32_t outstanding;    // paths started and not yet finished (atomic)
    uint32_t pad;
};
enum { SP_U_FREE = 0, SP_U_READY, SP_U_GENERATING, SP_U_DRAINING, SP_U_REDUCING };
struct SpShared {        // what the streams of a workgroup share (8 dwords)
    uint32_t lock;           // the unit bookkeeping of the serial sections (the two streams' can run at the same time)
    uint32_t cur;            // the unit slot new paths come from
    uint32_t tiles_done;     // the rank's tile queue is exhausted
    uint32_t abort;          // a bound tripped: every wave leaves the scheduler, the workgroup leaves the kernel
    uint32_t stall;          // consecutive serial sections (either stream's) whose cycle had nothing to run (HRT_SP_CYCLE_BOUND)
    uint32_t sections;       // serial sections so far (HRT_SP_GIVEUP_AFTER test builds)
    uint32_t pad[2];
};
static_assert(sizeof(SpCtl) % 16 == 0 && sizeof(SpShared) % 16 == 0 && sizeof(SpUnit) % 16 == 0, "the LDS regions behind the control blocks must stay 16-byte aligned");
static_assert(HRT_SP_STREAMS == 1 || HRT_SP_STREAMS == 2, "one or two streams");
static_assert(HRT_SP_UNITS >= 1 && HRT_SP_UNITS <= 4, "a path number carries its unit slot in its two top bits");
static_assert(!HRT_SP_PMQ || (HRT_SP_PM4 && HRT_SP_POOL <= 4096), "a T-queue entry is 12 bits of slot id + 4 bits of meshes");
static_assert(!HRT_SP_NOG5 || HRT_SP_UNIT <= 8192, "g7.w keeps 13 bits of the path's number in its unit beside the bounces left");
static_assert(HRT_SP_QCAP >= 512, "deferring partial chunks needs a queue that can hold a whole chunk whenever fewer than 64 slots are free (6 queues x 63 < QCAP - 64)");

static_assert((HRT_SP_POOL & (HRT_SP_POOL - 1)) == 0 && HRT_SP_POOL <= 65536, "slot ids are 16-bit and masked with HRT_SP_POOL - 1");
static_assert((HRT_SP_GLOBAL ? 0 : SP_FIELDS * HRT_SP_POOL * 4) + HRT_SP_NQ * HRT_SP_POOL * 2 + HRT_SP_MAXG * 8 + 512 <= 160 * 1024,
              "pool + queues do not fit the CU's 160 KB of LDS (with HRT_SP_GLOBAL=0 use -DHRT_SP_POOL=1024)");

struct SpLds {
    uint32_t *st;        // SP_FIELDS x POOL dwords
    uint16_t *q;         // HRT_SP_NQ queues x HRT_SP_QCAP entries of the stream in hand (streams follow each other)
    SpCtl *ctl;          // HRT_SP_STREAMS control blocks, then SpShared, the two SpUnit, 2 x HRT_SP_MAXG packed tile origins
};

#if HRT_SP_GLOBAL
#define SP_AT(field, slot) ((slot) * (uint32_t)SP_FIELDS + (uint32_t)(field))
#else
#define SP_AT(field, slot) ((uint32_t)(field) * (uint32_t)HRT_SP_POOL + (slot))
#endif
#ifndef HRT_SP_NT
#define HRT_SP_NT 0        // bit 0: path records are read, bit 1: written with the non-temporal hint (they stream through the CU: a record is
#endif                     //    touched once per visit, 512 KB per workgroup per cycle against a 32 KB L1)
template <class T>
struct SpRef {             // one dword of a path record
    T *p;
    __device__ __forceinline__ operator T() const { return (HRT_SP_NT & 1) ? __builtin_nontemporal_load(p) : *p; }
    __device__ __forceinline__ T operator=(T v) const {
        if (HRT_SP_NT & 2) __builtin_nontemporal_store(v, p);
        else *p = v;
        return v;
    }
};
__device__ __forceinline__ SpRef<float> spf(const SpLds &L, int field, uint32_t slot) { return SpRef<float>{reinterpret_cast<float *>(L.st) + SP_AT(field, slot)}; }
__device__ __forceinline__ SpRef<uint32_t> spu(const SpLds &L, int field, uint32_t slot) { return SpRef<uint32_t>{L.st + SP_AT(field, slot)}; }
// A path record is read and written in its eight aligned 16-byte GROUPS (one vector memory instruction each):
//   g0 {o.xyz, d.x}  g1 {d.y, d.z, hit t, hit kind|index}  g2 {hit a0, a1, triangle, meshes still to walk}
//   g3 {best triangle, t_entry, cursor, best t}  g4 {walk ref, bu, bv, meshes still to walk} (HRT_SP_PM4; else g3.x walk ref, g4.x triangle, g2.w meshes)
//   g5 {time, RNG key k0, k1, path number} (unused with HRT_SP_NOG5)
//   g6 {throughput.rgb, radiance.r}  g7 {radiance.g, .b, RNG position, bounces left | path number (sp_w7)}
// (Left to the load / store vectoriser, the per-field accesses of a hit visit became 11 loads -- three of them issued late,
// behind the first waits -- and 7 stores of mixed widths; by group they are 6 and 5.)
__device__ __forceinline__ uint4 sp_ld4(const SpLds &L, int g, uint32_t slot) {
#if HRT_SP_GLOBAL
    // a NATIVE vector in the global address space; the caller pins the groups of a visit together (SP_PIN) once all are
    // requested: hipcc otherwise takes them apart, drops and sinks components and re-merges the rest across group boundaries
    const v4u v = ((gu4)(L.st + SP_AT(4 * g, slot)))[0];
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return make_uint4(L.st[SP_AT(4 * g, slot)], L.st[SP_AT(4 * g + 1, slot)], L.st[SP_AT(4 * g + 2, slot)], L.st[SP_AT(4 * g + 3, slot)]);
#endif
}
__device__ __forceinline__ void sp_st4(const SpLds &L, int g, uint32_t slot, uint4 v) {
#if HRT_SP_GLOBAL
    v4u w;
    w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
    *((v4u __attribute__((address_space(1))) *)(L.st + SP_AT(4 * g, slot))) = w;
#else
    L.st[SP_AT(4 * g, slot)] = v.x; L.st[SP_AT(4 * g + 1, slot)] = v.y; L.st[SP_AT(4 * g + 2, slot)] = v.z; L.st[SP_AT(4 * g + 3, slot)] = v.w;
#endif
}
#define SP_PIN1(g) "+v"(g.x), "+v"(g.y), "+v"(g.z), "+v"(g.w)
// g7.w: the bounces left (3 bits) and, with HRT_SP_NOG5, the path's number beside them (13 bits of path-in-unit, the unit slot on top)
#if HRT_SP_NOG5
__device__ __forceinline__ uint32_t sp_w7(uint32_t left, uint32_t pnum) { return left | ((pnum & 0x1FFFu) << 3) | (pnum & 0xC0000000u); }
__device__ __forceinline__ uint32_t sp_w7_left(uint32_t w) { return w & 7u; }
__device__ __forceinline__ uint32_t sp_w7_pnum(uint32_t w) { return ((w >> 3) & 0x1FFFu) | (w & 0xC0000000u); }
#else
__device__ __forceinline__ uint32_t sp_w7(uint32_t left, uint32_t) { return left; }
__device__ __forceinline__ uint32_t sp_w7_left(uint32_t w) { return w; }
#endif
__device__ __forceinline__ uint4 sp_pack(float a, float b, float c, float d) { return make_uint4(__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)); }
__device__ __forceinline__ void sp_unpack_ray_hit(const uint4 g0, const uint4 g1, const uint4 g2, Ray &ray, Hit &h, uint32_t &pm) {
    ray.o = mk(__uint_as_float(g0.x), __uint_as_float(g0.y), __uint_as_float(g0.z));
    ray.d = mk(__uint_as_float(g0.w), __uint_as_float(g1.x), __uint_as_float(g1.y));
    h.t = __uint_as_float(g1.z); h.kind = g1.w >> 28; h.index = g1.w & 0x0FFFFFFFu;
    h.a0 = __uint_as_float(g2.x); h.a1 = __uint_as_float(g2.y); h.tri = g2.z; pm = g2.w;
}
// g0, g1, g2 of a ray with its closest hit so far
__device__ __forceinline__ void sp_store_ray_hit(const SpLds &L, uint32_t slot, const Ray &ray, const Hit &h, uint32_t pm) {
    sp_st4(L, 0, slot, sp_pack(ray.o.x, ray.o.y, ray.o.z, ray.d.x));
    sp_st4(L, 1, slot, make_uint4(__float_as_uint(ray.d.y), __float_as_uint(ray.d.z), __float_as_uint(h.t), (h.kind << 28) | h.index));
    sp_st4(L, 2, slot, make_uint4(__float_as_uint(h.a0), __float_as_uint(h.a1), h.tri, pm));
}
__device__ __forceinline__ uint16_t *spq(const SpLds &L, int which, uint32_t parity) { return L.q + (2 * which + parity) * HRT_SP_QCAP; }

// All appends of a chunk at once.  Every lane names the queue its path goes to (`to`: 0 = T, 1..4 = closest hit of kind
// to - 1, 5 = free list, SP_TO_NONE = nowhere); ONE LDS atomic, issued by the first lane of each queue on that queue's counter,
// reserves the entries, one ds_bpermute hands the bases round, one write stores the slot ids.  (One sp_push per queue was a
// chain of up to six atomic-with-return / shuffle pairs per chunk.)
#define SP_TO_NONE 6u
__device__ __forceinline__ void sp_push_all(const SpLds &L, SpCtl &C, uint32_t out, uint32_t to, uint32_t slot) {
    uint32_t rank = 0, count = 0, leader = 0;
#pragma unroll
    for (uint32_t q = 0; q < 6u; ++q) {
        const uint64_t m = __ballot(to == q);
        if (to == q) {
            rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            count = (uint32_t)__popcll(m);
            leader = (uint32_t)__builtin_ctzll(m | (1ull << 63));
        }
    }
    uint32_t base = 0;
    if (to < SP_TO_NONE && rank == 0u) base = atomicAdd(&C.cQ[to][out], count);
    base = __shfl(base, (int)leader);
    if (to < SP_TO_NONE) {
        const uint32_t pos = base + rank;
        // buffers (spq): T -> 0 from the front; kinds 1 (sphere) and 2 (square) share 1, kinds 0 (miss) and 3 (mesh) share 2, the
        // first of each pair from the front, the second from the back; free slots -> 3
        const uint32_t buf = to == 0u ? 0u : (to == 5u ? 3u : ((to == 2u || to == 3u) ? 1u : 2u));
        const bool back = to == 3u || to == 4u;
        spq(L, (int)buf, out)[back ? (uint32_t)HRT_SP_QCAP - 1u - pos : pos] = (uint16_t)slot;
    }
}

template <bool LIGHTS, bool EXACT = false, bool SPHF = false>
__device__ __forceinline__ void stream_body(const DRender &R) {
    extern __shared__ uint4 s_raw[];
    SpLds L;
#if HRT_SP_GLOBAL
    L.st = R.sp_pool + (size_t)blockIdx.x * ((size_t)SP_FIELDS * HRT_SP_POOL);
    L.q = reinterpret_cast<uint16_t *>(s_raw);
#else
    L.st = reinterpret_cast<uint32_t *>(s_raw);
    L.q = reinterpret_cast<uint16_t *>(L.st + SP_FIELDS * HRT_SP_POOL);
#endif
    L.ctl = reinterpret_cast<SpCtl *>(L.q + HRT_SP_NQ * HRT_SP_POOL);
    SpShared &SH = *reinterpret_cast<SpShared *>(L.ctl + HRT_SP_STREAMS);
    SpUnit *const U = reinterpret_cast<SpUnit *>(&SH + 1);                // the two work units in flight
    uint32_t *tile_xy = reinterpret_cast<uint32_t *>(U + HRT_SP_UNITS);              // [unit slot][tile]: x0 | y0 << 16, ~0: no such tile
    float *s_lut = reinterpret_cast<float *>(tile_xy + HRT_SP_UNITS * HRT_SP_MAXG);  // the u8 -> float tables (512 floats)
    float4 *s_tabs = reinterpret_cast<float4 *>(s_lut + 512);            // 16-byte aligned: every size above is a multiple of 16
    CtxT<EXACT, true, SPHF> cx;
    cx.S = (cscene)R.scene;
    const uint32_t tab_rows = cx.S->tab_rows;  // the scene's per-object tables: staged once, read per lane from LDS (CtxT)
    uint4 *s_units = reinterpret_cast<uint4 *>(s_tabs + tab_rows);
    cx.set_tables((lf4)s_tabs, (lf1)s_lut, cx.S);
    cx.lds = (lu4)s_units;
    cx.lds_n = R.lds_units;
    cx.err_abs = R.err_abs;
    cx.flags = R.flags;
    unsigned long long stamps_local[17];
    cx.st = stamps_local;
#ifdef HRT_WALK_SEG  // 16 u64 accumulators per wave behind the nodelets (hrt_api.hip reserves the 2 KB)
    {
        unsigned long long *wseg = reinterpret_cast<unsigned long long *>(s_units + R.lds_units) + (threadIdx.x >> 6) * 16u;
        if ((threadIdx.x & 63u) < 16u) wseg[threadIdx.x & 63u] = 0ull;
        cx.st = wseg;
    }
#endif
    ccam cam = (ccam)R.cam;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    {
        gu4 g_units = (gu4)cx.S->kd_units;
        for (uint32_t i = tid; i < cx.lds_n; i += HRT_SP_WG) s_units[i] = ld(g_units, i);
        gf4 g_tabs = (gf4)cx.S->tabs;
        for (uint32_t i = tid; i < tab_rows; i += HRT_SP_WG) s_tabs[i] = ld(g_tabs, i);
        if (tid < 512u) s_lut[tid] = c_u8_lut[tid];
    }
    uint16_t *const q_all = L.q;
    if (tid < HRT_SP_STREAMS) {
        SpCtl &C0 = L.ctl[tid];
        C0.cQ[0][0] = C0.cQ[0][1] = 0;
        for (int k = 0; k < 4; ++k) C0.cQ[1 + k][0] = C0.cQ[1 + k][1] = 0;
        C0.cQ[5][0] = HRT_SP_QCAP; C0.cQ[5][1] = 0;
        C0.parity = 1; C0.done = 0; C0.arrive = 0; C0.ready = 0; C0.cursor = 0; C0.ngen = 0; C0.gen_n0 = 0; C0.gen_slot = 0; C0.gen_s0 = 0;
        C0.red_slot = HRT_SP_UNITS; C0.red_j = 0; C0.red_s0 = 0; C0.red_ns = 0; C0.more = 1;
        if (tid == 0) {
            SH.lock = 0; SH.cur = 0; SH.tiles_done = 0; SH.abort = 0; SH.stall = 0; SH.sections = 0;
            for (int k = 0; k < HRT_SP_UNITS; ++k) { U[k].state = SP_U_FREE; U[k].j = 0; U[k].s0 = 0; U[k].ns = 0; U[k].gen_next = 0; U[k].gen_total = 0; U[k].outstanding = 0; }
        }
    }
    for (uint32_t i = tid; i < HRT_SP_POOL; i += HRT_SP_WG)  // every slot free, in its stream's free queue (parity 0)
        q_all[(i / HRT_SP_QCAP) * (HRT_SP_NQ * HRT_SP_QCAP) + (2 * 3 + 0) * HRT_SP_QCAP + (i % HRT_SP_QCAP)] = (uint16_t)i;
    const bool has_mesh = cx.S->n_meshes != 0u;
    const bool multi_mesh = cx.S->n_meshes > 1u;  // T chunks then mix lanes that wait for different meshes
#if HRT_SP_PMQ
    const bool pm_in_entry = cx.S->n_meshes <= 4u;  // the meshes to walk fit beside the slot id in a T-queue entry (12 + 4 bits)
#endif
    const bool prune = !EXACT && HRT_SP_PRUNE != 0 && cx.S->prune_ok != 0u;                // see HRT_SP_PRUNE
    const bool moving = cx.S->any_motion != 0u;                                              // a ray's time matters
    const bool sky_is_zero = cx.S->skybox_image < 0 && cx.S->dark_sky != 0;                // Scene.h:149-152: a miss adds nothing
    float *scratch = R.sp_scratch + (size_t)blockIdx.x * ((size_t)HRT_SP_UNITS * HRT_SP_UNIT * 3u);  // one part per unit slot
    const uint32_t glog = R.sp_group_log2, G = 1u << glog, blog = R.sp_band_log2;  // tiles per unit; or ONE row band of a tile, 8 x (8 >> blog) pixels
    const uint32_t upix = (64u << glog) >> blog, upix_log2 = 6u + glog - blog;     // pixels per unit
    const uint32_t items = R.tiles_owned << blog;                                  // what the tile queue hands out: tiles, or bands of tiles
#define SP_UNI(x) __builtin_amdgcn_readfirstlane(x)

#ifndef HRT_SP_SEG_KIND
#define HRT_SP_SEG_KIND 1  // which chunk class the diagnostic build stamps: 1 square hits, 2 T (KD walk), 3 sphere hits, 4 G (new paths), 5 mesh hits
#endif
#ifdef HRT_SP_SEG  // diagnostic build: where a square-hit chunk spends its clocks; every stamp first drains the wave's memory
                   // counters, so segments are serialised and the whole run is slower than the shipped kernel
    unsigned long long seg[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, seg_last = 0;
    bool seg_on = false;
#define SEG_START(on) do { seg_on = (on); if (seg_on) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); seg_last = __builtin_readcyclecounter(); } } while (0)
#define SEG(k) do { if (seg_on) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_readcyclecounter(); \
                                  seg[k] += t_ - seg_last; seg_last = t_; } } while (0)
#else
#define SEG_START(on) do { } while (0)
#define SEG(k) do { } while (0)
#endif
#ifdef HRT_SP_DEBUG
    unsigned long long dbg_work = 0, dbg_chunks = 0, dbg_cycles = 0, dbg_serial = 0, dbg_wait = 0, dbg_drain = 0, dbg_tail = 0, dbg_tail_cycles = 0;
    unsigned long long dbg_class[6] = {0, 0, 0, 0, 0, 0};  // clocks in T, mesh-hit, sphere-hit, square-hit, miss, G chunks
    const unsigned long long dbg_t0 = __builtin_readcyclecounter();
#endif
    // ---- the scheduler: HRT_SP_STREAMS streams of cycles over a PIPELINE of work units, no barrier anywhere inside.
    // A stream's cycle k: its control block (queue fills, new paths, a reduction to run) is valid once ready == k; the waves pull
    // its chunks from one cursor; a wave that finds none left waits for its own stores, ARRIVES (arrive += 1) and moves to the
    // other stream.  The wave whose arrival completes the cycle (arrive == waves x k) runs the serial section -- unit bookkeeping,
    // new paths, queue buffer swap, unused free slots -- and publishes ready = k + 1.  Waves visit the streams in the fixed order
    // A1 B1 A2 B2 ...; a stage only ever waits for all waves to have passed the previous stage of the same stream, so there is no
    // cyclic wait.  Every wait is bounded (abort -> HRT_ERR_DEVICE on the host).
    //
    // WORK UNITS.  A unit is G tiles x 64 pixels x the samples of one fold (<= HRT_SP_UNIT paths, its own part of the sample
    // scratch).  HRT_SP_UNITS are in flight: while the paths of one DRAIN (every path started, the long ones still bouncing), new paths
    // come from the next, so the pool stays full.  (With one unit at a time a quarter of the cycles ran a draining, half-empty
    // pool.)  When a unit's last path has finished, the next cycle of whichever stream notices carries its REDUCTION as ordinary
    // chunks: 64 (pixel, channel) columns each, the fold's samples added in sample order onto the running sums (main.cpp:193),
    // which live in out_tiles between folds; the last fold leaves the mean (main.cpp:195).  A path knows its unit slot (the two top
    // bits of its number).
    const uint32_t per_fold = min((uint32_t)HRT_SP_SCHUNK, (uint32_t)HRT_SP_UNIT / upix);  // samples per pixel of one fold
    auto serial_section = [&](SpCtl &C, uint16_t *qs) {  // one whole wave; control values by lane 0
        const uint32_t par = SP_UNI(C.parity) ^ 1u;  // the buffers the finished cycle appended to become the input
        const uint32_t free_in = SP_UNI(C.cQ[5][par]);
#if HRT_SP_DEFER
        const uint32_t want = free_in & ~63u;  // whole chunks of new paths only (the last paths of a fold come as they are)
#else
        const uint32_t want = free_in;
#endif
        if (lane == 0) {  // unit bookkeeping under the workgroup's lock: the two streams' serial sections can run at the same time
            uint32_t spins = 0;
            while (atomicCAS(&SH.lock, 0u, 1u) != 0u) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 22)) {  // the holder runs a few dozen LDS operations and one global atomic
                    if (R.stamps) { __hip_atomic_store(R.stamps + 15, 0xDEADull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); R.stamps[14] = 1; }
                    __hip_atomic_store(&SH.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    break;
                }
            }
            // 1. the reduction this stream ran in the cycle that has just ended is complete: the unit's next fold, or a free slot
            const uint32_t rs = C.red_slot;
            if (rs < HRT_SP_UNITS) {
                SpUnit &u = U[rs];
                const uint32_t s_next = u.s0 + u.ns;
                if (s_next < R.spp) {
                    u.s0 = s_next; u.ns = min(per_fold, R.spp - s_next); u.gen_next = 0; u.gen_total = upix * u.ns; u.state = SP_U_READY;
                } else {
                    u.state = SP_U_FREE;
                }
            }
            // 2. a unit whose every path has started and finished is reduced in the cycle now being prepared
            uint32_t red = HRT_SP_UNITS;
            for (uint32_t k = 0; k < HRT_SP_UNITS; ++k)
                if (red == HRT_SP_UNITS && U[k].state == SP_U_DRAINING && __hip_atomic_load(&U[k].outstanding, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) {
                    red = k;
                    U[k].state = SP_U_REDUCING;
                }
            C.red_slot = red;
            if (red < HRT_SP_UNITS) { C.red_j = U[red].j; C.red_s0 = U[red].s0; C.red_ns = U[red].ns; }
            // 3. new paths: from the unit that is generating, else the next fold that is ready, else a new unit off the tile queue
            uint32_t ngen = 0, n0 = 0, gs = 0;
            if (want != 0u) {
                uint32_t cur = SH.cur;
                if (U[cur].state != SP_U_GENERATING) {
                    cur = HRT_SP_UNITS;
                    for (uint32_t k = 0; k < HRT_SP_UNITS; ++k)
                        if (cur == HRT_SP_UNITS && U[k].state == SP_U_READY) cur = k;
                    if (cur == HRT_SP_UNITS && SH.tiles_done == 0u)
                        for (uint32_t k = 0; k < HRT_SP_UNITS; ++k)
                            if (cur == HRT_SP_UNITS && U[k].state == SP_U_FREE) {
                                const uint32_t j = atomicAdd(R.tile_counter, G);  // item: tile slot (G of them), or tile slot << blog | band
                                if (j >= items) { SH.tiles_done = 1u; break; }  // finite queue
                                SpUnit &u = U[k];
                                u.j = j; u.s0 = 0; u.ns = min(per_fold, R.spp); u.gen_next = 0; u.gen_total = upix * u.ns;
                                __hip_atomic_store(&u.outstanding, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                for (uint32_t t = 0; t < G; ++t) {
                                    uint32_t xy = 0xFFFFFFFFu;
                                    if (j + t < items) {
                                        const uint32_t tile = R.rank + ((j + t) >> blog) * R.world, band = (j + t) & ((1u << blog) - 1u);
                                        xy = ((tile % R.tiles_x) * 8u) | (((tile / R.tiles_x) * 8u + band * (8u >> blog)) << 16);  // the band's first row
                                    }
                                    tile_xy[k * HRT_SP_MAXG + t] = xy;
                                }
                                cur = k;
                            }
                    if (cur < HRT_SP_UNITS) { U[cur].state = SP_U_GENERATING; SH.cur = cur; }
                }
                if (cur < HRT_SP_UNITS) {
                    SpUnit &u = U[cur];
                    n0 = u.gen_next;
                    ngen = min(want, u.gen_total - n0);
                    u.gen_next = n0 + ngen;
                    atomicAdd(&u.outstanding, ngen);
                    if (u.gen_next == u.gen_total) u.state = SP_U_DRAINING;
                    gs = cur;
                }
            }
            bool generating = false, idle = SH.tiles_done != 0u;  // generating: paths can be started without anything having to finish first
            for (uint32_t k = 0; k < HRT_SP_UNITS; ++k) {
                generating = generating || U[k].state == SP_U_READY || U[k].state == SP_U_GENERATING || (U[k].state == SP_U_FREE && SH.tiles_done == 0u);
                idle = idle && U[k].state == SP_U_FREE;
            }
            C.ngen = ngen; C.gen_n0 = n0; C.gen_slot = gs; C.gen_s0 = U[gs].s0;
            C.more = generating ? 1u : 0u;
            C.done = idle ? 1u : 0u;  // (and nothing waiting in this stream: checked below)
            // Bounded: a scheduling bug must not spin the GPU.  The cycle being prepared runs something when it has queue entries (the
            // queue fills are final: every wave of the finished cycle has arrived), new paths or a reduction; only a run of cycles
            // with NOTHING -- this stream waiting for the other one, or a unit state that can never advance -- counts.
            uint32_t entries = ngen + (red < HRT_SP_UNITS ? 1u : 0u);  // what WILL run: whole chunks only while partial ones are deferred (below)
            for (int q = 0; q < 5; ++q) entries += (HRT_SP_DEFER && generating) ? (C.cQ[q][par] & ~63u) : C.cQ[q][par];
            const uint32_t stalled = entries != 0u ? (SH.stall = 0u) : ++SH.stall;
            const uint32_t sections = ++SH.sections;
            if (stalled > HRT_SP_CYCLE_BOUND || (HRT_SP_GIVEUP_AFTER != 0u && sections > HRT_SP_GIVEUP_AFTER)) {
                // The frame is lost: flag it for the host (hrt_check_last_launch / hrt_render return HRT_ERR_DEVICE) and take
                // the whole workgroup out of the kernel.  The other workgroups finish their tiles.
                if (R.stamps) {
                    __hip_atomic_store(R.stamps + 15, 0xDEADull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    R.stamps[14] = 2;  // diagnostic: which bound, and the unit slots at that moment
                    for (uint32_t k = 0; k < HRT_SP_UNITS; ++k) R.stamps[8 + k] = ((unsigned long long)U[k].state << 48) | ((unsigned long long)U[k].outstanding << 24) | U[k].gen_next;
                    R.stamps[12] = ((unsigned long long)SH.tiles_done << 32) | SH.cur; R.stamps[13] = ((unsigned long long)want << 32) | C.cQ[0][par];
                }
                __hip_atomic_store(&SH.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __hip_atomic_store(&SH.lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint32_t ngen = SP_UNI(C.ngen);
        uint16_t *qFi = qs + (2 * 3 + par) * HRT_SP_QCAP, *qFo = qs + (2 * 3 + (par ^ 1u)) * HRT_SP_QCAP;
        for (uint32_t i = lane; i < free_in - ngen; i += 64u) qFo[i] = qFi[ngen + i];  // carry unused free slots
        // While new paths can be started (a unit is generating, or a slot is free for the next tiles), a queue's last PARTIAL chunk
        // waits for the next cycle (it moves to the front of the output queue, so the oldest entries go first): the paths are
        // independent and every fold is ordered, so when a path is advanced changes nothing -- and the chunks that do run have
        // all 64 lanes filled.  When every unit slot is draining, new paths wait for old ones to finish: then everything runs.
        const bool defer = HRT_SP_DEFER && SP_UNI(C.more) != 0u;
        const uint32_t cT = SP_UNI(C.cQ[0][par]), rT = defer ? (cT & 63u) : 0u;
        if (lane < rT) qs[(2 * 0 + (par ^ 1u)) * HRT_SP_QCAP + lane] = qs[(2 * 0 + par) * HRT_SP_QCAP + cT - rT + lane];
        uint32_t cKv[4], rK[4], waiting = cT;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            cKv[k] = SP_UNI(C.cQ[1 + k][par]);
            rK[k] = defer ? (cKv[k] & 63u) : 0u;
            waiting += cKv[k];
            const int which = (k == 1 || k == 2) ? 1 : 2;  // the addressing of sp_push_all: kinds 1 and 0 from the front, 2 and 3 from the back
            const bool back = !(k == 1 || k == 0);
            const uint16_t *qi = qs + (2 * which + par) * HRT_SP_QCAP;
            uint16_t *qo = qs + (2 * which + (par ^ 1u)) * HRT_SP_QCAP;
            if (lane < rK[k]) {
                const uint32_t e = cKv[k] - rK[k] + lane;
                qo[back ? (uint32_t)HRT_SP_QCAP - 1u - lane : lane] = qi[back ? (uint32_t)HRT_SP_QCAP - 1u - e : e];
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { C.cQ[1 + k][par] = cKv[k] - rK[k]; C.cQ[1 + k][par ^ 1u] = rK[k]; }
            C.cQ[0][par] = cT - rT;
            C.cQ[0][par ^ 1u] = rT;
            C.cQ[5][par ^ 1u] = free_in - ngen;  // the unused free slots carry over, freed slots are appended after them
            C.cursor = 0; C.parity = par;
            // the stream ends when the launch has: no tile left, both unit slots free -- then nothing is in flight anywhere
            C.done = (C.done != 0u && ngen == 0u && waiting == 0u && C.red_slot == HRT_SP_UNITS) ? 1u : 0u;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // control block and free list are in LDS before the cycle is announced
        if (lane == 0) __hip_atomic_store(&C.ready, C.ready + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    __syncthreads();  // tables, queues and control blocks are set up
    if (tid < 64u)    // wave 0 opens the launch: cycle 1 of every stream (the first one takes the first unit off the tile queue)
        for (uint32_t st = 0; st < HRT_SP_STREAMS; ++st) serial_section(L.ctl[st], q_all + st * (HRT_SP_NQ * HRT_SP_QCAP));
    __syncthreads();
    uint32_t fin[HRT_SP_UNITS];  // paths of each unit slot this wave has finished since it last arrived
    for (int k = 0; k < HRT_SP_UNITS; ++k) fin[k] = 0u;

    uint32_t kcyc[HRT_SP_STREAMS], fin_mask = 0u;
    for (uint32_t st = 0; st < HRT_SP_STREAMS; ++st) kcyc[st] = 1u;
    for (uint32_t round = 0;; ++round) {  // stages A1 B1 A2 B2 ... of this wave
        const uint32_t st = HRT_SP_STREAMS == 1 ? 0u : (round & 1u);
        if (fin_mask == (1u << HRT_SP_STREAMS) - 1u) break;
        if (fin_mask & (1u << st)) continue;
        SpCtl &C = L.ctl[st];
        L.q = q_all + st * (HRT_SP_NQ * HRT_SP_QCAP);
        {   // wait for the control block of this stage (normally there already)
#ifdef HRT_SP_DEBUG
            const unsigned long long dbg_q0 = __builtin_readcyclecounter();
#endif
            uint32_t spins = 0, aborted = 0;
            const unsigned long long wait_t0 = __builtin_amdgcn_s_memrealtime();
            while (SP_UNI(__hip_atomic_load(&C.ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < kcyc[st]) {
                __builtin_amdgcn_s_sleep(8);
                aborted = SP_UNI(__hip_atomic_load(&SH.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (aborted) break;
                // A watchdog, by the 100 MHz real-time counter: the other waves of the cycle may legitimately work for SECONDS on a
                // single chunk (a lit scene whose meshes carry millions of reference-box entries: tools/fuzz_exact.py seed 1056 took
                // 132 s per frame and tripped the former bound of 2^24 spins, ~4 s) -- so this only ends a wait that nothing could explain
                if ((++spins & 1023u) == 0u && (spins >> 10) > 16u && __builtin_amdgcn_s_memrealtime() - wait_t0 > 100000000ull * HRT_SP_WAIT_SECONDS) {
                    if (lane == 0) {
                        if (R.stamps) { __hip_atomic_store(R.stamps + 15, 0xDEADull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); R.stamps[14] = 3; }
                        __hip_atomic_store(&SH.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    aborted = 1u;
                    break;
                }
            }
#ifdef HRT_SP_DEBUG
            dbg_wait += __builtin_readcyclecounter() - dbg_q0;
#endif
            if (aborted || SP_UNI(__hip_atomic_load(&SH.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) break;
        }
        if (SP_UNI(C.done)) { fin_mask |= 1u << st; continue; }
        const uint32_t parity = SP_UNI(C.parity);
        const uint32_t ngen = SP_UNI(C.ngen), n0 = SP_UNI(C.gen_n0), gs = SP_UNI(C.gen_slot), s0 = SP_UNI(C.gen_s0);
        const uint32_t red = SP_UNI(C.red_slot), nR = red < HRT_SP_UNITS ? (upix * 3u + 63u) >> 6 : 0u;  // reduction chunks: 64 columns each
        const uint32_t cTin = SP_UNI(C.cQ[0][parity]);
        const uint32_t cK0 = SP_UNI(C.cQ[1][parity]), cK1 = SP_UNI(C.cQ[2][parity]), cK2 = SP_UNI(C.cQ[3][parity]),
                       cK3 = SP_UNI(C.cQ[4][parity]);
        // chunk ranges of this cycle: T (longest) first, then mesh, sphere, square hits, misses, new paths last
        const uint32_t nT = HRT_SP_THALF ? (cTin + 31u) >> 5 : (cTin + 63u) >> 6, e3 = nT + ((cK3 + 63u) >> 6), e1 = e3 + ((cK1 + 63u) >> 6),
                       e2 = e1 + ((cK2 + 63u) >> 6), e0 = e2 + ((cK0 + 63u) >> 6), nG = (ngen + 63u) >> 6, total = e0 + nG;
        uint16_t *qTi = spq(L, 0, parity);
        uint16_t *qAi = spq(L, 1, parity), *qBi = spq(L, 2, parity);
        uint16_t *qFi = spq(L, 3, parity);
#ifdef HRT_SP_DEBUG
        const unsigned long long dbg_w0 = __builtin_readcyclecounter();
        ++dbg_cycles;
#endif

        for (;;) {  // chunks of this cycle: T first (longest), then S, then G
            uint32_t c = 0;
            if (lane == 0) c = atomicAdd(&C.cursor, 1u);
            c = SP_UNI(c);
            if (c >= nR + total) break;
            if (c < nR) {
                // ---------------- R: 64 (pixel, channel) columns of a finished fold, its samples added in sample order (main.cpp:193)
                // Plain loads: the scratch was written by waves of THIS workgroup (one CU), each of which waited for its stores
                // (s_waitcnt vmcnt(0)) before the arrival that took the unit's count of paths in flight to zero; a CU's vector
                // L1 is coherent with that CU's own stores (it is only other CUs' stores it never sees), which is also what the
                // path pool's plain loads and stores rely on.
                const uint32_t i = c * 64u + lane;  // pixel-of-unit * 3 + channel
                const uint32_t rj = SP_UNI(C.red_j), rs0 = SP_UNI(C.red_s0), rns = SP_UNI(C.red_ns);
                if (i < upix * 3u && rj + i / (192u >> blog) < items) {  // (rj counts tiles -- G of them, 192 columns each -- or bands: one per unit)
                    float *out = R.out_tiles + (size_t)rj * (192u >> blog) + i;
                    // the running sum: samples [0, rs0) of this launch, on top of the earlier launches' in progressive mode
                    float acc = (rs0 != 0u || R.accumulate) ? *out : 0.f;
                    const float *col = scratch + (size_t)red * ((size_t)HRT_SP_UNIT * 3u) + i;
                    const uint32_t stride = upix * 3u;
                    uint32_t sm = 0;
                    for (; sm + 8u <= rns; sm += 8u) {  // 8 loads in flight, added in sample order
                        float v[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = col[(size_t)(sm + k) * stride];
#pragma unroll
                        for (int k = 0; k < 8; ++k) acc += v[k];
                    }
                    for (; sm < rns; ++sm) acc += col[(size_t)sm * stride];
                    if (rs0 + rns >= R.spp) {  // the last fold: the pixel's value
                        const uint32_t q = i / 3u, p = q & 63u, txy = tile_xy[red * HRT_SP_MAXG + (q >> 6)];
                        const uint32_t px = (txy & 0xFFFFu) + (p & 7u), py = (txy >> 16) + (p >> 3);
                        if (!(px < R.w && py < R.h)) acc = 0.f;
                        else if (!R.accumulate) acc = acc / (float)R.spp;  // main.cpp:195; progressive mode keeps the sum
                    }
                    *out = acc;
                }
                continue;
            }
            c -= nR;
#ifdef HRT_SP_DEBUG
            ++dbg_chunks;
            const unsigned long long dbg_c0 = __builtin_readcyclecounter();
            const uint32_t dbg_k = c < nT ? 0u : (c < e3 ? 1u : (c < e1 ? 2u : (c < e2 ? 3u : (c < e0 ? 4u : 5u))));
#endif
            if (c < nT) {
                // ---------------- T: mesh walk
#if HRT_SP_TPRIO
                __builtin_amdgcn_s_setprio(HRT_SP_TPRIO);  // a walk is a chain of dependent loads with a few instructions between them: let them issue first
#endif
                const uint32_t e = HRT_SP_THALF ? c * 32u + lane : c * 64u + lane;
                const bool act = e < cTin && (!HRT_SP_THALF || lane < 32u);
                uint32_t slot = 0, kind = 0, pm = 0, ref_in = HRT_KD_NIL, pm_before = 0;
                bool walked = false;
                Ray ray;
                ray.o = mk(0.f, 0.f, 0.f); ray.d = mk(0.f, 0.f, 1.f); ray.time = 0.f;
                Hit h;
                h.kind = 0; h.index = 0; h.t = HRT_FLT_MAX; h.tri = 0; h.a0 = 0.f; h.a1 = 0.f;
                Walk w;
                w.ref = HRT_KD_NIL; w.t_entry = 0.f; w.kk = 0xFFFFu; w.best_t = HRT_FLT_MAX; w.best_tri = 0; w.bu = 0.f; w.bv = 0.f;
                SEG_START(HRT_SP_SEG_KIND == 2);
                if (act) {
#if HRT_SP_PMQ
                    const uint32_t entry = qTi[e];
                    slot = entry & (HRT_SP_POOL - 1u);
#else
                    slot = qTi[e] & (HRT_SP_POOL - 1u);
#endif
#if HRT_SP_PM4
                    // Four groups: a T visit needs the ray, the distance and kind of the best hit so far, the state of its walk and the
                    // meshes still to walk -- which the chunk that sent the path here wrote into the free dword of g4 -- but not the
                    // hit's a0 / a1 / triangle (g2): those it only ever REPLACES, when a mesh gives a closer hit.
                    uint4 g0 = sp_ld4(L, 0, slot), g1 = sp_ld4(L, 1, slot), g3 = sp_ld4(L, 3, slot), g4 = sp_ld4(L, 4, slot);
                    asm volatile("" : SP_PIN1(g0), SP_PIN1(g1), SP_PIN1(g3), SP_PIN1(g4));
                    {
                        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
                        sp_unpack_ray_hit(g0, g1, z, ray, h, pm);
                    }
                    pm = g4.w;
                    pm_before = __float_as_uint(h.t);  // (here: the best distance on entry)
#else
                    uint4 g0 = sp_ld4(L, 0, slot), g1 = sp_ld4(L, 1, slot), g2 = sp_ld4(L, 2, slot), g3 = sp_ld4(L, 3, slot),
                          g4 = sp_ld4(L, 4, slot);
#if HRT_SP_GLOBAL
                    asm volatile("" : SP_PIN1(g0), SP_PIN1(g1), SP_PIN1(g2), SP_PIN1(g3), SP_PIN1(g4));
#endif
                    sp_unpack_ray_hit(g0, g1, g2, ray, h, pm);  // (the mesh walk does not read ray.time)
                    pm_before = pm;
#endif
#if HRT_SP_PM4   // g3 {best triangle, t_entry, cursor, best t}  g4 {walk ref, bu, bv, meshes to walk}: the sender's ONE store of g4 also says "no walk in progress"
                    w.ref = g4.x; w.t_entry = __uint_as_float(g3.y); w.kk = g3.z; w.best_t = __uint_as_float(g3.w);
                    w.best_tri = g3.x; w.bu = __uint_as_float(g4.y); w.bv = __uint_as_float(g4.z);
#else
                    w.ref = g3.x; w.t_entry = __uint_as_float(g3.y); w.kk = g3.z; w.best_t = __uint_as_float(g3.w);
                    w.best_tri = g4.x; w.bu = __uint_as_float(g4.y); w.bv = __uint_as_float(g4.z);
#endif
#if HRT_SP_PMQ
                    // A path that comes from a hit / new-path chunk carries its meshes in the queue entry (the sender then wrote nothing into g4):
                    // no walk in progress.  One that comes back from a T visit has 0 there and its state in g3 / g4.
                    if (pm_in_entry && (entry >> 12) != 0u) { w.ref = HRT_KD_NIL; pm = entry >> 12; }
#endif
                    ref_in = w.ref;
                }
#ifdef HRT_SP_SEG
                asm volatile("" : "+v"(ray.o.x), "+v"(w.t_entry), "+v"(h.t));
#endif
                SEG(0);  // T: record loaded
                if (act) {
#if HRT_SP_VOTE
                    walked = walk_vote(cx, ray, pm, w, h, HRT_SP_STEPS);
#else
                    walked = multi_mesh ? walk_some_per_lane(cx, ray, pm, w, h, HRT_SP_TRIPS) : walk_some(cx, ray, pm, w, h, HRT_SP_TRIPS);
#endif
                }
                SEG(1);  // T: walk
                if (act) {
                    const uint32_t pm_in = pm_before;
#if HRT_SP_PM4
                    if (__float_as_uint(h.t) != pm_in) {  // a mesh gave a closer hit (g1 also carries d.y, d.z: rewritten as read)
#else
                    if (pm != pm_in) {  // a mesh was finished: the best hit may have changed (g1 also carries d.y, d.z: rewritten as read)
#endif
                        sp_st4(L, 1, slot, make_uint4(__float_as_uint(ray.d.y), __float_as_uint(ray.d.z), __float_as_uint(h.t), (h.kind << 28) | h.index));
                        sp_st4(L, 2, slot, make_uint4(__float_as_uint(h.a0), __float_as_uint(h.a1), h.tri, pm));
                    }
#if HRT_SP_PM4
                    if (!walked) {  // the state of the walk in progress, and the meshes still to walk
                        sp_st4(L, 3, slot, make_uint4(w.best_tri, __float_as_uint(w.t_entry), w.kk, __float_as_uint(w.best_t)));
                        sp_st4(L, 4, slot, make_uint4(w.ref, __float_as_uint(w.bu), __float_as_uint(w.bv), pm));
                    }
                    (void)ref_in;
#else
                    if (!walked) {  // the state of the walk in progress
                        sp_st4(L, 3, slot, make_uint4(w.ref, __float_as_uint(w.t_entry), w.kk, __float_as_uint(w.best_t)));
                        sp_st4(L, 4, slot, make_uint4(w.best_tri, __float_as_uint(w.bu), __float_as_uint(w.bv), 0u));
                    } else if (ref_in != HRT_KD_NIL) {
                        spu(L, SP_WREF, slot) = HRT_KD_NIL;  // invariant: SP_WREF is NIL whenever the path is not in a T queue
                    }
#endif
                    kind = h.kind;
                }
#if HRT_SP_TPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
                // unfinished: joins the next cycle's T chunks; finished: the closest-hit queue of its kind
                sp_push_all(L, C, parity ^ 1u, !act ? SP_TO_NONE : (walked ? 1u + kind : 0u), slot);  // (back to T: a bare slot id, the state is in g3 / g4)
                SEG(2);  // T: stores + appends
#ifdef HRT_SP_SEG
                if (seg_on) { seg[7] += 1; seg[6] += (unsigned long long)__popcll(__ballot(act)); seg[5] += (unsigned long long)__popcll(__ballot(act && walked)); }
#endif
            } else {
                // ---------------- S (shade + scatter, then next prims) and G (camera ray, then prims)
                const bool is_gen = c >= e0;
                // which closest-hit queue this chunk drains: its entries, its fill, and where entry e sits
                const uint32_t first = is_gen ? e0 : (c < e3 ? nT : (c < e1 ? e3 : (c < e2 ? e1 : e2)));
                const uint32_t fill = is_gen ? ngen : (c < e3 ? cK3 : (c < e1 ? cK1 : (c < e2 ? cK2 : cK0)));
                const bool from_back = !is_gen && (c < e3 || (c >= e1 && c < e2));  // mesh and square hits grow from the end
                const uint16_t *qHi = (c < e3 || c >= e2) ? qBi : qAi;
                const uint32_t e = (c - first) * 64u + lane;
                const bool act = e < fill;
                uint32_t slot = 0, kind = 0, fin_unit = 0, pnum = 0;
                bool trace = false, freed = false;  // trace: the path has a new ray to intersect; freed: its path has ended (unit fin_unit)
                bool ended = false, last_seg = false;  // ended: the sample's colour `rad` is final; last_seg: the new ray is the path's last segment
                f3 rad = mk(0.f, 0.f, 0.f);
                SEG_START((HRT_SP_SEG_KIND == 1 && !is_gen && c >= e1 && c < e2) || (HRT_SP_SEG_KIND == 3 && !is_gen && c >= e3 && c < e1) ||
                          (HRT_SP_SEG_KIND == 4 && is_gen) || (HRT_SP_SEG_KIND == 5 && !is_gen && c < e3));  // the chunk class in hand
                Ray ray;
                ray.o = mk(0.f, 0.f, 0.f); ray.d = mk(0.f, 0.f, 1.f); ray.time = 0.f;
                if (act && is_gen) {
                    slot = qFi[e] & (HRT_SP_POOL - 1u);
                    const uint32_t n = n0 + e;  // path n of the unit = (sample n / upix, pixel n % upix)
                    const uint32_t q = n & (upix - 1u), s = s0 + (n >> upix_log2);
                    const uint32_t p = q & 63u, txy = tile_xy[gs * HRT_SP_MAXG + (q >> 6)];
                    const uint32_t px = (txy & 0xFFFFu) + (p & 7u), py = (txy >> 16) + (p >> 3);
                    if (txy != 0xFFFFFFFFu && px < R.w && py < R.h) {
                        Rng rng;
                        rng.start(R.seed_lo, R.seed_hi, py * R.w + px, R.s0 + s);
                        const float u = ((float)px + rng.next()) / (float)R.w;
                        const float v = ((float)py + rng.next()) / (float)R.h;
                        const float tm = rng.next();
                        ray = camera_ray<EXACT>(cam, u, v, tm);
                        pnum = n | (gs << 30);
#if !HRT_SP_NOG5
                        sp_st4(L, 5, slot, make_uint4(__float_as_uint(tm), rng.k0, rng.k1, pnum));
#endif
                        sp_st4(L, 6, slot, sp_pack(1.f, 1.f, 1.f, 0.f));                                  // throughput 1, radiance 0
                        sp_st4(L, 7, slot, make_uint4(0u, 0u, rng.i, sp_w7(6u, pnum)));                  // MAXBOUNCES
                        trace = true;
                    } else {  // pixel outside a ragged image: the sample is zero, the slot stays free
                        float *o = scratch + (size_t)gs * ((size_t)HRT_SP_UNIT * 3u) + (size_t)n * 3u;
                        o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
                        freed = true;
                        fin_unit = gs;
                    }
                } else if (act) {
                    slot = qHi[from_back ? (uint32_t)HRT_SP_QCAP - 1u - e : e] & (HRT_SP_POOL - 1u);
#if HRT_SP_NOG5
                    // Five groups, not six: what g5 held -- the ray's time, the keys of the path's random stream, the path's number -- never
                    // changes after the path's first visit.  The number rides in g7.w beside the bounces left; the rest is recomputed
                    // from it (the pixel from the unit's tile table in LDS, the sample from the unit's fold; Rng::start; the time is draw 2
                    // of the stream, as in the G chunk): ~45 vector instructions for one vector memory instruction less per hit visit.
                    uint4 g0 = sp_ld4(L, 0, slot), g1 = sp_ld4(L, 1, slot), g2 = sp_ld4(L, 2, slot), g6 = sp_ld4(L, 6, slot), g7 = sp_ld4(L, 7, slot);
                    asm volatile("" : SP_PIN1(g0), SP_PIN1(g1), SP_PIN1(g2), SP_PIN1(g6), SP_PIN1(g7));
                    pnum = sp_w7_pnum(g7.w);
                    Rng key;
                    {
                        const uint32_t kn = pnum & 0x3FFFFFFFu, ku = pnum >> 30;
                        const uint32_t kq = kn & (upix - 1u), ks = U[ku].s0 + (kn >> upix_log2);
                        const uint32_t kxy = tile_xy[ku * HRT_SP_MAXG + (kq >> 6)];
                        const uint32_t kx = (kxy & 0xFFFFu) + (kq & 7u), ky = (kxy >> 16) + ((kq & 63u) >> 3);
                        key.start(R.seed_lo, R.seed_hi, ky * R.w + kx, R.s0 + ks);
                        key.i = 2u;
                    }
                    Hit h;
                    uint32_t pm_unused;
                    sp_unpack_ray_hit(g0, g1, g2, ray, h, pm_unused);
                    ray.time = moving ? key.next() : 0.f;  // (time x 0 == 0: without motion the time is never looked at)
#else
                    uint4 g0 = sp_ld4(L, 0, slot), g1 = sp_ld4(L, 1, slot), g2 = sp_ld4(L, 2, slot), g5 = sp_ld4(L, 5, slot),
                          g6 = sp_ld4(L, 6, slot), g7 = sp_ld4(L, 7, slot);
#if HRT_SP_GLOBAL
                    asm volatile("" : SP_PIN1(g0), SP_PIN1(g1), SP_PIN1(g2), SP_PIN1(g5), SP_PIN1(g6), SP_PIN1(g7));
#endif
                    Hit h;
                    uint32_t pm_unused;
                    sp_unpack_ray_hit(g0, g1, g2, ray, h, pm_unused);
                    ray.time = __uint_as_float(g5.x);
                    pnum = g5.w;
#endif
                    f3 thr = mk(__uint_as_float(g6.x), __uint_as_float(g6.y), __uint_as_float(g6.z));
                    rad = mk(__uint_as_float(g6.w), __uint_as_float(g7.x), __uint_as_float(g7.y));
                    int remaining = (int)sp_w7_left(g7.w);
#ifdef HRT_SP_SEG
                    asm volatile("" : "+v"(remaining), "+v"(ray.o.x), "+v"(thr.x), "+v"(rad.x));
#endif
                    SEG(0);  // record loaded
                    if (h.kind == 0u) {
                        rad = rad + thr * sky(cx, ray.d, remaining);
                        ended = true;
                    } else {
                        Rng rng;
#if HRT_SP_NOG5
                        rng.k0 = key.k0; rng.k1 = key.k1; rng.i = g7.z;
#else
                        rng.k0 = g5.y; rng.k1 = g5.z; rng.i = g7.z;
#endif
                        const Surface sf = shade(cx, ray, h);
                        SEG(1);  // shade: material rows, texel, normal map
                        f3 direct = mk(0.f, 0.f, 0.f);
                        if (LIGHTS) direct = direct_light(cx, sf, ray, rng);
                        SEG(8);  // direct light (shadow rays)
                        rad = rad + thr * (direct + sf.emission);
                        thr = thr * sf.albedo;
                        scatter(sf, ray, rng);
                        SEG(2);  // scatter
                        --remaining;
                        ended = (remaining == 0);
                        if ((HRT_SP_PRUNE & 1) && prune && thr.x == 0.f && thr.y == 0.f && thr.z == 0.f) ended = true;  // nothing can reach the sample any more
                        if (!ended) {
                            sp_st4(L, 6, slot, sp_pack(thr.x, thr.y, thr.z, rad.x));
                            sp_st4(L, 7, slot, make_uint4(__float_as_uint(rad.y), __float_as_uint(rad.z), rng.i, sp_w7((uint32_t)remaining, pnum)));
                            trace = true;
                            last_seg = (HRT_SP_PRUNE & 2) && !LIGHTS && prune && remaining == 1;
                        }
                    }
                }
                // spheres + squares + mesh gates for every lane of the chunk that has a new ray
                bool to_mesh = false;
                SEG(3);  // state written back / sample stored
                Hit hn;
                hn.kind = 0; hn.index = 0; hn.t = HRT_FLT_MAX; hn.tri = 0; hn.a0 = 0.f; hn.a1 = 0.f;
                uint32_t pmn = 0;
                if (trace) {
                    hn = prims_hit(cx, ray);
                    SEG(4);  // spheres + squares
                    pmn = has_mesh ? mesh_gates(cx, ray) : 0u;
                    SEG(5);  // mesh gates
                }
                if (SPHF && trace && hn.kind == 0u && pmn == 0u) {
                    // The builds for open scenes (a crowd of spheres under a sky): a ray that meets nothing ends its path HERE instead of
                    // travelling to a miss chunk (Scene.h:302-303; one visit in four of random_spheres).  Throughput, radiance and the
                    // bounces left were written to the record a moment ago by this lane (a new path: 1, 0, 6): read back, not kept live.
                    f3 thr = mk(1.f, 1.f, 1.f);
                    int remaining = 6;
                    if (!is_gen) {
                        const uint4 g6 = sp_ld4(L, 6, slot), g7 = sp_ld4(L, 7, slot);
                        thr = mk(__uint_as_float(g6.x), __uint_as_float(g6.y), __uint_as_float(g6.z));
                        rad = mk(__uint_as_float(g6.w), __uint_as_float(g7.x), __uint_as_float(g7.y));
                        remaining = (int)sp_w7_left(g7.w);
                    }
                    rad = rad + thr * sky(cx, ray.d, remaining);
                    trace = false; ended = true; last_seg = false;
                }
                if (last_seg) {  // HRT_SP_PRUNE bit 1: the closest hit of this ray is only asked whether it emits
                    bool dead;
                    if (hn.kind == 0u) {
                        dead = sky_is_zero;  // no sphere, no square: a mesh emits nothing and neither does this sky
                    } else {
                        const uint32_t mat = hn.kind == 1u ? __float_as_uint(ld(cx.ts, HRT_SPHERE_ROWS * hn.index + 1u).w)
                                                           : __float_as_uint(ld(cx.tq, HRT_QUAD_ROWS * hn.index + 4u).w);
                        dead = __float_as_uint(ld(cx.tm, HRT_MAT_ROWS * mat + 1u).w) == 0u;  // Material::emit, Material.cpp:13-15
                    }
                    if (dead) { trace = false; ended = true; }
                }
                if (ended) {  // Scene.h:348: the sample's colour, parked until its unit's ordered fold
                    const uint32_t n = min(pnum & 0x3FFFFFFFu, (uint32_t)HRT_SP_UNIT - 1u);  // the path's number; stays inside the scratch
                    fin_unit = pnum >> 30;
                    float *o = scratch + (size_t)fin_unit * ((size_t)HRT_SP_UNIT * 3u) + (size_t)n * 3u;
                    o[0] = rad.x / 6.f; o[1] = rad.y / 6.f; o[2] = rad.z / 6.f;
                    freed = true;
                }
                if (trace) {
                    sp_store_ray_hit(L, slot, ray, hn, pmn);
#if HRT_SP_PM4
#if HRT_SP_PMQ
                    if (!pm_in_entry)
#endif
                    if (pmn != 0u) sp_st4(L, 4, slot, make_uint4(HRT_KD_NIL, 0u, 0u, pmn));  // no walk in progress; the meshes to walk, where the T visit looks for them
#else
                    if (is_gen) spu(L, SP_WREF, slot) = HRT_KD_NIL;  // a fresh slot: no walk in progress (T keeps it so afterwards)
#endif
                    to_mesh = pmn != 0u;
                    kind = hn.kind;
                }
#if HRT_SP_PMQ
                sp_push_all(L, C, parity ^ 1u, trace ? (to_mesh ? 0u : 1u + kind) : (freed ? 5u : SP_TO_NONE), (trace && to_mesh && pm_in_entry) ? (slot | (pmn << 12)) : slot);
#else
                sp_push_all(L, C, parity ^ 1u, trace ? (to_mesh ? 0u : 1u + kind) : (freed ? 5u : SP_TO_NONE), slot);
#endif
#pragma unroll
                for (uint32_t k = 0; k < HRT_SP_UNITS; ++k) fin[k] += (uint32_t)__popcll(__ballot(freed && fin_unit == k));
                SEG(6);  // record stores + queue appends
#ifdef HRT_SP_SEG
                if (seg_on) seg[7] += 1;
#endif
            }
#ifdef HRT_SP_DEBUG
            dbg_class[dbg_k] += __builtin_readcyclecounter() - dbg_c0;
#endif
        }
#ifdef HRT_SP_DEBUG
        dbg_work += __builtin_readcyclecounter() - dbg_w0;
        if (ngen == 0u) { dbg_tail += __builtin_readcyclecounter() - dbg_w0; ++dbg_tail_cycles; }  // cycles of the drain: nothing left to start
#endif
        // this wave's part of the cycle is done: its records are in memory and its queue entries in LDS before it arrives
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (lane == 0) {  // ... and its finished samples in the scratch before their units see them finished
#pragma unroll
            for (uint32_t k = 0; k < HRT_SP_UNITS; ++k)
                if (fin[k]) atomicSub(&U[k].outstanding, fin[k]);
        }
#pragma unroll
        for (uint32_t k = 0; k < HRT_SP_UNITS; ++k) fin[k] = 0u;
        uint32_t arrived = 0;
        if (lane == 0) arrived = __hip_atomic_fetch_add(&C.arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        arrived = SP_UNI(arrived);
#ifdef HRT_SP_DEBUG
        const unsigned long long dbg_s0 = __builtin_readcyclecounter();
#endif
        if (arrived + 1u == (HRT_SP_WG / 64u) * kcyc[st]) serial_section(C, L.q);  // the last wave prepares the stream's next cycle
#ifdef HRT_SP_DEBUG
        dbg_serial += __builtin_readcyclecounter() - dbg_s0;
#endif
        ++kcyc[st];
    }
    L.q = q_all;
    __syncthreads();
    if (SP_UNI(__hip_atomic_load(&SH.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) return;  // uniform: read behind the barrier

#ifdef HRT_SP_SEG
    if (lane == 0 && R.stamps)
        for (int k = 0; k < 9; ++k) atomicAdd(R.stamps + k, seg[k]);
#endif
#ifdef HRT_WALK_SEG  // diagnostic build: where the trips of the KD walk spend their clocks (hrt_dual.hip mesh_walk)
    if (lane == 0 && R.stamps)
        for (int k = 0; k < 12; ++k) atomicAdd(R.stamps + k, cx.st[k]);
#endif
#ifdef HRT_SP_DEBUG
    if (lane == 0 && R.stamps) {  // per-wave sums: [0] clocks in chunk loops, [1] clocks alive, [2] cycles, [3] chunks, [4] clocks in the serial section
        atomicAdd(R.stamps + 0, dbg_work); atomicAdd(R.stamps + 1, __builtin_readcyclecounter() - dbg_t0);
        atomicAdd(R.stamps + 2, dbg_cycles); atomicAdd(R.stamps + 3, dbg_chunks); atomicAdd(R.stamps + 4, dbg_serial);
        for (int k = 0; k < 6; ++k) atomicAdd(R.stamps + 5 + k, dbg_class[k]);  // [5..10] clocks per chunk class
        atomicAdd(R.stamps + 11, dbg_wait); atomicAdd(R.stamps + 12, dbg_drain); atomicAdd(R.stamps + 13, dbg_tail); atomicAdd(R.stamps + 14, dbg_tail_cycles);  // waiting for a stream's next cycle; at the barrier that ends a fold
    }
#endif
#undef SP_UNI
}

}  // namespace hrtk

extern "C" __global__ void __launch_bounds__(HRT_SP_WG, HRT_SP_MINW) hrt_wgstream_kernel(const DRender R) { hrtk::stream_body<false>(R); }
extern "C" __global__ void __launch_bounds__(HRT_SP_WG, HRT_SP_MINW) hrt_wgstream_kernel_lights(const DRender R) { hrtk::stream_body<true>(R); }
// scenes with a crowd of spheres (HRT_SPHERE_FILTER_MIN..128): the builds that carry the spheres' pair filter (hrt_kernels.hip CtxT)
extern "C" __global__ void __launch_bounds__(HRT_SP_WG, HRT_SP_MINW) hrt_wgstream_kernel_sph(const DRender R) { hrtk::stream_body<false, false, true>(R); }
extern "C" __global__ void __launch_bounds__(HRT_SP_WG, HRT_SP_MINW) hrt_wgstream_kernel_lights_sph(const DRender R) { hrtk::stream_body<true, false, true>(R); }
// HRT_FLAG_EXACT_ONLY proof builds (no filters, no v_rcp_f32; CtxT in hrt_kernels.hip)
extern "C" __global__ void __launch_bounds__(HRT_SP_WG, HRT_SP_MINW) hrt_wgstream_kernel_exact(const DRender R) { hrtk::stream_body<false, true>(R); }
extern "C" __global__ void __launch_bounds__(HRT_SP_WG, HRT_SP_MINW) hrt_wgstream_kernel_lights_exact(const DRender R) { hrtk::stream_body<true, true>(R); }
