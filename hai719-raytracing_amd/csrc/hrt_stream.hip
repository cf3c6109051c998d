// hrt_stream_kernel -- the workgroup-streaming form of the trace megakernel (included by hrt_api.hip
// after hrt_kernels.hip, whose exact-arithmetic device functions it reuses unchanged).
//
// WHY.  In hrt_trace_kernel a lane owns a pixel and walks all stages of every bounce itself; the PMC
// profile (profiles/r01_pmc.json) shows 40 % VALU lane utilisation on Cornell+mesh because on any given
// bounce the 64 lanes of a wave want different things (new camera ray / mesh walk / sky / glass /
// diffuse ...).  Here the divergent rays are COMPACTED: a 1024-thread workgroup keeps a pool of
// HRT_SP_POOL paths in LDS (SoA, 24 dwords per path) and four stage queues of 16-bit slot ids.  One
// cycle =
//     GEN   free slot      -> camera ray (main.cpp:188-192)                  -> PRIMS queue
//     PRIMS ray            -> spheres + squares + mesh box gates             -> MESH or SHADE queue
//     MESH  ray + best hit -> rope KD walk of the gated meshes               -> SHADE queue
//     SHADE hit            -> sky | shade + direct light + scatter, path end -> PRIMS queue | free list
// Queues are double-buffered: a cycle consumes the "in" buffers, frozen at its start, and appends to
// the "out" buffers, so all four stages run in the SAME cycle with a single barrier pair.  The work
// of a cycle is cut into 64-entry chunks, each homogeneous in stage; the 16 waves pull chunks from one
// LDS cursor (dynamic balance inside the workgroup).  Appends are wave-aggregated: one __ballot, one
// LDS atomicAdd by the leader lane, positions by popcount of the lower lanes.
//
// DETERMINISM.  A path is keyed (pixel, sample) as before, so the schedule cannot change its random
// numbers or its arithmetic.  Finished samples are written to a per-workgroup scratch indexed by
// sample-major path number and folded into the pixel sum in sample order after the tile's chunk has
// drained: the fold is the reference's `image += color` order (main.cpp:193), bit for bit, whatever
// order paths finished in.
#include "hrt_device.h"

#ifndef HRT_SP_POOL
#define HRT_SP_POOL 1024   // paths resident per workgroup
#endif
#ifndef HRT_SP_WG
#define HRT_SP_WG 1024     // threads per workgroup (16 waves = 4 per SIMD, one workgroup per CU)
#endif
#define HRT_SP_SCHUNK 256  // samples per pixel traced between two ordered folds

namespace hrtk {

enum { SP_OX = 0, SP_OY, SP_OZ, SP_DX, SP_DY, SP_DZ, SP_TM, SP_TR, SP_TG, SP_TB, SP_RR, SP_RG, SP_RB,
       SP_K0, SP_K1, SP_RI, SP_N, SP_REM, SP_HT, SP_HID, SP_HA0, SP_HA1, SP_HTRI, SP_PM, SP_FIELDS };

struct SpCtl {           // control block in LDS
    uint32_t cP[2], cM[2], cS[2], cF[2];  // queue fills, [parity]
    uint32_t cursor;     // chunk cursor of the running cycle
    uint32_t ngen, gen_n0, paths_left;
    uint32_t nG, nP, nM, nS;  // chunks per stage this cycle
    uint32_t done, tile;
    uint32_t parity, cycles;  // which queue buffers are "in"; cycles spent on the current sample chunk
};

struct SpLds {
    uint32_t *st;        // SP_FIELDS x POOL dwords
    uint16_t *q;         // 8 queues x POOL: P0 P1 M0 M1 S0 S1 F0 F1
    SpCtl *ctl;
    float *run;          // 64 x 3 running pixel sums of the tile
};

__device__ __forceinline__ float &spf(const SpLds &L, int field, uint32_t slot) {
    return reinterpret_cast<float *>(L.st)[field * HRT_SP_POOL + slot];
}
__device__ __forceinline__ uint32_t &spu(const SpLds &L, int field, uint32_t slot) { return L.st[field * HRT_SP_POOL + slot]; }
__device__ __forceinline__ uint16_t *spq(const SpLds &L, int which, uint32_t parity) { return L.q + (2 * which + parity) * HRT_SP_POOL; }

// Wave-aggregated append of `slot` for the lanes with `want`: __ballot + one LDS atomic by the leader.
__device__ __forceinline__ void sp_push(uint16_t *q, uint32_t *count, bool want, uint32_t slot) {
    const uint64_t m = __ballot(want);
    if (m == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t leader = (uint32_t)__builtin_ctzll(m);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = __shfl(base, (int)leader);
    if (want) q[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)slot;
}

template <bool LIGHTS>
__device__ __forceinline__ void stream_body(const DRender &R) {
    extern __shared__ uint4 s_raw[];
    SpLds L;
    L.st = reinterpret_cast<uint32_t *>(s_raw);
    L.q = reinterpret_cast<uint16_t *>(L.st + SP_FIELDS * HRT_SP_POOL);
    L.ctl = reinterpret_cast<SpCtl *>(L.q + 8 * HRT_SP_POOL);
    L.run = reinterpret_cast<float *>(L.ctl + 1);
    uint4 *s_units = reinterpret_cast<uint4 *>(L.run + 64 * 3 + 64);  // 16-byte aligned: all sizes above are multiples of 16
    Ctx cx;
    cx.S = (cscene)R.scene;
    cx.lds = (lu4)s_units;
    cx.lds_n = R.lds_units;
    cx.err_abs = R.err_abs;
    unsigned long long stamps_local[17];
    cx.st = stamps_local;
    ccam cam = (ccam)R.cam;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    {
        gu4 g_units = (gu4)cx.S->kd_units;
        for (uint32_t i = tid; i < cx.lds_n; i += HRT_SP_WG) s_units[i] = ld(g_units, i);
    }
    SpCtl &C = *L.ctl;
    if (tid == 0) {
        C.cP[0] = C.cP[1] = C.cM[0] = C.cM[1] = C.cS[0] = C.cS[1] = 0;
        C.cF[0] = HRT_SP_POOL; C.cF[1] = 0;
        C.parity = 0; C.cycles = 0; C.done = 0;
    }
    for (uint32_t i = tid; i < HRT_SP_POOL; i += HRT_SP_WG) spq(L, 3, 0)[i] = (uint16_t)i;  // every slot free
    const bool has_mesh = cx.S->n_meshes != 0u;
    float *scratch = R.sp_scratch + (size_t)blockIdx.x * (64u * HRT_SP_SCHUNK * 3u);
    // Every control value is read from LDS after a barrier and made an SGPR with readfirstlane, so the
    // compiler sees provably wave-uniform loops around the barriers (no per-lane exec juggling).
#define SP_UNI(x) __builtin_amdgcn_readfirstlane(x)

#ifdef HRT_SP_DEBUG
    if (tid == 0 && blockIdx.x == 0 && R.stamps) { R.stamps[7] = 123ull; R.stamps[8] = R.tiles_owned; R.stamps[9] = R.spp; R.stamps[10] = (unsigned long long)cx.lds_n; }
#endif
    for (;;) {  // tiles
        __syncthreads();
        if (tid == 0) C.tile = atomicAdd(R.tile_counter, 1u);
        __syncthreads();
        const uint32_t j = SP_UNI(C.tile);
        if (j >= R.tiles_owned) break;  // finite queue: every workgroup gets here
        const uint32_t tile = R.rank + j * R.world;
        const uint32_t tx0 = (tile % R.tiles_x) * 8u, ty0 = (tile / R.tiles_x) * 8u;
        if (tid < 192) L.run[tid] = 0.f;

        for (uint32_t s0 = 0; s0 < R.spp; s0 += HRT_SP_SCHUNK) {  // sample chunks of the tile
            const uint32_t ns = min((uint32_t)HRT_SP_SCHUNK, R.spp - s0);
            __syncthreads();
            if (tid == 0) { C.paths_left = 64u * ns; C.gen_n0 = 0; C.cycles = 0; }
            for (;;) {  // cycles
                __syncthreads();
                if (tid == 0) {
                    const uint32_t parity = C.parity;
                    const uint32_t ngen = min(C.cF[parity], C.paths_left);
                    C.ngen = ngen;
                    C.nG = (ngen + 63u) >> 6; C.nP = (C.cP[parity] + 63u) >> 6;
                    C.nM = (C.cM[parity] + 63u) >> 6; C.nS = (C.cS[parity] + 63u) >> 6;
                    C.cursor = 0;
                    C.cP[parity ^ 1u] = 0; C.cM[parity ^ 1u] = 0; C.cS[parity ^ 1u] = 0;
                    C.cF[parity ^ 1u] = C.cF[parity] - ngen;  // the unused free slots carry over, SHADE appends after them
                    C.done = (ngen == 0u && C.cP[parity] == 0u && C.cM[parity] == 0u && C.cS[parity] == 0u) ? 1u : 0u;
                    if (++C.cycles > (1u << 16)) {  // bounded: a scheduling bug must not spin the GPU forever; the host reports it
                        if (R.stamps) R.stamps[15] = 0xDEADull;
                        C.done = 1u;
                    }
                }
                __syncthreads();
#ifdef HRT_SP_DEBUG
                if (tid == 0 && R.stamps && blockIdx.x == 0) { atomicAdd(R.stamps + 0, 1ull); atomicAdd(R.stamps + 1, (unsigned long long)C.ngen);
                    atomicAdd(R.stamps + 2, (unsigned long long)C.cP[C.parity]); atomicAdd(R.stamps + 3, (unsigned long long)C.cM[C.parity]);
                    atomicAdd(R.stamps + 4, (unsigned long long)C.cS[C.parity]); atomicAdd(R.stamps + 5, (unsigned long long)C.done); }
#endif
                if (SP_UNI(C.done)) break;
                const uint32_t parity = SP_UNI(C.parity);
                const uint32_t ngen = SP_UNI(C.ngen), cFin = SP_UNI(C.cF[parity]), n0 = SP_UNI(C.gen_n0);
                const uint32_t cPin = SP_UNI(C.cP[parity]), cMin = SP_UNI(C.cM[parity]), cSin = SP_UNI(C.cS[parity]);
                const uint32_t nM = SP_UNI(C.nM), nS = SP_UNI(C.nS), nP = SP_UNI(C.nP), total = nM + nS + nP + SP_UNI(C.nG);
                uint16_t *qPi = spq(L, 0, parity), *qPo = spq(L, 0, parity ^ 1u);
                uint16_t *qMi = spq(L, 1, parity), *qMo = spq(L, 1, parity ^ 1u);
                uint16_t *qSi = spq(L, 2, parity), *qSo = spq(L, 2, parity ^ 1u);
                uint16_t *qFi = spq(L, 3, parity), *qFo = spq(L, 3, parity ^ 1u);
                for (uint32_t i = tid; i < cFin - ngen; i += HRT_SP_WG) qFo[i] = qFi[ngen + i];  // carry unused free slots

                for (;;) {  // chunks of this cycle: MESH first (longest), then SHADE, PRIMS, GEN
                    uint32_t c = 0;
                    if (lane == 0) c = atomicAdd(&C.cursor, 1u);
                    c = __builtin_amdgcn_readfirstlane(c);
                    if (c >= total) break;
                    if (c < nM) {
                        // ---------------- MESH
                        const uint32_t e = c * 64u + lane;
                        const bool act = e < cMin;
                        uint32_t slot = 0;
                        if (act) {
                            slot = qMi[e] & (HRT_SP_POOL - 1u);
                            Ray ray;
                            ray.o = mk(spf(L, SP_OX, slot), spf(L, SP_OY, slot), spf(L, SP_OZ, slot));
                            ray.d = mk(spf(L, SP_DX, slot), spf(L, SP_DY, slot), spf(L, SP_DZ, slot));
                            ray.time = spf(L, SP_TM, slot);
                            Hit h;
                            const uint32_t hid = spu(L, SP_HID, slot);
                            h.kind = hid >> 28; h.index = hid & 0x0FFFFFFFu; h.t = spf(L, SP_HT, slot);
                            h.tri = 0; h.a0 = spf(L, SP_HA0, slot); h.a1 = spf(L, SP_HA1, slot);
                            meshes_hit(cx, ray, spu(L, SP_PM, slot), h);
                            spu(L, SP_HID, slot) = (h.kind << 28) | h.index;
                            spf(L, SP_HT, slot) = h.t; spf(L, SP_HA0, slot) = h.a0; spf(L, SP_HA1, slot) = h.a1;
                            spu(L, SP_HTRI, slot) = h.tri;
                        }
                        sp_push(qSo, &C.cS[parity ^ 1u], act, slot);
                    } else if (c < nM + nS) {
                        // ---------------- SHADE
                        const uint32_t e = (c - nM) * 64u + lane;
                        const bool act = e < cSin;
                        uint32_t slot = 0;
                        bool again = false, freed = false;
                        if (act) {
                            slot = qSi[e] & (HRT_SP_POOL - 1u);
                            Ray ray;
                            ray.o = mk(spf(L, SP_OX, slot), spf(L, SP_OY, slot), spf(L, SP_OZ, slot));
                            ray.d = mk(spf(L, SP_DX, slot), spf(L, SP_DY, slot), spf(L, SP_DZ, slot));
                            ray.time = spf(L, SP_TM, slot);
                            Hit h;
                            const uint32_t hid = spu(L, SP_HID, slot);
                            h.kind = hid >> 28; h.index = hid & 0x0FFFFFFFu; h.t = spf(L, SP_HT, slot);
                            h.tri = spu(L, SP_HTRI, slot); h.a0 = spf(L, SP_HA0, slot); h.a1 = spf(L, SP_HA1, slot);
                            f3 thr = mk(spf(L, SP_TR, slot), spf(L, SP_TG, slot), spf(L, SP_TB, slot));
                            f3 rad = mk(spf(L, SP_RR, slot), spf(L, SP_RG, slot), spf(L, SP_RB, slot));
                            int remaining = (int)spu(L, SP_REM, slot);
                            bool ended;
                            if (h.kind == 0u) {
                                rad = rad + thr * sky(cx.S, ray.d, remaining);
                                ended = true;
                            } else {
                                Rng rng;
                                rng.k0 = spu(L, SP_K0, slot); rng.k1 = spu(L, SP_K1, slot); rng.i = spu(L, SP_RI, slot);
                                const Surface sf = shade(cx.S, ray, h);
                                f3 direct = mk(0.f, 0.f, 0.f);
                                if (LIGHTS) direct = direct_light(cx, sf, ray, rng);
                                rad = rad + thr * (direct + sf.emission);
                                thr = thr * sf.albedo;
                                scatter(sf, ray, rng);
                                --remaining;
                                ended = (remaining == 0);
                                if (!ended) {
                                    spf(L, SP_OX, slot) = ray.o.x; spf(L, SP_OY, slot) = ray.o.y; spf(L, SP_OZ, slot) = ray.o.z;
                                    spf(L, SP_DX, slot) = ray.d.x; spf(L, SP_DY, slot) = ray.d.y; spf(L, SP_DZ, slot) = ray.d.z;
                                    spf(L, SP_TR, slot) = thr.x; spf(L, SP_TG, slot) = thr.y; spf(L, SP_TB, slot) = thr.z;
                                    spf(L, SP_RR, slot) = rad.x; spf(L, SP_RG, slot) = rad.y; spf(L, SP_RB, slot) = rad.z;
                                    spu(L, SP_RI, slot) = rng.i;
                                    spu(L, SP_REM, slot) = (uint32_t)remaining;
                                }
                            }
                            if (ended) {  // Scene.h:348: the sample's colour, parked until the ordered fold
#ifdef HRT_SP_DEBUG
                                if (R.stamps) atomicAdd(R.stamps + 6, 1ull);
#endif
                                const uint32_t n = min(spu(L, SP_N, slot), 64u * HRT_SP_SCHUNK - 1u);  // stays inside the scratch whatever happens
                                float *o = scratch + (size_t)n * 3u;
                                o[0] = rad.x / 6.f; o[1] = rad.y / 6.f; o[2] = rad.z / 6.f;
                                freed = true;
                            } else {
                                again = true;
                            }
                        }
                        sp_push(qPo, &C.cP[parity ^ 1u], again, slot);
                        sp_push(qFo, &C.cF[parity ^ 1u], freed, slot);
                    } else if (c < nM + nS + nP) {
                        // ---------------- PRIMS
                        const uint32_t e = (c - nM - nS) * 64u + lane;
                        const bool act = e < cPin;
                        uint32_t slot = 0;
                        bool to_mesh = false;
                        if (act) {
                            slot = qPi[e] & (HRT_SP_POOL - 1u);
                            Ray ray;
                            ray.o = mk(spf(L, SP_OX, slot), spf(L, SP_OY, slot), spf(L, SP_OZ, slot));
                            ray.d = mk(spf(L, SP_DX, slot), spf(L, SP_DY, slot), spf(L, SP_DZ, slot));
                            ray.time = spf(L, SP_TM, slot);
                            const Hit h = prims_hit(cx, ray);
                            const uint32_t pm = has_mesh ? mesh_gates(cx, ray) : 0u;
                            spu(L, SP_HID, slot) = (h.kind << 28) | h.index;
                            spf(L, SP_HT, slot) = h.t; spf(L, SP_HA0, slot) = h.a0; spf(L, SP_HA1, slot) = h.a1;
                            spu(L, SP_HTRI, slot) = 0u; spu(L, SP_PM, slot) = pm;
                            to_mesh = pm != 0u;
                        }
                        sp_push(qMo, &C.cM[parity ^ 1u], act && to_mesh, slot);
                        sp_push(qSo, &C.cS[parity ^ 1u], act && !to_mesh, slot);
                    } else {
                        // ---------------- GEN: path n of the chunk = (sample n / 64, pixel n % 64)
                        const uint32_t e = (c - nM - nS - nP) * 64u + lane;
                        const bool act = e < ngen;
                        uint32_t slot = 0;
                        bool started = false, freed = false;
                        if (act) {
                            slot = qFi[e] & (HRT_SP_POOL - 1u);
                            const uint32_t n = n0 + e;
                            const uint32_t p = n & 63u, s = s0 + (n >> 6);
                            const uint32_t px = tx0 + (p & 7u), py = ty0 + (p >> 3);
                            if (px < R.w && py < R.h) {
                                Rng rng;
                                rng.start(R.seed_lo, R.seed_hi, py * R.w + px, s);
                                const float u = ((float)px + rng.next()) / (float)R.w;
                                const float v = ((float)py + rng.next()) / (float)R.h;
                                const float tm = rng.next();
                                const Ray ray = camera_ray(cam, u, v, tm);
                                spf(L, SP_OX, slot) = ray.o.x; spf(L, SP_OY, slot) = ray.o.y; spf(L, SP_OZ, slot) = ray.o.z;
                                spf(L, SP_DX, slot) = ray.d.x; spf(L, SP_DY, slot) = ray.d.y; spf(L, SP_DZ, slot) = ray.d.z;
                                spf(L, SP_TM, slot) = tm;
                                spf(L, SP_TR, slot) = 1.f; spf(L, SP_TG, slot) = 1.f; spf(L, SP_TB, slot) = 1.f;
                                spf(L, SP_RR, slot) = 0.f; spf(L, SP_RG, slot) = 0.f; spf(L, SP_RB, slot) = 0.f;
                                spu(L, SP_K0, slot) = rng.k0; spu(L, SP_K1, slot) = rng.k1; spu(L, SP_RI, slot) = rng.i;
                                spu(L, SP_N, slot) = n; spu(L, SP_REM, slot) = 6u;  // MAXBOUNCES
                                started = true;
                            } else {  // pixel outside a ragged image: the sample is zero, the slot stays free
                                float *o = scratch + (size_t)n * 3u;
                                o[0] = 0.f; o[1] = 0.f; o[2] = 0.f;
                                freed = true;
                            }
                        }
                        sp_push(qPo, &C.cP[parity ^ 1u], started, slot);
                        sp_push(qFo, &C.cF[parity ^ 1u], freed, slot);
                    }
                }
                __syncthreads();
#ifdef HRT_SP_DEBUG
                if (tid == 0 && blockIdx.x == 0 && R.stamps && C.cycles == 1u) {  // snapshot after the first cycle
                    R.stamps[7] = C.cP[0]; R.stamps[8] = C.cP[1]; R.stamps[9] = C.cF[0]; R.stamps[10] = C.cF[1];
                    R.stamps[11] = parity; R.stamps[12] = ngen; R.stamps[13] = total; R.stamps[14] = C.cursor;
                }
#endif
                if (tid == 0) { C.paths_left -= ngen; C.gen_n0 += ngen; C.parity = parity ^ 1u; }
            }
            // the chunk has drained: fold its samples into the pixel sums in sample order (main.cpp:193).
            // The scratch was written by this workgroup's own waves on this CU; the barrier above orders it.
            __threadfence_block();
            __syncthreads();
            if (tid < 192) {
                const uint32_t p = tid / 3u, ch = tid % 3u;
                float acc = L.run[tid];
                // agent-scope relaxed loads (global_load ... sc1): served by L2, never by a stale L1 line of an
                // earlier chunk that other waves of this workgroup have since overwritten
                for (uint32_t s = 0; s < ns; ++s)
                    acc += __hip_atomic_load(scratch + ((size_t)s * 64u + p) * 3u + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                L.run[tid] = acc;
            }
        }
        __syncthreads();
        if (tid < 192) {
            const uint32_t p = tid / 3u;
            const uint32_t px = tx0 + (p & 7u), py = ty0 + (p >> 3);
            float c = 0.f;
            if (px < R.w && py < R.h) c = L.run[tid] / (float)R.spp;  // main.cpp:195
            R.out_tiles[(size_t)j * 192u + tid] = c;
        }
    }
}

}  // namespace hrtk

extern "C" __global__ void __launch_bounds__(HRT_SP_WG) hrt_wgstream_kernel(const DRender R) { hrtk::stream_body<false>(R); }
extern "C" __global__ void __launch_bounds__(HRT_SP_WG) hrt_wgstream_kernel_lights(const DRender R) { hrtk::stream_body<true>(R); }
