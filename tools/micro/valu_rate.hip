// What does one wave64 vector instruction cost on MI355X?  (Round 3: the "VALU busy" figure of the PMC summaries priced every
// vector instruction at 4 cycles of its SIMD -- 64 lanes on 16 -- and came out ABOVE 1 for random_spheres; and the packed fp32
// instructions had turned out to be no gain in the trace kernels.)
//
// One 1024-thread workgroup per CU (4 waves per SIMD, the streaming kernel's shape); every wave runs 32 768 iterations of 32
// INDEPENDENT instructions of one kind (eight accumulators: dependent-issue latency is hidden).  Per variant one JSON line:
// ns_per_wave_inst_per_simd = kernel time (HIP events) / the instructions of the 4 waves of a SIMD, and the s_memtime ticks of the
// waves (the shader clock: ticks of the longest wave / kernel time = the clock the variant ran at, 1.83 GHz under 64-lane
// v_fma_f32, 2.39 GHz when little switches).
//   ./valu_rate          v_fma_f32 under various lane masks, v_add_f32, v_pk_fma_f32, v_fma_f64, v_rcp_f32, v_mul_lo_u32, and
//                        v_fma_f32 / v_pk_fma_f32 with ONE wave per SIMD
//   ./valu_rate lanes    v_fma_f32 / v_add_f32 / v_mul_lo_u32 by the NUMBER of enabled lanes
// Measured (profiles/r03_valu_rate.json): v_add / v_fma_f32 2.7 cycles per instruction per SIMD, v_pk_fma_f32 = v_fma_f64 =
// v_mul_lo_u32 4.3, v_rcp_f32 8.2; a lone wave issues one v_fma_f32 per 5.4 cycles.
//
//   hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP4X(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int KIND>
__global__ void __launch_bounds__(1024) rate_kernel(uint64_t *cycles, float *sink, uint32_t iters, uint64_t lane_mask) {
    float a[8];
    double d[8];
    v2f p[8];
    uint32_t u[8];
    const float k = 1.0000001f, c = 1e-9f;
    for (int i = 0; i < 8; ++i) { a[i] = 1.f + (float)threadIdx.x * 1e-6f + (float)i; d[i] = a[i]; p[i].x = a[i]; p[i].y = a[i] + 1.f; u[i] = threadIdx.x * 2654435761u + (uint32_t)i; }
    const bool on = (lane_mask >> (threadIdx.x & 63u)) & 1ull;
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    if (on) {
        for (uint32_t it = 0; it < iters; ++it) {
            if (KIND == 0) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k), "v"(c));
                REP4X(OP)
#undef OP
            } else if (KIND == 1) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
                REP4X(OP)
#undef OP
            } else if (KIND == 2) {
#define OP(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(d[(i + 1) & 7]), "v"(d[(i + 2) & 7]));
                REP4X(OP)
#undef OP
            } else if (KIND == 3) {
#define OP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                REP4X(OP)
#undef OP
            } else if (KIND == 4) {
#define OP(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                REP4X(OP)
#undef OP
            } else if (KIND == 5) {
#define OP(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                REP4X(OP)
#undef OP
            }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i] + p[i].x + p[i].y + (float)u[i];
    if (s == 123.456f) sink[0] = s;
    if ((threadIdx.x & 63u) == 0u) cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

#define CHECK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

template <int KIND>
int run(const char *name, uint64_t mask, int block, uint64_t *d_cycles, float *d_sink) {
    const uint32_t iters = 32768;
    const int grid = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(grid), dim3(block), 0, 0, d_cycles, d_sink, 64u, mask);  // warm-up
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(grid), dim3(block), 0, 0, d_cycles, d_sink, iters, mask);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int waves = grid * (block / 64);
    std::vector<uint64_t> cyc(waves);
    CHECK(hipMemcpy(cyc.data(), d_cycles, waves * sizeof(uint64_t), hipMemcpyDeviceToHost));
    uint64_t mx = 0;
    double mean = 0;
    for (uint64_t c : cyc) { mx = c > mx ? c : mx; mean += (double)c / waves; }
    const double insts_per_wave = (double)iters * 32.0, waves_per_simd = block / 64 / 4.0;
    // counter ticks: __builtin_readcyclecounter = s_memtime, the shader clock
    std::printf("{\"variant\": \"%s\", \"waves_per_simd\": %.0f, \"kernel_ms\": %.4f, \"ns_per_wave_inst_per_simd\": %.4f, "
                "\"counter_ticks_mean\": %.0f, \"counter_ticks_max\": %llu, \"cycles_per_wave_inst_per_simd_at_2400MHz\": %.3f}\n",
                name, waves_per_simd < 1 ? 1.0 : waves_per_simd, ms, ms * 1e6 / (insts_per_wave * (waves_per_simd < 1 ? 1.0 : waves_per_simd)), mean, (unsigned long long)mx,
                ms * 1e6 / (insts_per_wave * (waves_per_simd < 1 ? 1.0 : waves_per_simd)) * 2.4);
    return 0;
}

int main(int argc, char **argv) {
    uint64_t *d_cycles = nullptr;
    float *d_sink = nullptr;
    CHECK(hipMalloc((void **)&d_cycles, 256 * 16 * sizeof(uint64_t)));
    CHECK(hipMalloc((void **)&d_sink, 4));
    const uint64_t all = ~0ull;
    if (argc > 1 && std::string(argv[1]) == "lanes") {   // the cost of an instruction by the NUMBER of enabled lanes
        const int counts[] = {64, 48, 32, 24, 20, 16, 14, 12, 10, 9, 8, 6, 4, 2, 1};
        for (int n : counts) {
            char name[64];
            std::snprintf(name, sizeof name, "v_fma_f32, lanes 0-%d", n - 1);
            if (run<0>(name, n == 64 ? all : ((1ull << n) - 1ull), 1024, d_cycles, d_sink)) return 1;
        }
        for (int n : {16, 8, 4, 1}) {
            char name[64];
            std::snprintf(name, sizeof name, "v_add_f32, lanes 0-%d", n - 1);
            if (run<5>(name, (1ull << n) - 1ull, 1024, d_cycles, d_sink)) return 1;
        }
        for (int n : {16, 8, 1}) {
            char name[64];
            std::snprintf(name, sizeof name, "v_mul_lo_u32, lanes 0-%d", n - 1);
            if (run<4>(name, (1ull << n) - 1ull, 1024, d_cycles, d_sink)) return 1;
        }
        return 0;
    }
    if (run<0>("v_fma_f32, lane 0 (first)", 1ull, 1024, d_cycles, d_sink)) return 1;
    if (run<0>("v_fma_f32, 64 lanes", all, 1024, d_cycles, d_sink)) return 1;
    if (run<0>("v_fma_f32, lanes 0-47", 0xFFFFFFFFFFFFull, 1024, d_cycles, d_sink)) return 1;
    if (run<0>("v_fma_f32, lanes 0-7", 0xFFull, 1024, d_cycles, d_sink)) return 1;
    if (run<0>("v_fma_f32, lanes 0-3", 0xFull, 1024, d_cycles, d_sink)) return 1;
    if (run<0>("v_fma_f32, lanes 0-1", 0x3ull, 1024, d_cycles, d_sink)) return 1;
    if (run<0>("v_fma_f32, lanes 0-15 and 32-47", 0x0000FFFF0000FFFFull, 1024, d_cycles, d_sink)) return 1;
    if (run<0>("v_fma_f32, 4 lanes of every 16", 0x000F000F000F000Full, 1024, d_cycles, d_sink)) return 1;
    if (run<0>("v_fma_f32, 8 lanes of every 16", 0x00FF00FF00FF00FFull, 1024, d_cycles, d_sink)) return 1;
    if (run<0>("v_fma_f32, lanes 0-31", 0xFFFFFFFFull, 1024, d_cycles, d_sink)) return 1;
    if (run<0>("v_fma_f32, lanes 0-15", 0xFFFFull, 1024, d_cycles, d_sink)) return 1;
    if (run<0>("v_fma_f32, lane 0", 1ull, 1024, d_cycles, d_sink)) return 1;
    if (run<0>("v_fma_f32, every second lane", 0x5555555555555555ull, 1024, d_cycles, d_sink)) return 1;
    if (run<0>("v_fma_f32, one lane of every 16", 0x0001000100010001ull, 1024, d_cycles, d_sink)) return 1;
    if (run<5>("v_add_f32, 64 lanes", all, 1024, d_cycles, d_sink)) return 1;
    if (run<1>("v_pk_fma_f32, 64 lanes", all, 1024, d_cycles, d_sink)) return 1;
    if (run<1>("v_pk_fma_f32, lanes 0-31", 0xFFFFFFFFull, 1024, d_cycles, d_sink)) return 1;
    if (run<2>("v_fma_f64, 64 lanes", all, 1024, d_cycles, d_sink)) return 1;
    if (run<3>("v_rcp_f32, 64 lanes", all, 1024, d_cycles, d_sink)) return 1;
    if (run<4>("v_mul_lo_u32, 64 lanes", all, 1024, d_cycles, d_sink)) return 1;
    if (run<0>("v_fma_f32, 64 lanes, ONE wave per SIMD", all, 256, d_cycles, d_sink)) return 1;
    if (run<1>("v_pk_fma_f32, 64 lanes, ONE wave per SIMD", all, 256, d_cycles, d_sink)) return 1;
    return 0;
}
