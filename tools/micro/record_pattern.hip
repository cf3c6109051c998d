// Calibration of the L2 <-> fabric counters on the path pool's OWN access pattern (the guide:
// "calibrate on a known byte count in your own access pattern before trusting an absolute").
//
// A hit visit of hrt_wgstream_kernel reads six aligned 16-byte groups of one 128-byte path record per lane
// (g0 g1 g2 g5 g6 g7: hrt_stream.hip sp_ld4, native global_load_dwordx4, pinned together) and writes five (g0 g1 g2 g6 g7:
// sp_st4), every lane of a wave on a different record.  This program does exactly that -- the same instruction forms, 64
// lanes on 64 pseudo-random records per visit -- over a pool of a chosen size, so that the bytes are KNOWN:
//   per visit and lane   96 B requested by loads (both 64-byte halves of the 128-byte line are touched), 80 B stored
// Run under rocprofv3 with one counter group per run (tools/calibrate_traffic.sh):
//   FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum |
//   TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
// with the pool (a) 96 MB: larger than the 32 MB of L2, inside the 256 MB Infinity Cache -- the regime the kernel's 128 MB
// pool + scratch is sized for -- and (b) 2 GB: larger than the Infinity Cache, every visit a DRAM access.  Prints one JSON
// line with the visit count and the known bytes; the script divides the counters by them.
//
//   hipcc -O3 --offload-arch=gfx950 -o record_pattern record_pattern.hip && ./record_pattern <pool MiB> <visits per lane>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef v4u __attribute__((address_space(1))) *gu4w;

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// one "hit visit" per iteration: 6 group loads of one record, all requested before the first use; 5 group stores
extern "C" __global__ void __launch_bounds__(1024) record_visits(uint32_t *pool, uint32_t n_records, uint32_t visits, uint32_t *sink) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t v = 0; v < visits; ++v) {
        const uint32_t slot = mix32(gid * 0x9E3779B1u + v * 0x85EBCA77u + 1u) % n_records;
        gu4w rec = (gu4w)(pool + (size_t)slot * 32u);
        v4u g0 = rec[0], g1 = rec[1], g2 = rec[2], g5 = rec[5], g6 = rec[6], g7 = rec[7];
        asm volatile("" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g5), "+v"(g6), "+v"(g7));
        acc += g0.x + g1.y + g2.z + g5.w + g6.x + g7.y;
        g0.x += 1u; g1.y += acc; g2.z ^= acc; g6.x += 3u; g7.y += g5.x;
        rec[0] = g0; rec[1] = g1; rec[2] = g2; rec[6] = g6; rec[7] = g7;
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;  // keeps the loads alive
}

#define CHECK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    const size_t mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 96;
    const uint32_t visits = argc > 2 ? (uint32_t)atoi(argv[2]) : 64u;
    const size_t bytes = mib << 20;
    const uint32_t n_records = (uint32_t)(bytes / 128u);
    uint32_t *pool = nullptr, *sink = nullptr;
    CHECK(hipMalloc((void **)&pool, bytes));
    CHECK(hipMalloc((void **)&sink, 4));
    CHECK(hipMemset(pool, 1, bytes));
    const int grid = 256, block = 1024;  // one 16-wave workgroup per CU, as the streaming kernel
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(record_visits, dim3(grid), dim3(block), 0, 0, pool, n_records, 2u, sink);  // first touch
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(record_visits, dim3(grid), dim3(block), 0, 0, pool, n_records, visits, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double lane_visits = (double)grid * block * visits;
    std::printf("{\"pool_mib\": %zu, \"records\": %u, \"lane_visits\": %.0f, \"requested_read_bytes\": %.0f, \"line_read_bytes\": %.0f, "
                "\"stored_bytes\": %.0f, \"ms\": %.3f, \"note\": \"the SECOND launch of record_visits is the measured one (the first is 2 visits per lane)\"}\n",
                mib, n_records, lane_visits, lane_visits * 96.0, lane_visits * 128.0, lane_visits * 80.0, ms);
    return 0;
}
