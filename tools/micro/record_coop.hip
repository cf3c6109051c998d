// Does it matter how many LINES one vector memory instruction touches?  (Round 3, DESIGN.md 9 (iii): an extra one-dword gather per
// chunk cost the streaming kernel 9-17 %; a hit visit is 5 group loads + 5 group stores, every lane on its own 128-byte record.)
//
// The same bytes moved two ways, one 1024-thread workgroup per CU, every wave a "chunk" of 64 pseudo-random records per visit:
//   gather   lane l loads groups 0 1 2 6 7 of ITS record (5 x global_load_dwordx4, 64 lines each), then stores them back
//   coop     lane l of instruction i loads piece (64 i + l) mod 5 of record (64 i + l) div 5: the same 5 instructions, the same 320
//            16-byte pieces, but each instruction touches 13 records (lines) instead of 64; stores likewise
// (coop leaves every lane with pieces of OTHER lanes' records: the kernel would turn them round through LDS; not done here --
// this measures the memory side only.)  Pool: 128 MiB (the streaming kernel's: past L2, inside the Infinity Cache).
//
//   hipcc -O3 --offload-arch=gfx950 -o record_coop record_coop.hip && ./record_coop [pool MiB] [visits]
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
// record r (0..63) of the chunk wave `w` works on in visit `v`
__device__ __forceinline__ uint32_t slot_of(uint32_t w, uint32_t v, uint32_t r, uint32_t n_records) {
    return mix32((w * 64u + r) * 0x9E3779B1u + v * 0x85EBCA77u + 1u) % n_records;
}
__device__ __forceinline__ uint32_t group_of(uint32_t piece) { return piece < 3u ? piece : piece + 3u; }  // pieces 0..4 -> groups 0 1 2 6 7

template <bool COOP>
__global__ void __launch_bounds__(1024) visits_kernel(uint32_t *pool, uint32_t n_records, uint32_t visits, uint32_t *sink) {
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0;
    for (uint32_t v = 0; v < visits; ++v) {
        gu4w at[5];
#pragma unroll
        for (uint32_t i = 0; i < 5u; ++i) {
            const uint32_t k = COOP ? 64u * i + lane : 5u * lane + i;       // piece number within the chunk
            at[i] = (gu4w)(pool + (size_t)slot_of(wave, v, k / 5u, n_records) * 32u) + group_of(k % 5u);
        }
        v4u g[5];
#pragma unroll
        for (uint32_t i = 0; i < 5u; ++i) g[i] = *at[i];
        asm volatile("" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]));
#pragma unroll
        for (uint32_t i = 0; i < 5u; ++i) { acc += g[i].x; g[i].y += acc; }
#pragma unroll
        for (uint32_t i = 0; i < 5u; ++i) *at[i] = g[i];
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

#define CHECK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

template <bool COOP>
int run(const char *name, uint32_t *pool, uint32_t n_records, uint32_t visits, uint32_t *sink) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(visits_kernel<COOP>, dim3(256), dim3(1024), 0, 0, pool, n_records, 2u, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(visits_kernel<COOP>, dim3(256), dim3(1024), 0, 0, pool, n_records, visits, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double chunk_visits = 256.0 * 16.0 * visits;
    std::printf("{\"pattern\": \"%s\", \"ms\": %.3f, \"ns_per_chunk_visit_per_cu\": %.1f, \"line_gbps\": %.1f}\n", name, ms, ms * 1e6 / (16.0 * visits),
                chunk_visits * 64.0 * 128.0 * 2.0 / (ms * 1e-3) / 1e9);
    return 0;
}

int main(int argc, char **argv) {
    const size_t mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 128;
    const uint32_t visits = argc > 2 ? (uint32_t)atoi(argv[2]) : 256u;
    const size_t bytes = mib << 20;
    const uint32_t n_records = (uint32_t)(bytes / 128u);
    uint32_t *pool = nullptr, *sink = nullptr;
    CHECK(hipMalloc((void **)&pool, bytes));
    CHECK(hipMalloc((void **)&sink, 4));
    CHECK(hipMemset(pool, 1, bytes));
    for (int rep = 0; rep < 2; ++rep) {
        if (run<false>("gather: every lane its own record, 5 loads + 5 stores of 64 lines each", pool, n_records, visits, sink)) return 1;
        if (run<true>("coop: the same pieces, 13 records per instruction", pool, n_records, visits, sink)) return 1;
    }
    return 0;
}
