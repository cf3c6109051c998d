// micro-test of the wave-aggregated LDS queue push used by hrt_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
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
__global__ void k(uint32_t *out, int mode) {
    extern __shared__ uint4 raw[];
    uint32_t *st = reinterpret_cast<uint32_t *>(raw);
    uint16_t *q = reinterpret_cast<uint16_t *>(st + 1024);
    uint32_t *cnt = reinterpret_cast<uint32_t *>(q + 2048);
    if (threadIdx.x == 0) { cnt[0] = 0; cnt[1] = 0; }
    __syncthreads();
    uint32_t parity = mode & 1;   // runtime value, like in the kernel
    bool want = (mode & 2) ? ((threadIdx.x % 3) != 0) : true;
    sp_push(q + parity * 1024, &cnt[parity], want, threadIdx.x);
    __syncthreads();
    if (threadIdx.x == 0) { out[0] = cnt[0]; out[1] = cnt[1]; }
    uint32_t c = cnt[parity];
    for (uint32_t i = threadIdx.x; i < c && i < 1024; i += blockDim.x) out[2 + i] = q[parity * 1024 + i];
}
int main() {
    uint32_t *d; hipMalloc(&d, 4096 * 4);
    for (int mode = 0; mode < 4; ++mode) {
        hipMemset(d, 0xFF, 4096 * 4);
        hipLaunchKernelGGL(k, dim3(1), dim3(512), 8192 + 4096 + 64, 0, d, mode);
        uint32_t h[1100]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        uint32_t c = h[mode & 1];
        unsigned long long sum = 0; for (uint32_t i = 0; i < c && i < 1024; ++i) sum += h[2 + i];
        printf("mode %d: counts %u %u  sum %llu  first %u %u %u\n", mode, h[0], h[1], sum, h[2], h[3], h[4]);
    }
    return 0;
}
