// Output stage of the path on the device: what main.cpp does after the last sample.
//   hrt_finalize_kernel     running sums -> pixel means (main.cpp:195) and gamma (Functions.cpp:56-60)
//   hrt_ppm6_kernel         (int)(255.f * min(1.f, c)) per channel as one byte (binary PPM)
//   hrt_ppm3_*_kernel       the same integers as the reference's ASCII file, byte for byte (main.cpp:258-262)
// All three are streaming byte/word kernels: HBM-bound, one pass over the frame (two for P3).
#include <hip/hip_runtime.h>
#include <stdint.h>

// sum / n in fp32 exactly as `image[i] /= nsamples` (Vec3::operator/= divides each component by (float)n),
// then pow(c, 1/2.2) in double.  Same arithmetic as the one-shot path (trace kernel + hrt_gamma_kernel).
extern "C" __global__ void hrt_finalize_kernel(const float *__restrict__ sums, float *__restrict__ out, uint32_t n,
                                               uint32_t total_samples, uint32_t gamma) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float c = sums[i] / (float)total_samples;
    if (gamma) c = (float)pow((double)c, 1.0 / 2.2);
    out[i] = c;
}

// main.cpp:259: (int)(255.f * std::min<float>(1.f, c)).  std::min(a, b) is (b < a) ? b : a, so a NaN
// channel becomes 1.f; the conversion truncates toward zero (v_cvt_i32_f32 saturates like cvttss2si
// at -inf: INT_MIN).
__device__ __forceinline__ int hrt_ppm_value(float c) {
    const float m = (c < 1.f) ? c : 1.f;
    return (int)(255.f * m);
}

extern "C" __global__ void hrt_ppm6_kernel(const float *__restrict__ frame, uint32_t n, unsigned char *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int v = hrt_ppm_value(frame[i]);
    out[i] = (unsigned char)(v < 0 ? 0 : v);  // a byte cannot hold the reference's negative text; clamp
}

__device__ __forceinline__ uint32_t hrt_dec_len(int v) {  // characters of `ostream << v` plus the separating space
    uint32_t a = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
    uint32_t n = 1;
    while (a >= 10u) { a /= 10u; ++n; }
    return n + (v < 0 ? 1u : 0u) + 1u;
}

__device__ __forceinline__ uint32_t hrt_pixel_len(const float *__restrict__ frame, uint32_t px) {
    return hrt_dec_len(hrt_ppm_value(frame[3u * px])) + hrt_dec_len(hrt_ppm_value(frame[3u * px + 1u])) +
           hrt_dec_len(hrt_ppm_value(frame[3u * px + 2u]));
}

// Pass 1: text length of every pixel; one block = 1024 pixels (4 per thread); block totals for the host-side scan.
extern "C" __global__ void __launch_bounds__(256) hrt_ppm3_measure_kernel(const float *__restrict__ frame, uint32_t npix,
                                                                           uint32_t *__restrict__ len,
                                                                           uint32_t *__restrict__ block_total) {
    __shared__ uint32_t part[256];
    const uint32_t first = blockIdx.x * 1024u + threadIdx.x * 4u;
    uint32_t mine = 0;
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t px = first + k;
        if (px < npix) {
            const uint32_t l = hrt_pixel_len(frame, px);
            len[px] = l;
            mine += l;
        }
    }
    part[threadIdx.x] = mine;
    __syncthreads();
    for (uint32_t s = 128u; s > 0u; s >>= 1) {
        if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_total[blockIdx.x] = part[0];
}

__device__ __forceinline__ unsigned char *hrt_put_dec(unsigned char *o, int v) {
    uint32_t a = v < 0 ? 0u - (uint32_t)v : (uint32_t)v;
    unsigned char d[10];
    int n = 0;
    do { d[n++] = (unsigned char)('0' + a % 10u); a /= 10u; } while (a);
    if (v < 0) *o++ = '-';
    while (n) *o++ = d[--n];
    *o++ = ' ';
    return o;
}

// Pass 2: exclusive scan of the block's 1024 lengths (thread partials through LDS), then every thread writes
// the text of its four pixels at base[block] + offset.
extern "C" __global__ void __launch_bounds__(256) hrt_ppm3_write_kernel(const float *__restrict__ frame, uint32_t npix,
                                                                         const uint32_t *__restrict__ len,
                                                                         const unsigned long long *__restrict__ base,
                                                                         unsigned char *__restrict__ out) {
    __shared__ uint32_t scan[256];
    const uint32_t first = blockIdx.x * 1024u + threadIdx.x * 4u;
    uint32_t mine = 0;
    for (uint32_t k = 0; k < 4u; ++k)
        if (first + k < npix) mine += len[first + k];
    scan[threadIdx.x] = mine;
    __syncthreads();
    for (uint32_t s = 1u; s < 256u; s <<= 1) {  // Hillis-Steele inclusive scan
        const uint32_t v = threadIdx.x >= s ? scan[threadIdx.x - s] : 0u;
        __syncthreads();
        scan[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned char *o = out + base[blockIdx.x] + (scan[threadIdx.x] - mine);
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t px = first + k;
        if (px < npix) {
            o = hrt_put_dec(o, hrt_ppm_value(frame[3u * px]));
            o = hrt_put_dec(o, hrt_ppm_value(frame[3u * px + 1u]));
            o = hrt_put_dec(o, hrt_ppm_value(frame[3u * px + 2u]));
        }
    }
}
