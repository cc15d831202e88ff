// How fast can a CU stream random SEGMENTS of a 100 MB table (the packed CSR of ML-25M shape: Infinity-Cache resident)?
// Wave-instruction = one segment: 64 lanes x {4, 8, 16} bytes, contiguous; 16 independent segment loads in flight per wave;
// 16 (or 32) waves per CU; the data is only summed (no LDS).  Prints GB/s per CU and chip-wide, and segments / us / CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int W> struct Vec;
template <> struct Vec<1> { using T = uint32_t; };
template <> struct Vec<2> { using T = uint2; };
template <> struct Vec<4> { using T = uint4; };
__device__ inline uint32_t fold(uint32_t v) { return v; }
__device__ inline uint32_t fold(uint2 v) { return v.x ^ v.y; }
__device__ inline uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }
template <int W>
__global__ __launch_bounds__(1024) void k(const uint32_t* __restrict__ table, uint32_t n_words, int iters, uint32_t* out) {
    using T = typename Vec<W>::T;
    const int lane = threadIdx.x & 63;
    uint32_t s = (blockIdx.x * 1024u + (threadIdx.x >> 6) * 64u) * 2654435761u + 777u;   // wave-uniform generator
    uint32_t acc = 0;
    const uint32_t seg_words = 64 * W;
    const uint32_t n_seg = n_words / seg_words - 1;
    for (int i = 0; i < iters; i++) {
        T v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            s = s * 1664525u + 1013904223u;
            const uint32_t seg = __builtin_amdgcn_readfirstlane((s >> 8) % n_seg);
            v[q] = *reinterpret_cast<const T*>(table + (size_t)seg * seg_words + 1 /* not line-aligned, like a CSR slice */ * 0 + lane * W);
        }
#pragma unroll
        for (int q = 0; q < 16; q++) acc ^= fold(v[q]);
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int W>
void run(const uint32_t* d, uint32_t n_words, int threads, uint32_t* out) {
    const int iters = 400;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<W><<<256, threads>>>(d, n_words, 10, out);
    hipEventRecord(a);
    k<W><<<256, threads>>>(d, n_words, iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double segs = 256.0 * (threads / 64) * iters * 16;
    const double bytes = segs * 256.0 * W;
    printf("%2d B/lane  %4d threads/CU : %7.3f ms  %7.1f GB/s per CU  %6.2f TB/s chip  %6.1f segments/us/CU\n", 4 * W, threads, ms,
           bytes / (ms * 1e-3) / 256 / 1e9, bytes / (ms * 1e-3) / 1e12, segs / (ms * 1e3) / 256);
}
int main() {
    const uint32_t n_words = 25u << 20;   // 100 MB
    uint32_t *d, *out; hipMalloc(&d, (size_t)n_words * 4); hipMalloc(&out, 4);
    hipMemset(d, 1, (size_t)n_words * 4);
    for (int threads : {512, 1024}) { run<1>(d, n_words, threads, out); run<2>(d, n_words, threads, out); run<4>(d, n_words, threads, out); }
    return 0;
}
