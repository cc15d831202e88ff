// Throughput of LDS atomics on gfx950: every CU runs one 1024-thread workgroup; each lane adds to pseudo-random columns of a
// 16K-entry LDS array (the row kernel's access shape: one entry per lane, 64 distinct addresses per wave-instruction).
// Prints lane-adds per clock per CU (2.4 GHz nominal) for f32 / f64 / u32 / u64, random and unit-stride addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <class T> __device__ void add(T* p, T v) { atomicAdd(p, v); }
template <class T, bool RANDOM>
__global__ __launch_bounds__(1024) void k(int iters, T* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    T* acc = reinterpret_cast<T*>(raw);
    const int N = 16384;
    for (int t = threadIdx.x; t < N; t += blockDim.x) acc[t] = (T)0;
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    int idx = threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            if (RANDOM) { s = s * 1664525u + 1013904223u; idx = (s >> 10) & (N - 1); }
            else idx = (idx + 64 * 7) & (N - 1);
            add<T>(&acc[idx], (T)1);
        }
    }
    __syncthreads();
    T sum = (T)0;
    for (int t = threadIdx.x; t < N; t += blockDim.x) sum += acc[t];
    if (sum == (T)12345) out[blockIdx.x] = sum;
}
template <class T, bool RANDOM>
void run(const char* name) {
    T* out; hipMalloc(&out, 256 * sizeof(T));
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<T, RANDOM>), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * sizeof(T));
    const int iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<T, RANDOM><<<256, 1024, 16384 * sizeof(T)>>>(10, out);
    hipEventRecord(a);
    k<T, RANDOM><<<256, 1024, 16384 * sizeof(T)>>>(iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double lane_adds = 256.0 * 1024 * iters * 8;
    printf("%-22s %8.3f ms  %.3e lane-adds/s  = %.2f lane-adds per clock per CU (2.4 GHz)  = %.1f clk per wave-instruction\n", name, ms,
           lane_adds / (ms * 1e-3), lane_adds / (ms * 1e-3) / 256 / 2.4e9, 64.0 / (lane_adds / (ms * 1e-3) / 256 / 2.4e9));
    hipFree(out);
}
int main() {
    run<float, true>("f32 random"); run<double, true>("f64 random"); run<unsigned int, true>("u32 random"); run<unsigned long long, true>("u64 random");
    run<float, false>("f32 unit-stride"); run<double, false>("f64 unit-stride"); run<unsigned int, false>("u32 unit-stride"); run<unsigned long long, false>("u64 unit-stride");
    return 0;
}
