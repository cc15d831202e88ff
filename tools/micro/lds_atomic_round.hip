// Does ds_add_f32 round to nearest?  One wave adds K pseudo-random positive floats to one LDS word (lane 0 only, sequentially)
// and to a second word from all 64 lanes; compared with the fp64 sum and with a plain fp32 register sum (RNE).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const float* v, int K, float* out) {
    __shared__ float acc[2];
    if (threadIdx.x < 2) acc[threadIdx.x] = 0.f;
    __syncthreads();
    float reg = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < K; i++) { atomicAdd(&acc[0], v[i]); reg += v[i]; }
    for (int i = threadIdx.x; i < K; i += blockDim.x) atomicAdd(&acc[1], v[i]);
    __syncthreads();
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = acc[1]; out[2] = reg; }
}
int main() {
    for (int K : {1000, 10000, 100000, 1000000}) {
        std::vector<float> h(K);
        double exact = 0;
        unsigned s = 12345;
        for (int i = 0; i < K; i++) { s = s * 1664525u + 1013904223u; h[i] = 1e-4f * (1.0f + (s >> 8) * (1.0f / 16777216.0f)); exact += h[i]; }
        float *d, *o; hipMalloc(&d, K * 4); hipMalloc(&o, 12);
        hipMemcpy(d, h.data(), K * 4, hipMemcpyHostToDevice);
        k<<<1, 64>>>(d, K, o);
        float r[3]; hipMemcpy(r, o, 12, hipMemcpyDeviceToHost);
        printf("K=%7d  lds sequential rel err %+.3e   lds 64 lanes %+.3e   fp32 register sum %+.3e\n", K, (r[0] - exact) / exact, (r[1] - exact) / exact, (r[2] - exact) / exact);
        hipFree(d); hipFree(o);
    }
    return 0;
}
