#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
template <int MODE> __global__ void k(uint32_t* o, int iters)
{
    uint32_t c0 = threadIdx.x * 2654435761u + 1, c2 = threadIdx.x * 40503u + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            if (MODE == 0) {
                const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
                const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
                c0 = hi1 ^ lo0 ^ r; c2 = hi0 ^ lo1;
            } else {
                const uint64_t p0 = (uint64_t)c0 * 0xD2511F53ull, p1 = (uint64_t)c2 * 0xCD9E8D57ull;
                c0 = (uint32_t)(p1 >> 32) ^ (uint32_t)p0 ^ r; c2 = (uint32_t)(p0 >> 32) ^ (uint32_t)p1;
            }
        }
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = c0 ^ c2;
}
template <int MODE> void run(uint32_t* d, int waves_per_simd)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256 * 4 * waves_per_simd), dim3(64), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d, %d wave(s)/SIMD: %.2f ns per Philox round (2 products)\n", MODE, waves_per_simd, ms * 1e6 / (iters * 10.0));
}
int main() { uint32_t* d; hipMalloc(&d, 256 * 4 * 4 * 64 * 4); run<0>(d, 1); run<1>(d, 1); run<0>(d, 2); run<1>(d, 2); return 0; }
