// Micro-benchmark: can ONE wave per SIMD overlap its own MFMAs with independent VALU work?
// Per loop trip: 12 x v_mfma_f32_32x32x16_f16 (two accumulator chains) and 48 x v_fma_f32 (8 independent chains),
// issued (a) MFMAs first then the VALU block, (b) interleaved 1 MFMA : 4 VALU, (c) MFMAs only, (d) VALU only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

#define MF(c) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
#define V4 asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(k1), "v"(k2));
#define W4 asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(k1), "v"(k2));

template <int MODE>
__global__ void k(float* out, int iters)
{
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * j); }
    float16v c0 = {0}, c1 = {0};
    float x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7, k1 = 0.999f, k2 = 0.001f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // sequential: 12 MFMA then 48 VALU
            MF(c0) MF(c1) MF(c0) MF(c1) MF(c0) MF(c1) MF(c0) MF(c1) MF(c0) MF(c1) MF(c0) MF(c1)
            V4 W4 V4 W4 V4 W4 V4 W4 V4 W4 V4 W4
        } else if (MODE == 1) {   // interleaved
            MF(c0) V4 MF(c1) W4 MF(c0) V4 MF(c1) W4 MF(c0) V4 MF(c1) W4 MF(c0) V4 MF(c1) W4 MF(c0) V4 MF(c1) W4 MF(c0) V4 MF(c1) W4
        } else if (MODE == 2) {   // MFMA only
            MF(c0) MF(c1) MF(c0) MF(c1) MF(c0) MF(c1) MF(c0) MF(c1) MF(c0) MF(c1) MF(c0) MF(c1)
        } else {                  // VALU only
            V4 W4 V4 W4 V4 W4 V4 W4 V4 W4 V4 W4
        }
    }
    float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    for (int j = 0; j < 16; ++j) s += c0[j] + c1[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> float run(float* out, int blocks, int iters)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}

int main()
{
    float* out; hipMalloc(&out, 4096 * 64 * 4);
    const int iters = 20000;
    for (int blocks : {1024, 2048}) {
        float t0 = run<0>(out, blocks, iters), t1 = run<1>(out, blocks, iters), t2 = run<2>(out, blocks, iters), t3 = run<3>(out, blocks, iters);
        auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / iters; };
        printf("waves %d: per trip (12 MFMA + 48 VALU) cycles@2.4GHz: sequential %.0f  interleaved %.0f  | MFMA only %.0f (%.1f per MFMA)  VALU only %.0f (%.1f per VALU)\n",
               blocks, cyc(t0), cyc(t1), cyc(t2), cyc(t2) / 12, cyc(t3), cyc(t3) / 48);
    }
    return 0;
}
