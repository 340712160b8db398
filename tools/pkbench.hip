// Micro-benchmark: issue cost of v_fma_f32 vs v_pk_fma_f32 for ONE wave per SIMD (the regime of the env
// kernels at 65 536 envs).  Same number of floating-point FMAs in both kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2_ __attribute__((ext_vector_type(2)));

__global__ void k_scalar(float* out, float a, float b, int iters)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
        x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
        x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ void k_packed(float* out, float a, float b, int iters)
{
    float2_ x0 = {(float)threadIdx.x, threadIdx.x + 1.f}, x1 = x0 + 2.f, x2 = x0 + 4.f, x3 = x0 + 6.f;
    const float2_ av = {a, a}, bv = {b, b};
    for (int i = 0; i < iters; ++i) {
        x0 = __builtin_elementwise_fma(x0, av, bv); x1 = __builtin_elementwise_fma(x1, av, bv);
        x2 = __builtin_elementwise_fma(x2, av, bv); x3 = __builtin_elementwise_fma(x3, av, bv);
    }
    float2_ s = x0 + x1 + x2 + x3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

template <typename K> float timeit(K k, int blocks, float* out, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, 0.999f, 0.001f, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, 0.999f, 0.001f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

int main()
{
    float* out; hipMalloc(&out, 8192 * 64 * 4);
    const int iters = 20000;
    for (int blocks : {1024, 2048, 4096}) {
        float a = timeit(k_scalar, blocks, out, iters), b = timeit(k_packed, blocks, out, iters);
        // per wave: scalar issues 8*iters v_fma, packed 4*iters v_pk_fma
        printf("waves %d: scalar %.3f ms (%.2f cyc/instr @2.4GHz)  packed %.3f ms (%.2f cyc/instr)\n", blocks, a,
               a * 1e-3 * 2.4e9 / (8.0 * iters) / ((blocks + 1023) / 1024), b, b * 1e-3 * 2.4e9 / (4.0 * iters) / ((blocks + 1023) / 1024));
    }
    return 0;
}
