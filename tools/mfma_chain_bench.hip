// Micro-benchmark (round 3): does a chain of v_mfma_f32_32x32x16_f16 that accumulate into the SAME tile (each one's SrcC is the previous
// one's result) run slower than MFMAs that alternate between two tiles?  One wave per SIMD, 240 MFMAs, operands in registers.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/mfma_chain_bench.hip -o build/wsdiag/mfma_chain_bench && build/wsdiag/mfma_chain_bench
// Result on MI355X (profiles/r03_mfma_chain_bench.txt): see the printed cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <int TILES, int VALU_BETWEEN>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters)
{
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * (threadIdx.x % 7 + j)); b[j] = (_Float16)(0.02f * (threadIdx.x % 5 + j)); }
    float16v c[4];
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 16; ++j) c[t][j] = 0.0f;
    float x = 1.0f + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            c[m % TILES] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[m % TILES], 0, 0, 0);
            asm volatile("" : "+v"(c[m % TILES]));
#pragma unroll
            for (int v = 0; v < VALU_BETWEEN; ++v) { x = __builtin_fmaf(x, 0.999f, 0.001f); asm volatile("" : "+v"(x)); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = x;
    for (int t = 0; t < 4; ++t) s += c[t][0] + c[t][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int TILES, int VALU_BETWEEN>
void run(const char* what)
{
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<TILES, VALU_BETWEEN>), dim3(256), dim3(256), 0, 0, out, cyc, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<TILES, VALU_BETWEEN>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s %6.2f ns per MFMA (events), %6.1f memtime ticks per MFMA\n", what, ms * 1e6 / (iters * 24.0), (double)c / (iters * 24.0));
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<1, 0>("one accumulator tile (every MFMA depends on the one before)");
    run<2, 0>("two tiles alternating");
    run<4, 0>("four tiles alternating");
    run<1, 5>("one tile, 5 dependent VALU between MFMAs");
    run<2, 5>("two tiles alternating, 5 dependent VALU between MFMAs");
    run<1, 7>("one tile, 7 VALU between");
    run<2, 7>("two tiles alternating, 7 VALU between");
    return 0;
}
