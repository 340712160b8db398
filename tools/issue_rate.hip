// Micro-benchmark: issue interval of ONE wave per SIMD for different VALU encodings (is the lone-wave limit instruction
// FETCH - bytes per instruction - or issue?).  8 independent chains each; 4-byte VOP2 (v_fmac_f32_e32, v_mul_f32_e32)
// against 8-byte VOP3 (v_fma_f32) and VOP3 with a 32-bit literal (12 bytes).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(x0) X(x1) X(x2) X(x3) X(x4) X(x5) X(x6) X(x7)
#define FMA3(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(k1), "v"(k2));
#define FMAC2(x) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x) : "v"(k1), "v"(k2));
#define MUL2(x) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(x) : "v"(k1));
#define FMALIT(x) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3a83126f" : "+v"(x) : "v"(k1));
#define FMAABS(x) asm volatile("v_fma_f32 %0, |%0|, %1, -%2" : "+v"(x) : "v"(k1), "v"(k2));
#define FMASG(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "s"(sk1), "v"(k2));

template <int MODE>
__global__ void k(float* out, int iters, float sk1)
{
    float x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3, x4 = 4, x5 = 5, x6 = 6, x7 = 7, k1 = 0.999f, k2 = 0.001f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { REP8(FMA3) REP8(FMA3) REP8(FMA3) REP8(FMA3) }
        if (MODE == 1) { REP8(FMAC2) REP8(FMAC2) REP8(FMAC2) REP8(FMAC2) }
        if (MODE == 2) { REP8(MUL2) REP8(MUL2) REP8(MUL2) REP8(MUL2) }
        if (MODE == 3) { REP8(FMALIT) REP8(FMALIT) REP8(FMALIT) REP8(FMALIT) }
        if (MODE == 4) { REP8(FMAABS) REP8(FMAABS) REP8(FMAABS) REP8(FMAABS) }
        if (MODE == 5) { REP8(FMASG) REP8(FMASG) REP8(FMASG) REP8(FMASG) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE> float run(float* out, int blocks, int iters)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 0.999f); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, iters, 0.999f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}

int main()
{
    float* out; (void)hipMalloc(&out, 4096 * 64 * 4);
    const int iters = 20000;
    const char* names[6] = {"v_fma_f32 (VOP3, 8 B)", "v_fmac_f32_e32 (VOP2, 4 B)", "v_mul_f32_e32 (VOP2, 4 B)", "v_fmaak_f32 + literal (VOP2, 8 B)",
                            "v_fma_f32 |abs| -neg (VOP3)", "v_fma_f32 with an SGPR operand"};
    for (int blocks : {1024, 2048, 4096}) {
        float t[6] = {run<0>(out, blocks, iters), run<1>(out, blocks, iters), run<2>(out, blocks, iters), run<3>(out, blocks, iters),
                      run<4>(out, blocks, iters), run<5>(out, blocks, iters)};
        for (int m = 0; m < 6; ++m)
            printf("waves %4d  %-32s %.2f ns per instruction per wave (%.2f cycles @ 2.4 GHz)\n", blocks, names[m],
                   t[m] * 1e6 / (32.0 * iters), t[m] * 1e-3 * 2.4e9 / (32.0 * iters));
    }
    return 0;
}
