// Reproducer: is `v_permlane32_swap vX, vY` immediately followed by an MFMA that READS vX (SrcB) safe on gfx950?
// Variant NOPS = 0: the swap of the fourth fragment register is directly followed by the MFMA; NOPS = n: n wait states between.
// Each lane's expected result is computed by the same kernel with a generous gap.  Build + run on MI355X:
//   hipcc -O2 --offload-arch=gfx950 tools/permlane_mfma_hazard.hip -o build/wsdiag/permlane_hazard && build/wsdiag/permlane_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <int NOPS>
__global__ void k(float* out, int iters)
{
    const int lane = threadIdx.x & 63;
    float16v total = {0};
    for (int it = 0; it < iters; ++it) {
        half8 w, p, q;
        for (int j = 0; j < 8; ++j) {
            w[j] = (_Float16)(0.125f * ((lane + j + it) % 7) - 0.25f);
            p[j] = (_Float16)(0.0625f * ((lane * 3 + j + it) % 11));
            q[j] = (_Float16)(0.03125f * ((lane * 5 + j * 2 + it) % 13));
        }
        uint4 pu = __builtin_bit_cast(uint4, p), qu = __builtin_bit_cast(uint4, q);
        float16v acc = {0};
        // fixed registers (the MFMA needs register tuples): p -> v[20:23], q -> v[24:27], w -> v[28:31], result v[40:55]
#define HZ_BODY(GAP)                                                                                                              \
        asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %2\n\tv_mov_b32 v22, %3\n\tv_mov_b32 v23, %4\n\t"                      \
                     "v_mov_b32 v24, %5\n\tv_mov_b32 v25, %6\n\tv_mov_b32 v26, %7\n\tv_mov_b32 v27, %8\n\t"                      \
                     "v_mov_b32 v28, %9\n\tv_mov_b32 v29, %10\n\tv_mov_b32 v30, %11\n\tv_mov_b32 v31, %12\n\t" PRE                  \
                     "v_permlane32_swap_b32 v20, v24\n\tv_permlane32_swap_b32 v21, v25\n\tv_permlane32_swap_b32 v22, v26\n\t"   \
                     "v_permlane32_swap_b32 v23, v27\n\t" GAP                                                                    \
                     "v_mfma_f32_32x32x16_f16 v[40:55], v[28:31], v[20:23], 0\n\ts_nop 15\n\ts_nop 15\n\t"                      \
                     "v_mov_b32 %0, v40"                                                                                          \
                     : "=v"(r0) : "v"(pu.x), "v"(pu.y), "v"(pu.z), "v"(pu.w), "v"(qu.x), "v"(qu.y), "v"(qu.z), "v"(qu.w),          \
                       "v"(wu.x), "v"(wu.y), "v"(wu.z), "v"(wu.w)                                                                 \
                     : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v40", "v41", "v42",   \
                       "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55")
        const uint4 wu = __builtin_bit_cast(uint4, w);
        float r0;
#define PRE "s_nop 7\n\t"
        if (NOPS == 0) HZ_BODY("");
        else if (NOPS == 1) HZ_BODY("s_nop 0\n\t");
        else if (NOPS == 2) HZ_BODY("s_nop 1\n\t");
        else if (NOPS == 16) HZ_BODY("s_nop 15\n\t");
#undef PRE
#define PRE "v_mov_b32 v27, %8\n\tv_mov_b32 v23, %4\n\tv_mov_b32 v24, %5\n\tv_mov_b32 v20, %1\n\t"      /* VALU writes right before the swaps that read them */
        else if (NOPS == 100) HZ_BODY("s_nop 15\n\t");
#undef PRE
        acc[0] = r0;
        for (int j = 0; j < 16; ++j) total[j] += acc[j];
    }
    for (int j = 0; j < 16; ++j) out[(blockIdx.x * blockDim.x + threadIdx.x) * 16 + j] = total[j];
}

template <int NOPS> static void run(float* d, float* h, int n)
{
    hipLaunchKernelGGL(k<NOPS>, dim3(1024), dim3(64), 0, 0, d, 50);
    hipMemcpy(h, d, (size_t)n * 4, hipMemcpyDeviceToHost);
}

int main()
{
    const int n = 1024 * 64 * 16;
    float *d, *ref = new float[n], *got = new float[n];
    hipMalloc(&d, (size_t)n * 4);
    run<16>(d, ref, n);
    const int nops[4] = {0, 1, 2, 100};
    for (int v = 0; v < 4; ++v) {
        if (v == 0) run<0>(d, got, n); else if (v == 1) run<1>(d, got, n); else if (v == 2) run<2>(d, got, n); else run<100>(d, got, n);
        long bad = 0; double worst = 0;
        for (int i = 0; i < n; ++i) { const double e = fabs((double)got[i] - ref[i]); if (e != 0) { ++bad; if (e > worst) worst = e; } }
        if (nops[v] < 100) printf("%d wait state(s) between v_permlane32_swap and the MFMA reading its vdst: %ld of %d values differ (max |diff| %.4g)\n", nops[v], bad, n, worst);
        else printf("v_mov writes of the swap operands directly in front of the swaps (0 wait states): %ld of %d values differ (max |diff| %.4g)\n", bad, n, worst);
    }
    return 0;
}
