// Micro-benchmark (round 4, VERDICT r03 item 3 ii): the 16-row third block of an 80-wide layer of the split-f16 evaluation (mlp_eval_x) as
// shipped - ten slots of three v_mfma_f32_32x32x16_f16, half of every accumulator tile padding - against a 16x16x32 tiling of the same
// block: 4 env tiles x 3 k-steps x 3 split products = 36 v_mfma_f32_16x16x32_f16 (half the matrix cycles), which needs its B operands
// re-arranged from the 32-env fragment layout into 16-env rows (and its results back): cross-lane moves (v_permlane32_swap /
// v_permlane16_swap, one register each).  Both forms carry the SAME activation work beside their MFMAs - the eight units (4 activations
// each: 4 mul, 4 max, 2 cvt_pk, 4 fma_mix, 2 cvt_pk) of the block before, as in the shipped pipeline - so what is compared is
// what a lone wave's instruction issue and the matrix pipe make of each stream.  Operands stay in registers (weights from LDS cost both forms
// the same reads).  One wave per SIMD, 256 workgroups of 256 threads.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/block2_tiling_bench.hip -o build/wsdiag/block2_tiling_bench
// Result on MI355X: profiles/r04_block2_tiling_bench.txt.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

// one unit of the shipped activation + split (dpenv_policy_dev.h: leaky_split4_hi / _lo), in two halves
__device__ __forceinline__ void unit_hi(const float x[4], const float m[4], float t[4], uint32_t H[2])
{
    asm volatile("v_max_f32 %2, %6, %10\n\tv_max_f32 %3, %7, %11\n\tv_max_f32 %4, %8, %12\n\tv_max_f32 %5, %9, %13\n\t"
                 "v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5"
                 : "=&v"(H[0]), "=&v"(H[1]), "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
                 : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]));
}
__device__ __forceinline__ void unit_lo(float t[4], const uint32_t H[2], uint32_t L[2])
{
    asm volatile("v_fma_mix_f32 %2, %6, %8, %2 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %3, %6, %8, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                 "v_fma_mix_f32 %4, %7, %8, %4 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %5, %7, %8, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                 "v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5"
                 : "=&v"(L[0]), "=&v"(L[1]), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3])
                 : "v"(H[0]), "v"(H[1]), "v"(-1.0f));
}

// FORM 0: shipped.  10 slots x (MFMA32, 4 VALU, MFMA32, 6 VALU, MFMA32, 6 VALU), units on 8 of the 10 slots.
// FORM 1: 12 groups x 3 MFMA16; the 8 units spread over the groups in the same three pieces; MOVES cross-lane moves spread over the groups.
template <int FORM, int MOVES>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters)
{
    half8 wh, wl, bh, bl;
    for (int j = 0; j < 8; ++j) {
        wh[j] = (_Float16)(0.01f * (threadIdx.x % 7 + j)); wl[j] = (_Float16)(1e-4f * (threadIdx.x % 3 + j));
        bh[j] = (_Float16)(0.02f * (threadIdx.x % 5 + j)); bl[j] = (_Float16)(1e-4f * (threadIdx.x % 11 + j));
    }
    float16v c0, c1, src;
    for (int j = 0; j < 16; ++j) { c0[j] = 0.0f; c1[j] = 0.0f; src[j] = 0.1f * (float)(j - 7) + 0.001f * threadIdx.x; }
    float4v d[4];
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 4; ++j) d[t][j] = 0.0f;
    uint32_t H[2] = {0, 0}, L[2] = {0, 0}, mv0 = threadIdx.x, mv1 = threadIdx.x * 3u;
    uint32_t sink = 0;
    const float leak = 0.2f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (FORM == 0) {
#pragma unroll
            for (int s = 0; s < 10; ++s) {
                float16v& c = (s & 1) ? c1 : c0;
                const bool has = s < 8;
                float x[4], m[4], t[4];
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh, c, 0, 0, 0); asm volatile("" : "+v"(c));
                if (has) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { x[q] = src[(4 * s + q) & 15]; m[q] = x[q] * leak; }
                }
                __builtin_amdgcn_sched_barrier(0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl, c, 0, 0, 0); asm volatile("" : "+v"(c));
                if (has) unit_hi(x, m, t, H);
                __builtin_amdgcn_sched_barrier(0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, c, 0, 0, 0); asm volatile("" : "+v"(c));
                if (has) { unit_lo(t, H, L); sink ^= H[0] ^ H[1] ^ L[0] ^ L[1]; }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            constexpr int G = 12;                       // 4 env tiles x 3 k-steps
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float4v& c = d[g & 3];
                // 8 units in 24 pieces over 36 MFMA gaps: the piece of gap (g, j) - units ride on the first 8 groups
                const bool has = g < 8;
                float x[4], m[4], t[4];
                // cross-lane moves of this group: MOVES / G, split over the three gaps
                constexpr int MPG = MOVES / G;
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh, c, 0, 0, 0); asm volatile("" : "+v"(c));
                if (has) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { x[q] = src[(4 * g + q) & 15]; m[q] = x[q] * leak; }
                }
#pragma unroll
                for (int v = 0; v < MPG / 3; ++v) { auto r = __builtin_amdgcn_permlane32_swap(mv0, mv1, false, false); mv0 = r[0]; mv1 = r[1]; }
                __builtin_amdgcn_sched_barrier(0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bl, c, 0, 0, 0); asm volatile("" : "+v"(c));
                if (has) unit_hi(x, m, t, H);
#pragma unroll
                for (int v = 0; v < MPG / 3; ++v) { auto r = __builtin_amdgcn_permlane16_swap(mv0, mv1, false, false); mv0 = r[0]; mv1 = r[1]; }
                __builtin_amdgcn_sched_barrier(0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, bh, c, 0, 0, 0); asm volatile("" : "+v"(c));
                if (has) { unit_lo(t, H, L); sink ^= H[0] ^ H[1] ^ L[0] ^ L[1]; }
#pragma unroll
                for (int v = 0; v < MPG - 2 * (MPG / 3); ++v) { auto r = __builtin_amdgcn_permlane32_swap(mv0, mv1, false, false); mv0 = r[0]; mv1 = r[1]; }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = c0[0] + c1[3] + (float)(sink & 7u) + (float)((mv0 ^ mv1) & 3u);
    for (int t = 0; t < 4; ++t) s += d[t][0] + d[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int FORM, int MOVES>
double run(const char* what)
{
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<FORM, MOVES>), dim3(256), dim3(256), 0, 0, out, cyc, 200);
    hipLaunchKernelGGL((k<FORM, MOVES>), dim3(256), dim3(256), 0, 0, out, cyc, 200);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<FORM, MOVES>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / iters;
    printf("%-100s %8.1f ns per block\n", what, ns);
    hipFree(out); hipFree(cyc);
    return ns;
}

int main()
{
    printf("third row-block of one 80-wide hidden layer, split arithmetic, with the activation work of the block before riding along (one wave per SIMD)\n");
    const double a = run<0, 0>("shipped: 10 slots x 3 v_mfma_f32_32x32x16_f16 (half of each tile padding) + 8 activation units");
    const double b0 = run<1, 0>("16x16x32: 36 v_mfma_f32_16x16x32_f16 + 8 activation units, NO operand re-arrangement (not a usable kernel: the bound)");
    const double b1 = run<1, 96>("16x16x32: ... + 96 cross-lane moves (one per operand word of the 12 B-operand pairs)");
    const double b2 = run<1, 144>("16x16x32: ... + 144 cross-lane moves (operands + the 16 x 64 results back into the 32-env fragment layout)");
    const double b3 = run<1, 192>("16x16x32: ... + 192 cross-lane moves (two swaps per operand word)");
    printf("relative to shipped: bound %.2f, 96 moves %.2f, 144 moves %.2f, 192 moves %.2f\n", b0 / a, b1 / a, b2 / a, b3 / a);
    return 0;
}
