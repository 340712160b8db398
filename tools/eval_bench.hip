// Micro-benchmark: cycles of ONE network evaluation (mlp_eval, 9-80-80-80-7, f16) by one wave per SIMD, weights in LDS -
// the network wave of the closed loop with nothing beside it.  Optionally a second, VALU-only wave per SIMD (the env wave's
// stand-in: a dependent v_fma chain) to see what the pair costs each other.
// With -DDPENV_EVAL_NO_VALU the activation / packing VALU is compiled out: the MFMA + LDS skeleton of the evaluation (round 2: 3 260 of the
// 3 856 cycles of a whole evaluation; its 76 MFMAs are 2 432).
//   hipcc -O3 --offload-arch=gfx950 -I ml4ca_amd/csrc -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 \
//         tools/eval_bench.hip -o build/wsdiag/eval_bench && build/wsdiag/eval_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "dpenv_policy_dev.h"
using namespace dpenv;

// PARTNER: 0 = network waves alone, 1 = network (s_setprio 3) + VALU waves, 3 = the same at equal priority, 2 = VALU waves alone
template <int KA, int PARTNER>
__global__ __launch_bounds__(512) void k(const uint4* frags, const float* bias, int nfrag, int nblk, int n_hidden, int iters, float* out,
                                         unsigned long long* cyc)
{
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < nfrag * 64; i += blockDim.x) lds[i] = frags[i];
    float* lb = (float*)(lds + nfrag * 64);
    for (int i = threadIdx.x; i < nblk * 32; i += blockDim.x) lb[i] = bias[i];
    if (threadIdx.x < 4) ((int*)(lb + nblk * 32))[threadIdx.x] = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 4) {                                   // partner waves (PARTNER only): a VALU-bound stream
        float x0 = lane, x1 = 1, x2 = 2, x3 = 3;
        const unsigned long long p0 = __builtin_readcyclecounter();
        volatile int* done = (volatile int*)(lb + nblk * 32);
        int trips = 0;
        // 16 independent-enough v_fma per poll of the flag, the same loop body in every mode
        for (int i = 0; (done[wave & 3] == 0) && (PARTNER != 2 || i < iters * 40); ++i, ++trips) {
            asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(0.999f), "v"(0.001f));
            asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(0.999f), "v"(0.001f));
            asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(0.999f), "v"(0.001f));
            asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(0.999f), "v"(0.001f));
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
        if (threadIdx.x == 256 && blockIdx.x == 0) { cyc[1] = __builtin_readcyclecounter() - p0; cyc[2] = (unsigned long long)trips; }
        return;
    }
    if (PARTNER == 2) return;
    if (PARTNER == 1) __builtin_amdgcn_s_setprio(3);
    half8 in0, in1;
    for (int j = 0; j < 8; ++j) { in0[j] = (_Float16)(0.01f * (lane + j)); in1[j] = (_Float16)(0.02f * j); }
    float o[8], acc = 0.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        mlp_eval<KA>(lds, lb, n_hidden, in0, in1, (_Float16)0.2f, o);
        acc += o[0] + o[5];
        in0[0] = (_Float16)(acc * 1e-3f);              // serialise consecutive evaluations, like the rollout does
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0) ((volatile int*)(lb + nblk * 32))[wave] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// the fp32-faithful (split-f16) evaluation alone: same harness, weights' low image = a second copy of the fragments
template <int KA>
__global__ __launch_bounds__(256) void kx(const uint4* frags, const float* bias, int nfrag, int nblk, int n_hidden, int iters, float* out,
                                          unsigned long long* cyc)
{
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < 2 * nfrag * 64; i += blockDim.x) lds[i] = frags[i % (nfrag * 64)];
    float* lb = (float*)(lds + 2 * nfrag * 64);
    for (int i = threadIdx.x; i < nblk * 32; i += blockDim.x) lb[i] = bias[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    SplitIn in;
    for (int j = 0; j < 8; ++j) { in.h0[j] = (_Float16)(0.01f * (lane + j)); in.h1[j] = (_Float16)(0.02f * j); in.l0[j] = (_Float16)(1e-4f * j); in.l1[j] = (_Float16)(2e-4f * j); }
    float o[8], acc = 0.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        mlp_eval_x<KA>(lds, lds + nfrag * 64, lb, n_hidden, in, 0.2f, o);
        acc += o[0] + o[5];
        in.h0[0] = (_Float16)(acc * 1e-3f);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

static void run_x(int iters)
{
    const int KS = 5, nh = 3, nfrag = 3 + 3 * KS * (nh - 1) + KS, nblk = 3 * (nh - 1) + 1;
    std::vector<uint16_t> hf((size_t)nfrag * 64 * 8);
    for (size_t i = 0; i < hf.size(); ++i) hf[i] = (uint16_t)(0x2000 + (i * 37) % 0x0800);
    std::vector<float> hb((size_t)nblk * 32, 0.01f);
    uint4* df; float* db; float* dout; unsigned long long* dc;
    hipMalloc(&df, hf.size() * 2); hipMalloc(&db, hb.size() * 4); hipMalloc(&dout, 256 * 512 * 4); hipMalloc(&dc, 24);
    hipMemcpy(df, hf.data(), hf.size() * 2, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    const size_t ldsb = (size_t)2 * nfrag * 64 * 16 + (size_t)nblk * 32 * 4 + 16;
    hipFuncSetAttribute((const void*)kx<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((kx<5>), dim3(256), dim3(256), ldsb, 0, df, db, nfrag, nblk, nh, iters, dout, dc);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("fp32-faithful evaluation alone: %.3f us per evaluation (events), %.0f ticks; 231 MFMAs = 7 392 matrix-pipe cycles\n", ms * 1e3 / iters, (double)c / iters);
}

template <int PARTNER> static void run(int iters)
{
    const int KS = 5, nh = 3, nfrag = 3 + 3 * KS * (nh - 1) + KS, nblk = 3 * (nh - 1) + 1;
    std::vector<uint16_t> hf((size_t)nfrag * 64 * 8);
    for (size_t i = 0; i < hf.size(); ++i) hf[i] = (uint16_t)(0x2000 + (i * 37) % 0x0800);     // small positive halves
    std::vector<float> hb((size_t)nblk * 32, 0.01f);
    uint4* df; float* db; float* dout; unsigned long long* dc;
    hipMalloc(&df, hf.size() * 2); hipMalloc(&db, hb.size() * 4); hipMalloc(&dout, 256 * 512 * 4); hipMalloc(&dc, 24); hipMemset(dc, 0, 24);
    hipMemcpy(df, hf.data(), hf.size() * 2, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    const size_t ldsb = (size_t)nfrag * 64 * 16 + (size_t)nblk * 32 * 4 + 16;
    hipFuncSetAttribute((const void*)k<5, PARTNER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = PARTNER ? 512 : 256;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<5, PARTNER>), dim3(256), dim3(threads), ldsb, 0, df, db, nfrag, nblk, nh, iters, dout, dc);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[3]; hipMemcpy(c, dc, 24, hipMemcpyDeviceToHost);
    printf("mode %d  kernel %.1f us; network wave %.0f ticks per evaluation; VALU wave: %.0f v_fma in that time = %.1f v_fma per us\n",
           PARTNER, ms * 1e3, (double)c[0] / iters, (double)c[2] * 16, (double)c[2] * 16 / (ms * 1e3));
}

int main()
{
    run<0>(2000);      // network waves alone
    run<1>(2000);      // + a VALU wave per SIMD, network waves at s_setprio 3: the VALU wave runs until its network wave is done
    run<3>(2000);      // the same without s_setprio
    run<2>(2000);      // VALU waves alone (fixed trip count)
    run_x(1000);
    return 0;
}
