// Reproducer attempt for the interaction seen in the two-wave policy rollout: a wave running packed
// fp32 arithmetic (v_pk_fma_f32) while ANOTHER wave of the same workgroup runs MFMAs on the same SIMD.  Wave 0 advances
// the same recurrence twice - packed and scalar - and counts the lanes where they part; wave 1 runs back-to-back MFMAs
// (mode 1) or a VALU loop (mode 0).  Workgroups of 128 threads, enough of them to put both waves on every SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(128) void k(unsigned* bad_lanes, float* sink, int iters, int mfma_on)
{
    __shared__ uint4 frag[64 * 12];
    for (int j = threadIdx.x; j < 64 * 12; j += 128) frag[j] = make_uint4(0x3c003c00u + j, 0x38003800u, 0x34003400u, 0x30003000u);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    if (threadIdx.x >= 64) {
        half8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (lane + j)); b[j] = (_Float16)(0.002f * j); }
        float16v c0 = {0}, c1 = {0};
        float y = lane;
        for (int i = 0; i < iters; ++i) {
            if (mfma_on) {
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    const half8 w = __builtin_bit_cast(half8, frag[((i + r) % 12) * 64 + lane]);     // weight fragment from LDS
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(w), "v"(b));
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(w), "v"(a));
                }
            } else {
#pragma unroll
                for (int r = 0; r < 48; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y) : "v"(0.999f), "v"(0.001f));
            }
        }
        float s = y;
        for (int j = 0; j < 16; ++j) s += c0[j] + c1[j];
        sink[blockIdx.x * 64 + lane] = s;
        return;
    }
    float2v p[4];
    float q[8];
    for (int j = 0; j < 4; ++j) { p[j] = float2v{0.1f * lane + j, 0.2f * lane - j}; q[2 * j] = p[j][0]; q[2 * j + 1] = p[j][1]; }
    const float2v ka = {0.99990f, 1.00010f}, kb = {0.0003f, -0.0002f};
    unsigned bad = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(p[j]) : "s"(ka), "v"(kb));     // SGPR-pair operand, as the compiler emitted
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[j]) : "v"(ka));
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p[j]) : "v"(kb));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[2 * j]) : "v"(ka[0]), "v"(kb[0]));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(q[2 * j + 1]) : "v"(ka[1]), "v"(kb[1]));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = q[j] * ka[0];                          // op_sel_hi:[1,0]: both halves times ka.lo
#pragma unroll
            for (int j = 0; j < 4; ++j) { q[2 * j] = q[2 * j] - kb[0]; q[2 * j + 1] = q[2 * j + 1] - kb[1]; }
        }
        bool diff = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) diff = diff || (p[j][0] != q[2 * j]) || (p[j][1] != q[2 * j + 1]);
        if (diff) {
            ++bad;
#pragma unroll
            for (int j = 0; j < 4; ++j) { p[j][0] = q[2 * j]; p[j][1] = q[2 * j + 1]; }      // re-synchronise
        }
    }
    bad_lanes[blockIdx.x * 64 + lane] = bad;
}

int main()
{
    const int blocks = 1024, iters = 20000;
    unsigned* d; float* sink;
    (void)hipMalloc(&d, blocks * 64 * 4); (void)hipMalloc(&sink, blocks * 64 * 4);
    unsigned* h = new unsigned[blocks * 64];
    for (int mode = 0; mode < 2; ++mode) {
        unsigned long long events = 0, lanes_hi = 0, lanes_lo = 0;
        for (int rep = 0; rep < 20; ++rep) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(128), 0, 0, d, sink, iters, mode);
            (void)hipMemcpy(h, d, blocks * 64 * 4, hipMemcpyDeviceToHost);
            for (int i = 0; i < blocks * 64; ++i) if (h[i]) { events += h[i]; if ((i & 63) >= 48) ++lanes_hi; else ++lanes_lo; }
        }
        printf("partner wave runs %s: packed != scalar events %llu (lanes 48-63: %llu lane-launches, lanes 0-47: %llu) over 20 launches x %d workgroups x %d iterations\n",
               mode ? "MFMAs" : "VALU", events, lanes_hi, lanes_lo, blocks, iters);
    }
    return 0;
}
