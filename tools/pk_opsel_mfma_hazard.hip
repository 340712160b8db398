// Stand-alone reproducer of the hardware interaction found in the two-wave closed-loop kernel:
//
//   wave E:  v_pk_fma_f32 vdst[lo:hi], src0, src1, src2  op_sel:[0,1,0]    (the LOW result multiplies by src1's HIGH register;
//                                                                           likewise op_sel:[0,0,1], the low result adding src2's high)
//   wave M:  v_mfma_f32_32x32x16_f16 followed by a VALU instruction that READS its result   (same SIMD, other wave)
//
// Now and then wave E's instruction delivers  vdst.lo = src2.lo  in lanes 48-63 - the product term is lost - while vdst.hi and
// lanes 0-47 are right.  Not seen with the swizzle on src0 (op_sel:[1,0,0], same arithmetic), nor when wave M's MFMA results
// are not read by VALU, nor beside plain VALU / LDS / permlane work.  No wait-state rule of the ISA covers it (it is between
// two waves); the library is therefore built without packed fp32 arithmetic (-fno-slp-vectorize, tests/test_abi_cpu.py).
// 512-thread workgroups: waves 0-3 run the packed recurrence with a scalar shadow (v_fma_f32), waves 4-7 the partner work, so
// that every SIMD hosts one of each (a workgroup's waves are dealt to the four SIMDs cyclically).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/pk_opsel_mfma_hazard.hip -o build/pk_opsel && build/pk_opsel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

struct Ev { unsigned block, lane, iter, which; float got, want; };

// partner: mode 0 = VALU only, 1 = MFMA + VALU reading its result, 2 = MFMA + VALU that does not read MFMA results
__global__ __launch_bounds__(512) void k(Ev* ev, unsigned* nev, float* sink, int iters, int mode, int variant)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 4) {
        half8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (lane + j)); b[j] = (_Float16)(0.002f * j); }
        float16v c0 = {0}, c1 = {0};
        float y = lane;
        for (int i = 0; i < iters; ++i) {
            if (mode == 1) {
                // MFMA whose result is read by VALU
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float16v zero = {0};
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=v"(c0) : "v"(a), "v"(b), "v"(zero));
#pragma unroll
                    for (int j = 0; j < 16; ++j) y = fmaf(c0[j], 1e-3f, y);
                }
            } else if (mode == 2) {
                // MFMAs whose results no VALU instruction reads inside the loop, interleaved with independent VALU
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
#pragma unroll
                    for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y) : "v"(0.999f), "v"(0.001f));
                }
            } else {
#pragma unroll
                for (int r = 0; r < 48; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(y) : "v"(0.999f), "v"(0.001f));
            }
        }
        float s = y;
        for (int j = 0; j < 16; ++j) s += c0[j] + c1[j];
        sink[blockIdx.x * 256 + (wave - 4) * 64 + lane] = s;
        return;
    }
    // packed recurrence: uv <- k1 * uv + k2 ; acc.lo += cs * uv.hi (op_sel:[0,1,0]) ; acc.hi += sn * uv.hi ; then the other pair
    float2v uv = {0.01f * lane - 0.3f, 0.02f * lane + 0.1f}, acc = {0.0f, 0.0f};
    const float2v cssn = {0.8f + 0.001f * lane, 0.6f - 0.001f * lane}, nsc = {-0.6f + 0.001f * lane, 0.8f + 0.001f * lane};
    const float2v k1 = {0.9990f, 0.9985f}, k2 = {0.0004f, -0.0003f};
    float su = uv[0], sv = uv[1], sa0 = 0.0f, sa1 = 0.0f;       // scalar shadow
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(uv) : "v"(k1), "v"(k2));
            if (variant == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(cssn), "v"(uv));   // lo: cs * uv.hi, hi: sn * uv.hi
            else asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel:[1,0,0]" : "+v"(acc) : "v"(cssn), "v"(uv));               // same products, swizzle on src0
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(nsc), "v"(uv));                  // lo: -sn * uv.lo, hi: cs * uv.lo
            asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(su) : "v"(k1[0]), "v"(k2[0]));
            asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(sv) : "v"(k1[1]), "v"(k2[1]));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sa0) : "v"(cssn[0]), "v"(sv));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sa1) : "v"(cssn[1]), "v"(sv));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sa0) : "v"(nsc[0]), "v"(su));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(sa1) : "v"(nsc[1]), "v"(su));
        }
        const bool d0 = acc[0] != sa0, d1 = acc[1] != sa1, d2 = uv[0] != su || uv[1] != sv;
        if (d0 || d1 || d2) {
            const unsigned slot = atomicAdd(nev, 1u);
            if (slot < 4096u) ev[slot] = Ev{blockIdx.x, (unsigned)(wave * 64 + lane), (unsigned)i, (unsigned)(d0 | (d1 << 1) | (d2 << 2)), acc[0], sa0};
            acc[0] = sa0; acc[1] = sa1; uv[0] = su; uv[1] = sv;       // re-synchronise
        }
        if (fabsf(sa0) > 1.0e3f) { acc[0] = acc[1] = sa0 = sa1 = 0.0f; }
    }
    if (ev == nullptr) sink[blockIdx.x * 256 + wave * 64 + lane] = acc[0] + acc[1];
}

int main(int argc, char** argv)
{
    const int blocks = 256, iters = argc > 1 ? atoi(argv[1]) : 20000;
    Ev* d; unsigned* n; float* sink;
    (void)hipMalloc(&d, 4096 * sizeof(Ev)); (void)hipMalloc(&n, 4); (void)hipMalloc(&sink, blocks * 256 * 4);
    Ev* h = new Ev[4096];
    for (int variant = 0; variant < 2; ++variant)
        for (int mode = 0; mode < 3; ++mode) {
            unsigned long long events = 0, hi = 0, lo = 0, which[8] = {0};
            for (int rep = 0; rep < 10; ++rep) {
                (void)hipMemset(n, 0, 4);
                hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, n, sink, iters, mode, variant);
                unsigned cnt = 0;
                (void)hipMemcpy(&cnt, n, 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(h, d, sizeof(Ev) * (cnt < 4096 ? cnt : 4096), hipMemcpyDeviceToHost);
                events += cnt;
                for (unsigned e = 0; e < (cnt < 4096 ? cnt : 4096); ++e) { ((h[e].lane & 63) >= 48 ? hi : lo)++; which[h[e].which & 7]++; }
                if (rep == 0 && cnt) printf("    first event: block %u lane %u iter %u which %u got %.9g want %.9g\n", h[0].block, h[0].lane & 63, h[0].iter, h[0].which, h[0].got, h[0].want);
            }
            printf("%s, partner wave runs %s: packed != scalar events %llu (lanes 48-63: %llu, lanes 0-47: %llu; acc.lo only %llu, acc.hi only %llu, other %llu) over 10 launches x %d workgroups x %d iterations x 8\n",
                   variant == 0 ? "op_sel:[0,1,0] (lo <- src1.hi)" : "op_sel:[1,0,0] (lo <- src0.hi)", mode == 0 ? "VALU only" : (mode == 1 ? "MFMA -> VALU reads the result" : "MFMA + independent VALU"), events, hi, lo, which[1], which[2],
                   events - which[1] - which[2], blocks, iters);
        }
    return 0;
}
