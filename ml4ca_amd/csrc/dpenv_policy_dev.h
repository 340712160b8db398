// dpenv_policy_dev.h - device code shared by the policy translation units of libdpenv.so:
//   dpenv_policy.hip    f16 network arithmetic (fast mode): forward pass, one-wave and two-wave closed-loop rollouts
//   dpenv_policy_x.hip  split-f16 ("fp32-faithful") network arithmetic: forward pass and one-wave closed-loop rollout
// See dpenv_policy.hip for the mapping of the MLP to the matrix cores.
#ifndef DPENV_POLICY_DEV_H
#define DPENV_POLICY_DEV_H
#include "dpenv_env_dev.h"

namespace dpenv {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));


constexpr int PBLOCK = 256;          // 4 waves share one LDS image of the weights
constexpr int PWAVES = PBLOCK / 64;

// ---- weight image (one per network; the split arithmetic has a second one for the low parts) -----------------------------------
// MFMA A-operand fragments of 16 bytes per lane, stored COMPACTLY: a fragment holds 32 output rows (lane & 31) x 8 input slots per
// lane half, but
//   * the third row-block of an 80-wide layer (KS = 5) has 16 real rows: rows 80..95 are padding whose products nobody reads
//     (registers 8..15 of that accumulator tile are never consumed) - stored as 2 x 16 entries, lanes r and r + 16 read the same one;
//   * the output layer has at most 8 real rows - stored as 2 x 8 entries.
// 28.75 KiB instead of 38 KiB per image for the shipped 9-80-80-80 shape, which is what lets all four images of the split
// arithmetic AND the two-wave form's mailboxes share the 160 KiB LDS.  Entry offsets (16-byte units) inside an image:
//   first layer   [0, L0):        block 0 | block 1 | block 2 (B2 entries)
//   hidden layer  LH each:        block 0: KS x 64 | block 1: KS x 64 | block 2: KS x B2
//   output layer  KS x 16
// pack_policy_kernel (dpenv_policy.hip) writes it, FragAddr reads it.
template <int KS>
struct FragAddr {
    static constexpr int B2 = (KS == 5) ? 32 : 64;          // entries of a block-2 fragment
    static constexpr int L0 = 128 + B2;                     // first layer
    static constexpr int LH = KS * (128 + B2);              // a hidden -> hidden layer
    static constexpr int LO = KS * 16;                      // the output layer
    int lane, lane2, lane_o;                                // this lane's entry inside a full / block-2 / output fragment
    __device__ __forceinline__ explicit FragAddr(int l)
        : lane(l), lane2(KS == 5 ? (((l >> 5) << 4) | (l & 15)) : l), lane_o(((l >> 5) << 3) | (l & 7)) {}
    __device__ __forceinline__ static half8 ld(const uint4* W, int e) { return __builtin_bit_cast(half8, W[e]); }
    __device__ __forceinline__ half8 first(const uint4* W, int mo) const { return ld(W, mo < 2 ? mo * 64 + lane : 128 + lane2); }
    __device__ __forceinline__ half8 hid(const uint4* W, int e0, int mo, int ks) const
    {
        return ld(W, mo < 2 ? e0 + (mo * KS + ks) * 64 + lane : e0 + 2 * KS * 64 + ks * B2 + lane2);
    }
    __device__ __forceinline__ half8 out(const uint4* W, int e0, int ks) const { return ld(W, e0 + ks * 16 + lane_o); }
    // block 0 of the layer at e0, which is a hidden layer or the output layer (is_out: uniform)
    __device__ __forceinline__ half8 blk0(const uint4* W, int e0, bool is_out, int ks) const
    {
        return ld(W, is_out ? e0 + ks * 16 + lane_o : e0 + ks * 64 + lane);
    }
};
__host__ __device__ constexpr int frag_image_entries(int ks, int n_hidden)
{
    return (128 + (ks == 5 ? 32 : 64)) + (n_hidden - 1) * ks * (128 + (ks == 5 ? 32 : 64)) + ks * 16;
}

// bias tile of one 32-row block in accumulator layout: register r of a lane in half h holds output row
// 8 (r >> 2) + 4 h + (r & 3); all lanes of a half read the same 64 bytes (LDS broadcast)
__device__ __forceinline__ float16v ldbias(const float* B, int blk, int lane)
{
    const float4* p = (const float4*)(B + blk * 32 + (lane >> 5) * 16);
    const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    const float16v r = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    return r;
}

// hidden activation + f32 -> f16 of registers 8s .. 8s+7 of an accumulator tile = the B fragment of k-step s.
// ACT_LEAKY: leaky-relu of slope `leak` (0 = relu), the reference's default ('leaky', train.py:24,31); ACT_TANH: its
// '--activation tanh' option (Spinning Up's own default), evaluated in f32 as 1 - 2 / (exp(2x) + 1).
constexpr int ACT_LEAKY = 0, ACT_TANH = 1;

template <int ACT>
__device__ __forceinline__ half8 act_pack(const float16v& acc, int s, _Float16 leak)
{
#ifdef DPENV_EVAL_NO_VALU          // tools/eval_bench.hip only: the MFMA + LDS skeleton of an evaluation (results meaningless)
    const uint4 raw = {__float_as_uint(acc[8 * s]), __float_as_uint(acc[8 * s + 1]), __float_as_uint(acc[8 * s + 2]), __float_as_uint(acc[8 * s + 3])};
    return __builtin_bit_cast(half8, raw);
#endif
    half8 r;
    const half2v lk = {leak, leak};
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        if (ACT == ACT_TANH) {
            float t[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float e = __builtin_amdgcn_exp2f(acc[8 * s + j + q] * 2.8853900817779268f);      // exp(2x)
                t[q] = fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
            }
            const float2v tf = {t[0], t[1]};
            const half2v h = __builtin_convertvector(tf, half2v);
            r[j] = h[0]; r[j + 1] = h[1];
        } else {
            // v_cvt_pk_f16_f32 (gfx950: round-to-nearest-even, unlike v_cvt_pkrtz), then packed f16 mul + max:
            // 1.5 instructions per activation
            const float2v af = {acc[8 * s + j], acc[8 * s + j + 1]};
            half2v h = __builtin_convertvector(af, half2v);
            h = __builtin_elementwise_max(h, h * lk);
            r[j] = h[0]; r[j + 1] = h[1];
        }
    }
    return r;
}

// issue order hint for one pipeline stage: the LDS reads of the block after next first (KS weight fragments + 4 for the
// bias tile), then N x (1 MFMA, 4 VALU) - the MFMAs of the block being multiplied against the packing of the block
// before it
template <int KS>
__device__ __forceinline__ void interleave_stage()
{
    __builtin_amdgcn_sched_group_barrier(0x100, KS + 4, 0);
#pragma unroll
    for (int k = 0; k < 2 * KS; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
    }
    __builtin_amdgcn_sched_barrier(0);          // a stage draws only on its own instructions
}

// Evaluate one MLP for the 64 envs of this wave.  in0 / in1: first-layer B fragments of env tiles 0-31 / 32-63.
// out[j], j < 8: output row j of the lane's OWN env.
//
// Register discipline: a layer's output is never held as a whole f32 tile set.  Each 32-row block (two
// accumulator tiles, one per env tile) is activated and packed to f16 as soon as its MFMAs are done, straight
// into the next layer's B fragments; weight fragments are fetched from LDS one row-block ahead.  That keeps the
// evaluation under the 256 architectural VGPRs, so the accumulators stay out of the AGPR half (every AGPR value
// a VALU instruction needs costs a v_accvgpr_read).
//
// Schedule: ONE software pipeline over all row-blocks of all layers - the MFMAs of a block are issued between the
// activation / packing VALU of the block before it, ACROSS layer boundaries too.  A layer's first block needs fragments
// 0..3 of its input for its first four k-steps and fragment 4 (5) only for the last; those last fragments come from the
// LAST row-block of the layer before.  So the first eight MFMAs of a layer run while that block is being packed, and the
// matrix pipe never waits for a whole layer to be packed (round 2; before, every layer boundary exposed an MFMA drain, a
// packing tail and the LDS latency of the next layer's weights: ~22 % of a layer's time).  The MFMA chain of every
// accumulator and the packing are unchanged: results are bit-identical to the layer-by-layer order.
// Input and output fragment sets alternate between two register arrays (X, Y) by layer, statically.
// KA = KS + 16 ACT: k-steps of 16 hidden features (5 or 6) and the hidden activation, as one template parameter
template <int KA>
__device__ __forceinline__ void mlp_eval(const uint4* W, const float* B, int n_hidden, half8 in0, half8 in1, _Float16 leak, float out[8])
{
    constexpr int KS = KA & 15, ACT = KA >> 4;
    constexpr int PACK_VALU = (ACT == ACT_TANH) ? 44 : 12;                   // VALU instructions of one act_pack (8 activations)
    const int lane = threadIdx.x & 63;
    const float16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    half8 X[KS][2], Y[KS][2], w[KS], wn[KS];
    float16v cb, cbn, p0, p1, q0, q1;
    const FragAddr<KS> fa(lane);
    int e0 = FragAddr<KS>::L0, bblk = 0;                                     // image entry / bias tile of the layer being entered
    int hleft = n_hidden - 1;                                                // hidden -> hidden layers still to come
    // ---- first layer: one k-step, three row-blocks; the weights of the NEXT layer's block 0 are fetched behind it
#pragma unroll
    for (int mo = 0; mo < 3; ++mo) w[mo] = fa.first(W, mo);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wn[ks] = fa.blk0(W, e0, hleft == 0, ks);
    cbn = ldbias(B, bblk, lane);
    {
        const float16v a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0], in0, zero, 0, 0, 0);
        const float16v a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[0], in1, zero, 0, 0, 0);
        const float16v c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1], in0, zero, 0, 0, 0);
        const float16v c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[1], in1, zero, 0, 0, 0);
        q0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[2], in0, zero, 0, 0, 0);
        q1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[2], in1, zero, 0, 0, 0);
        X[0][0] = act_pack<ACT>(a0, 0, leak); X[0][1] = act_pack<ACT>(a1, 0, leak);
        X[1][0] = act_pack<ACT>(a0, 1, leak); X[1][1] = act_pack<ACT>(a1, 1, leak);
        X[2][0] = act_pack<ACT>(c0, 0, leak); X[2][1] = act_pack<ACT>(c1, 0, leak);
        X[3][0] = act_pack<ACT>(c0, 1, leak); X[3][1] = act_pack<ACT>(c1, 1, leak);
    }
    __builtin_amdgcn_sched_barrier(0);
    // Entering a layer: wn / cbn hold its block-0 weights and bias tile; I[0..3] are packed; q0 / q1 are the accumulators of the
    // previous layer's last block (MFMAs issued), still owed to I[4] (and I[5]).
    // head(): block 0 of the layer - k-steps 0..3 beside the packing of q, then the remaining k-steps.
    auto head = [&](half8 (&I)[KS][2], bool more) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) w[ks] = wn[ks];
        cb = cbn;
        if (more) {                                                          // a hidden layer: its block 1 is fetched now
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wn[ks] = fa.hid(W, e0, 1, ks);
            cbn = ldbias(B, bblk + 1, lane);
        }
        p0 = cb; p1 = cb;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            p0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ks], I[ks][0], p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ks], I[ks][1], p1, 0, 0, 0);
        }
        I[4][0] = act_pack<ACT>(q0, 0, leak); I[4][1] = act_pack<ACT>(q1, 0, leak);
        if (5 < KS) { I[KS - 1][0] = act_pack<ACT>(q0, 1, leak); I[KS - 1][1] = act_pack<ACT>(q1, 1, leak); }
        __builtin_amdgcn_sched_group_barrier(0x100, KS + 4, 0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, (2 * (KS - 4) * PACK_VALU + 7) / 8, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 4; ks < KS; ++ks) {
            p0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ks], I[ks][0], p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ks], I[ks][1], p1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // hidden(): one hidden -> hidden layer, input fragments I, output fragments O (0..3 packed on return, 4.. owed by q)
    auto hidden = [&](half8 (&I)[KS][2], half8 (&O)[KS][2]) __attribute__((always_inline)) {
        head(I, true);
        const bool next_out = (--hleft == 0);
#pragma unroll
        for (int mo = 1; mo < 3; ++mo) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) w[ks] = wn[ks];
            cb = cbn;
            // block 2 is fetched during block 1; during block 2, block 0 of the layer after this one (hidden or output)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wn[ks] = (mo == 1) ? fa.hid(W, e0, 2, ks) : fa.blk0(W, e0 + FragAddr<KS>::LH, next_out, ks);
            cbn = ldbias(B, bblk + mo + 1, lane);
            float16v c0 = cb, c1 = cb;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ks], I[ks][0], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ks], I[ks][1], c1, 0, 0, 0);
            }
            O[2 * (mo - 1)][0] = act_pack<ACT>(p0, 0, leak); O[2 * (mo - 1)][1] = act_pack<ACT>(p1, 0, leak);
            O[2 * (mo - 1) + 1][0] = act_pack<ACT>(p0, 1, leak); O[2 * (mo - 1) + 1][1] = act_pack<ACT>(p1, 1, leak);
            __builtin_amdgcn_sched_group_barrier(0x100, KS + 4, 0);
#pragma unroll
            for (int k = 0; k < 2 * KS; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (4 * PACK_VALU + 2 * KS - 1) / (2 * KS), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            p0 = c0; p1 = c1;
        }
        q0 = p0; q1 = p1;
        e0 += FragAddr<KS>::LH;
        bblk += 3;
    };
    auto output = [&](half8 (&I)[KS][2]) __attribute__((always_inline)) {
        head(I, false);
        // rows 0..3 sit in registers 0..3 of lane half 0, rows 4..7 in registers 0..3 of lane half 1, for the 32 envs
        // of each tile: one permlane32 swap per register brings every env's 8 rows home to its own lane
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(p0[j]), __float_as_uint(p1[j]), false, false);
            out[j] = __uint_as_float(r[0]);
            out[4 + j] = __uint_as_float(r[1]);
        }
    };
    int l = 1;
    for (; l + 1 < n_hidden; l += 2) { hidden(X, Y); hidden(Y, X); }
    if (l < n_hidden) { hidden(X, Y); output(Y); }
    else output(X);
}

// ---------------------------------------------------------------------------------------------
// Actor AND critic of the same observation as one interleaved routine.
//
// Evaluated one after the other (mlp_eval twice) every row-block is "12 MFMAs, wait for them, ~48 VALU of
// leaky-relu + f16 packing" - the activation needs the tile the matrix pipe has just been given, so MFMA and VALU
// never overlap and a lone wave per SIMD pays for both in full (PMC: matrix pipe 27 % busy, VALU issue the rest).
// The two networks are independent, so their row-blocks are alternated P0 V0 P1 V1 P2 V2 per layer and software-
// pipelined across that sequence: while the matrix pipe works through the MFMAs of block i+1 the wave issues the
// activation/packing VALU of block i (sched_group_barrier: 1 MFMA, 4 VALU, ...).  Across a layer boundary the pending
// block is the critic's last one, which the actor's first block of the next layer does not depend on.
// Same MFMA chains and same packing as mlp_eval: results are bit-identical to two separate evaluations.
// ---------------------------------------------------------------------------------------------
struct Acc2 {
    float16v c0, c1;
};

template <int KS>
__device__ __forceinline__ Acc2 mfma_block(const half8 (&w)[KS], const half8 (&b)[KS][2], const float16v& cinit)
{
    Acc2 r = {cinit, cinit};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        r.c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ks], b[ks][0], r.c0, 0, 0, 0);
        r.c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[ks], b[ks][1], r.c1, 0, 0, 0);
    }
    return r;
}

template <int KA>
__device__ __forceinline__ void pack_block(const Acc2& a, half8 (&dst)[KA & 15][2], int mo, _Float16 leak)
{
    constexpr int KS = KA & 15, ACT = KA >> 4;
    dst[2 * mo][0] = act_pack<ACT>(a.c0, 0, leak); dst[2 * mo][1] = act_pack<ACT>(a.c1, 0, leak);
    if (2 * mo + 1 < KS) { dst[2 * mo + 1][0] = act_pack<ACT>(a.c0, 1, leak); dst[2 * mo + 1][1] = act_pack<ACT>(a.c1, 1, leak); }
}

template <int KA>
__device__ __forceinline__ void mlp_eval2(const uint4* Wp, const uint4* Wv, const float* Bp, const float* Bv, int n_hidden,
                                          half8 in0, half8 in1, _Float16 leak, float outp[8], float outv[8])
{
    constexpr int KS = KA & 15;
    const int lane = threadIdx.x & 63;
    const FragAddr<KS> fa(lane);
    half8 bP[KS][2], bV[KS][2], nP[KS][2], nV[KS][2], w[KS], wn[KS];
    Acc2 pend;                                   // the block whose activation/packing is still to be issued
    // ---- first layer: one k-step per block (2 MFMAs against ~48 VALU): VALU-bound whatever the order -------------
    {
        half8 wp[3], wv[3];
#pragma unroll
        for (int mo = 0; mo < 3; ++mo) { wp[mo] = fa.first(Wp, mo); wv[mo] = fa.first(Wv, mo); }
        const float16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        auto first = [&](const half8& wf) {
            Acc2 r;
            r.c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, in0, zero, 0, 0, 0);
            r.c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, in1, zero, 0, 0, 0);
            return r;
        };
        Acc2 a = first(wp[0]);
#pragma unroll
        for (int mo = 0; mo < 3; ++mo) {
            Acc2 v = first(wv[mo]);
            pack_block<KA>(a, bP, mo, leak);
            if (mo < 2) a = first(wp[mo + 1]);
            if (mo < 2) pack_block<KA>(v, bV, mo, leak); else pend = v;
        }
    }
    // bV[4] (and bV[5]) are still pending in `pend`
    int e0 = FragAddr<KS>::L0, bblk = 0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) w[ks] = fa.blk0(Wp, e0, n_hidden == 1, ks);
    float16v cb = ldbias(Bp, bblk, lane), cbn;
    for (int l = 1; l < n_hidden; ++l) {
        const bool next_out = (l + 1 == n_hidden);
        // stage P0: needs bP only; the critic's last block of the layer before is packed underneath it
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wn[ks] = fa.hid(Wv, e0, 0, ks);
        cbn = ldbias(Bv, bblk, lane);
        Acc2 cur = mfma_block<KS>(w, bP, cb);
        pack_block<KA>(pend, bV, 2, leak);
        interleave_stage<KS>();
        Acc2 prev = cur;
#pragma unroll
        for (int mo = 0; mo < 3; ++mo) {
            // stage V_mo under the packing of P_mo
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) w[ks] = wn[ks];
            cb = cbn;
            if (mo < 2) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) wn[ks] = fa.hid(Wp, e0, mo + 1, ks);
                cbn = ldbias(Bp, bblk + mo + 1, lane);
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) wn[ks] = fa.blk0(Wp, e0 + FragAddr<KS>::LH, next_out, ks);   // next layer's (or the output's) P0
                cbn = ldbias(Bp, bblk + 3, lane);
            }
            cur = mfma_block<KS>(w, bV, cb);
            pack_block<KA>(prev, nP, mo, leak);
            interleave_stage<KS>();
            prev = cur;
            if (mo < 2) {
                // stage P_{mo+1} under the packing of V_mo
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) w[ks] = wn[ks];
                cb = cbn;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) wn[ks] = fa.hid(Wv, e0, mo + 1, ks);
                cbn = ldbias(Bv, bblk + mo + 1, lane);
                cur = mfma_block<KS>(w, bP, cb);
                pack_block<KA>(prev, nV, mo, leak);
                interleave_stage<KS>();
                prev = cur;
            }
        }
        pend = prev;                              // V2 of this layer
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { bP[ks][0] = nP[ks][0]; bP[ks][1] = nP[ks][1]; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { bV[ks][0] = nV[ks][0]; bV[ks][1] = nV[ks][1]; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) w[ks] = wn[ks];
        cb = cbn;
        e0 += FragAddr<KS>::LH;
        bblk += 3;
    }
    // ---- output layer: one row-block per network --------------------------------------------------------------
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wn[ks] = fa.out(Wv, e0, ks);
    cbn = ldbias(Bv, bblk, lane);
    const Acc2 op = mfma_block<KS>(w, bP, cb);
    pack_block<KA>(pend, bV, 2, leak);
    interleave_stage<KS>();
    const Acc2 ov = mfma_block<KS>(wn, bV, cbn);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(op.c0[j]), __float_as_uint(op.c1[j]), false, false);
        outp[j] = __uint_as_float(r[0]);
        outp[4 + j] = __uint_as_float(r[1]);
        const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(ov.c0[j]), __float_as_uint(ov.c1[j]), false, false);
        outv[j] = __uint_as_float(q[0]);
        outv[4 + j] = __uint_as_float(q[1]);
    }
}

// first-layer B fragments from the per-lane observation row: input slot k < OD = obs[k], slot 15 = 1 (bias)
template <int OD>
__device__ __forceinline__ void obs_to_frags(const float o[9], half8& in0, half8& in1)
{
    // The f32 observation is what gets rounded to f16, whichever kernel produced it: without the barrier the compiler
    // folds the last FMA of an observation element computed in this kernel into v_fma_mixlo_f16 (ONE rounding), while a
    // row that came through memory or an LDS mailbox is rounded twice - about one env in a thousand then sees a
    // different f16 input, and the launch forms of the rollout stop being bit-identical.
    float x[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        x[k] = o[k];
        if (k < OD) asm volatile("" : "+v"(x[k]));
    }
    half8 P, Q;
#pragma unroll
    for (int k = 0; k < 8; ++k) P[k] = (_Float16)(k < OD ? x[k] : 0.0f);
#pragma unroll
    for (int k = 0; k < 8; ++k) Q[k] = (_Float16)((8 + k) < OD ? x[(8 + k) < 9 ? (8 + k) : 8] : 0.0f);
    Q[7] = (_Float16)1.0f;
    const uint4 p = __builtin_bit_cast(uint4, P), q = __builtin_bit_cast(uint4, Q);
    uint4 a, b;
    auto r0 = __builtin_amdgcn_permlane32_swap(p.x, q.x, false, false); a.x = r0[0]; b.x = r0[1];
    auto r1 = __builtin_amdgcn_permlane32_swap(p.y, q.y, false, false); a.y = r1[0]; b.y = r1[1];
    auto r2 = __builtin_amdgcn_permlane32_swap(p.z, q.z, false, false); a.z = r2[0]; b.z = r2[1];
    auto r3 = __builtin_amdgcn_permlane32_swap(p.w, q.w, false, false); a.w = r3[0]; b.w = r3[1];
    in0 = __builtin_bit_cast(half8, a);   // envs 0..31: lanes 0..31 carry slots 0..7, lanes 32..63 slots 8..15
    in1 = __builtin_bit_cast(half8, b);   // envs 32..63
}

// a = mu + std * xi (core.py:85) and its log-likelihood (gaussian_likelihood, core.py:42-46); c = consts of PolicyArgs in registers
template <int A>
struct PolicyConsts {
    float std[A], inv_std_eps[A], logp_const[A];
};

template <int A>
__device__ __forceinline__ PolicyConsts<A> load_policy_consts(const PolicyArgs& pa)
{
    PolicyConsts<A> c;
#pragma unroll
    for (int k = 0; k < A; ++k) { c.std[k] = pa.consts[k]; c.inv_std_eps[k] = pa.consts[8 + k]; c.logp_const[k] = pa.consts[16 + k]; }
    return c;
}

template <int A>
__device__ __forceinline__ float sample_action(const PolicyConsts<A>& c, const float mu[A], const float xi[A], float act[A])
{
    float logp = 0.0f;
#pragma unroll
    for (int k = 0; k < A; ++k) {
        act[k] = fmaf(c.std[k], xi[k], mu[k]);
        const float z = (act[k] - mu[k]) * c.inv_std_eps[k];
        logp += fmaf(-0.5f * z, z, c.logp_const[k]);
    }
    return logp;
}

// the log-likelihood of sample_action / mean_action from (a, mu) alone: same operations in the same order (a - mu is exactly
// zero for the mean action, and fma(-0, 0, c) = c), so a kernel can take it off the path between sampling and stepping
template <int A>
__device__ __forceinline__ float action_logp(const PolicyConsts<A>& c, const float mu[A], const float act[A])
{
    float logp = 0.0f;
#pragma unroll
    for (int k = 0; k < A; ++k) {
        const float z = (act[k] - mu[k]) * c.inv_std_eps[k];
        logp += fmaf(-0.5f * z, z, c.logp_const[k]);
    }
    return logp;
}

template <int A>
__device__ __forceinline__ float mean_action(const PolicyConsts<A>& c, const float mu[A], float act[A])
{
    float logp = 0.0f;
#pragma unroll
    for (int k = 0; k < A; ++k) { act[k] = mu[k]; logp += c.logp_const[k]; }
    return logp;
}

// this lane's vessel: its own per-env block (dpenv_set_vessel_params / domain randomisation), its class block from the table in HBM,
// or the single class of the kernel arguments - loaded once per launch; the T-step kernels' staging area is the register file
__device__ __forceinline__ Vessel launch_vessel_plain(const StepArgs& a, int il)
{
    if (a.env_tab) return vessel_from_env(a.env_tab, a.env_stride, il);
    return (a.n_classes > 1) ? vessel_from_table(a.class_tab, a.class_id[il]) : vessel_from_args(a.v0);
}
// ... and pinned in vector registers for the whole launch
__device__ __forceinline__ Vessel launch_vessel(const StepArgs& a, int il)
{
    Vessel ve = launch_vessel_plain(a, il);
    pin_vessel_in_vgprs(ve);
    return ve;
}

// LDS image: [2 or 4][nent] weight-fragment entries (16 B each, FragAddr) | [2][nblk][32] bias floats | wave-private row staging
template <int THREADS>
__device__ __forceinline__ void stage_weights_n(uint4* lds, const PolicyArgs& pa)
{
    const int total = 2 * pa.nent * (1 + pa.split);
    for (int k = threadIdx.x; k < total; k += THREADS) lds[k] = pa.frags[k];
    float* lb = (float*)(lds + total);
    for (int k = threadIdx.x; k < 2 * pa.nblk * 32; k += THREADS) lb[k] = pa.bias[k];
    __syncthreads();
}
__device__ __forceinline__ void stage_weights(uint4* lds, const PolicyArgs& pa) { stage_weights_n<PBLOCK>(lds, pa); }

__device__ __forceinline__ int policy_lds_io_offset_floats(const PolicyArgs& pa)      // after fragments and biases
{
    return 2 * pa.nent * 4 * (1 + pa.split) + 2 * pa.nblk * 32;
}

// wave-private AoS row I/O through LDS for a 64-env slice of a 256-thread workgroup
template <int W>
__device__ __forceinline__ void wave_store_rows(float* lds_w, void* dst, int64_t row0_elems, int64_t rem, const float* v, int lane,
                                                bool bf16 = false)
{
    lds_order<64>();
#pragma unroll
    for (int k = 0; k < W; ++k) lds_w[lane * W + k] = v[k];
    lds_order<64>();
    store_rows<W, 64>(dst, row0_elems, rem, bf16, lds_w, lane);
}

template <int W>
__device__ __forceinline__ void wave_rows_from_regs(float* lds_w, const float pre[W], float row[W], int lane)
{
    lds_order<64>();
#pragma unroll
    for (int j = 0; j < W; ++j) lds_w[j * 64 + lane] = pre[j];
    lds_order<64>();
#pragma unroll
    for (int k = 0; k < W; ++k) row[k] = lds_w[lane * W + k];
}


// a lane's own row, straight to memory (36 / 28-byte rows: slower stores than the LDS-transposed ones, same bytes) - where the LDS
// has no room for row staging
template <int W>
__device__ __forceinline__ void store_row_direct(void* dst, int64_t row, const float* v, bool bf16)
{
    if (bf16) {
        uint16_t* p = (uint16_t*)dst + row * W;
#pragma unroll
        for (int k = 0; k < W; ++k) p[k] = f2bf(v[k]);
    } else {
        float* p = (float*)dst + row * W;
#pragma unroll
        for (int k = 0; k < W; ++k) p[k] = v[k];
    }
}


// =============================================================================================
//  Split-f16 ("fp32-faithful") evaluation - DPENV_POLICY_F32.
//
//  The reference's networks are fp32 (core.py:29-33,80-107).  The matrix cores' fp32 MFMA runs at 1/16 of the f16 rate, so the
//  exact mode keeps the f16 instruction and splits both operands instead: W = Wh + Wl and x = xh + xl with
//  xh = f16(x), xl = f16(x - xh) (22 significand bits together), and  W x ~= Wh xh + Wh xl + Wl xh  - three MFMAs per
//  product, f32 accumulation, the dropped Wl xl term is 2^-22 relative.  Activations (leaky-relu / tanh) are evaluated in
//  f32 on the accumulator and split again.  Result: within ~1e-6 of an fp32 evaluation (tests: 1e-5 of the output scale),
//  at 3x the matrix work and ~3x the packing work of the f16 mode.  Same fragment layout as the f16 mode: the LDS image
//  carries a second set of fragments for Wl.
// =============================================================================================
// h -> (hi, lo) for two values: hi = f16(h) (v_cvt_pk_f16_f32, round to nearest even), lo = f16(h - f32(hi)); the difference is exact
__device__ __forceinline__ void split_pair(float h0, float h1, uint32_t& hi, uint32_t& lo)
{
    const float2v hv = {h0, h1};
    const half2v hh = __builtin_convertvector(hv, half2v);
    hi = __builtin_bit_cast(uint32_t, hh);
    const float2v lv = {h0 - (float)hh[0], h1 - (float)hh[1]};
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(lv, half2v));
}

// Leaky-relu + split of FOUR accumulator values: 4 x (mul, max), 2 x cvt_pk (hi), 4 x v_fma_mix_f32 (h - f32(hi half), the half
// widened inside the FMA), 2 x cvt_pk (lo) = 4 instructions per activation.  The part after the multiply is one asm statement
// because the compiler (a) puts a canonicalising v_max_f32(x, x) in front of every fmaxf under IEEE mode, (b) expands the difference
// into convert + subtract, and (c) pads each separate asm statement with an s_nop - together 7 per activation.
// HAZARD NOTE: x are MFMA results, and the hazard recogniser does not count an asm statement as a VALU reader of them (it would
// read the accumulator before the last MFMA of the chain has landed: seen as a 2e-5 error, the Wl xh term missing).  The
// multiply s x is therefore left to the compiler - it is a VALU read of every x in front of the asm, so the wait states are in place.
// No packed-fp32 instruction in here; a NaN propagates through max / fma_mix into the outputs as before.
__device__ __forceinline__ void leaky_split4(const float x[4], float leak, uint32_t H[2], uint32_t L[2])
{
    float t0, t1, t2, t3;
    const float m0 = x[0] * leak, m1 = x[1] * leak, m2 = x[2] * leak, m3 = x[3] * leak;
    asm("v_max_f32 %4, %8, %12\n\tv_max_f32 %5, %9, %13\n\tv_max_f32 %6, %10, %14\n\tv_max_f32 %7, %11, %15\n\t"
        "v_cvt_pk_f16_f32 %0, %4, %5\n\tv_cvt_pk_f16_f32 %1, %6, %7\n\t"
        "v_fma_mix_f32 %4, %0, %16, %4 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %5, %0, %16, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %6, %1, %16, %6 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %7, %1, %16, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_cvt_pk_f16_f32 %2, %4, %5\n\tv_cvt_pk_f16_f32 %3, %6, %7"
        : "=&v"(H[0]), "=&v"(H[1]), "=&v"(L[0]), "=&v"(L[1]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(m0), "v"(m1), "v"(m2), "v"(m3), "v"(-1.0f));
}

template <int ACT>
__device__ __forceinline__ void act_split(const float16v& acc, int s, float leak, half8& hi, half8& lo)
{
    uint32_t H[4], L[4];
    if (ACT == ACT_TANH) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            float h[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float e = __builtin_amdgcn_exp2f(acc[8 * s + j + q] * 2.8853900817779268f);      // exp(2x)
                h[q] = fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
            }
            split_pair(h[0], h[1], H[j >> 1], L[j >> 1]);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float x[4] = {acc[8 * s + 4 * q], acc[8 * s + 4 * q + 1], acc[8 * s + 4 * q + 2], acc[8 * s + 4 * q + 3]};
            leaky_split4(x, leak, H + 2 * q, L + 2 * q);
        }
    }
    const uint4 hq = {H[0], H[1], H[2], H[3]}, lq = {L[0], L[1], L[2], L[3]};
    hi = __builtin_bit_cast(half8, hq);
    lo = __builtin_bit_cast(half8, lq);
}

__device__ __forceinline__ float16v mfma3(const half8& wh, const half8& wl, const half8& bh, const half8& bl, float16v c)
{
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, c, 0, 0, 0);
    return c;
}

struct SplitIn {
    half8 h0, h1, l0, l1;      // first-layer B fragments of env tiles 0-31 / 32-63, high and low parts
};

template <int OD>
__device__ __forceinline__ void obs_to_frags_x(const float o[9], SplitIn& in)
{
    uint32_t H[8], L[8];
    // as in obs_to_frags: what is split is the MATERIALISED f32 observation, whichever kernel (or wave) produced it - otherwise the
    // compiler may fold an observation's last FMA into the f16 conversion in one launch form and not in another
    float x[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        x[k] = o[k];
        if (k < OD) asm volatile("" : "+v"(x[k]));
    }
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        const float x0 = (k < OD) ? x[k < 9 ? k : 8] : 0.0f;
        const float x1 = (k + 1 < OD) ? x[k + 1 < 9 ? k + 1 : 8] : (k + 1 == 15 ? 1.0f : 0.0f);     // slot 15: the bias input (its low part is 0)
        split_pair(x0, x1, H[k >> 1], L[k >> 1]);
    }
    const uint4 ph = {H[0], H[1], H[2], H[3]}, qh = {H[4], H[5], H[6], H[7]}, pl = {L[0], L[1], L[2], L[3]}, ql = {L[4], L[5], L[6], L[7]};
    const half8 Ph = __builtin_bit_cast(half8, ph), Qh = __builtin_bit_cast(half8, qh), Pl = __builtin_bit_cast(half8, pl), Ql = __builtin_bit_cast(half8, ql);
    auto swap4 = [](const half8& P, const half8& Q, half8& a, half8& b) {
        const uint4 p = __builtin_bit_cast(uint4, P), q = __builtin_bit_cast(uint4, Q);
        uint4 x, y;
        auto r0 = __builtin_amdgcn_permlane32_swap(p.x, q.x, false, false); x.x = r0[0]; y.x = r0[1];
        auto r1 = __builtin_amdgcn_permlane32_swap(p.y, q.y, false, false); x.y = r1[0]; y.y = r1[1];
        auto r2 = __builtin_amdgcn_permlane32_swap(p.z, q.z, false, false); x.z = r2[0]; y.z = r2[1];
        auto r3 = __builtin_amdgcn_permlane32_swap(p.w, q.w, false, false); x.w = r3[0]; y.w = r3[1];
        a = __builtin_bit_cast(half8, x); b = __builtin_bit_cast(half8, y);
    };
    swap4(Ph, Qh, in.h0, in.h1);
    swap4(Pl, Ql, in.l0, in.l1);
}

// The two halves of leaky_split4 as separate statements, so that an MFMA can be issued between them (mlp_eval_x): same
// instructions, same results.  `m` = leak * x comes from the compiler (the VALU read of the accumulators that carries the wait states).
__device__ __forceinline__ void leaky_split4_hi(const float x[4], const float m[4], float t[4], uint32_t H[2])
{
    asm volatile("v_max_f32 %2, %6, %10\n\tv_max_f32 %3, %7, %11\n\tv_max_f32 %4, %8, %12\n\tv_max_f32 %5, %9, %13\n\t"
        "v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5"
        : "=&v"(H[0]), "=&v"(H[1]), "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]));
}
__device__ __forceinline__ void leaky_split4_lo(float t[4], const uint32_t H[2], uint32_t L[2])
{
    asm volatile("v_fma_mix_f32 %2, %6, %8, %2 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %3, %6, %8, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %4, %7, %8, %4 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %5, %7, %8, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5"
        : "=&v"(L[0]), "=&v"(L[1]), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3])
        : "v"(H[0]), "v"(H[1]), "v"(-1.0f));
}

// HAZARD NOTE 2 (found when the evaluation was interleaved, round 2): an asm statement is invisible to the MFMA hazard recogniser
// as a WRITER too.  With 80-wide layers only registers 0..7 of a layer's last accumulator tile are ever read (rows 80..95 are padding),
// so the register allocator considers registers 8..15 dead as soon as the tile's MFMAs are ISSUED and hands them to the next asm
// statement as temporaries - whose v_max / v_fma_mix results the still-running MFMA then overwrites (seen: actor mean of env tile 0
// wrong by 0.1-0.4 in one inlined copy of the evaluation).  keep_alive() is an empty asm that reads whole tiles: placed after the
// units that consume a tile (and after the MFMAs that read a bias tile as SrcC) it keeps every register of the tile allocated until
// the matrix pipe is done with it.  tests/test_abi_cpu.py scans the ISA of this unit for asm-written registers inside the
// destination of a recent MFMA.
__device__ __forceinline__ void keep_alive(const float16v& a) { asm volatile("" : : "v"(a)); }

// One split-f16 MLP for the 64 envs of a wave.  A wave issues in order and the matrix pipe takes one MFMA at a time, so MFMAs and
// the activation / split VALU only overlap if they ALTERNATE in the instruction stream.  The evaluation is therefore written as a
// sequence of slots - one k-step of one env tile: MFMA (Wh xh), 4 VALU, MFMA (Wh xl), 6 VALU, MFMA (Wl xh), 6 VALU - where the VALU
// of a slot is one unit (four activations) of the row-block computed one stage earlier; `sched_barrier`s pin the order.  Like
// mlp_eval it is one pipeline across layers: a layer's block 0 runs its first four k-steps beside the split of the previous layer's
// last block.  Before (all MFMAs of a block, then all VALU of a block): matrix pipe and VALU took turns, 15.5 us per closed-loop
// step; the MFMA chain of every accumulator and every activation's arithmetic are unchanged, so results are bit-identical.
template <int KA>
__device__ __forceinline__ void mlp_eval_x(const uint4* Wh, const uint4* Wl, const float* B, int n_hidden, const SplitIn& in, float leak,
                                           float out[8])
{
    constexpr int KS = KA & 15, ACT = KA >> 4;
    const int lane = threadIdx.x & 63;
    const float16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    struct Frag { uint32_t h[4], l[4]; };                 // one B fragment: high and low f16 parts, 8 features each
    Frag X[KS][2], Y[KS][2];
    float16v p0, p1, q0, q1;
    // the same bias tiles through an offset the compiler cannot see through: two reads stay two reads (see head()).  The OFFSET is
    // laundered, not the pointer: an asm-laundered pointer loses its LDS address space and every read through it becomes a
    // flat_load (seen as 28 VMEM reads per evaluation in the PMC counters of the first round-3 build)
    int b2_off = 0;
    asm volatile("" : "+v"(b2_off));
    const float* B2 = B + b2_off;
    auto FH = [](const Frag& f) { const uint4 q = {f.h[0], f.h[1], f.h[2], f.h[3]}; return __builtin_bit_cast(half8, q); };
    auto FL = [](const Frag& f) { const uint4 q = {f.l[0], f.l[1], f.l[2], f.l[3]}; return __builtin_bit_cast(half8, q); };
    // slot: three MFMAs of accumulator c with weights (wh, wl) and input fragment b; between them, unit `qq` (four activations:
    // registers 8 s + 4 qq .. + 3) of the finished accumulator `acc` goes into words 2 qq, 2 qq + 1 of fragment d
    auto slot = [&](float16v& c, const half8& wh, const half8& wl, const Frag& b, bool has, const float16v& acc, int sreg, int qq,
                    Frag& d) __attribute__((always_inline)) {
        float x[4], m[4], t[4];
        // the empty asm after each MFMA pins it in program order (an MFMA has no side effects: without it the compiler is free to
        // sink a whole chain to its first use, past every sched_barrier)
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, FH(b), c, 0, 0, 0);
        asm volatile("" : "+v"(c));
        if (has) {
            if (ACT == ACT_TANH) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float e = __builtin_amdgcn_exp2f(acc[8 * sreg + 4 * qq + k] * 2.8853900817779268f);      // exp(2x)
                    x[k] = fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) { x[k] = acc[8 * sreg + 4 * qq + k]; m[k] = x[k] * leak; }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, FL(b), c, 0, 0, 0);
        asm volatile("" : "+v"(c));
        if (has) {
            if (ACT == ACT_TANH) { split_pair(x[0], x[1], d.h[2 * qq], d.l[2 * qq]); }
            else leaky_split4_hi(x, m, t, &d.h[2 * qq]);
        }
        __builtin_amdgcn_sched_barrier(0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, FH(b), c, 0, 0, 0);
        asm volatile("" : "+v"(c));
        if (has) {
            if (ACT == ACT_TANH) { split_pair(x[2], x[3], d.h[2 * qq + 1], d.l[2 * qq + 1]); }
            else leaky_split4_lo(t, &d.h[2 * qq], &d.l[2 * qq]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // a unit without an MFMA to hide behind (first layer only)
    auto bare_unit = [&](const float16v& acc, int sreg, int qq, Frag& d) __attribute__((always_inline)) {
        float x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = acc[8 * sreg + 4 * qq + k];
        if (ACT == ACT_TANH) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float e = __builtin_amdgcn_exp2f(x[k] * 2.8853900817779268f);
                x[k] = fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
            }
            split_pair(x[0], x[1], d.h[2 * qq], d.l[2 * qq]);
            split_pair(x[2], x[3], d.h[2 * qq + 1], d.l[2 * qq + 1]);
        } else {
            leaky_split4(x, leak, &d.h[2 * qq], &d.l[2 * qq]);
        }
    };
    const FragAddr<KS> fa(lane);
    int e0 = FragAddr<KS>::L0, bblk = 0;
    // What a row-block needs before its first MFMA - its two bias tiles (the accumulators start from them) and its first weight
    // fragments, twelve LDS reads - is fetched during the LAST k-step of the block before it (round 3).  Those slots carry no
    // activation work, and the accumulators of the block before that one have just been used up, so the fetch costs no register
    // the evaluation does not already own; issued at the block's own start it left the matrix pipe idle for the LDS latency ten
    // times per evaluation.
    float16v nb0, nb1;
    half8 nw_h, nw_l;
    auto fetch_blk0 = [&](int e, int blk, bool is_out) __attribute__((always_inline)) {
        nb0 = ldbias(B, blk, lane); nb1 = ldbias(B2, blk, lane);
        nw_h = fa.blk0(Wh, e, is_out, 0); nw_l = fa.blk0(Wl, e, is_out, 0);
    };
    auto fetch_hid = [&](int e, int blk, int mo) __attribute__((always_inline)) {
        nb0 = ldbias(B, blk + mo, lane); nb1 = ldbias(B2, blk + mo, lane);
        nw_h = fa.hid(Wh, e, mo, 0); nw_l = fa.hid(Wl, e, mo, 0);
    };
    // ---- first layer (one k-step): blocks 0 and 1, then block 2 beside the split of block 0; the split of block 1 is exposed
    {
        Frag I0, I1;
        const uint4 h0 = __builtin_bit_cast(uint4, in.h0), l0 = __builtin_bit_cast(uint4, in.l0);
        const uint4 h1 = __builtin_bit_cast(uint4, in.h1), l1 = __builtin_bit_cast(uint4, in.l1);
        I0.h[0] = h0.x; I0.h[1] = h0.y; I0.h[2] = h0.z; I0.h[3] = h0.w; I0.l[0] = l0.x; I0.l[1] = l0.y; I0.l[2] = l0.z; I0.l[3] = l0.w;
        I1.h[0] = h1.x; I1.h[1] = h1.y; I1.h[2] = h1.z; I1.h[3] = h1.w; I1.l[0] = l1.x; I1.l[1] = l1.y; I1.l[2] = l1.z; I1.l[3] = l1.w;
        const half8 w0h = fa.first(Wh, 0), w0l = fa.first(Wl, 0), w1h = fa.first(Wh, 1), w1l = fa.first(Wl, 1);
        const half8 w2h = fa.first(Wh, 2), w2l = fa.first(Wl, 2);
        float16v a0 = zero, a1 = zero, c0 = zero, c1 = zero;
        Frag none;
        slot(a0, w0h, w0l, I0, false, zero, 0, 0, none);
        slot(a1, w0h, w0l, I1, false, zero, 0, 0, none);
        slot(c0, w1h, w1l, I0, true, a0, 0, 0, X[0][0]);
        slot(c1, w1h, w1l, I1, true, a0, 0, 1, X[0][0]);
        q0 = zero; q1 = zero;
        slot(q0, w2h, w2l, I0, true, a0, 1, 0, X[1][0]);
        slot(q1, w2h, w2l, I1, true, a0, 1, 1, X[1][0]);
        fetch_blk0(e0, bblk, n_hidden == 1);                 // the head of the next layer, under the exposed units below
        bare_unit(a1, 0, 0, X[0][1]); bare_unit(a1, 0, 1, X[0][1]);
        bare_unit(a1, 1, 0, X[1][1]); bare_unit(a1, 1, 1, X[1][1]);
#pragma unroll
        for (int sreg = 0; sreg < 2; ++sreg) {
            bare_unit(c0, sreg, 0, X[2 + sreg][0]); bare_unit(c0, sreg, 1, X[2 + sreg][0]);
            bare_unit(c1, sreg, 0, X[2 + sreg][1]); bare_unit(c1, sreg, 1, X[2 + sreg][1]);
        }
        keep_alive(a0); keep_alive(a1); keep_alive(c0); keep_alive(c1);
    }
    __builtin_amdgcn_sched_barrier(0);
    // head: block 0 of the layer being entered.  I[0..3] are complete; q0 / q1 (the last block of the layer before) still owe I[4]
    // (and I[5]): their units ride on the first k-steps.  Weights: one k-step (wh, wl) is fetched while the one before is multiplied.
    auto head = [&](Frag (&I)[KS][2], bool is_out) __attribute__((always_inline)) {
        // each accumulator starts from its OWN read of the bias tile (a 64-byte LDS broadcast per lane half): a shared copy would
        // be 16 more live registers for the whole block and 32 v_mov - this evaluation runs at the edge of the register file
        p0 = nb0;
        p1 = nb1;
        half8 wh = nw_h, wl = nw_l;
        constexpr int NU = 4 * (KS - 4);                                    // pending units: 4 (KS = 5) or 8 (KS = 6)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            half8 nwh = wh, nwl = wl;
            if (ks + 1 < KS) { nwh = fa.blk0(Wh, e0, is_out, ks + 1); nwl = fa.blk0(Wl, e0, is_out, ks + 1); }
            // q0 / q1 have been read for the last time one k-step ago and the matrix pipe finished with them long before
            if (ks == NU / 2) { keep_alive(q0); keep_alive(q1); }
            if (ks == KS - 1 && !is_out) fetch_hid(e0, bblk, 1);
            const int u0 = 2 * ks, u1 = 2 * ks + 1;                          // unit u: tile u & 1, quarter (u >> 1) & 1, half u >> 2
            slot(p0, wh, wl, I[ks][0], ks < 4 && u0 < NU, (u0 & 1) ? q1 : q0, u0 >> 2, (u0 >> 1) & 1, I[4 + (u0 >> 2) < KS ? 4 + (u0 >> 2) : 4][u0 & 1]);
            slot(p1, wh, wl, I[ks][1], ks < 4 && u1 < NU, (u1 & 1) ? q1 : q0, u1 >> 2, (u1 >> 1) & 1, I[4 + (u1 >> 2) < KS ? 4 + (u1 >> 2) : 4][u1 & 1]);
            wh = nwh; wl = nwl;
        }
    };
    auto hidden = [&](Frag (&I)[KS][2], Frag (&O)[KS][2], bool next_is_out) __attribute__((always_inline)) {
        head(I, false);
#pragma unroll
        for (int mo = 1; mo < 3; ++mo) {
            float16v c0 = nb0, c1 = nb1;
            half8 wh = nw_h, wl = nw_l;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                half8 nwh = wh, nwl = wl;
                if (ks + 1 < KS) { nwh = fa.hid(Wh, e0, mo, ks + 1); nwl = fa.hid(Wl, e0, mo, ks + 1); }
                if (ks == 4) { keep_alive(p0); keep_alive(p1); }             // their eight units rode on k-steps 0..3
                if (ks == KS - 1) {
                    if (mo == 1) fetch_hid(e0, bblk, 2);
                    else fetch_blk0(e0 + FragAddr<KS>::LH, bblk + 3, next_is_out);
                }
                // the eight units of block mo - 1 (p0 / p1) -> O[2 (mo - 1)], O[2 (mo - 1) + 1]
                const int u0 = 2 * ks, u1 = 2 * ks + 1;
                slot(c0, wh, wl, I[ks][0], u0 < 8, (u0 & 1) ? p1 : p0, (u0 >> 2) & 1, (u0 >> 1) & 1, O[2 * (mo - 1) + ((u0 >> 2) & 1)][u0 & 1]);
                slot(c1, wh, wl, I[ks][1], u1 < 8, (u1 & 1) ? p1 : p0, (u1 >> 2) & 1, (u1 >> 1) & 1, O[2 * (mo - 1) + ((u1 >> 2) & 1)][u1 & 1]);
                wh = nwh; wl = nwl;
            }
            p0 = c0; p1 = c1;
        }
        q0 = p0; q1 = p1;
        e0 += FragAddr<KS>::LH;
        bblk += 3;
    };
    auto output = [&](Frag (&I)[KS][2]) __attribute__((always_inline)) {
        head(I, true);
        keep_alive(p0); keep_alive(p1);                  // only rows 0..7 are read below; the next asm statement may be close
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(p0[j]), __float_as_uint(p1[j]), false, false);
            out[j] = __uint_as_float(r[0]);
            out[4 + j] = __uint_as_float(r[1]);
        }
    };
    int l = 1;
    for (; l + 1 < n_hidden; l += 2) { hidden(X, Y, false); hidden(Y, X, !(l + 2 < n_hidden)); }
    if (l < n_hidden) { hidden(X, Y, true); output(Y); }
    else output(X);
}

}  // namespace dpenv

#endif
