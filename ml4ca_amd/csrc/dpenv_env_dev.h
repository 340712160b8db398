// dpenv_env_dev.h - device-side building blocks shared by the kernels of libdpenv.so
// (dpenv_kernels.hip: step / rollout / reset ...; dpenv_policy.hip: policy-in-the-loop rollout).
// Everything here is __device__ __forceinline__ code in namespace dpenv; see dpenv_kernels.hip for the
// reference citations of the path as a whole.
#ifndef DPENV_ENV_DEV_H
#define DPENV_ENV_DEV_H
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "dpenv_dev.h"

namespace dpenv {

constexpr float kPi = 3.14159265358979323846f;

// ---- Philox4x32-10 (Salmon et al. SC'11): counter-based RNG of the reset sampler -------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one v_mad_u64_u32 per 32 x 32 -> 64 product (the 32-bit integer multiplier is a quarter-rate unit shared by the
        // SIMD's waves: separate v_mul_hi_u32 + v_mul_lo_u32 cost 14.6 ns of it per round, the 64-bit form 11.5 ns)
        const uint64_t p0 = (uint64_t)c0 * 0xD2511F53ull, p1 = (uint64_t)c2 * 0xCD9E8D57ull;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u01_sym(uint32_t w)
{
    // 24-bit uniform mapped to [-1, 1): every step is exact in fp32
    return 2.0f * (float)(w >> 8) * (1.0f / 16777216.0f) - 1.0f;
}

template <int MODE> struct ModeTraits;
template <> struct ModeTraits<MODE_FULL> { static constexpr int A = 6; };
template <> struct ModeTraits<MODE_SIMPLE> { static constexpr int A = 3; };
template <> struct ModeTraits<MODE_LIMITED> { static constexpr int A = 5; };
template <> struct ModeTraits<MODE_FINAL_WRAP> { static constexpr int A = 5; };
template <> struct ModeTraits<MODE_FINAL_CONT> { static constexpr int A = 7; };

// state-space bounds, ENV:26,337,361,386 (intended per-variant values)
template <int MODE> __device__ __forceinline__ void ss_bounds(float b[6])
{
    b[0] = 8.0f; b[1] = 8.0f; b[2] = kPi * 0.5f; b[3] = 1.4f; b[4] = 0.30f; b[5] = 0.52f;
    if (MODE == MODE_SIMPLE) { b[3] = 1.75f; b[5] = 0.51f; }
    if (MODE == MODE_LIMITED || MODE == MODE_FINAL_WRAP || MODE == MODE_FINAL_CONT) b[2] = 45.0f * kPi / 180.0f;
}

// default azimuth commands, ENV:58,341-346,366-371,394-399
template <int MODE> __device__ __forceinline__ void default_angles(float& a_bow, float& a_port, float& a_star)
{
    a_bow = 0.0f; a_port = 0.0f; a_star = 0.0f;
    if (MODE == MODE_SIMPLE) { a_bow = kPi * 0.5f; a_port = -3.0f * kPi * 0.25f; a_star = 3.0f * kPi * 0.25f; }
    if (MODE == MODE_LIMITED || MODE == MODE_FINAL_WRAP || MODE == MODE_FINAL_CONT) a_bow = kPi * 0.5f;
}

// mathematics.py:14-17 wrap_angle in the cancellation-free form x - 2ref*floor((x+ref)/(2ref));
// deg=true is what errorFrame.py:29,31 actually calls (quirk Q1: degrees constant on radians).
__device__ __forceinline__ float wrap_angle(float x, bool deg)
{
    const float ref = deg ? 180.0f : kPi;
    const float inv = deg ? (1.0f / 360.0f) : (0.5f / kPi);
    const float k = floorf((x + ref) * inv);
    return fmaf(-k, 2.0f * ref, x);
}

__device__ __forceinline__ float clipf(float v, float b) { return fminf(fmaxf(v, -b), b); }

// ---- lean transcendental code (validated in tools/lean_math_check.py against float64 libm) -------------
// sincos: 3-constant Cody-Waite reduction by pi/2 + degree-7/6 kernels (Cephes single-precision
// coefficients); max abs error 9.2e-8 for |x| <= 3e4 (a heading of 4775 turns).  Larger arguments - where a
// float heading has lost all sub-radian meaning anyway - are first folded by 2 pi in plain fp32 so that the
// result stays a valid (if inaccurate) rotation; no library call, so the kernels stay call-free.
__device__ __forceinline__ void sincos_lean(float x, float& s, float& c)
{
    if (__builtin_expect(fabsf(x) > 30000.0f, 0)) x = fmaf(-rintf(x * (0.5f / kPi)), 2.0f * kPi, x);
    const float kf = rintf(x * 0.6366197723675814f);
    float r = fmaf(kf, -1.5703125f, x);
    r = fmaf(kf, -4.837512969970703125e-4f, r);
    r = fmaf(kf, -7.54978995489188216e-8f, r);
    const float r2 = r * r;
    float p = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    p = fmaf(r2, p, -1.6666654611e-1f);
    const float sv = fmaf(r * r2, p, r);
    float q = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    q = fmaf(r2, q, 4.166664568298827e-2f);
    const float cv = fmaf(r2 * r2, q, fmaf(r2, -0.5f, 1.0f));
    const int n = (int)kf;
    const float so = (n & 1) ? cv : sv;
    const float co = (n & 1) ? sv : cv;
    s = (n & 2) ? -so : so;
    c = ((n + 1) & 2) ? -co : co;
}

// atan2: a = min/max in [0,1], odd polynomial of degree 17 in a (coefficients fitted in
// tools/lean_math_check.py: max abs error 2.6e-7, 2 ulp), octant fix-up, sign of y (signed zeros kept:
// atan2(-0, -1) = -pi like numpy's arctan2, which customEnv.py:231-232 calls).
__device__ __forceinline__ float atan2_lean(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float a = (mx == 0.0f) ? 0.0f : mn * __builtin_amdgcn_rcpf(mx);
    const float s = a * a;
    float r = 2.6222818114e-03f;
    r = fmaf(r, s, -1.5132710144e-02f);
    r = fmaf(r, s, 4.1122186800e-02f);
    r = fmaf(r, s, -7.3667379326e-02f);
    r = fmaf(r, s, 1.0573949200e-01f);
    r = fmaf(r, s, -1.4185980238e-01f);
    r = fmaf(r, s, 1.9990397277e-01f);
    r = fmaf(r, s, -3.3332987079e-01f);
    r = fmaf(r * s, a, a);
    r = (ay > ax) ? (kPi * 0.5f - r) : r;
    r = (__float_as_uint(x) >> 31) ? (kPi - r) : r;
    return copysignf(r, y);
}

__device__ __forceinline__ float sqrt_hw(float x) { return __builtin_amdgcn_sqrtf(x); }   // v_sqrt_f32, <= 1 ulp

struct Vessel {
    float m11, m22, m23, inv11, i22, i23, i33;
    float Xu, Xuu, Yv, Yvv, Yr, Nv, Nr, Nrr, Nuv, Yur;
    float Kf[3], Kr[3], lx[3], ly[3];
};

__device__ __forceinline__ Vessel vessel_from_args(const VesselDev& d)
{
    Vessel v;
    v.m11 = d.p[VD_M11]; v.m22 = d.p[VD_M22]; v.m23 = d.p[VD_M23];
    v.inv11 = d.p[VD_INV11]; v.i22 = d.p[VD_I22]; v.i23 = d.p[VD_I23]; v.i33 = d.p[VD_I33];
    v.Xu = d.p[VD_XU]; v.Xuu = d.p[VD_XUU]; v.Yv = d.p[VD_YV]; v.Yvv = d.p[VD_YVV];
    v.Yr = d.p[VD_YR]; v.Nv = d.p[VD_NV]; v.Nr = d.p[VD_NR]; v.Nrr = d.p[VD_NRR];
    v.Nuv = d.p[VD_NUV]; v.Yur = d.p[VD_YUR];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v.Kf[i] = d.p[VD_KF + i]; v.Kr[i] = d.p[VD_KR + i]; v.lx[i] = d.p[VD_LX + i]; v.ly[i] = d.p[VD_LY + i];
    }
    return v;
}

// Keep the (wave-uniform) parameter block in VGPRs.  gfx9 VALU instructions read at most one SGPR, so
// two-parameter FMAs need a v_mov each time they execute, and 31 live SGPR parameters push the long
// rollout loop into SGPR->VGPR-lane spills (v_writelane/v_readlane).  An empty asm with a "+v" constraint
// pins each value in a vector register once.
__device__ __forceinline__ void pin_vgpr(float& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void pin_vessel_in_vgprs(Vessel& v)
{
    pin_vgpr(v.m11); pin_vgpr(v.m22); pin_vgpr(v.m23); pin_vgpr(v.inv11); pin_vgpr(v.i22); pin_vgpr(v.i23); pin_vgpr(v.i33);
    pin_vgpr(v.Xu); pin_vgpr(v.Xuu); pin_vgpr(v.Yv); pin_vgpr(v.Yvv); pin_vgpr(v.Yr); pin_vgpr(v.Nv); pin_vgpr(v.Nr);
    pin_vgpr(v.Nrr); pin_vgpr(v.Nuv); pin_vgpr(v.Yur);
#pragma unroll
    for (int i = 0; i < 3; ++i) { pin_vgpr(v.Kf[i]); pin_vgpr(v.Kr[i]); pin_vgpr(v.lx[i]); pin_vgpr(v.ly[i]); }
}

// LDS image of the class table is [param][class]: for a fixed parameter, lanes of different
// classes hit different banks (n_classes <= 32 distinct banks) and lanes of one class broadcast.
__device__ __forceinline__ Vessel vessel_from_lds(const float* tab, int ncls, int cls)
{
    Vessel v;
    v.m11 = tab[VD_M11 * ncls + cls]; v.m22 = tab[VD_M22 * ncls + cls]; v.m23 = tab[VD_M23 * ncls + cls];
    v.inv11 = tab[VD_INV11 * ncls + cls]; v.i22 = tab[VD_I22 * ncls + cls]; v.i23 = tab[VD_I23 * ncls + cls];
    v.i33 = tab[VD_I33 * ncls + cls];
    v.Xu = tab[VD_XU * ncls + cls]; v.Xuu = tab[VD_XUU * ncls + cls]; v.Yv = tab[VD_YV * ncls + cls];
    v.Yvv = tab[VD_YVV * ncls + cls]; v.Yr = tab[VD_YR * ncls + cls]; v.Nv = tab[VD_NV * ncls + cls];
    v.Nr = tab[VD_NR * ncls + cls]; v.Nrr = tab[VD_NRR * ncls + cls];
    v.Nuv = tab[VD_NUV * ncls + cls]; v.Yur = tab[VD_YUR * ncls + cls];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v.Kf[i] = tab[(VD_KF + i) * ncls + cls]; v.Kr[i] = tab[(VD_KR + i) * ncls + cls];
        v.lx[i] = tab[(VD_LX + i) * ncls + cls]; v.ly[i] = tab[(VD_LY + i) * ncls + cls];
    }
    return v;
}

// T-step kernels load a lane's class block ONCE per launch straight from the [class][param] table in HBM (29 dwords per
// lane per launch; the table is a few KiB and L2-resident): for them the register file is the staging area.
__device__ __forceinline__ Vessel vessel_from_table(const float* tab, int cls)
{
    VesselDev d;
#pragma unroll
    for (int k = 0; k < VD_COUNT; ++k) d.p[k] = tab[cls * VD_COUNT + k];
    return vessel_from_args(d);
}

// ---- per-env parameter blocks (dpenv_dev.h: float4 ET[ENV_GROUPS][stride]) ------------------------------------------------
// A lane's own block straight into registers: eight coalesced 16-byte loads (1 KiB per wave-instruction), no LDS.
__device__ __forceinline__ Vessel vessel_from_env(const float4* tab, int stride, int il)
{
    VesselDev d;
    float4 q[ENV_GROUPS];
#pragma unroll
    for (int g = 0; g < ENV_GROUPS; ++g) q[g] = tab[(int64_t)g * stride + il];
#pragma unroll
    for (int k = 0; k < VD_COUNT; ++k) {
        const float4& v = q[k >> 2];
        d.p[k] = (k & 3) == 0 ? v.x : (k & 3) == 1 ? v.y : (k & 3) == 2 ? v.z : v.w;
    }
    return vessel_from_args(d);
}

// The same block from an LDS image [group][lane] of float4 (ds_read_b128: a lane's 16 bytes, conflict-free)
__device__ __forceinline__ Vessel vessel_from_env_lds(const float4* lds, int lane)
{
    VesselDev d;
#pragma unroll
    for (int g = 0; g < ENV_GROUPS; ++g) {
        const float4 v = lds[g * 64 + lane];
        if (4 * g + 0 < VD_COUNT) d.p[4 * g + 0] = v.x;
        if (4 * g + 1 < VD_COUNT) d.p[4 * g + 1] = v.y;
        if (4 * g + 2 < VD_COUNT) d.p[4 * g + 2] = v.z;
        if (4 * g + 3 < VD_COUNT) d.p[4 * g + 3] = v.w;
    }
    return vessel_from_args(d);
}

// raw public parameters (DPENV_P_* order, include/dpenv.h) -> one per-env block: the SAME float operations in the same order as the
// host's derive_vessel (dpenv_api.hip), so that an env given its class's parameters integrates with the
// class path's constants bit for bit.  fp32 division is IEEE-correct here (no -ffast-math; hipcc's default correctly rounded divide).
// A block that is not a vessel (mass matrix not positive definite, non-finite entries) becomes NaN: the env then reports
// DPENV_DONE_FAULT at its first step instead of integrating garbage.
__device__ __forceinline__ void derive_env_block(const float raw[RAND_NPARAM], float d[ENV_BLOCK_FLOATS])
{
    const float m11 = raw[0], m22 = raw[1], m23 = raw[2], m33 = raw[3];
    const float fdet = m22 * m33 - m23 * m23;
    d[VD_M11] = m11; d[VD_M22] = m22; d[VD_M23] = m23;
    d[VD_INV11] = 1.0f / m11;
    d[VD_I22] = m33 / fdet; d[VD_I23] = -m23 / fdet; d[VD_I33] = m22 / fdet;
    d[VD_XU] = raw[4]; d[VD_XUU] = raw[5]; d[VD_YV] = raw[6]; d[VD_YVV] = raw[7]; d[VD_YR] = raw[8];
    d[VD_NV] = raw[9]; d[VD_NR] = raw[10]; d[VD_NRR] = raw[11];
    d[VD_NUV] = raw[24]; d[VD_YUR] = raw[25];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        d[VD_KF + k] = raw[12 + k]; d[VD_KR + k] = raw[15 + k]; d[VD_LX + k] = raw[18 + k]; d[VD_LY + k] = raw[21 + k];
    }
    d[VD_M33] = m33;
#pragma unroll
    for (int k = VD_M33 + 1; k < ENV_BLOCK_FLOATS; ++k) d[k] = 0.0f;
    float chk = 0.0f;
#pragma unroll
    for (int k = 0; k < RAND_NPARAM; ++k) chk += raw[k];
    bool ok = (m11 > 0.0f) && (m22 > 0.0f) && (fdet > 0.0f) && (fabsf(chk) <= 3.0e38f);
#pragma unroll
    for (int k = 26; k < RAND_NPARAM; ++k) ok = ok && (raw[k] >= 0.0f);      // thrust-loss coefficients (kept in the loss table, not in d)
    if (!ok) {
#pragma unroll
        for (int k = 0; k < ENV_BLOCK_FLOATS; ++k) d[k] = __builtin_nanf("");
    }
}

__device__ __forceinline__ void store_env_block(float4* tab, int stride, int i, const float d[ENV_BLOCK_FLOATS])
{
#pragma unroll
    for (int g = 0; g < ENV_GROUPS; ++g) tab[(int64_t)g * stride + i] = make_float4(d[4 * g], d[4 * g + 1], d[4 * g + 2], d[4 * g + 3]);
}

// Domain randomisation: the hull of episode `episode` of env `gid`.  Public parameter p = nominal[p] x (1 + range[p] u),
// u uniform in [-1, 1) with 16 bits (steps of 2^-15: every operation up to the product is exact or a single rounding), from four
// Philox4x32-10 blocks keyed by the seed with counter (global env id, episode, tag 0x48000000 | block): a pure function of the env and
// of its episode, like the pose sample (tags 0, 1) and the reset thrust (tag 2) - so a sharded run draws the hulls of a single-process run.
// A parameter's 16 bits are half (q & 1) of word (q & 7) >> 1 of block q >> 3, q = its SLOT: the parameters are numbered in the order
// of the packed block (m11 m22 m23 m33 Xu Klr[3] | Xuu Yv Yvv Yr Nv Nr Nrr Nuv | Yur Kf Kr lx_bow | lx_port lx_star ly Klf[3]), so that
// Philox block b yields exactly float4 groups 2b and 2b + 1 of the block: the draw is STREAMED - one Philox block, eight parameters, two
// groups handed to `emit(g, float4)` - and never holds more than a dozen values (a reset sits inside kernels that have no registers to
// spare).  Groups 8 and 9 (emitted last, from values of blocks 0 and 3) are the thrust-loss table's (dpenv_dev.h LOSS_GROUPS).
// The slot table is part of the definition of the draw (include/dpenv.h, dpenv_set_vessel_randomisation).
struct HullKey {            // what the draw needs of the kernel arguments
    const float* rand_tab;
    uint32_t seed_lo, seed_hi;
};
__device__ __forceinline__ HullKey hull_key(const StepArgs& a) { return HullKey{a.rand_tab, a.seed_lo, a.seed_hi}; }

__device__ __forceinline__ float rand_param(const HullKey& a, const uint32_t w[4], int p, int q)
{
    const uint32_t h16 = (w[(q & 7) >> 1] >> (16 * (q & 1))) & 0xffffu;
    const float u = (float)h16 * (1.0f / 32768.0f) - 1.0f;
    const float sc = 1.0f + a.rand_tab[32 + p] * u;          // two roundings (multiply, add), never contracted: plain IEEE, reproducible on a host
    return a.rand_tab[p] * sc;
}

template <class Emit>
__device__ __forceinline__ void draw_env_groups(const HullKey& a, int64_t gid, uint32_t episode, Emit&& emit)
{
    const uint32_t g0 = (uint32_t)((uint64_t)gid & 0xffffffffu), g1 = (uint32_t)((uint64_t)gid >> 32);
    const float nan = __builtin_nanf("");
    uint32_t w[4];
    float m33, klr0, klr1, klr2, klf0, klf1, klf2;
    bool ok;
    {
        philox4x32_10(g0, g1, episode, 0x48000000u, a.seed_lo, a.seed_hi, w);
        const float m11 = rand_param(a, w, 0, 0), m22 = rand_param(a, w, 1, 1), m23 = rand_param(a, w, 2, 2);
        m33 = rand_param(a, w, 3, 3);
        const float Xu = rand_param(a, w, 4, 4);
        klr0 = rand_param(a, w, 29, 5); klr1 = rand_param(a, w, 30, 6); klr2 = rand_param(a, w, 31, 7);
        // the mass-matrix inverse with derive_env_block's operations (the range was checked on the host: every hull is a vessel;
        // a non-finite nominal entry set behind the library's back still ends as NaN blocks, i.e. as DPENV_DONE_FAULT)
        const float fdet = m22 * m33 - m23 * m23;
        ok = (m11 > 0.0f) && (m22 > 0.0f) && (fdet > 0.0f);
        emit(0, ok ? make_float4(m11, m22, m23, 1.0f / m11) : make_float4(nan, nan, nan, nan));
        emit(1, ok ? make_float4(m33 / fdet, -m23 / fdet, m22 / fdet, Xu) : make_float4(nan, nan, nan, nan));
    }
    {
        philox4x32_10(g0, g1, episode, 0x48000001u, a.seed_lo, a.seed_hi, w);
        emit(2, make_float4(rand_param(a, w, 5, 8), rand_param(a, w, 6, 9), rand_param(a, w, 7, 10), rand_param(a, w, 8, 11)));        // Xuu Yv Yvv Yr
        emit(3, make_float4(rand_param(a, w, 9, 12), rand_param(a, w, 10, 13), rand_param(a, w, 11, 14), rand_param(a, w, 24, 15)));   // Nv Nr Nrr Nuv
    }
    {
        philox4x32_10(g0, g1, episode, 0x48000002u, a.seed_lo, a.seed_hi, w);
        emit(4, make_float4(rand_param(a, w, 25, 16), rand_param(a, w, 12, 17), rand_param(a, w, 13, 18), rand_param(a, w, 14, 19)));  // Yur Kf
        emit(5, make_float4(rand_param(a, w, 15, 20), rand_param(a, w, 16, 21), rand_param(a, w, 17, 22), rand_param(a, w, 18, 23)));  // Kr lx_bow
    }
    {
        philox4x32_10(g0, g1, episode, 0x48000003u, a.seed_lo, a.seed_hi, w);
        emit(6, make_float4(rand_param(a, w, 19, 24), rand_param(a, w, 20, 25), rand_param(a, w, 21, 26), rand_param(a, w, 22, 27)));  // lx_port lx_star ly_bow ly_port
        emit(7, make_float4(rand_param(a, w, 23, 28), ok ? m33 : nan, 0.0f, 0.0f));                                                     // ly_star | raw m33
        klf0 = rand_param(a, w, 26, 29); klf1 = rand_param(a, w, 27, 30); klf2 = rand_param(a, w, 28, 31);
    }
    emit(8, make_float4(klf0, klf1, klf2, klr0));                                                                                       // the loss table's two groups
    emit(9, make_float4(klr1, klr2, 0.0f, 0.0f));
}

// where group g of a draw goes: the env's vessel block (g < ENV_GROUPS) or its thrust-loss coefficients behind it (ET has DRAW_GROUPS rows)
__device__ __forceinline__ void store_draw_group(float4* env_tab, int stride, int i, int g, const float4& q)
{
    env_tab[(int64_t)g * stride + i] = q;
}

// group g of a packed block -> the fields of a Vessel (VD order, dpenv_dev.h)
__device__ __forceinline__ void vessel_set_group(Vessel& v, int g, const float4& q)
{
    switch (g) {
    case 0: v.m11 = q.x; v.m22 = q.y; v.m23 = q.z; v.inv11 = q.w; break;
    case 1: v.i22 = q.x; v.i23 = q.y; v.i33 = q.z; v.Xu = q.w; break;
    case 2: v.Xuu = q.x; v.Yv = q.y; v.Yvv = q.z; v.Yr = q.w; break;
    case 3: v.Nv = q.x; v.Nr = q.y; v.Nrr = q.z; v.Nuv = q.w; break;
    case 4: v.Yur = q.x; v.Kf[0] = q.y; v.Kf[1] = q.z; v.Kf[2] = q.w; break;
    case 5: v.Kr[0] = q.x; v.Kr[1] = q.y; v.Kr[2] = q.z; v.lx[0] = q.w; break;
    case 6: v.lx[1] = q.x; v.lx[2] = q.y; v.ly[0] = q.z; v.ly[1] = q.w; break;
    case 7: v.ly[2] = q.x; break;
    default: break;                                          // groups 8, 9: the thrust-loss table's, not the vessel's
    }
}

// a reset with the randomisation on: draw the new episode's hull group by group, into the table ...
__device__ __forceinline__ void redraw_vessel_table(const StepArgs& a, int i, uint32_t episode)
{
    draw_env_groups(hull_key(a), a.env_id_base + i, episode, [&](int g, const float4& q) { store_draw_group(a.env_tab, a.env_stride, i, g, q); });
}
// ... and into the registers the launch runs on with
__device__ __forceinline__ void redraw_vessel(const StepArgs& a, int i, uint32_t episode, Vessel& ve)
{
    draw_env_groups(hull_key(a), a.env_id_base + i, episode, [&](int g, const float4& q) {
        store_draw_group(a.env_tab, a.env_stride, i, g, q);
        vessel_set_group(ve, g, q);
    });
}
// The two-wave closed-loop kernels run at the edge of the register file (243-256 VGPRs, dpenv_policy_ws.h).  Three ways of giving their
// reset branch the hull draw behind a RUN-TIME switch were built and read off the ISA (round 5): inlined - 180-370 B of scratch per lane in
// the hot loop; pre-drawn into a side table while the env wave waits for the actor - 476 B; a real function call - the 256-env f16 form
// went from 18 scratch instructions to 86.  So there the randomisation is its own INSTANTIATION (template flag RND, the shipped training
// configuration only: dpenv_policy_ws.h), and inside it the draw is a function call - the only one in the library: its registers are its
// own, what it clobbers is saved around it inside the cold branch, and the caller re-reads the block from the table group by group.
static __device__ __attribute__((noinline)) void redraw_vessel_table_call(const float* rand_tab, uint32_t seed_lo, uint32_t seed_hi, float4* env_tab,
                                                                          int env_stride, int64_t gid, int i, uint32_t episode)
{
    draw_env_groups(HullKey{rand_tab, seed_lo, seed_hi}, gid, episode, [&](int g, const float4& q) { store_draw_group(env_tab, env_stride, i, g, q); });
}
__device__ __forceinline__ void redraw_vessel_cold(const StepArgs& a, int i, uint32_t episode, Vessel& ve)
{
    redraw_vessel_table_call(a.rand_tab, a.seed_lo, a.seed_hi, a.env_tab, a.env_stride, a.env_id_base + i, i, episode);
#pragma unroll
    for (int g = 0; g < ENV_GROUPS; ++g) vessel_set_group(ve, g, a.env_tab[(int64_t)g * a.env_stride + i]);
}

// BUILD-OWNED inflow thrust loss (round 5): F_i = K_i n_i|n_i| - Kl_i |n_i| u_a,i, u_a,i = the inflow along thruster i's
// axis at its position, (u, v, r) = the velocity through the water at the start of the env step - the linear open-water characteristic
// K_T(J) = K_T0 (1 - J / J0) (Fossen 2011, eq. 9.7) - never past zero thrust.  Coefficients forward / reverse like the gains.
struct ThrustLoss {
    float klf[3], klr[3];
    float u, v, r;
};

// StepArgs.loss_on (LOSS_*, dpenv_dev.h) for the kernels that read the per-env TABLE: LOSS_TABLE = apply the table's coefficients.  The host sets it
// when it knows that some env has one AND while it does not know (dpenv_set_vessel_params recorded into a graph, or its answer - a word the
// packing kernel leaves behind the table - still on its way: dpenv_api.hip resolve_loss): an env without a coefficient gets F - (0 |n|) u_a = F,
// the rows of the kernels that do not carry the code bit for bit (tested), so assuming a loss costs 32 B per env-step and never a wrong row.
// A plain kernel argument: the table loads below are issued with the step's opening burst (round 6 tried a device-side flag read here; the loads
// then sat behind it and the general per-env step kernel lost 3 %, profiles/LAB_NOTES.md).
__device__ __forceinline__ bool thrust_loss_on(const StepArgs& a) { return a.loss_on == LOSS_TABLE; }
// `il` argument of env_step_chain / env_step: where the step's thrust-loss coefficients come from.  A compile-time constant in every kernel
// but the one-wave closed loop (which always passes its env index and decides at run time).
constexpr int IL_NONE = -1;        // no loss code compiled in: the kernel is what it was before the loss existed
constexpr int IL_SHARED = -2;      // the single class's coefficients, StepArgs.kl (kernel arguments): VES_ARGS_LOSS / the closed loop's HULL_SHARED_LOSS

// SupervisedTau.py:42-83: tau = B(alpha) F, F_i = K_i n_i |n_i|
// sc != nullptr: sin/cos of the port and starboard azimuths are already known (sc = {sin_p, cos_p, sin_s, cos_s})
// tl != nullptr (wave-uniform): the inflow thrust loss above is applied; nullptr - the default hull - is the reference's law, untouched
__device__ __forceinline__ void thrust_map(const Vessel& ve, const float n[3], const float al[3], float& tx, float& ty,
                                           float& tn, const float* sc = nullptr, const ThrustLoss* tl = nullptr)
{
    tx = 0.0f; ty = 0.0f; tn = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float K = (n[i] >= 0.0f) ? ve.Kf[i] : ve.Kr[i];
        float F = K * fabsf(n[i]) * n[i];
        float sa, ca;
        if (i > 0 && sc != nullptr) {
            sa = sc[2 * (i - 1)]; ca = sc[2 * (i - 1) + 1];
        } else if (i == 0 && al[0] == kPi * 0.5f) {
            // the bow thruster sits at its reset default pi/2 in simple/limited/final (customEnv.py:397):
            // sin/cos of float(pi/2), no evaluation needed
            sa = 1.0f; ca = -4.371139e-08f;
        } else {
            sincos_lean(al[i], sa, ca);
        }
        if (tl != nullptr) {
            const bool ahead = n[i] >= 0.0f;
            const float ua = fmaf(fmaf(-ve.ly[i], tl->r, tl->u), ca, fmaf(ve.lx[i], tl->r, tl->v) * sa);
            F = fmaf(-((ahead ? tl->klf[i] : tl->klr[i]) * fabsf(n[i])), ua, F);
            F = ahead ? fmaxf(F, 0.0f) : fminf(F, 0.0f);
        }
        tx = fmaf(ca, F, tx);
        ty = fmaf(sa, F, ty);
        tn = fmaf(fmaf(ve.lx[i], sa, -(ve.ly[i] * ca)), F, tn);
    }
}

// observation, errorFrame.py:25-32 + ENV:196-205
__device__ __forceinline__ void make_obs(float N, float E, float psi, float u, float v, float r, float refN, float refE,
                                         float refPsi, const float pt[3], bool deg, float o[9], float& sr, float& cr,
                                         bool& rot_is_psi)
{
    const float eN = N - refN, eE = E - refE;
    const float rot = wrap_angle(psi, deg);
    rot_is_psi = (rot == psi);     // true unless the (degree-mode) wrap fired: sr, cr are then sin/cos of psi itself
    sincos_lean(rot, sr, cr);
    o[0] = fmaf(cr, eN, sr * eE);
    o[1] = fmaf(cr, eE, -(sr * eN));
    o[2] = wrap_angle(psi - refPsi, deg);
    o[3] = u; o[4] = v; o[5] = r;
    o[6] = pt[0] * 0.01f; o[7] = pt[1] * 0.01f; o[8] = pt[2] * 0.01f;
}

// training reset sampler, ENV:143-145 + simtools.py:109-123, Philox keyed (seed; global env id, episode)
template <int MODE>
__device__ __forceinline__ void sample_reset(const StepArgs& a, int64_t gid, uint32_t episode, float eta[3], float nu[3])
{
    float b[6];
    ss_bounds<MODE>(b);
    const float fr = a.reset_fraction, fv = 0.30f * a.reset_fraction;
    uint32_t w0[4], w1[4];
    const uint32_t g0 = (uint32_t)((uint64_t)gid & 0xffffffffu), g1 = (uint32_t)((uint64_t)gid >> 32);
    philox4x32_10(g0, g1, episode, 0u, a.seed_lo, a.seed_hi, w0);
    philox4x32_10(g0, g1, episode, 1u, a.seed_lo, a.seed_hi, w1);
    eta[0] = (b[0] * fr) * u01_sym(w0[0]);
    eta[1] = (b[1] * fr) * u01_sym(w0[1]);
    eta[2] = (b[2] * fr) * u01_sym(w0[2]);
    nu[0] = (b[3] * fv) * u01_sym(w0[3]);
    nu[1] = (b[4] * fv) * u01_sym(w1[0]);
    nu[2] = (b[5] * fv) * u01_sym(w1[1]);
}

// Ordering point for the LDS transposes.  With one wave per workgroup (BLOCK == 64) the staging area is
// wave-private: DS operations of a wave execute in issue order, so a compiler-level fence is all that is
// needed.  A real __syncthreads() would also drain vmcnt(0), i.e. stall on every outstanding global
// store and prefetch - measured at ~2 us per env step in the fused rollout.
template <int THREADS> __device__ __forceinline__ void lds_order()
{
    if (THREADS == 64) __builtin_amdgcn_wave_barrier();
    else __syncthreads();
}

// Current of one env: constant, or a first-order Gauss-Markov (Ornstein-Uhlenbeck) process around its mean,
// advanced once per env step (config 5; build-defined).  Noise: Philox keyed by the seed,
// counter (global env id, draw index, tag) -> Box-Muller.
struct Current {
    float vc, beta;       // present speed [m/s] and NED direction [rad]
    float vcN, vcE;       // NED components
    uint32_t ctr;         // draws made so far
};

__device__ __forceinline__ void current_components(Current& c)
{
    float sb, cb;
    sincos_lean(c.beta, sb, cb);
    c.vcN = c.vc * cb; c.vcE = c.vc * sb;
}

// Two standard normals from two Philox words (Box-Muller): u1 in (0, 1), u2 in [0, 1), both 24-bit.
__device__ __forceinline__ void box_muller(uint32_t wa, uint32_t wb, float& z0, float& z1)
{
    const float u1 = ((float)(wa >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0, 1)
    const float u2 = (float)(wb >> 8) * (1.0f / 16777216.0f);            // [0, 1)
    const float rad = sqrt_hw(-2.0f * logf(u1));
    float s2, c2;
    sincos_lean(2.0f * kPi * u2, s2, c2);
    z0 = rad * c2; z1 = rad * s2;
}

// The same transform on the hardware transcendentals (v_log_f32, v_sin_f32 / v_cos_f32 take their argument in revolutions, i.e.
// u2 itself): 8 instructions per pair instead of ~75.  Used for the policy's exploration noise, the one stream that is drawn
// every env step inside the closed loop (4 pairs per step were ~0.5 us of a SIMD's 7 us); within ~2e-6 of box_muller, which the
// reset and drift draws keep.  The reference's own noise is an unseeded TF stream (core.py:85): only the distribution can match.
__device__ __forceinline__ void box_muller_hw(uint32_t wa, uint32_t wb, float& z0, float& z1)
{
    const float u1 = ((float)(wa >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0, 1)
    const float u2 = (float)(wb >> 8) * (1.0f / 16777216.0f);            // [0, 1)
    const float rad = sqrt_hw(-1.3862943611198906f * __builtin_amdgcn_logf(u1));     // -2 ln 2 log2(u1)
    z0 = rad * __builtin_amdgcn_cosf(u2); z1 = rad * __builtin_amdgcn_sinf(u2);
}

__device__ __forceinline__ void current_drift_step(const StepArgs& a, Current& c, float vc0, float beta0, int64_t gid)
{
    uint32_t w[4];
    philox4x32_10((uint32_t)((uint64_t)gid & 0xffffffffu), (uint32_t)((uint64_t)gid >> 32), c.ctr, 0xC0000000u, a.seed_lo,
                  a.seed_hi, w);
    c.ctr += 1u;
    float zc, zs;
    box_muller_hw(w[0], w[1], zc, zs);       // drawn every env step where the current drifts: the hardware transcendentals (above)
    c.vc = fmaf(a.drift_sv, zc, fmaf(a.drift_a, vc0 - c.vc, c.vc));
    c.beta = fmaf(a.drift_sb, zs, fmaf(a.drift_a, beta0 - c.beta, c.beta));
    current_components(c);
}

// Per-episode randomisation of the current (round 6; dpenv_set_current_randomisation): every reset - explicit, auto, reset_at_end - draws the new
// episode's current around the env's nominal one,
//     V_c = max(0, V_nom + range_v u1),   beta_c = beta_nom + range_b u2,   u1, u2 = u01_sym of words 0, 1 of
// Philox4x32-10 keyed by the seed with counter (global env id, episode, tag 3): a function of the env and of its episode like the pose sample
// (tags 0, 1), the reset thrust (tag 2) and the hull (0x48000000 | block) - independent of the rank count and of the launch form.  The drawn
// values become both the present current and the mean the drift reverts to.  Lives where the hull re-draw lives: in the reset kernel and in
// the GENERAL per-env kernels (dpenv_dev.h VES_ENV_RND, the closed loop's HULL_ENV_RND and one-wave forms); every other kernel is untouched.
// current_draw: the draw itself, inlined where the caller has registers to spare (the reset kernel, dpenv_step's reset wave and one-wave reset
// branch, the fused rollout's reset branch); current_redraw_call: the same as a function call, like the hull draw's (redraw_vessel_table_call
// above), for the closed-loop kernels that run at the edge of the register file - the Philox rounds use the callee's registers, two floats come back.
__device__ __forceinline__ float2 current_draw(const float* nom, int nom_stride, float range_v, float range_b, uint32_t seed_lo, uint32_t seed_hi,
                                               int64_t gid, int i, uint32_t episode)
{
    uint32_t w[4];
    philox4x32_10((uint32_t)((uint64_t)gid & 0xffffffffu), (uint32_t)((uint64_t)gid >> 32), episode, 3u, seed_lo, seed_hi, w);
    const float v = fmaxf(nom[i] + range_v * u01_sym(w[0]), 0.0f);           // (a product and a sum, never contracted: -ffp-contract=off)
    const float b = nom[nom_stride + i] + range_b * u01_sym(w[1]);
    return make_float2(v, b);
}
__device__ __forceinline__ float2 current_draw(const StepArgs& a, int i, int il, uint32_t episode)
{
    return current_draw(a.cur_nom, a.cur_nom_stride, a.cur_range_v, a.cur_range_b, a.seed_lo, a.seed_hi, a.env_id_base + i, il, episode);
}
static __device__ __attribute__((noinline)) float2 current_redraw_call(const float* nom, int nom_stride, float range_v, float range_b, uint32_t seed_lo,
                                                                       uint32_t seed_hi, int64_t gid, int i, uint32_t episode)
{
    return current_draw(nom, nom_stride, range_v, range_b, seed_lo, seed_hi, gid, i, episode);
}
// the caller's registers: the present current, its components, the drift's means.  The caller stores them (cur_vc, cur_beta, cur_vc0,
// cur_beta0) where it stores the drifting current: at once in the one-step kernels, at the end of the launch in the T-step kernels.
__device__ __forceinline__ void current_redraw(const StepArgs& a, int i, uint32_t episode, Current& c, float& vc0, float& beta0)
{
    const float2 d = current_redraw_call(a.cur_nom, a.cur_nom_stride, a.cur_range_v, a.cur_range_b, a.seed_lo, a.seed_hi, a.env_id_base + i, i, episode);
    c.vc = d.x; c.beta = d.y;
    vc0 = d.x; beta0 = d.y;
    current_components(c);
}
__device__ __forceinline__ void current_redraw_inline(const StepArgs& a, int i, uint32_t episode, Current& c, float& vc0, float& beta0)
{
    const float2 d = current_draw(a, i, i, episode);
    c.vc = d.x; c.beta = d.y;
    vc0 = d.x; beta0 = d.y;
    current_components(c);
}
__device__ __forceinline__ void store_current(const StepArgs& a, int i, const Current& c, float vc0, float beta0, bool means)
{
    a.cur_vc[i] = c.vc; a.cur_beta[i] = c.beta;
    if (means) { a.cur_vc0[i] = vc0; a.cur_beta0[i] = beta0; }
}

// Exploration noise of the policy (core.py:85 tf.random_normal inside the graph): A standard normals for draw `ctr` of env
// `gid`, Philox keyed by the seed with counter (global env id, per-env draw index, tag 0xA0000000 | block) -> a function of
// the env and of how many actions it has sampled so far, not of the rank count or of the launch geometry.
template <int A>
__device__ __forceinline__ void policy_noise(const StepArgs& a, int64_t gid, uint32_t ctr, float xi[A])
{
    const uint32_t g0 = (uint32_t)((uint64_t)gid & 0xffffffffu), g1 = (uint32_t)((uint64_t)gid >> 32);
    float z[8];
    uint32_t w[4];
    philox4x32_10(g0, g1, ctr, 0xA0000000u, a.seed_lo, a.seed_hi, w);
    box_muller_hw(w[0], w[1], z[0], z[1]);
    box_muller_hw(w[2], w[3], z[2], z[3]);
    if (A > 4) {
        philox4x32_10(g0, g1, ctr, 0xA0000001u, a.seed_lo, a.seed_hi, w);
        box_muller_hw(w[0], w[1], z[4], z[5]);
        box_muller_hw(w[2], w[3], z[6], z[7]);
    }
#pragma unroll
    for (int k = 0; k < A; ++k) xi[k] = z[k];
}

// customEnv.py:179-188 (reset_acts): the episode starts with previous thrust clip(scale(N(0, 0.1))) instead of zero - thrust
// only, azimuths keep their defaults.  Draw keyed (seed; global env id, episode, tag 2) next to the pose / velocity draws.
__device__ __forceinline__ void sample_reset_thrust(const StepArgs& a, int64_t gid, uint32_t episode, float pt[3])
{
    uint32_t w[4];
    philox4x32_10((uint32_t)((uint64_t)gid & 0xffffffffu), (uint32_t)((uint64_t)gid >> 32), episode, 2u, a.seed_lo, a.seed_hi, w);
    float z0, z1, z2, z3;
    box_muller(w[0], w[1], z0, z1);
    box_muller(w[2], w[3], z2, z3);
    pt[0] = clipf((0.1f * z0) * 100.0f, 100.0f);          // np.random.normal(0, 0.1) then scale_and_clip (ENV:181-182,215-225)
    pt[1] = clipf((0.1f * z1) * 100.0f, 100.0f);
    pt[2] = clipf((0.1f * z2) * 100.0f, 100.0f);
}

__device__ __forceinline__ uint16_t f2bf(float x)
{
    // plain cast: v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
    return __builtin_bit_cast(uint16_t, __float2bfloat16(x));
}

// Coalesced write of a workgroup's THREADS x OD staged elements (LDS image [j*THREADS + tid]) to
// dst[base ...]; `rem` = elements remaining in the destination from `base` (uniform).  Full workgroups take
// the unpredicated path: uniform base pointer + 32-bit lane offset.
template <int OD, int THREADS>
__device__ __forceinline__ void store_rows(void* dst, int64_t base, int64_t rem, bool bf16, const float* lds, int tid)
{
    const bool full = rem >= (int64_t)THREADS * OD;
    if (bf16) {
        uint16_t* p = (uint16_t*)dst + base;
        if (full) {
#pragma unroll
            for (int j = 0; j < OD; ++j) p[(unsigned)(j * THREADS + tid)] = f2bf(lds[j * THREADS + tid]);
        } else {
#pragma unroll
            for (int j = 0; j < OD; ++j)
                if (j * THREADS + tid < rem) p[(unsigned)(j * THREADS + tid)] = f2bf(lds[j * THREADS + tid]);
        }
    } else {
        float* p = (float*)dst + base;
        if (full) {
#pragma unroll
            for (int j = 0; j < OD; ++j) p[(unsigned)(j * THREADS + tid)] = lds[j * THREADS + tid];
        } else {
#pragma unroll
            for (int j = 0; j < OD; ++j)
                if (j * THREADS + tid < rem) p[(unsigned)(j * THREADS + tid)] = lds[j * THREADS + tid];
        }
    }
}

// Coalesced read of a workgroup's THREADS x A action elements into registers (element j*THREADS + tid).
// Out-of-range elements of the last workgroup are clamped to the last valid one instead of being selected
// to zero: a select on the loaded value would force a wait after every load (the lanes that read them are
// dead and never store).
template <int A, int THREADS>
__device__ __forceinline__ void load_rows(const float* src_block, int64_t rem, int tid, float pre[A])
{
    const int lim = (int)(rem < (int64_t)THREADS * A ? rem : (int64_t)THREADS * A) - 1;   // uniform
#pragma unroll
    for (int j = 0; j < A; ++j) {
        const int e = j * THREADS + tid;
        pre[j] = src_block[(unsigned)(e < lim ? e : lim)];
    }
}

// Write one observation row per lane.  AOS: through LDS so that the global stores are coalesced
// (row stride OD is odd for OD = 9 -> conflict-free; OD = 6 costs a 2-way conflict).
template <int OD>
__device__ __forceinline__ void store_obs(const StepArgs& a, void* dst, const float o[9], int i, bool live, float* lds)
{
    const int tid = threadIdx.x;
    const int n = a.n;
    if (a.obs_layout == LAYOUT_SOA) {
        if (live) {
            if (a.obs_bf16) {
                uint16_t* p = (uint16_t*)dst;
#pragma unroll
                for (int k = 0; k < OD; ++k) p[(int64_t)k * n + i] = f2bf(o[k]);
            } else {
                float* p = (float*)dst;
#pragma unroll
                for (int k = 0; k < OD; ++k) p[(int64_t)k * n + i] = o[k];
            }
        }
        return;
    }
    lds_order<BLOCK>();   // previous users of the LDS staging area are done
#pragma unroll
    for (int k = 0; k < OD; ++k) lds[tid * OD + k] = o[k];
    lds_order<BLOCK>();
    const int64_t base = (int64_t)blockIdx.x * (BLOCK * OD);          // uniform
    const int64_t rem = (int64_t)n * OD - base;                           // elements left from this block's start
    store_rows<OD, BLOCK>(dst, base, rem, a.obs_bf16, lds, tid);
}

// =============================================================================================
//  one environment step in registers (shared by step_kernel and rollout_kernel)
// =============================================================================================
struct Env {              // per-lane state carried between env steps
    float N, E, psi, u, v, r;
    float refN, refE, refPsi;
    float pt[3];          // previous thrust command, percent (ENV:126)
    float ang[3];         // azimuth command in force: bow, port, star (ENV:122)
    int steps;            // steps taken in the running episode
    float sn, cs;         // sin/cos of psi (cache: refreshed by every observation, reused by the next plant step)
};

struct StepOut {
    float o[9];           // observation of this step (terminal one if the env finished)
    float reward;
    float parts[4];
    uint32_t d;           // DONE_* bits
};

// ENV:104-133 for one env in three pieces, so that a launch can give them to different waves (rollout_ws_kernel):
// env_decode (action -> commands and wrench), env_plant (the integrator), env_done / env_reward (termination, reward,
// late new_ref).  env_step is their composition.
struct Wrench {
    float thr[3];         // thrust commands, percent
    float tx, ty, tn;     // body-frame wrench of the three thrusters
};

// ---- action decode ENV:104-110, scale_and_clip ENV:215-225, command map ENV:117-122, force map --------------
// DEFER_ANG (MODE_FINAL_CONT only): the two atan2 of the stern azimuths are NOT evaluated here - the force map uses the normalised
// heads directly, the angles themselves are only bookkeeping (next step's ang_prev) and reward (azimuth-rate penalty) - so a kernel
// whose next observation is on a critical path takes them later with stern_angles() (env_step_finish); ang[1], ang[2] are left alone.
template <int MODE, bool DEFER_ANG = false>
__device__ __forceinline__ void env_decode_cmd(float ang[3], const float* act, float thr[3])
{
    // the COMMANDS of a step (thrust percent, azimuths): all the reward's penalties and the azimuth bookkeeping need; the force map
    // (below) needs more.  Split out so that a wave that only keeps the books (rollout_ws_kernel's row wave) evaluates exactly these
    // expressions and nothing else.
    thr[0] = clipf(act[0] * 100.0f, 100.0f);
    thr[1] = clipf(act[1] * 100.0f, 100.0f);
    thr[2] = clipf(act[2] * 100.0f, 100.0f);
    if (MODE == MODE_FULL) {
        ang[0] = clipf(act[3] * kPi, kPi); ang[1] = clipf(act[4] * kPi, kPi); ang[2] = clipf(act[5] * kPi, kPi);
    } else if (MODE == MODE_LIMITED) {
        ang[1] = clipf(act[3] * (kPi * 0.5f), kPi * 0.5f); ang[2] = clipf(act[4] * (kPi * 0.5f), kPi * 0.5f);
    } else if (MODE == MODE_FINAL_WRAP) {
        // ENV:237-244: wrap_angle(a*pi, deg=False)/pi, evaluated in units of pi (exact in fp32)
        const float w3 = act[3] - 2.0f * floorf((act[3] + 1.0f) * 0.5f);
        const float w4 = act[4] - 2.0f * floorf((act[4] + 1.0f) * 0.5f);
        ang[1] = clipf(w3 * kPi, kPi); ang[2] = clipf(w4 * kPi, kPi);
    } else if (MODE == MODE_FINAL_CONT) {
        // ENV:227-235: atan2(sin_head, cos_head)/pi, then *pi and clip
        if (!DEFER_ANG) { ang[1] = clipf(atan2_lean(act[3], act[4]), kPi); ang[2] = clipf(atan2_lean(act[5], act[6]), kPi); }
    }
}

template <int MODE, bool DEFER_ANG = false>
__device__ __forceinline__ void env_decode(const Vessel& ve, float ang[3], const float* act, Wrench& w, const ThrustLoss* tl = nullptr)
{
    env_decode_cmd<MODE, DEFER_ANG>(ang, act, w.thr);
    if (MODE == MODE_FINAL_CONT) {
        // the azimuth is atan2 of the two heads, so its sine and cosine are the normalised heads themselves:
        // no sincos of the angle just computed ((0, 0) -> angle 0 -> (0, 1))
        float sc[4];
        const float np2 = fmaf(act[3], act[3], act[4] * act[4]), ns2 = fmaf(act[5], act[5], act[6] * act[6]);
        const float ip = __builtin_amdgcn_rsqf(np2), is = __builtin_amdgcn_rsqf(ns2);
        sc[0] = (np2 > 0.0f) ? act[3] * ip : 0.0f; sc[1] = (np2 > 0.0f) ? act[4] * ip : 1.0f;
        sc[2] = (ns2 > 0.0f) ? act[5] * is : 0.0f; sc[3] = (ns2 > 0.0f) ? act[6] * is : 1.0f;
        thrust_map(ve, w.thr, ang, w.tx, w.ty, w.tn, sc, tl);
    } else {
        thrust_map(ve, w.thr, ang, w.tx, w.ty, w.tn, nullptr, tl);
    }
}

// ---- plant: BUILD-OWNED 3-DOF model, n_substeps semi-implicit Euler steps ---------------
// (sn, cs) = sin/cos of psi on entry; on exit the state after the step.  The caller refreshes (sn, cs) for the heading
// reached and converts the velocity back to over-ground with it (env_plant_finish).
__device__ __forceinline__ void env_plant(const StepArgs& a, const Vessel& ve, float tx, float ty, float tn, bool cur, float vcN,
                                          float vcE, float& N, float& E, float& psi, float& u, float& v, float& r, float sn, float cs)
{
    if (cur) {
        u -= fmaf(cs, vcN, sn * vcE);      // relative velocity nu_r = nu - R(psi)^T v_c
        v -= fmaf(cs, vcE, -(sn * vcN));
    }
    const float h = a.h;
    const float hA = h * ve.inv11, h22 = h * ve.i22, h23 = h * ve.i23, h33 = h * ve.i33;
    // the Coriolis term c23 = m11 u only ever multiplies r (sway) and v (yaw): fold it into the
    // speed-proportional cross-flow coefficients so that it costs nothing per sub-step
    const float yur = ve.Yur + ve.m11;      // fy gets -(Yr + (Yur + m11) u) r
    const float nuv = ve.Nuv - ve.m11;      // fn gets -(Nv + (Nuv - m11) u) v
    const int nsub = a.hold_plant ? 0 : a.n_substeps;
    float aN = 0.0f, aE = 0.0f;
    // unrolled x10 (20 sub-steps = 2 trips): the loop counter / compare / branch are SALU issue slots of the same lone wave
#pragma unroll 10
    for (int k = 0; k < nsub; ++k) {
        // C(nu)nu with c13 = -q, q = m22 v + m23 r
        const float q = fmaf(ve.m22, v, ve.m23 * r);
        float fx = fmaf(-fmaf(ve.Xuu, fabsf(u), ve.Xu), u, tx);
        fx = fmaf(q, r, fx);
        float fy = fmaf(-fmaf(ve.Yvv, fabsf(v), ve.Yv), v, ty);
        fy = fmaf(-fmaf(yur, u, ve.Yr), r, fy);
        float fn = fmaf(-q, u, tn);
        fn = fmaf(-fmaf(nuv, u, ve.Nv), v, fn);
        fn = fmaf(-fmaf(ve.Nrr, fabsf(r), ve.Nr), r, fn);
        u = fmaf(hA, fx, u);
        v = fmaf(h22, fy, fmaf(h23, fn, v));
        r = fmaf(h23, fy, fmaf(h33, fn, r));
        // kinematics with the old heading and the new velocity: the NED velocity over water is summed here and the
        // position advanced once after the loop (h * sum: one rounding of N, E per env step instead of twenty)
        aN = fmaf(-sn, v, fmaf(cs, u, aN));
        aE = fmaf(cs, v, fmaf(sn, u, aE));
        // heading: psi += d, d = h r; its sin/cos by the rotation (1 - d^2/2, d), i.e. exact to second order - one
        // order above the integrator's own - and re-seeded from the exact sin/cos of psi at every env step.
        // cs' = cs - sn d - cs d^2/2 and sn' = sn + cs d - sn d^2/2 in Horner form
        const float d = h * r;
        const float e = -0.5f * d;
        psi += d;
        const float c2 = fmaf(fmaf(e, cs, -sn), d, cs);
        const float s2n = fmaf(fmaf(e, sn, cs), d, sn);
        cs = c2; sn = s2n;
    }
    N = fmaf(h, aN, N);
    E = fmaf(h, aE, E);
    if (cur) {
        const float hn = h * (float)nsub;       // the current carries the hull along for the whole step
        N = fmaf(hn, vcN, N);
        E = fmaf(hn, vcE, E);
    }
}

// back to velocity over ground with the exact sin/cos (se, ce) of the heading reached
__device__ __forceinline__ void env_plant_finish(bool cur, float vcN, float vcE, float se, float ce, float& u, float& v)
{
    if (cur) {
        u += fmaf(ce, vcN, se * vcE);
        v += fmaf(ce, vcE, -(se * vcN));
    }
}

// ---- reward ENV:253-325 -------------------------------------------------------------------------------------------
// o = the observation of this step (o[0..5] used), thr = this step's thrust commands, pt_old / ang_prev = the commands in force before
// it, ang_new = the azimuths commanded by it
template <int MODE, bool EXT>
__device__ __forceinline__ void env_reward(const StepArgs& a, const float* o, const float thr[3], const float pt_old[3],
                                           const float ang_new[3], const float ang_prev[3], StepOut& out)
{
    float p_der = 0.0f;
    const float p_vel = -sqrt_hw(fmaf(o[3] * o[3], 0.5f, fmaf(o[4] * o[4], 0.5f, o[5] * o[5])));   // ENV:267-273
    const float rr2 = fmaf(o[0], o[0], o[1] * o[1]);
    const float yaw = o[2] * (180.0f / kPi);                                                    // ENV:281
    // exp(x) as the hardware's exp2(x log2 e): relative error <= ~1e-7 (1 + |x|), far inside the reward tolerance
    const float multivar = 2.0f * __builtin_amdgcn_exp2f((-0.5f * 1.4426950408889634f) * fmaf(yaw * yaw, 1.0f / 25.0f, rr2));   // ENV:283, covar ENV:86-88
    const float special = sqrt_hw(fmaf(yaw * 0.25f, yaw * 0.25f, rr2));                         // ENV:287
    const float p_pos = multivar + fmaxf(-1.0f, fmaf(-0.1f, special, 1.0f)) + 0.5f;             // ENV:288-290
    const float p_thr = -(fabsf(thr[0]) * 0.20f + fabsf(thr[1]) * 0.30f + fabsf(thr[2]) * 0.30f) * 0.01f;   // ENV:292-302
    if (EXT) {
        const float inv_dt = a.inv_dt;
        float pen = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) pen -= fabsf((thr[k] - pt_old[k]) * inv_dt * 0.01f) * 0.05f;   // ENV:310-313
        const float inv_bnd = (MODE == MODE_LIMITED) ? (2.0f / kPi) : (1.0f / kPi);               // ENV:319
        const float angpen = -(fabsf((ang_new[1] - ang_prev[1]) * inv_dt * inv_bnd) * 0.01f +
                               fabsf((ang_new[2] - ang_prev[2]) * inv_dt * inv_bnd) * 0.01f);     // ENV:315-320 (bow coeff 0)
        p_der = pen + fmaxf(-1.0f, angpen);                                                       // ENV:322-323
    }
    out.parts[0] = p_vel; out.parts[1] = p_pos; out.parts[2] = p_thr; out.parts[3] = p_der;
    out.reward = p_vel + p_pos + p_thr + p_der;   // ENV:263
}

// ---- termination ENV:207-213, fault check, late new_ref ENV:131, bookkeeping ENV:126 -----------------------------------
// o[0..5] = the observation of this step; act only for the non-finite check
template <int MODE>
__device__ __forceinline__ void env_done(const StepArgs& a, Env& s, const float* act, const float* o, const float thr[3], bool has_ref,
                                         float nrN, float nrE, float nrP, StepOut& out)
{
    uint32_t d = 0;
    if (a.terminate) {
        float b[6];
        ss_bounds<MODE>(b);
        bool t = false;
#pragma unroll
        for (int k = 0; k < 6; ++k) t = t || (fabsf(o[k]) > b[k]);   // ENV:207-213, strict >
        d = t ? DONE_TERMINAL : 0u;
    }
    {
        // any NaN/Inf in the state or in the action poisons the sum (a NaN thrust would otherwise be clipped to a
        // legal command by fminf/fmaxf and vanish)
        float chk = s.N + s.E + s.psi + s.u + s.v + s.r;
#pragma unroll
        for (int k = 0; k < ModeTraits<MODE>::A; ++k) chk += act[k];
        if (!(fabsf(chk) <= 3.0e38f)) d |= DONE_TERMINAL | DONE_FAULT;
    }
    if (has_ref) { s.refN = nrN; s.refE = nrE; s.refPsi = nrP; }   // ENV:131: visible from the next step (Q4)
    s.steps += 1;
    if (a.max_ep_len > 0 && s.steps >= a.max_ep_len) d |= DONE_TIMELIMIT;   // ppo.py:304
    s.pt[0] = thr[0]; s.pt[1] = thr[1]; s.pt[2] = thr[2];   // ENV:126
    out.d = d;
}

// what the reward of a step still needs once its observation and done bits are out (env_step_chain -> env_step_finish)
struct StepRest {
    float thr[3], pt_old[3], ang_prev[3], ang_new[3];
};

// ENV:104-133 up to and including the observation and the termination bits - everything the NEXT policy input depends on.
// DEFER: leave the reward (and, for the continuous-angle variant, the two atan2 behind the azimuth bookkeeping) to env_step_finish,
// so that a kernel can hand the observation over first.  cur = constant current (vcN, vcE in NED) present.
// il: the env's (clamped) index, for the thrust-loss table; IL_SHARED = the single class's coefficients (kernel arguments); IL_NONE (a compile-time
// constant in every kernel but the general per-env ones, the shared-loss ones and the one-wave closed loop) = the loss code is not even
// compiled in, those kernels are what they were without it.
template <int MODE, bool EXT, bool DEFER>
__device__ __forceinline__ void env_step_chain(const StepArgs& a, const Vessel& ve, Env& s, const float* act, bool has_ref,
                                               float nrN, float nrE, float nrP, bool cur, float vcN, float vcE, StepOut& out, StepRest& rest,
                                               int il = -1)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) { rest.ang_prev[k] = s.ang[k]; rest.pt_old[k] = s.pt[k]; }   // ENV:102
    Wrench w;
    if (il == IL_SHARED || (il >= 0 && thrust_loss_on(a))) {     // launch-uniform
        ThrustLoss tl;
        if (il == IL_SHARED) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { tl.klf[k] = a.kl[k]; tl.klr[k] = a.kl[3 + k]; }
        } else {
            const float4 q0 = a.env_tab[(int64_t)ENV_GROUPS * a.env_stride + il], q1 = a.env_tab[(int64_t)(ENV_GROUPS + 1) * a.env_stride + il];
            tl.klf[0] = q0.x; tl.klf[1] = q0.y; tl.klf[2] = q0.z; tl.klr[0] = q0.w; tl.klr[1] = q1.x; tl.klr[2] = q1.y;
        }
        tl.u = s.u; tl.v = s.v; tl.r = s.r;
        if (cur) {                                       // through the water: nu_r = nu - R(psi)^T v_c (env_plant forms the same difference)
            tl.u -= fmaf(s.cs, vcN, s.sn * vcE);
            tl.v -= fmaf(s.cs, vcE, -(s.sn * vcN));
        }
        env_decode<MODE, DEFER>(ve, s.ang, act, w, &tl);
    } else {
        env_decode<MODE, DEFER>(ve, s.ang, act, w);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { rest.thr[k] = w.thr[k]; rest.ang_new[k] = s.ang[k]; }      // ang_new[1..2] still the old ones if deferred
    float N = s.N, E = s.E, psi = s.psi, u = s.u, v = s.v, r = s.r;
    env_plant(a, ve, w.tx, w.ty, w.tn, cur, vcN, vcE, N, E, psi, u, v, r, s.sn, s.cs);
    // exact sin/cos of the heading reached: needed by the observation, by the current term and by the next step
    const bool deg = (a.wrap_mode == WRAP_REFERENCE);
    float* o = out.o;
    bool same;
    float se, ce;
    {
        // the observation only depends on nu through o[3..5]; compute the frame first, patch nu after
        make_obs(N, E, psi, 0.0f, 0.0f, 0.0f, s.refN, s.refE, s.refPsi, rest.pt_old, deg, o, se, ce, same);
        if (!same) sincos_lean(psi, se, ce);
    }
    env_plant_finish(cur, vcN, vcE, se, ce, u, v);
    o[3] = u; o[4] = v; o[5] = r;
    s.N = N; s.E = E; s.psi = psi; s.u = u; s.v = v; s.r = r;
    s.sn = se; s.cs = ce;
    env_done<MODE>(a, s, act, o, rest.thr, has_ref, nrN, nrE, nrP, out);
}

// the rest of the step: azimuth bookkeeping of the continuous-angle variant (if it was deferred) and the reward.  `untouched`: the env was
// NOT re-drawn since env_step_chain (a reset has put the default azimuths into s.ang: they stay).  out.o must still hold the
// observation env_step_chain produced.
template <int MODE, bool EXT, bool DEFER>
__device__ __forceinline__ void env_step_finish(const StepArgs& a, Env& s, const float* act, StepRest& rest, bool untouched, StepOut& out)
{
    if (DEFER && MODE == MODE_FINAL_CONT) {
        rest.ang_new[1] = clipf(atan2_lean(act[3], act[4]), kPi);       // ENV:227-235
        rest.ang_new[2] = clipf(atan2_lean(act[5], act[6]), kPi);
        if (untouched) { s.ang[1] = rest.ang_new[1]; s.ang[2] = rest.ang_new[2]; }
    }
    env_reward<MODE, EXT>(a, out.o, rest.thr, rest.pt_old, rest.ang_new, rest.ang_prev, out);
}

// ENV:104-133 for one env: decode, command map, plant, observation, reward, termination, late new_ref.
template <int MODE, bool EXT>
__device__ __forceinline__ void env_step(const StepArgs& a, const Vessel& ve, Env& s, const float* act, bool has_ref,
                                         float nrN, float nrE, float nrP, bool cur, float vcN, float vcE, StepOut& out, int il = -1)
{
    StepRest rest;
    env_step_chain<MODE, EXT, false>(a, ve, s, act, has_ref, nrN, nrE, nrP, cur, vcN, vcE, out, rest, il);
    env_step_finish<MODE, EXT, false>(a, s, act, rest, true, out);
}

// auto-reset of one finished env: ENV:135-194 with the training sampler; returns the new episode's first obs.
// Two pieces, so that a kernel with idle time can take the draw off its critical path: reset_draw (the Philox / Box-Muller part, a
// pure function of (seed, global env id, episode)) and reset_apply (state assignment + first observation).
struct ResetDraw {
    float eta[3], nu[3], pt[3];
};

template <int MODE>
__device__ __forceinline__ void reset_draw(const StepArgs& a, int64_t gid, uint32_t episode, ResetDraw& d)
{
    sample_reset<MODE>(a, gid, episode, d.eta, d.nu);
    d.pt[0] = d.pt[1] = d.pt[2] = 0.0f;                         // ENV:190
    if (a.reset_acts) sample_reset_thrust(a, gid, episode, d.pt);   // ENV:179-188
}

template <int MODE>
__device__ __forceinline__ void reset_apply(const StepArgs& a, Env& s, const ResetDraw& d, float o_new[9])
{
    s.N = d.eta[0]; s.E = d.eta[1]; s.psi = d.eta[2]; s.u = d.nu[0]; s.v = d.nu[1]; s.r = d.nu[2];
    s.pt[0] = d.pt[0]; s.pt[1] = d.pt[1]; s.pt[2] = d.pt[2];
    default_angles<MODE>(s.ang[0], s.ang[1], s.ang[2]);         // ENV:173-177,192
    s.steps = 0;
    bool same;
    make_obs(s.N, s.E, s.psi, s.u, s.v, s.r, s.refN, s.refE, s.refPsi, s.pt, a.wrap_mode == WRAP_REFERENCE, o_new, s.sn, s.cs,
             same);
    if (!same) sincos_lean(s.psi, s.sn, s.cs);
}

template <int MODE>
__device__ __forceinline__ void env_auto_reset(const StepArgs& a, Env& s, int64_t gid, uint32_t episode, float o_new[9])
{
    ResetDraw d;
    reset_draw<MODE>(a, gid, episode, d);
    reset_apply<MODE>(a, s, d, o_new);
}

__device__ __forceinline__ void load_env(const float4* S0, const float4* S1, const float4* S2, const float4* RF, int il, Env& s)
{
    const float4 s0 = S0[il], s1 = S1[il], s2 = S2[il], rf = RF[il];
    s.N = s0.x; s.E = s0.y; s.psi = s0.z; s.u = s0.w; s.v = s1.x; s.r = s1.y;
    s.ang[0] = rf.w; s.ang[1] = s1.z; s.ang[2] = s1.w;
    s.pt[0] = s2.x; s.pt[1] = s2.y; s.pt[2] = s2.z; s.steps = __float_as_int(s2.w);
    s.refN = rf.x; s.refE = rf.y; s.refPsi = rf.z;
}

__device__ __forceinline__ void load_env(const StepArgs& a, int il, Env& s) { load_env(a.S0, a.S1, a.S2, a.RF, il, s); }

__device__ __forceinline__ void store_env(const StepArgs& a, int i, const Env& s, bool rf_dirty)
{
    a.S0[i] = make_float4(s.N, s.E, s.psi, s.u);
    a.S1[i] = make_float4(s.v, s.r, s.ang[1], s.ang[2]);
    a.S2[i] = make_float4(s.pt[0], s.pt[1], s.pt[2], __int_as_float(s.steps));
    if (rf_dirty) a.RF[i] = make_float4(s.refN, s.refE, s.refPsi, s.ang[0]);
}

}  // namespace dpenv

#endif
