"""Numerical validation of the lean device math (sincos / atan2) used by dpenv_kernels.hip.

Emulates the fp32 FMA sequences with float64 intermediates rounded to float32 after every operation
(an FMA is a*b+c in float64 rounded once), and reports max abs / ulp error against float64 libm.
Also derives the atan polynomial coefficients (Chebyshev-node least squares + a few Remez-style
reweighting rounds) so that no magic numbers are taken on trust.
"""
import numpy as np

f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def mul(a, b):
    return (a.astype(np.float64) * b.astype(np.float64)).astype(f32)


C1, C2, C3 = f32(1.5703125), f32(4.837512969970703125e-4), f32(7.54978995489188216e-8)
S1, S2, S3 = f32(-1.6666654611e-1), f32(8.3321608736e-3), f32(-1.9515295891e-4)
Q1, Q2, Q3 = f32(4.166664568298827e-2), f32(-1.388731625493765e-3), f32(2.443315711809948e-5)


def sincos_lean(x):
    x = x.astype(f32)
    kf = np.rint(mul(x, np.full_like(x, f32(0.6366197723675814))))
    r = fma(kf, np.full_like(x, -C1), x)
    r = fma(kf, np.full_like(x, -C2), r)
    r = fma(kf, np.full_like(x, -C3), r)
    r2 = mul(r, r)
    p = fma(r2, np.full_like(x, S3), np.full_like(x, S2))
    p = fma(r2, p, np.full_like(x, S1))
    s = fma(mul(r, r2), p, r)
    q = fma(r2, np.full_like(x, Q3), np.full_like(x, Q2))
    q = fma(r2, q, np.full_like(x, Q1))
    c = fma(mul(r2, r2), q, fma(r2, np.full_like(x, f32(-0.5)), np.full_like(x, f32(1.0))))
    n = kf.astype(np.int64) & 3
    so = np.where(n & 1, c, s)
    co = np.where(n & 1, s, c)
    so = np.where(n & 2, -so, so)
    co = np.where((n + 1) & 2, -co, co)
    return so.astype(f32), co.astype(f32)


def fit_atan(deg):
    """coefficients c_k of atan(a) ~= a * sum_k c_k s^k, s = a^2, a in [0, 1]; c_0 fixed to 1."""
    a = np.cos(np.linspace(0, np.pi, 4001)) * 0.5 + 0.5
    a = a[a > 1e-6]
    s = a * a
    y = (np.arctan(a) / a - 1.0) / s          # = sum_{k>=1} c_k s^(k-1)
    V = np.vander(s, deg, increasing=True)
    w = np.ones_like(a)
    for _ in range(40):                        # crude Remez: reweight toward the max error
        c = np.linalg.lstsq(V * w[:, None], y * w, rcond=None)[0]
        err = np.abs((V @ c - y) * s * a)
        w = w * (1 + 2 * err / err.max())
    return np.concatenate([[1.0], c])


def atan2_lean(y, x, coef):
    y = y.astype(f32)
    x = x.astype(f32)
    ax, ay = np.abs(x), np.abs(y)
    mx, mn = np.maximum(ax, ay), np.minimum(ax, ay)
    with np.errstate(divide='ignore', invalid='ignore'):
        a = np.where(mx == 0, f32(0), (mn.astype(np.float64) / mx.astype(np.float64)).astype(f32))
    s = mul(a, a)
    cf = [f32(v) for v in coef]
    r = np.full_like(a, cf[-1])
    for k in range(len(cf) - 2, 0, -1):
        r = fma(r, s, np.full_like(a, cf[k]))
    r = fma(mul(r, s), a, a)
    r = np.where(ay > ax, f32(np.pi / 2) - r, r).astype(f32)
    r = np.where(np.signbit(x), f32(np.pi) - r, r).astype(f32)
    return np.copysign(r, y).astype(f32)


def ulp(ref):
    return np.spacing(np.abs(ref).astype(f32)).astype(np.float64)


if __name__ == '__main__':
    rng = np.random.RandomState(0)
    for lo, hi in [(-np.pi, np.pi), (-8, 8), (-200, 200), (-30000, 30000)]:
        x = rng.uniform(lo, hi, 2000000).astype(f32)
        s, c = sincos_lean(x)
        es = np.abs(s - np.sin(x.astype(np.float64)))
        ec = np.abs(c - np.cos(x.astype(np.float64)))
        print('sincos [%g, %g]: max abs err sin %.3e cos %.3e' % (lo, hi, es.max(), ec.max()))
    s, c = sincos_lean(np.array([0.0, np.pi / 2, -np.pi / 2, np.pi], f32))
    print('special', s, c)
    for deg in (7, 8, 9):
        coef = fit_atan(deg)
        y = rng.normal(size=2000000).astype(f32)
        x = rng.normal(size=2000000).astype(f32)
        r = atan2_lean(y, x, coef)
        ref = np.arctan2(y.astype(np.float64), x.astype(np.float64))
        e = np.abs(r - ref)
        print('atan2 deg %d: max abs err %.3e  max ulp %.2f' % (deg, e.max(), (e / ulp(ref)).max()))
        print('   coef', ', '.join('%.10ef' % v for v in coef))
    coef = fit_atan(9)
    yy = np.array([0.0, -0.0, 0.0, -0.0, 1.0, -1.0, 0.0, -0.0], f32)
    xx = np.array([0.0, 0.0, -1.0, -1.0, 0.0, 0.0, 1.0, 1.0], f32)
    print('atan2 special', atan2_lean(yy, xx, coef), np.arctan2(yy, xx))
