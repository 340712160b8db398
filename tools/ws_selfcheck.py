"""Run the two-wave closed-loop kernel of a DPENV_WS_SELFCHECK build (DPENV_LIB=...) and print what its in-kernel
double evaluation of env_step caught: which quantity of which lane differed between the evaluation that ran beside the
partner wave's MFMAs and the one that ran after them.  Usage: DPENV_LIB=build/wsdiag/selfcheck_slp.so python tools/ws_selfcheck.py [reps]"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from tests import helpers as H
from tests.test_gpu_policy import make_ac
from ml4ca_amd.policy import policy_rollout
from ml4ca_amd import _lib

FIELDS = ['N', 'E', 'psi', 'u', 'v', 'r', 'sn', 'cs', 'reward', 'o0', 'o1', 'o2', 'o3', 'o4', 'o5', 'o6', 'o7', 'o8', 'done', 'ang_port']
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n, T = 2000 + 11, 45
kw = dict(auto_reset=True, max_ep_len=2 * n_steps, seed=8, n_steps=n_steps)
lib = _lib.load()
buf = torch.zeros(4 + 2000 * 60, dtype=torch.int32, device='cuda:0')
lib.dpenv_debug_set_buffer.argtypes = [C.c_void_p]
lib.dpenv_debug_set_buffer(buf.data_ptr())
seen = 0
events = {}
lane_hist = np.zeros(64, int)
field_hist = np.zeros(20, int)
for rep in range(reps):
    env, _ = H.make_pair('final_cont', n, **kw)
    ac = make_ac(9, 7, (80, 80, 80), seed=2, device=env.device).upload(env)
    g = torch.Generator(device=env.device).manual_seed(1)
    env.reset()
    refs = torch.randn((2, 3, n), generator=g, device=env.device)
    policy_rollout(env, T, noise=None, switch_steps=(3, 30), refs=refs)
    torch.cuda.synchronize()
    cnt = int(buf[0])
    if cnt > seen:
        recs = buf[4:4 + min(cnt, 2000) * 60].cpu().numpy().view(np.uint32).reshape(-1, 60)
        for r in recs[seen:min(cnt, 2000)]:
            i, t, mask, lane = int(r[0]), int(r[1]), int(r[2]), int(r[3])
            fa = r[4:24].view(np.float32); fb = r[24:44].view(np.float32)
            names = [FIELDS[k] for k in range(20) if mask >> k & 1]
            lane_hist[lane] += 1
            events.setdefault((rep, i // 64, t), []).append((fa.copy(), fb.copy(), r[44:50].view(np.float32).copy(), r[50:57].view(np.float32).copy()))
            for k in range(20):
                if mask >> k & 1:
                    field_hist[k] += 1
            if seen < 60:
                # which term of the position sum would explain the difference?  dN = h * (term): cs u and -sn v of the state reached
                h = 0.01
                print('rep %d env %d lane %d t %d fields %s | dN/h %.6g  cs*u %.6g  -sn*v %.6g | dE/h %.6g sn*u %.6g cs*v %.6g' % (
                    rep, i, lane, t, names, (float(fa[0]) - float(fb[0])) / h, fb[7] * fb[3], -fb[6] * fb[4],
                    (float(fa[1]) - float(fb[1])) / h, fb[6] * fb[3], fb[7] * fb[4]))
                for k in range(20):
                    if mask >> k & 1:
                        print('    %-8s beside-MFMA %.9g  alone %.9g  diff %.3g' % (FIELDS[k], fa[k], fb[k], float(fa[k]) - float(fb[k])))
            seen += 1
print('launches', reps, 'n_substeps', n_steps, 'mismatch records', seen, 'wave-step events ~', seen / 16.0, 'per launch', seen / 16.0 / reps)
print('by lane:', {l: int(c) for l, c in enumerate(lane_hist) if c})
print('by field:', {FIELDS[k]: int(c) for k, c in enumerate(field_hist) if c})

# ---- which sub-step lost its term?  float64 re-statement of env_plant from the recorded pre-state and action --------
vp = _lib.default_vessel().astype(np.float64)
P = _lib.P
m11, m22, m23, m33 = vp[P['M11']], vp[P['M22']], vp[P['M23']], vp[P['M33']]
det = m22 * m33 - m23 * m23
inv11, i22, i23, i33 = 1.0 / m11, m33 / det, -m23 / det, m22 / det
Xu, Xuu, Yv, Yvv, Yr, Nv, Nr, Nrr, Nuv, Yur = (vp[P[k]] for k in ('XU', 'XUU', 'YV', 'YVV', 'YR', 'NV', 'NR', 'NRR', 'NUV', 'YUR'))
hist_k = {}
for (rep, wave, t), lst in sorted(events.items()):
    pre = np.array([e[2] for e in lst], np.float64)          # [m][6]
    act = np.array([e[3] for e in lst], np.float64)
    dN = np.array([float(e[0][0]) - float(e[1][0]) for e in lst])
    thr = np.clip(act[:, 0:3] * 100.0, -100, 100)
    ang = np.stack([np.full(len(lst), np.pi / 2), np.arctan2(act[:, 3], act[:, 4]), np.arctan2(act[:, 5], act[:, 6])], 1)
    Kf = np.array([vp[P['KF_BOW']], vp[P['KF_PORT']], vp[P['KF_STAR']]]); Kr = np.array([vp[P['KR_BOW']], vp[P['KR_PORT']], vp[P['KR_STAR']]])
    lx = np.array([vp[P['LX_BOW']], vp[P['LX_PORT']], vp[P['LX_STAR']]]); ly = np.array([vp[P['LY_BOW']], vp[P['LY_PORT']], vp[P['LY_STAR']]])
    F = np.where(thr >= 0, Kf, Kr) * np.abs(thr) * thr
    tx = (np.cos(ang) * F).sum(1); ty = (np.sin(ang) * F).sum(1); tn = ((lx * np.sin(ang) - ly * np.cos(ang)) * F).sum(1)
    psi, u, v, r = pre[:, 2].copy(), pre[:, 3].copy(), pre[:, 4].copy(), pre[:, 5].copy()
    sn, cs = np.sin(psi), np.cos(psi)
    h = 0.01
    yur, nuv = Yur + m11, Nuv - m11
    terms = []
    for k in range(n_steps):
        q = m22 * v + m23 * r
        fx = -(Xuu * abs(u) + Xu) * u + tx + q * r
        fy = -(Yvv * abs(v) + Yv) * v + ty - (yur * u + Yr) * r
        fn = -q * u + tn - (nuv * u + Nv) * v - (Nrr * abs(r) + Nr) * r
        u = u + h * inv11 * fx
        v, r = v + h * i22 * fy + h * i23 * fn, r + h * i23 * fy + h * i33 * fn
        terms.append((cs * u, -sn * v))
        d = h * r
        psi = psi + d
        cs, sn = cs - sn * d - cs * d * d / 2, sn + cs * d - sn * d * d / 2
    errs = [float(np.sum((dN / h + tk[0]) ** 2)) for tk in terms]
    errs2 = [float(np.sum((dN / h + tk[1]) ** 2)) for tk in terms]
    k = int(np.argmin(errs))
    hist_k[k] = hist_k.get(k, 0) + 1
    if len(hist_k) and sum(hist_k.values()) <= 12:
        srt = np.argsort(errs)
        print('event rep %d wave %d t %d lanes %d: best sub-step k=%d (rms %.3g), next k=%d (rms %.3g); best for the -sn*v term rms %.3g' % (
            rep, wave, t, len(lst), k, (errs[k] / len(lst)) ** 0.5, int(srt[1]) if len(srt) > 1 else -1, (errs[int(srt[1])] / len(lst)) ** 0.5 if len(srt) > 1 else 0, (min(errs2) / len(lst)) ** 0.5))
print('sub-step index of the lost cs*u term (0-based):', dict(sorted(hist_k.items())))
