"""The C ABI from plain C (no Python in the process that steps the env): builds examples/c_abi_demo.c with gcc against
include/dpenv.h + libdpenv.so + the HIP runtime and runs it on the GPU."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_program_steps_the_env(tmp_path):
    exe = str(tmp_path / 'c_abi_demo')
    lib = os.path.join(ROOT, 'ml4ca_amd', 'lib')
    subprocess.check_call(['gcc', '-std=c11', '-O2', '-I', os.path.join(ROOT, 'include'), '-I', '/opt/rocm/include',
                           os.path.join(ROOT, 'examples', 'c_abi_demo.c'), '-L', lib, '-ldpenv', '-L', '/opt/rocm/lib', '-lamdhip64',
                           '-Wl,-rpath,' + lib, '-Wl,-rpath,/opt/rocm/lib', '-o', exe])
    out = subprocess.check_output([exe, '65536', '1000', '4096'], timeout=120).decode()
    m = re.search(r'= ([0-9.e+]+) env-steps/s .* mean reward (-?[0-9.]+); faults (\d+)', out)
    assert m, out
    assert float(m.group(1)) > 1e9 and int(m.group(3)) == 0 and -3.0 < float(m.group(2)) < 3.5, out
    # the fused entry points from the same C program (dpenv_rollout with a setpoint switch, dpenv_gae_stats, dpenv_adv_apply_stats): its
    # checksums against the same calls through the Python binding - same library, same kernels, same bits
    import numpy as np
    import torch
    import ml4ca_amd
    from ml4ca_amd import rollout as RO
    f = re.search(r'fused: (\d+) envs x (\d+) steps .* switch at step (\d+); checksums obs (\S+) rew (\S+) done (\d+) gae_stats (\S+) (\S+) '
                  r'adv_norm_sum (\S+) adv_norm_sq (\S+) ret (\S+)', out)
    assert f, out
    n, T, sw = int(f.group(1)), int(f.group(2)), int(f.group(3))
    vals = np.empty(T * n * 7, np.float32)
    a_, c_ = np.uint64(1664525), np.uint64(1013904223)
    s = np.uint64(2024)
    for i in range(vals.size):                                                  # the demo's LCG
        s = (s * a_ + c_) & np.uint64(0xffffffff)
        vals[i] = (np.float32(int(s) >> 8) / np.float32(16777216.0) - np.float32(0.5)) * np.float32(1.6)
    env = ml4ca_amd.BatchedRevoltEnv(n, auto_reset=True, seed=7)
    env.reset()
    refs = torch.tensor([1.5, -0.5, 0.1], device=env.device).reshape(1, 3, 1).repeat(1, 1, n).contiguous()
    obs, rew, done = env.rollout(torch.from_numpy(vals.reshape(T, n, 7)).to(env.device), switch_steps=(sw,), refs=refs)
    stats = torch.zeros(2, dtype=torch.float64, device=env.device)
    adv, ret = RO.gae(rew, (rew * 0.5).contiguous(), end=done, stats=stats)
    st = stats.cpu().numpy().copy()
    RO.normalize_advantages(adv, stats=stats)
    got = [float(obs.double().sum()), float(rew.double().sum()), int((done != 0).sum()), st[0], st[1], float(adv.double().sum()),
           float((adv.double() ** 2).sum()), float(ret.double().sum())]
    want = [float(f.group(k)) for k in (4, 5)] + [int(f.group(6))] + [float(f.group(k)) for k in (7, 8, 9, 10, 11)]
    assert got[2] == want[2] and got[2] > 0, (got, want)
    assert got[3] == want[3] and got[4] == want[4], (got, want)                 # the device-side statistics: bit for bit
    for g_, w_ in zip(got, want):                                               # host sums of float32 rows in double (order of summation differs)
        assert abs(g_ - w_) <= 1e-9 * max(1.0, abs(w_)), (got, want)
    assert abs(got[5]) < 1e-2 * n * T and abs(got[6] / (n * T) - 1.0) < 1e-3   # normalised: mean 0, variance 1
    # per-env vessels from the same C program: the thrust-loss preset through dpenv_create (read back exactly), explicit per-env blocks with
    # their own loss coefficients, then the randomisation around the preset - checksums against the same calls through the Python binding
    assert 'refused calls left the handle alone' in out, out
    v = re.search(r'vessels: (\d+) envs x (\d+) steps .* preset read back (\w+); refused calls left the handle alone; per-env blocks: checksums obs (\S+) rew (\S+) table (\S+); '
                  r'randomised: obs (\S+) rew (\S+) table (\S+)', out)
    assert v and v.group(3) == 'exactly', out
    n3, S3 = int(v.group(1)), int(v.group(2))

    def lcg(seed, count):
        x, vals = seed, np.empty(count, np.float32)
        for i in range(count):
            x = (x * 1664525 + 1013904223) & 0xffffffff
            vals[i] = np.float32(x >> 8) / np.float32(16777216.0) - np.float32(0.5)
        return vals

    preset = ml4ca_amd.default_vessel('thrust_loss')
    env = ml4ca_amd.BatchedRevoltEnv(n3, auto_reset=True, max_ep_len=16, seed=11, vessel_params=preset)     # customEnv.py:83: 16 -> 8 env steps, the demo's cfg.max_ep_len
    assert np.array_equal(env.get_vessel_params().cpu().numpy(), np.tile(preset[:, None], (1, n3)))
    tab = (preset[:, None] * (np.float32(1.0) + np.float32(0.2) * lcg(4711, 32 * n3).reshape(32, n3))).astype(np.float32)
    env.set_vessel_params(torch.from_numpy(tab).to(env.device))
    for phase in range(2):
        if phase == 1:
            env.set_vessel_randomisation(0.15, nominal=preset)
        env.reset()
        acts = (lcg(99 + phase, S3 * n3 * 7) * np.float32(1.6)).reshape(S3, n3, 7)
        osum = rsum = 0.0
        for t in range(S3):
            o, r, _, _ = env.step(torch.from_numpy(acts[t]).to(env.device))
            osum += float(o.double().sum())
            rsum += float(r.double().sum())
        tsum = float(env.get_vessel_params().double().sum())
        for g_, w_ in zip((osum, rsum, tsum), (float(v.group(4 + 3 * phase)), float(v.group(5 + 3 * phase)), float(v.group(6 + 3 * phase)))):
            assert abs(g_ - w_) <= 1e-9 * max(1.0, abs(w_)), (phase, g_, w_)
