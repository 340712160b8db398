"""GPU tests (-m gpu) of the inflow thrust loss (round 5; BUILD-OWNED plant feature, no reference counterpart beyond the two sets of steady
speeds customEnv.py:13-18 records): F = K n|n| - Kl |n| u_a, parameters 26..31 of the public vector (DPENV_P_KLF_* / DPENV_P_KLR_*), carried as
two more rows of the per-env table and applied by the GENERAL per-env kernels only (VES_ENV_RND of dpenv_step / dpenv_rollout, the RND
instantiations and the one-wave forms of the closed loop).  Checked against the CPU oracle's restatement (oracle/dpenv_oracle_impl.h dpo_plant)
and, where no coefficient is in force, bit for bit against the kernels that do not carry the code."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H
from tests import tolerances as TOL
from tests.test_gpu_vessel_env import make_ac, oracle_step_subset, spread_indices, torch_

pytestmark = pytest.mark.gpu


def preset():
    import ml4ca_amd
    return np.asarray(ml4ca_amd.default_vessel('thrust_loss'), np.float32)


def test_65536_hulls_with_their_own_loss_coefficients_against_the_oracle():
    """every env its own hull AND its own six coefficients (every fourth env none), a current (the loss sees the speed through the water),
    states fast enough for the zero-thrust clamp to act; three steps, the oracle on > 4 096 spread envs"""
    torch = torch_()
    n = 65536 + 37
    rng = np.random.RandomState(41)
    hulls = H.random_hulls(rng, n, loss=0.15)
    env, orc = H.make_pair('final_cont', n, current=True)
    env.set_vessel_params(H.to_dev(hulls))
    assert np.array_equal(env.get_vessel_params().cpu().numpy(), hulls)          # the coefficients survive the packing too
    vc = rng.uniform(0, 0.3, n).astype(np.float32)
    beta = rng.uniform(-3, 3, n).astype(np.float32)
    env.set_current(H.to_dev(vc), H.to_dev(beta))
    idx = spread_indices(n, rng)
    st = H.random_state(rng, n, spread=0.6)
    st[3] = rng.uniform(-1.3, 1.3, n)                        # surge up to the bound: u_a large enough to cancel small thrusts
    ctr = np.zeros((2, n), np.int32)
    env.set_state(H.to_dev(st), H.to_dev(ctr))
    moved = 0
    for k in range(3):
        act = H.random_actions(rng, n, 7)
        g_st, g_ctr = env.get_state()
        obs, rew, done, _ = env.step(H.to_dev(act))
        g_st2, _ = env.get_state()
        torch.cuda.synchronize()
        (oo, orw, od), ost, _ = oracle_step_subset(orc, idx, g_st.cpu().numpy(), g_ctr.cpu().numpy(), act, hulls,
                                                   current=np.ascontiguousarray(np.stack([vc[idx], beta[idx]])))
        TOL.assert_close(obs.cpu().numpy()[idx], oo, TOL.OBS_FLOOR, what='obs step %d' % k)
        TOL.assert_close(rew.cpu().numpy()[idx], orw, TOL.REWARD_FLOOR, what='reward step %d' % k)
        TOL.assert_close(g_st2.cpu().numpy()[0:3, idx].T, ost[0:3].T, TOL.ETA_FLOOR, what='eta step %d' % k)
        TOL.assert_close(g_st2.cpu().numpy()[3:6, idx].T, ost[3:6].T, TOL.NU_FLOOR, what='nu step %d' % k)
        assert bool(TOL.done_agrees(done.cpu().numpy()[idx], od, oo, env.real_ss_bounds).all())
        # the loss is really in force: the same hulls without their coefficients give other velocities
        nol = hulls.copy()
        nol[26:32] = 0.0
        (o2, _, _), _, _ = oracle_step_subset(orc, idx, g_st.cpu().numpy(), g_ctr.cpu().numpy(), act, nol, current=np.ascontiguousarray(np.stack([vc[idx], beta[idx]])))
        differ = np.abs(o2[:, 3:6] - oo[:, 3:6]).max(axis=1) > 1e-5
        has = hulls[26:32, idx].max(axis=0) > 0
        assert differ[has].mean() > 0.9 and not differ[~has].any()
        moved += int(differ.sum())
    assert moved > 0


@pytest.mark.parametrize('mode,ext', [(m, e) for m in ('full', 'simple', 'limited', 'final_wrap', 'final_cont') for e in (True, False) if not (m == 'simple' and e)])
def test_thrust_loss_in_every_variant(mode, ext):
    torch = torch_()
    n = 1000 + 7
    rng = np.random.RandomState(3)
    hulls = H.random_hulls(rng, n, loss=0.1)
    env, orc = H.make_pair(mode, n, ext=ext)
    env.set_vessel_params(H.to_dev(hulls))
    st = H.random_state(rng, n, spread=0.5)
    st[3] = rng.uniform(-1.0, 1.0, n)
    ctr = np.zeros((2, n), np.int32)
    env.set_state(H.to_dev(st), H.to_dev(ctr))
    for k in range(3):
        act = H.random_actions(rng, n, env.num_actions)
        g_st, g_ctr = env.get_state()
        obs, rew, done, _ = env.step(H.to_dev(act))
        ost, octr = np.ascontiguousarray(g_st.cpu().numpy()), np.ascontiguousarray(g_ctr.cpu().numpy())
        oo, orw, od = orc.step(ost, octr, act, vessel_env=hulls)
        TOL.assert_close(obs.cpu().numpy(), oo, TOL.OBS_FLOOR[:env.num_states], what='%s obs step %d' % (mode, k))
        TOL.assert_close(rew.cpu().numpy(), orw, TOL.REWARD_FLOOR, what='%s reward step %d' % (mode, k))
        assert bool(TOL.done_agrees(done.cpu().numpy(), od, oo, env.real_ss_bounds).all())


def test_envs_without_a_coefficient_get_the_rows_of_the_kernels_without_the_loss_code():
    """One env of the batch has a coefficient: the general per-env kernels run for all of them (dpenv_dev.h) - the others' rows must be
    those of the plain per-env kernels bit for bit (-(0 |n|) u_a + F = F exactly).  Step, fused rollout (both forms) and the closed loop."""
    from ml4ca_amd.policy import policy_rollout
    torch = torch_()
    n, T = 3000 + 5, 24
    rng = np.random.RandomState(8)
    hulls = H.random_hulls(rng, n)
    lossy = hulls.copy()
    lossy[26:32, 77] = 0.08
    keep = np.ones(n, bool)
    keep[77] = False
    keep = torch.as_tensor(keep, device='cuda')
    st, ctr = H.to_dev(H.random_state(rng, n, 0.5)), H.to_dev(np.zeros((2, n), np.int32))
    acts = H.to_dev((rng.standard_normal((T, n, 7)) * 0.7).astype(np.float32))
    for one_wave in (False, True):
        kw = dict(auto_reset=True, max_ep_len=9, seed=4, step_one_wave=one_wave)
        a, _ = H.make_pair('final_cont', n, **kw)
        b, _ = H.make_pair('final_cont', n, **kw)
        a.set_vessel_params(H.to_dev(hulls))
        b.set_vessel_params(H.to_dev(lossy))
        for e in (a, b):
            e.set_state(st, ctr)
        for t in range(6):
            ra, rb = a.step(acts[t]), b.step(acts[t])
            assert all(torch.equal(x[keep], y[keep]) for x, y in zip(ra[:3], rb[:3])), t
        assert not torch.equal(a.get_state()[0][:, 77], b.get_state()[0][:, 77])
        for e in (a, b):
            e.set_state(st, ctr)
        oa, ob = a.rollout(acts), b.rollout(acts)
        assert all(torch.equal(x[:, keep], y[:, keep]) for x, y in zip(oa, ob))
    for prec, form in (('f16', 'two_wave'), ('f32', 'two_wave'), ('f16', 'one_wave')):
        outs = []
        for tab in (hulls, lossy):
            e, _ = H.make_pair('final_cont', n, auto_reset=True, max_ep_len=9, seed=4)
            e.set_vessel_params(H.to_dev(tab))
            e.reset()
            make_ac(9, 7, (80, 80, 80), seed=2, device=e.device).upload(e, precision=prec, launch_form=form)
            outs.append(policy_rollout(e, T, sample=True))
        for k in ('obs', 'act', 'rew', 'done', 'val', 'logp'):
            assert torch.equal(outs[0][k][:, keep], outs[1][k][:, keep]), (prec, form, k)


def test_the_preset_through_the_constructor_and_its_steady_speeds():
    """dpenv_default_vessel_ex(DPENV_VESSEL_THRUST_LOSS) as the single class (round 6: hull and coefficients as kernel arguments, the shared training form), steps like
    the oracle with that vessel, also on the way to the reference's second set of recorded speeds, +1.4 / -1.1 m/s (customEnv.py:17)."""
    import ml4ca_amd
    torch = torch_()
    p = preset()
    assert p[27] > 0 and p[30] > 0 and p[26] == 0 and p[16] < p[13]
    n = 512
    env, orc = H.make_pair('full', n, vessel_params=p, terminate=False, time_limit=False)
    assert np.array_equal(env.get_vessel_params().cpu().numpy(), np.tile(p[:, None], (1, n)))
    rng = np.random.RandomState(2)
    st = H.random_state(rng, n, 0.4)
    ctr = np.zeros((2, n), np.int32)
    env.set_state(H.to_dev(st), H.to_dev(ctr))
    act = H.random_actions(rng, n, 6)
    obs, rew, _, _ = env.step(H.to_dev(act))
    oo, orw, _ = orc.step(st, ctr, act)
    TOL.assert_close(obs.cpu().numpy(), oo, TOL.OBS_FLOOR, what='obs')
    TOL.assert_close(rew.cpu().numpy(), orw, TOL.REWARD_FLOOR, what='reward')
    # full ahead (first half of the batch) / full astern: stern thrusters +-100 %, azimuths 0, bow off  (full variant: n0 n1 n2 a0 a1 a2) - the
    # approach to the recorded speeds, +1.4 / -1.1 m/s (customEnv.py:17), step for step with the oracle.  100 steps (20 s): this hull is
    # not course-stable at speed, and where the oracle's exactly symmetric arithmetic keeps r = 0 for ever, the kernel's lean sincos leaves
    # r ~ 1e-11 that grows into a turn after ~40 s; the steady speeds themselves are pinned on the float64 oracle (tests/test_host_cpu.py)
    a = np.zeros((n, 6), np.float32)
    a[: n // 2, 1:3] = 1.0
    a[n // 2:, 1:3] = -1.0
    z = np.zeros_like(st)
    env.set_state(H.to_dev(z), H.to_dev(ctr))
    ost, octr = z.copy(), ctr.copy()
    ad = H.to_dev(a)
    for _ in range(100):
        env.step(ad)
        orc.step(ost, octr, a)
    g = env.get_state()[0].cpu().numpy()
    TOL.assert_close(g[3:6].T, ost[3:6].T, TOL.NU_FLOOR, what='nu after 100 steps')
    assert 1.25 < g[3, 0] < 1.40 and -1.10 < g[3, -1] < -0.90 and abs(g[5]).max() < 1e-5
    # classes cannot carry the coefficients; negative ones are no vessel
    with pytest.raises(Exception):
        ml4ca_amd.BatchedRevoltEnv(64, vessel_params=np.stack([p, p]))
    bad = p.copy()
    bad[27] = -0.01
    with pytest.raises(Exception):
        ml4ca_amd.BatchedRevoltEnv(64, vessel_params=bad)


@pytest.mark.parametrize('one_wave', [False, True])
def test_randomised_loss_coefficients_are_redrawn_like_the_oracle(one_wave):
    """the randomisation around the thrust-loss preset: the six coefficients are drawn with the hull (Philox words of their own) by the
    explicit reset and by the auto-reset of dpenv_step - table and rows against the oracle"""
    torch = torch_()
    n, rel = 1500 + 11, 0.2
    rng = np.random.RandomState(5)
    p = preset()
    env, orc = H.make_pair('final_cont', n, auto_reset=True, max_ep_len=7, seed=33, step_one_wave=one_wave)
    env.set_vessel_randomisation(rel, nominal=p)
    rt = H.rand_table(rel, nominal=p)
    hulls = np.ascontiguousarray(np.tile(rt[:O.NPARAM, None], (1, n)))
    ost, octr = orc.new_state(n)
    env.reset()
    orc.reset(ost, octr, vessel_env=hulls, rand_tab=rt)
    assert np.array_equal(env.get_vessel_params().cpu().numpy(), hulls)
    k = hulls[[27, 28, 30, 31]] / rt[[27, 28, 30, 31], None]
    assert 1 - rel - 1e-6 <= k.min() < 1 - 0.9 * rel and 1 + 0.9 * rel < k.max() <= 1 + rel + 1e-6 and not (hulls[[26, 29]] != 0).any()
    assert abs(np.corrcoef(hulls[27], hulls[13])[0, 1]) < 0.1 and abs(np.corrcoef(hulls[27], hulls[30])[0, 1]) < 0.1
    for t in range(16):
        act = H.random_actions(rng, n, 7, scale=1.0)
        g_st, g_ctr = env.get_state()
        ost, octr = np.ascontiguousarray(g_st.cpu().numpy()), np.ascontiguousarray(g_ctr.cpu().numpy())
        obs, rew, done, _ = env.step(H.to_dev(act))
        oo, orw, od = orc.step(ost, octr, act, vessel_env=hulls, rand_tab=rt)
        assert bool(TOL.done_agrees(done.cpu().numpy(), od, oo, env.real_ss_bounds).all())
        agree = (done.cpu().numpy() != 0) == (od != 0)
        assert agree.mean() > 0.999
        assert np.array_equal(env.get_vessel_params().cpu().numpy()[:, agree], hulls[:, agree]), t
        TOL.assert_close(obs.cpu().numpy()[agree], oo[agree], TOL.OBS_FLOOR, what='obs step %d' % t)
        TOL.assert_close(rew.cpu().numpy()[agree], orw[agree], TOL.REWARD_FLOOR, what='reward step %d' % t)
        hulls = np.ascontiguousarray(env.get_vessel_params().cpu().numpy())


@pytest.mark.parametrize('randomise', [0.0, 0.2])
@pytest.mark.parametrize('one_wave', [False, True])
def test_fused_rollout_with_thrust_loss_equals_single_steps(one_wave, randomise):
    torch = torch_()
    n, T = 1500 + 3, 40
    kw = dict(auto_reset=True, max_ep_len=9, seed=6, reset_acts=True)
    envs = []
    for ow in (one_wave, False):
        e, _ = H.make_pair('final_cont', n, step_one_wave=ow, **kw)
        if randomise > 0:
            e.set_vessel_randomisation(randomise, nominal=preset())
        else:
            e.set_vessel_params(H.to_dev(H.random_hulls(np.random.RandomState(1), n, loss=0.1)))
        e.reset()
        envs.append(e)
    a, b = envs
    g = torch.Generator(device=a.device).manual_seed(1)
    acts = torch.randn((T, n, 7), generator=g, device=a.device) * 0.8
    o, r, d = a.rollout(acts)
    for t in range(T):
        o1, r1, d1, _ = b.step(acts[t])
        assert torch.equal(o1, o[t]) and torch.equal(r1, r[t]) and torch.equal(d1, d[t]), t
    sa, sb = a.get_state(), b.get_state()
    assert torch.equal(sa[0], sb[0]) and torch.equal(sa[1], sb[1]) and int(sa[1][1].min()) >= 4
    assert torch.equal(a.get_vessel_params(), b.get_vessel_params())


@pytest.mark.parametrize('randomise', [0.0, 0.15])
@pytest.mark.parametrize('precision,form', [('f16', 'two_wave'), ('f32_actor', 'two_wave'), ('f32', 'two_wave'), ('f16', 'one_wave'), ('f32', 'one_wave')])
@pytest.mark.parametrize('n', [1000 + 9, 40000])
def test_closed_loop_with_thrust_loss_replays_through_single_steps(precision, form, n, randomise):
    """every closed-loop form (both workgroup geometries) on hulls with a thrust loss - fixed per-env hulls, and the randomisation around
    the preset -, with a drifting current: the stored actions replayed through dpenv_step give the same rows, state and table"""
    from ml4ca_amd.policy import policy_rollout, policy_launch_form
    torch = torch_()
    T = 30
    kw = dict(auto_reset=True, max_ep_len=11, seed=12, reset_acts=True, current=True, current_drift=True)
    envs = []
    for _ in range(2):
        e, _ = H.make_pair('final_cont', n, **kw)
        if randomise > 0:
            e.set_vessel_randomisation(randomise, nominal=preset())
        else:
            e.set_vessel_params(H.to_dev(H.random_hulls(np.random.RandomState(1), n, loss=0.1)))
        e.set_current(torch.full((n,), 0.15, device=e.device), torch.full((n,), 1.0, device=e.device))
        e.reset()
        envs.append(e)
    env, env2 = envs
    make_ac(9, 7, (80, 80, 80), seed=2, device=env.device).upload(env, precision=precision, launch_form=form)
    assert policy_launch_form(env)[0] == form
    out = policy_rollout(env, T, sample=True)
    for t in range(T):
        o, r, d, _ = env2.step(out['act'][t].contiguous())
        nxt = out['obs'][t + 1] if t + 1 < T else out['last_obs']
        assert torch.equal(r, out['rew'][t]) and torch.equal(d, out['done'][t]) and torch.equal(o, nxt), t
    sa, sb = env.get_state(), env2.get_state()
    assert torch.equal(sa[0], sb[0]) and torch.equal(sa[1], sb[1]) and int(sa[1][1].min()) >= 3
    assert torch.equal(env.get_vessel_params(), env2.get_vessel_params())


def test_thrust_loss_in_other_closed_loop_variants_runs_the_one_wave_kernels():
    from ml4ca_amd.policy import policy_rollout
    torch = torch_()
    n, T = 700, 24
    for mode, ext, act in (('limited', False, 'leaky'), ('final_cont', True, 'tanh')):
        envs = []
        for _ in range(2):
            e, _ = H.make_pair(mode, n, auto_reset=True, max_ep_len=7, seed=3, ext=ext)
            e.set_vessel_params(preset())                    # one vector: every env the same block
            e.reset()
            envs.append(e)
        env, env2 = envs
        make_ac(env.num_states, env.num_actions, (80, 80, 80), seed=2, device=env.device, activation=act).upload(env, precision='f16')
        out = policy_rollout(env, T, sample=True)
        for t in range(T):
            o, r, d, _ = env2.step(out['act'][t].contiguous())
            nxt = out['obs'][t + 1] if t + 1 < T else out['last_obs']
            assert torch.equal(r, out['rew'][t]) and torch.equal(d, out['done'][t]) and torch.equal(o, nxt), (mode, t)


# (round 6: dpenv_set_vessel_params no longer reads a word back and CAN be recorded into a graph - tests/test_gpu_round6.py
#  test_set_vessel_params_recorded_into_a_graph replaces the refusal test that stood here)
