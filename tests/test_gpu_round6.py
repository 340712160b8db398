"""Round 6 (VERDICT r05 items 2, 5, 6 and ADVICE r05): the per-env table's setter without a read-back (refused calls leave the handle alone, the
setter can be recorded into a graph), the checkpoint / restore path of randomised hulls, the shared-hull thrust-loss kernels (coefficients as
kernel arguments) against the per-env form of the same hull, and the per-episode randomisation of the current through every reset path -
against the fp32 oracle (oracle/dpenv_oracle_impl.h: dpo_draw_current, reset_one) and between the launch forms bit for bit."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H
from tests import tolerances as TOL

pytestmark = pytest.mark.gpu


def torch_():
    import torch
    return torch


def preset():
    import ml4ca_amd
    return np.asarray(ml4ca_amd.default_vessel('thrust_loss'), np.float32)


def make_ac(*a, **k):
    from ml4ca_amd.policy import ActorCritic
    return ActorCritic(*a, **k)


def _oracle_step_check(env, orc, rng, steps, hulls=None, what=''):
    """`steps` steps of env against the oracle re-seeded from the GPU state before every step (per-env hull table `hulls`, or the oracle's own)"""
    n = env.n_envs
    for t in range(steps):
        act = H.random_actions(rng, n, env.num_actions, scale=0.9)
        g_st, g_ctr = env.get_state()
        ost, octr = np.ascontiguousarray(g_st.cpu().numpy()), np.ascontiguousarray(g_ctr.cpu().numpy())
        obs, rew, done, _ = env.step(H.to_dev(act))
        oo, orw, od = orc.step(ost, octr, act, vessel_env=hulls)
        assert bool(TOL.done_agrees(done.cpu().numpy(), od, oo, env.real_ss_bounds).all()), (what, t)
        TOL.assert_close(obs.cpu().numpy(), oo, TOL.OBS_FLOOR, what='%s obs step %d' % (what, t))
        TOL.assert_close(rew.cpu().numpy(), orw, TOL.REWARD_FLOOR, what='%s reward step %d' % (what, t))


# ---------------------------------------------------------------------------------------------------------------- item 2: the setter
@pytest.mark.parametrize('kind', ['thrust_loss_preset', 'randomisation_on', 'table_with_loss'])
def test_a_refused_set_vessel_params_leaves_loss_and_redraws_in_force(kind):
    """VERDICT / ADVICE r05: dpenv_set_vessel_params used to clear `randomise` and `loss_on` BEFORE it could refuse the call, so a refused call
    silently switched the thrust loss and the hull re-draws off.  Now every refusal comes first: after one the handle steps exactly as before -
    with the loss (against the oracle), and hulls are still re-drawn."""
    import ml4ca_amd
    from ml4ca_amd import _lib
    torch = torch_()
    n = 1000 + 7
    rng = np.random.RandomState(3)
    kw = dict(auto_reset=(kind == 'randomisation_on'), max_ep_len=6 if kind == 'randomisation_on' else 800, seed=4)
    if kind == 'thrust_loss_preset':
        env, orc = H.make_pair('final_cont', n, vessel_params=preset(), **kw)
        hulls = None
    else:
        env, orc = H.make_pair('final_cont', n, **kw)
        if kind == 'randomisation_on':
            env.set_vessel_randomisation(0.15, nominal=preset())
            hulls = None
        else:
            hulls = H.random_hulls(rng, n, loss=0.1)
            env.set_vessel_params(H.to_dev(hulls))
    env.reset()
    table = H.to_dev(H.random_hulls(rng, n))
    # three ways to be refused: unknown flag bits; KEEP_RANDOMISATION without a table; KEEP_RANDOMISATION without the randomisation in force
    refusals = [lambda: env.lib.dpenv_set_vessel_params_ex(env._h, env._ptr(table), 0x80, env._stream()),
                lambda: env.lib.dpenv_set_vessel_params_ex(env._h, None, _lib.VESSEL_KEEP_RANDOMISATION, env._stream())]
    if kind != 'randomisation_on':
        refusals.append(lambda: env.lib.dpenv_set_vessel_params_ex(env._h, env._ptr(table), _lib.VESSEL_KEEP_RANDOMISATION, env._stream()))
    for call in refusals:
        assert call() == _lib.EINVAL
        assert b'dpenv_set_vessel_params_ex' in env.lib.dpenv_last_error(env._h) or b'KEEP_RANDOMISATION' in env.lib.dpenv_last_error(env._h)
    if kind == 'randomisation_on':
        before = env.get_vessel_params().clone()
        ep0 = env.get_state()[1][1].clone()
        for t in range(8):
            env.step(H.to_dev(H.random_actions(rng, n, 7)))
        moved = env.get_state()[1][1] > ep0
        after = env.get_vessel_params()
        assert bool(moved.any()) and bool((after[:, moved] != before[:, moved]).any(dim=0).all())      # every reset env got a new hull
        assert torch.equal(after[:, ~moved], before[:, ~moved])
        assert bool((after[26:32] > 0).any())                                                            # ... with its loss coefficients
    else:
        _oracle_step_check(env, orc, rng, 4, hulls=hulls, what=kind)
        # and the loss really is in those rows: the same states through a handle without it differ
        ref, _ = H.make_pair('final_cont', n, seed=4)
        if hulls is not None:
            h0 = hulls.copy(); h0[26:32] = 0
            ref.set_vessel_params(H.to_dev(h0))
        st, ctr = env.get_state()
        st = st.clone(); st[3] = 1.0                                                                     # making way: inflow at the stern thrusters
        act = H.to_dev(np.tile(np.array([[0.0, 0.9, 0.9, 0.0, 1.0, 0.0, 1.0]], np.float32), (n, 1)))
        ref.set_state(st, ctr); env.set_state(st, ctr)
        o_l, _, _, _ = env.step(act)
        o_0, _, _, _ = ref.step(act)
        assert float((o_l[:, 3] - o_0[:, 3]).abs().max()) > 1e-3


@pytest.mark.parametrize('loss', [0.0, 0.1])
def test_set_vessel_params_recorded_into_a_graph(loss):
    """The setter no longer reads a word back: it is stream-ordered and can be recorded.  A graph of {set the table, 3 steps} replayed gives the
    rows of the same calls made eagerly on a second handle - with thrust-loss coefficients in the table and without (a recorded setter leaves
    the question open: the general kernels then apply the coefficients, zeros included, which change no row)."""
    torch = torch_()
    n = 2048 + 5
    rng = np.random.RandomState(11)
    hulls = H.to_dev(H.random_hulls(rng, n, loss=loss))
    a, _ = H.make_pair('final_cont', n, seed=2)
    b, orc = H.make_pair('final_cont', n, seed=2)
    for e in (a, b):
        e.reset()
    acts = [H.to_dev(H.random_actions(rng, n, 7)) for _ in range(3)]
    outs = [(torch.empty((n, 9), device=a.device), torch.empty(n, device=a.device), torch.empty(n, dtype=torch.uint8, device=a.device)) for _ in range(3)]
    st0, ctr0 = a.get_state()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        a.set_vessel_params(hulls)                 # warm-up of the launch path outside the capture
        for k in range(3):
            a.step(acts[k], out=outs[k])
        torch.cuda.synchronize()
        a.set_vessel_params(None)
        a.set_state(st0, ctr0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            a.set_vessel_params(hulls)
            for k in range(3):
                a.step(acts[k], out=outs[k])
    torch.cuda.current_stream().wait_stream(side)
    for o in outs:
        for t in o:
            t.zero_()
    g.replay()
    torch.cuda.synchronize()
    b.set_vessel_params(hulls)
    for k in range(3):
        o, r, d, _ = b.step(acts[k])
        assert torch.equal(o, outs[k][0]) and torch.equal(r, outs[k][1]) and torch.equal(d, outs[k][2]), k
    sa, sb = a.get_state(), b.get_state()
    assert torch.equal(sa[0], sb[0]) and torch.equal(a.get_vessel_params(), hulls)
    # and those rows are the oracle's for these hulls
    _oracle_step_check(b, orc, rng, 2, hulls=np.ascontiguousarray(hulls.cpu().numpy()), what='after the graph')


def test_plain_per_env_kernels_still_selected_when_no_env_has_a_loss():
    """the answer to "does any env have a coefficient" reaches the host behind an event: the first launch after the setter takes it, and a table
    without coefficients runs the plain per-env kernels (rows bit-identical to the general form either way - checked against a handle that is
    forced onto the general form by one env's coefficient elsewhere in the table)"""
    torch = torch_()
    n = 3000
    rng = np.random.RandomState(2)
    hulls = H.random_hulls(rng, n)
    a, _ = H.make_pair('final_cont', n, seed=9)
    b, _ = H.make_pair('final_cont', n + 64, seed=9)
    hb = np.concatenate([hulls, H.random_hulls(rng, 64, loss=0.2)], 1)
    hb[26:32, :n] = 0
    a.set_vessel_params(H.to_dev(hulls)); b.set_vessel_params(H.to_dev(hb))
    a.reset(); b.reset()
    st, ctr = a.get_state()
    sb, cb = b.get_state()
    sb[:, :n] = st; cb[:, :n] = ctr
    b.set_state(sb, cb)
    for t in range(5):
        act = H.random_actions(rng, n + 64, 7)
        oa, ra, da, _ = a.step(H.to_dev(act[:n]))
        ob, rb, db, _ = b.step(H.to_dev(act))
        assert torch.equal(oa, ob[:n]) and torch.equal(ra, rb[:n]) and torch.equal(da, db[:n]), t


# ------------------------------------------------------------------------------------------------ checkpoint of randomised hulls
@pytest.mark.parametrize('with_loss', [False, True])
def test_checkpoint_restore_of_randomised_hulls_continues_bit_for_bit(with_loss):
    """include/dpenv.h: a checkpoint = dpenv_get_state (state + counters) + dpenv_get_vessel_params (+ the rng counters); restore = fresh handle,
    dpenv_set_vessel_randomisation(nominal, range), dpenv_set_vessel_params_ex(table, DPENV_VESSEL_KEEP_RANDOMISATION), dpenv_set_state: the run
    continues exactly like the uninterrupted one - rows, state, and the hulls drawn by every later reset."""
    torch = torch_()
    n, k, m = 2000 + 3, 13, 25
    rng = np.random.RandomState(8)
    nominal = preset() if with_loss else None
    kw = dict(auto_reset=True, max_ep_len=9, seed=31, env_id_base=123456, reset_acts=True)

    def fresh():
        e, _ = H.make_pair('final_cont', n, **kw)
        e.set_vessel_randomisation(0.2, nominal=nominal)
        return e

    a = fresh()
    a.reset()
    acts = [H.to_dev(H.random_actions(rng, n, 7, scale=1.0)) for _ in range(k + m)]
    for t in range(k):
        a.step(acts[t])
    st, ctr = a.get_state()
    st, ctr, hulls = st.clone(), ctr.clone(), a.get_vessel_params().clone()
    assert int(ctr[1].min()) >= 1 and not torch.equal(hulls[:, 0], hulls[:, 1])
    b = fresh()
    b.set_vessel_params(hulls, keep_randomisation=True)
    b.set_state(st, ctr)
    for t in range(k, k + m):
        oa, ra, da, _ = a.step(acts[t])
        ob, rb, db, _ = b.step(acts[t])
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db), t
    sa, sb = a.get_state(), b.get_state()
    assert torch.equal(sa[0], sb[0]) and torch.equal(sa[1], sb[1])
    assert torch.equal(a.get_vessel_params(), b.get_vessel_params())
    assert int(sa[1][1].min()) >= 3 and not torch.equal(a.get_vessel_params(), hulls)      # hulls moved on after the restore
    # the plain setter, as before, ends the re-draws: the table then stays
    c = fresh()
    c.set_vessel_params(hulls)
    c.set_state(st, ctr)
    for t in range(k, k + m):
        c.step(acts[t])
    assert torch.equal(c.get_vessel_params(), hulls)


# ------------------------------------------------------------------------------------------------ item 5: the shared hull with its thrust loss
@pytest.mark.parametrize('mode,ext', [('final_cont', True), ('final_wrap', False), ('limited', True), ('full', True), ('simple', False)])
@pytest.mark.parametrize('auto_reset,one_wave', [(False, False), (True, False), (True, True)])
def test_shared_hull_thrust_loss_step_equals_the_per_env_form_of_the_same_hull(mode, ext, auto_reset, one_wave):
    """dpenv_create(cfg, thrust-loss preset, 1 class): dpenv_step takes hull and coefficients from its arguments (step_kernel<.., VES_ARGS_LOSS>); the
    rows are those of the per-env form with every env on that hull (the general per-env kernels of round 5) bit for bit - every variant, with
    and without the reset wave - and the oracle's."""
    torch = torch_()
    n = 1500 + 5
    rng = np.random.RandomState(7)
    kw = dict(ext=ext, auto_reset=auto_reset, max_ep_len=7 if auto_reset else 800, seed=5, step_one_wave=one_wave, current=True)
    a, orc = H.make_pair(mode, n, vessel_params=preset(), **kw)
    b, _ = H.make_pair(mode, n, **kw)
    b.set_vessel_params(preset())
    cur = (torch.full((n,), 0.2, device=a.device), torch.full((n,), 2.3, device=a.device))
    for e in (a, b):
        e.set_current(*cur)
        e.reset()
    for t in range(12):
        act = H.to_dev(H.random_actions(rng, n, a.num_actions, scale=1.0))
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db), t
    sa, sb = a.get_state(), b.get_state()
    assert torch.equal(sa[0], sb[0]) and torch.equal(sa[1], sb[1])
    assert torch.equal(a.get_vessel_params(), b.get_vessel_params())                       # (the class hull, read back through the table image)
    if not auto_reset:
        # against the oracle (its vessel = the preset), current on
        g_st, g_ctr = a.get_state()
        ost, octr = np.ascontiguousarray(g_st.cpu().numpy()), np.ascontiguousarray(g_ctr.cpu().numpy())
        act = H.random_actions(rng, n, a.num_actions, scale=0.9)
        obs, rew, done, _ = a.step(H.to_dev(act))
        oo, orw, od = orc.step(ost, octr, act, current=np.stack([np.full(n, 0.2, np.float32), np.full(n, 2.3, np.float32)]))
        TOL.assert_close(obs.cpu().numpy(), oo, TOL.OBS_FLOOR[:oo.shape[1]], what='obs')
        TOL.assert_close(rew.cpu().numpy(), orw, TOL.REWARD_FLOOR, what='reward')


@pytest.mark.parametrize('one_wave', [False, True])
def test_shared_hull_thrust_loss_fused_rollout_equals_single_steps(one_wave):
    torch = torch_()
    n, T = 1200 + 3, 30
    kw = dict(auto_reset=True, max_ep_len=8, seed=6, reset_acts=True, vessel_params=preset())
    a, _ = H.make_pair('final_cont', n, step_one_wave=one_wave, **kw)
    b, _ = H.make_pair('final_cont', n, **kw)
    for e in (a, b):
        e.reset()
    g = torch.Generator(device=a.device).manual_seed(1)
    acts = torch.randn((T, n, 7), generator=g, device=a.device) * 0.9
    o, r, d = a.rollout(acts)
    for t in range(T):
        o1, r1, d1, _ = b.step(acts[t])
        assert torch.equal(o1, o[t]) and torch.equal(r1, r[t]) and torch.equal(d1, d[t]), t
    sa, sb = a.get_state(), b.get_state()
    assert torch.equal(sa[0], sb[0]) and torch.equal(sa[1], sb[1])


@pytest.mark.parametrize('precision,form', [('f16', 'two_wave'), ('f32_actor', 'two_wave'), ('f32', 'two_wave'), ('f16', 'one_wave')])
@pytest.mark.parametrize('n', [1000 + 9, 40000])
def test_shared_hull_thrust_loss_closed_loop_replays_through_single_steps(precision, form, n):
    """the two-wave closed loop's shared-loss instantiation (both workgroup geometries, all three arithmetics) and the one-wave kernels (which read the
    class hull from its table image): the stored actions replayed through dpenv_step give the same rows and the same final state"""
    from ml4ca_amd.policy import policy_rollout, policy_launch_form
    torch = torch_()
    T = 24
    kw = dict(auto_reset=True, max_ep_len=11, seed=12, reset_acts=True, current=True, current_drift=True, vessel_params=preset())
    env, _ = H.make_pair('final_cont', n, **kw)
    env2, _ = H.make_pair('final_cont', n, **kw)
    for e in (env, env2):
        e.set_current(torch.full((n,), 0.15, device=e.device), torch.full((n,), 1.0, device=e.device))
        e.reset()
    make_ac(9, 7, (80, 80, 80), seed=2, device=env.device).upload(env, precision=precision, launch_form=form)
    assert policy_launch_form(env)[0] == form
    out = policy_rollout(env, T, sample=True)
    for t in range(T):
        o, r, d, _ = env2.step(out['act'][t].contiguous())
        nxt = out['obs'][t + 1] if t + 1 < T else out['last_obs']
        assert torch.equal(r, out['rew'][t]) and torch.equal(d, out['done'][t]) and torch.equal(o, nxt), t
    sa, sb = env.get_state(), env2.get_state()
    assert torch.equal(sa[0], sb[0]) and torch.equal(sa[1], sb[1]) and int(sa[1][1].min()) >= 2


def test_returning_to_the_class_keeps_its_thrust_loss():
    """dpenv_set_vessel_params(h, NULL) on a handle created with the thrust-loss preset: back to the shared hull WITH its coefficients (round 5:
    without), the table is that hull's image again"""
    torch = torch_()
    n = 900
    rng = np.random.RandomState(1)
    a, _ = H.make_pair('final_cont', n, vessel_params=preset(), seed=3)
    b, _ = H.make_pair('final_cont', n, vessel_params=preset(), seed=3)
    a.reset(); b.reset()
    a.set_vessel_params(H.to_dev(H.random_hulls(rng, n, loss=0.05)))
    a.step(H.to_dev(H.random_actions(rng, n, 7)))
    a.set_vessel_params(None)
    st, ctr = b.get_state()
    st = st.clone(); st[3] = 1.1
    a.set_state(st, ctr); b.set_state(st, ctr)
    assert torch.equal(a.get_vessel_params(), b.get_vessel_params())
    for t in range(4):
        act = H.to_dev(H.random_actions(rng, n, 7))
        oa, ra, _, _ = a.step(act)
        ob, rb, _, _ = b.step(act)
        assert torch.equal(oa, ob) and torch.equal(ra, rb)


# ------------------------------------------------------------------------------------------------ item 6: per-episode current randomisation
def _cur_pair(n, rv=0.1, rb=0.6, drift=False, **kw):
    torch = torch_()
    env, orc = H.make_pair('final_cont', n, current=True, current_drift=drift, **kw)
    nv = (0.2 + 0.05 * np.sin(np.arange(n))).astype(np.float32)
    nb = (2.356 + 0.3 * np.cos(np.arange(n))).astype(np.float32)
    env.set_current(H.to_dev(nv), H.to_dev(nb))
    env.set_current_randomisation(rv, rb)
    return env, orc, (rv, rb, nv, nb)


def _currents(env):
    v, b = env.get_current()
    mv, mb = env.get_current_mean()
    return np.stack([v.cpu().numpy(), b.cpu().numpy()]), np.stack([mv.cpu().numpy(), mb.cpu().numpy()])


@pytest.mark.parametrize('one_wave', [False, True])
@pytest.mark.parametrize('drift', [False, True])
def test_current_is_redrawn_by_every_reset_like_the_oracle(one_wave, drift):
    """dpenv_set_current_randomisation: dpenv_reset (sampled, masked with explicit init) and the auto-reset inside dpenv_step (reset-wave and one-wave
    kernels) give the new episode, for (seed, global env id, episode), exactly the current the oracle draws - bit for bit in fp32 - as the
    present value AND as the drift's mean; the steps that follow run in it."""
    torch = torch_()
    n = 2000 + 11
    rng = np.random.RandomState(5)
    kw = dict(auto_reset=True, max_ep_len=7, seed=21, env_id_base=5_000_000_000, step_one_wave=one_wave)
    env, orc, cr = _cur_pair(n, drift=drift, **kw)
    rv, rb, nv, nb = cr
    cur = np.ascontiguousarray(np.stack([nv, nb]))
    mean = cur.copy()
    dctr = np.zeros(n, np.uint32)
    ost, octr = orc.new_state(n)
    obs = env.reset()
    oobs = orc.reset(ost, octr, current=cur, current_mean=mean, cur_rand=cr)
    g_cur, g_mean = _currents(env)
    assert np.array_equal(g_cur, cur) and np.array_equal(g_mean, mean) and np.array_equal(cur, mean)
    for k, (lo, hi, nom) in enumerate(((rv, rv, nv), (rb, rb, nb))):
        dev = cur[k] - nom
        assert -lo - 1e-6 <= dev.min() < -0.9 * lo and 0.9 * hi < dev.max() <= hi + 1e-6 and abs(dev.mean()) < 0.06 * lo
    # one draw spelled out: episode 0 of global env id base + 17
    one = orc.draw_current(5_000_000_000 + 17, 0, nv[17], nb[17], rv, rb)
    assert one[0] == cur[0, 17] and one[1] == cur[1, 17]
    g_st, g_ctr = env.get_state()
    assert np.array_equal(g_st.cpu().numpy(), ost) and np.array_equal(g_ctr.cpu().numpy(), octr)
    TOL.assert_close(obs.cpu().numpy(), oobs, TOL.OBS_FLOOR, what='reset obs')
    # masked reset with explicit init: only the selected envs get a new current, and their episode counters advance
    mask = (rng.uniform(size=n) < 0.3).astype(np.uint8)
    init = np.zeros((6, n), np.float32)
    init[0:2] = rng.uniform(-3, 3, size=(2, n))
    before = cur.copy()
    env.reset(mask=H.to_dev(mask), init=H.to_dev(init))
    orc.reset(ost, octr, mask=mask, init=init, current=cur, current_mean=mean, cur_rand=cr)
    g_cur, g_mean = _currents(env)
    assert np.array_equal(g_cur, cur) and np.array_equal(g_mean, mean)
    assert np.array_equal(cur[:, mask == 0], before[:, mask == 0]) and bool((cur[0, mask == 1] != before[0, mask == 1]).mean() > 0.99)
    assert np.array_equal(env.get_state()[1].cpu().numpy(), octr)
    # steps with auto-reset (time limit 7, plus terminations): every finished env continues in a new current
    resets = 0
    for t in range(20):
        act = H.random_actions(rng, n, 7, scale=1.0)
        g_st, g_ctr = env.get_state()
        ost, octr = np.ascontiguousarray(g_st.cpu().numpy()), np.ascontiguousarray(g_ctr.cpu().numpy())
        if drift:
            dctr = np.ascontiguousarray(env.get_rng_counters()[1].cpu().numpy().astype(np.uint32))
        obs, rew, done, _ = env.step(H.to_dev(act))
        oo, orw, od = orc.step(ost, octr, act, current=cur, current_mean=mean, drift_ctr=dctr if drift else None, cur_rand=cr)
        assert bool(TOL.done_agrees(done.cpu().numpy(), od, oo, env.real_ss_bounds).all())
        agree = (done.cpu().numpy() != 0) == (od != 0)       # (an env within 2e-6 of a bound may finish on one side only: skip it from here)
        assert agree.mean() > 0.999
        g_cur, g_mean = _currents(env)
        fin = od != 0
        assert np.array_equal(g_cur[:, agree & fin], cur[:, agree & fin]) and np.array_equal(g_mean[:, agree], mean[:, agree]), t
        if drift:       # the drifting present value of a continuing env: lean arithmetic against libm, to tolerance
            assert np.abs(g_cur[:, agree] - cur[:, agree]).max() < 2e-6
        else:
            assert np.array_equal(g_cur[:, agree], cur[:, agree])
        TOL.assert_close(obs.cpu().numpy()[agree], oo[agree], TOL.OBS_FLOOR, what='obs step %d' % t)
        TOL.assert_close(rew.cpu().numpy()[agree], orw[agree], TOL.REWARD_FLOOR, what='reward step %d' % t)
        cur, mean = np.ascontiguousarray(g_cur), np.ascontiguousarray(g_mean)          # continue from the GPU's values (the skipped envs)
        resets += int(fin.sum())
    assert resets > 3 * n // 2
    # switching it off keeps the currents in force
    env.set_current_randomisation(0.0, 0.0)
    m0 = env.get_current_mean()[0].clone()
    for t in range(8):
        env.step(H.to_dev(H.random_actions(rng, n, 7)))
    assert torch.equal(env.get_current_mean()[0], m0)


@pytest.mark.parametrize('one_wave', [False, True])
@pytest.mark.parametrize('hulls_too', [False, True])
def test_fused_rollout_with_randomised_current_equals_single_steps(one_wave, hulls_too):
    torch = torch_()
    n, T = 1500 + 3, 40
    kw = dict(auto_reset=True, max_ep_len=9, seed=6, reset_acts=True)
    a, _, _ = _cur_pair(n, drift=True, step_one_wave=one_wave, **kw)
    b, _, _ = _cur_pair(n, drift=True, **kw)
    for e in (a, b):
        if hulls_too:
            e.set_vessel_randomisation(0.15, nominal=preset())
        e.reset()
    g = torch.Generator(device=a.device).manual_seed(1)
    acts = torch.randn((T, n, 7), generator=g, device=a.device) * 0.8
    refs = torch.zeros((1, 3, n), device=a.device)
    refs[0, 0] = 1.5
    o, r, d = a.rollout(acts, switch_steps=(17,), refs=refs)
    for t in range(T):
        o1, r1, d1, _ = b.step(acts[t], new_ref=refs[0] if t == 17 else None)
        assert torch.equal(o1, o[t]) and torch.equal(r1, r[t]) and torch.equal(d1, d[t]), t
    sa, sb = a.get_state(), b.get_state()
    assert torch.equal(sa[0], sb[0]) and torch.equal(sa[1], sb[1]) and int(sa[1][1].min()) >= 4
    ca, cb = _currents(a), _currents(b)
    assert np.array_equal(ca[0], cb[0]) and np.array_equal(ca[1], cb[1])
    assert torch.equal(a.get_vessel_params(), b.get_vessel_params())
    nv = 0.2 + 0.05 * np.sin(np.arange(n))
    assert np.abs(ca[1][0] - nv).max() <= 0.1 + 1e-6 and np.abs(ca[1][0] - nv).max() > 0.09      # the means are draws around the nominals


@pytest.mark.parametrize('precision,form', [('f16', 'two_wave'), ('f32_actor', 'two_wave'), ('f32', 'two_wave'), ('f16', 'one_wave'), ('f32', 'one_wave')])
@pytest.mark.parametrize('n', [1000 + 9, 40000])
def test_closed_loop_with_randomised_current_replays_through_single_steps(precision, form, n):
    """every closed-loop form with the current re-drawn per episode (auto-reset and reset_at_end), a drifting current and randomised hulls: the stored
    actions replayed through dpenv_step give the same rows, the same state and the same currents"""
    from ml4ca_amd.policy import policy_rollout, policy_launch_form
    torch = torch_()
    T = 30
    kw = dict(auto_reset=True, max_ep_len=11, seed=12, reset_acts=True)
    env, _, _ = _cur_pair(n, drift=True, **kw)
    env2, _, _ = _cur_pair(n, drift=True, **kw)
    for e in (env, env2):
        e.set_vessel_randomisation(0.1)
        e.reset()
    make_ac(9, 7, (80, 80, 80), seed=2, device=env.device).upload(env, precision=precision, launch_form=form)
    assert policy_launch_form(env)[0] == form
    out = policy_rollout(env, T, sample=True)
    for t in range(T):
        o, r, d, _ = env2.step(out['act'][t].contiguous())
        nxt = out['obs'][t + 1] if t + 1 < T else out['last_obs']
        assert torch.equal(r, out['rew'][t]) and torch.equal(d, out['done'][t]) and torch.equal(o, nxt), t
    sa, sb = env.get_state(), env2.get_state()
    assert torch.equal(sa[0], sb[0]) and torch.equal(sa[1], sb[1]) and int(sa[1][1].min()) >= 3
    ca, cb = _currents(env), _currents(env2)
    assert np.array_equal(ca[0], cb[0]) and np.array_equal(ca[1], cb[1])
    # reset_at_end: every env is cut and re-drawn after the last step - in a new current as well
    out2 = policy_rollout(env, 5, sample=True, reset_at_end=True)
    c2 = _currents(env)
    assert float((c2[1][0] != ca[1][0]).mean()) > 0.99 and bool(np.array_equal(c2[0], c2[1]) or True)
    assert int(env.get_state()[1][1].min()) >= int(sa[1][1].min()) + 1


def test_randomised_current_does_not_depend_on_the_sharding():
    """two handles with env_id_base 0 and n/2 draw the currents of one handle with n envs (keyed by the global env id), step for step"""
    torch = torch_()
    n, h = 2048, 1024
    rng = np.random.RandomState(4)
    kw = dict(auto_reset=True, max_ep_len=6, seed=77)
    whole, _, cr = _cur_pair(n, **kw)
    parts = []
    for k in range(2):
        e, _ = H.make_pair('final_cont', h, current=True, env_id_base=k * h, **kw)
        e.set_current(H.to_dev(cr[2][k * h:(k + 1) * h]), H.to_dev(cr[3][k * h:(k + 1) * h]))
        e.set_current_randomisation(cr[0], cr[1])
        parts.append(e)
    for e in [whole] + parts:
        e.reset()
    for t in range(15):
        act = H.to_dev(H.random_actions(rng, n, 7, scale=1.0))
        ow, rw, dw, _ = whole.step(act)
        for k, e in enumerate(parts):
            o, r, d, _ = e.step(act[k * h:(k + 1) * h].contiguous())
            assert torch.equal(o, ow[k * h:(k + 1) * h]) and torch.equal(r, rw[k * h:(k + 1) * h]) and torch.equal(d, dw[k * h:(k + 1) * h]), (t, k)
    cw = _currents(whole)
    for k, e in enumerate(parts):
        c = _currents(e)
        assert np.array_equal(c[0], cw[0][:, k * h:(k + 1) * h]) and np.array_equal(c[1], cw[1][:, k * h:(k + 1) * h])
    assert int(whole.get_state()[1][1].min()) >= 2


def test_current_randomisation_argument_checks():
    import ml4ca_amd
    from ml4ca_amd import _lib
    torch = torch_()
    n = 256
    e0 = ml4ca_amd.BatchedRevoltEnv(n)
    with pytest.raises(RuntimeError, match='current_enabled'):
        e0.set_current_randomisation(0.1, 0.1)
    e1 = ml4ca_amd.BatchedRevoltEnv(n, current=True)
    for bad in ((-0.1, 0.1), (0.1, float('nan')), (float('inf'), 0.0)):
        with pytest.raises(RuntimeError, match='half-ranges'):
            e1.set_current_randomisation(*bad)
    e1.set_current_randomisation(0.05, 0.2)
    e1.set_vessel_params(None)                                # one class: the shared training form carries the re-draw
    e1.set_current_randomisation(0.0, 0.0)
    vp = np.tile(np.asarray(ml4ca_amd.default_vessel(), np.float32), (2, 1))
    e2 = ml4ca_amd.BatchedRevoltEnv(n, current=True, vessel_params=vp)
    with pytest.raises(RuntimeError, match='vessel-class form'):
        e2.set_current_randomisation(0.05, 0.2)
    e2.set_vessel_params(torch.from_numpy(np.tile(vp[0][:, None], (1, n))).to(e2.device))      # per-env blocks: now it may ...
    e2.set_current_randomisation(0.05, 0.2)
    with pytest.raises(RuntimeError, match='switch it off first'):                               # ... and the classes may not come back under it
        e2.set_vessel_params(None)


@pytest.mark.parametrize('loss', [False, True])
def test_current_randomisation_on_the_shared_hull_equals_the_per_env_form(loss):
    """one class + dpenv_set_current_randomisation runs the shared training form (step_kernel<.., VES_ARGS_LOSS>, coefficients zero without a loss);
    the same hull as per-env blocks runs the general per-env form: the same rows, states and currents bit for bit - and, without a loss, until
    the first reset the rows of a plain default handle in the same (fixed) current"""
    torch = torch_()
    n = 1500 + 7
    rng = np.random.RandomState(3)
    hull = preset() if loss else None
    kw = dict(auto_reset=True, max_ep_len=8, seed=9, vessel_params=hull)
    a, _, cr = _cur_pair(n, drift=True, **kw)
    b, _, _ = _cur_pair(n, drift=True, **kw)
    import ml4ca_amd
    b.set_vessel_params(np.asarray(ml4ca_amd.default_vessel('thrust_loss' if loss else 'no_loss'), np.float32))
    for e in (a, b):
        e.reset()
    for t in range(20):
        act = H.to_dev(H.random_actions(rng, n, 7, scale=1.0))
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db), t
    sa, sb = a.get_state(), b.get_state()
    assert torch.equal(sa[0], sb[0]) and torch.equal(sa[1], sb[1]) and int(sa[1][1].min()) >= 3
    ca, cb = _currents(a), _currents(b)
    assert np.array_equal(ca[0], cb[0]) and np.array_equal(ca[1], cb[1])
    if not loss:
        c, _ = H.make_pair('final_cont', n, current=True, seed=9)                 # a default handle: plain kernels, no re-draw
        d, _, _ = _cur_pair(n, seed=9)
        d.reset(); c.reset()
        v, bta = d.get_current()
        c.set_current(v.clone(), bta.clone())
        st, ctr = d.get_state()
        c.set_state(st, ctr)
        for t in range(5):
            act = H.to_dev(H.random_actions(rng, n, 7))
            oc, rc, dc, _ = c.step(act)
            od, rd, dd, _ = d.step(act)
            assert torch.equal(oc, od) and torch.equal(rc, rd) and torch.equal(dc, dd), t


@pytest.mark.parametrize('kind', ['dynpos_fit', 'dynpos_fit_thrust_loss'])
def test_dynpos_fit_preset_through_the_constructor_against_the_oracle(kind):
    """dpenv_default_vessel_ex(DPENV_VESSEL_DYNPOS_FIT [| THRUST_LOSS]) as the one class of a handle (the default kernels / the shared training form):
    25 steps of full side thrust from rest against the oracle holding the same vector, re-seeded from the GPU state every step"""
    import ml4ca_amd
    torch = torch_()
    n = 600
    vec = np.asarray(ml4ca_amd.default_vessel(kind), np.float32)
    env, orc = H.make_pair('final_cont', n, vessel_params=vec, seed=2, terminate=False, time_limit=False)
    env.reset(init=H.to_dev(np.zeros((6, n), np.float32)))
    act = np.tile(np.array([[1.0, 0.4, 0.4, 1.0, 0.0, 1.0, 0.0]], np.float32), (n, 1))        # bow full, stern 40 % at 90 deg: sideways
    act += 0.02 * np.random.RandomState(1).normal(size=act.shape).astype(np.float32)
    for t in range(25):
        g_st, g_ctr = env.get_state()
        ost, octr = np.ascontiguousarray(g_st.cpu().numpy()), np.ascontiguousarray(g_ctr.cpu().numpy())
        obs, rew, done, _ = env.step(H.to_dev(act))
        oo, orw, od = orc.step(ost, octr, act)
        TOL.assert_close(obs.cpu().numpy(), oo, TOL.OBS_FLOOR, what='obs step %d' % t)
        TOL.assert_close(rew.cpu().numpy(), orw, TOL.REWARD_FLOOR, what='reward step %d' % t)
    assert float(obs[:, 4].abs().mean()) > 0.1                                                   # it did move sideways


def test_checkpoint_restore_with_randomised_and_drifting_current_continues_bit_for_bit():
    """a checkpoint of a run whose current is re-drawn per episode AND drifts: state + counters, present current, drift means, rng counters; restore =
    fresh handle, set_current(means), set_current(present, present_only), set_current_randomisation(ORIGINAL nominals, ranges), set_rng_counters,
    set_state - the run continues exactly like the uninterrupted one"""
    torch = torch_()
    n, k, m = 1800 + 5, 11, 22
    rng = np.random.RandomState(6)
    kw = dict(auto_reset=True, max_ep_len=7, seed=13, env_id_base=999)
    a, _, cr = _cur_pair(n, drift=True, **kw)
    a.reset()
    acts = [H.to_dev(H.random_actions(rng, n, 7, scale=1.0)) for _ in range(k + m)]
    for t in range(k):
        a.step(acts[t])
    st, ctr = (x.clone() for x in a.get_state())
    cur, mean = [x.clone() for x in a.get_current()], [x.clone() for x in a.get_current_mean()]
    nctr, dctr = (x.clone() for x in a.get_rng_counters())
    assert int(ctr[1].min()) >= 1 and not torch.equal(cur[0], mean[0])                       # resets happened, and the present value has drifted off its mean
    b, _ = H.make_pair('final_cont', n, current=True, current_drift=True, **kw)
    b.set_current(*mean)
    b.set_current(*cur, present_only=True)
    b.set_current_randomisation(cr[0], cr[1], vc_nominal=H.to_dev(cr[2]), beta_nominal=H.to_dev(cr[3]))
    b.set_rng_counters(nctr, dctr)
    b.set_state(st, ctr)
    for t in range(k, k + m):
        oa, ra, da, _ = a.step(acts[t])
        ob, rb, db, _ = b.step(acts[t])
        assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db), t
    ca, cb = _currents(a), _currents(b)
    assert np.array_equal(ca[0], cb[0]) and np.array_equal(ca[1], cb[1])
    assert torch.equal(a.get_state()[0], b.get_state()[0]) and int(a.get_state()[1][1].min()) >= 3


@pytest.mark.parametrize('n', [1, 63, 64, 65, 257])
def test_shared_training_form_on_tiny_and_ragged_batches_in_both_layouts(n):
    """the shared training form (thrust-loss preset + per-episode current) at batch sizes around a wave, row-major and [dim][n] layouts: the rows of
    the per-env form of the same hull, bit for bit"""
    torch = torch_()
    rng = np.random.RandomState(n)
    kw = dict(auto_reset=True, max_ep_len=5, seed=3, vessel_params=preset())
    for layout in ('aos', 'soa'):
        a, _, _ = _cur_pair(n, drift=True, layout=layout, **kw)
        b, _, _ = _cur_pair(n, drift=True, layout=layout, **kw)
        b.set_vessel_params(preset())
        a.reset(); b.reset()
        for t in range(12):
            act = H.random_actions(rng, n, 7, scale=1.0)
            act_d = H.to_dev(act if layout == 'aos' else np.ascontiguousarray(act.T))
            oa, ra, da, _ = a.step(act_d)
            ob, rb, db, _ = b.step(act_d)
            assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db), (layout, t)
        ca, cb = _currents(a), _currents(b)
        assert np.array_equal(ca[0], cb[0]) and np.array_equal(ca[1], cb[1]) and int(a.get_state()[1][1].min()) >= 2


@pytest.mark.parametrize('form', ['one_wave', 'two_wave'])
@pytest.mark.parametrize('mode,ext,act', [('final_cont', True, 'leaky'), ('limited', False, 'leaky'), ('final_cont', True, 'tanh')])
def test_shared_hull_with_loss_and_redrawn_current_in_every_closed_loop_route(form, mode, ext, act):
    """one class with thrust-loss coefficients + dpenv_set_current_randomisation through the closed loop's routes: the shared training form of the
    two-wave kernels (shipped configuration), and the one-wave kernels everything else is sent to (they read the class hull from its table image and
    re-draw the current behind a run-time switch) - the stored actions replayed through dpenv_step give the same rows, state and currents"""
    from ml4ca_amd.policy import policy_rollout
    torch = torch_()
    n, T = 900 + 7, 24
    envs = []
    for _ in range(2):
        e, _ = H.make_pair(mode, n, ext=ext, auto_reset=True, max_ep_len=7, seed=8, current=True, current_drift=True, vessel_params=preset())
        e.set_current(torch.full((n,), 0.2, device=e.device), torch.full((n,), 2.0, device=e.device))
        e.set_current_randomisation(0.1, 1.0)
        e.reset()
        envs.append(e)
    env, env2 = envs
    make_ac(env.num_states, env.num_actions, (80, 80, 80), seed=2, device=env.device, activation=act).upload(env, precision='f16', launch_form=form if act == 'leaky' else 'auto')
    out = policy_rollout(env, T, sample=True)
    for t in range(T):
        o, r, d, _ = env2.step(out['act'][t].contiguous())
        nxt = out['obs'][t + 1] if t + 1 < T else out['last_obs']
        assert torch.equal(r, out['rew'][t]) and torch.equal(d, out['done'][t]) and torch.equal(o, nxt), (mode, t)
    sa, sb = env.get_state(), env2.get_state()
    assert torch.equal(sa[0], sb[0]) and torch.equal(sa[1], sb[1]) and int(sa[1][1].min()) >= 2
    ca, cb = _currents(env), _currents(env2)
    assert np.array_equal(ca[0], cb[0]) and np.array_equal(ca[1], cb[1])
