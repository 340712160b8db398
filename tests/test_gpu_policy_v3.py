"""Round-3 GPU tests of the closed-loop kernels:
two-wave launch form of the split ("fp32-faithful") arithmetics, the 128-env workgroup geometry, the reference's
epoch boundary (reset_at_end, ppo.py:305-322), pieces of one episode (pipelined exchange), checkpoint of the draw counters,
uploads that do not disturb launches in flight."""
import numpy as np
import pytest

from tests import helpers as H
from tests import tolerances as TOL

pytestmark = pytest.mark.gpu

ROWS = ('obs', 'act', 'rew', 'done', 'logp', 'val', 'boot', 'last_obs', 'last_val')


def torch_():
    import torch
    return torch


def make_ac(*a, **kw):
    from ml4ca_amd.policy import ActorCritic
    return ActorCritic(*a, **kw)


def _pair_of_launches(n, T, precision, forms=('two_wave', 'one_wave'), seed=12, hidden=(80, 80, 80), mode='final_cont', ext=True, **extra):
    """the same sampled rollout (drifting current, reset_acts, auto-reset through a short time limit) in two launch forms"""
    from ml4ca_amd.policy import policy_rollout
    torch = torch_()
    kw = dict(auto_reset=True, max_ep_len=6, seed=seed, reset_acts=True, current=True, current_drift=True, ext=ext)
    kw.update(extra)
    outs, states = [], []
    for form in forms:
        env, _ = H.make_pair(mode, n, **kw)
        env.set_current(torch.full((n,), 0.15, device=env.device), torch.full((n,), 1.0, device=env.device))
        env.reset()
        make_ac(env.num_states, env.num_actions, hidden, seed=5, device=env.device).upload(env, precision=precision, launch_form=form)
        outs.append(policy_rollout(env, T, sample=True))
        states.append(env.get_state() + env.get_rng_counters() + env.get_current())
    return outs, states


@pytest.mark.parametrize('precision', ['f32_actor', 'f32', 'f16'])
@pytest.mark.parametrize('n', [2000 + 9, 65536])
def test_two_wave_form_writes_the_one_wave_rows(precision, n):
    """DPENV_LAUNCH_TWO_WAVE for every arithmetic (round 3: the split ones too) against DPENV_LAUNCH_ONE_WAVE: every row of a
    sampled closed-loop launch with in-kernel noise, drifting current, reset_acts and cut episodes, the final state, the draw
    counters - bit for bit.  2 009 envs run in the 128-env workgroup geometry (a SIMD per wave), 65 536 in the 256-env one."""
    from ml4ca_amd import DpenvError
    torch = torch_()
    T = 14
    try:
        (a, b), (sa, sb) = _pair_of_launches(n, T, precision)
    except DpenvError as e:
        if precision == 'f32' and 'exceed the 160 KiB' in str(e):
            pytest.skip('DPENV_POLICY_F32 two-wave does not fit the LDS in this build: %s' % e)
        raise
    for k in ROWS:
        assert torch.equal(a[k], b[k]), (precision, n, k)
    for x, y in zip(sa, sb):
        assert torch.equal(x, y)
    done = a['done']
    assert int(((done & 2) != 0).sum()) >= n and bool(torch.isfinite(a['boot']).all()) and bool((a['boot'] != 0).any())


def test_epoch_boundary_in_both_launch_forms():
    """reset_at_end with drifting current, reset_acts and vessel classes, every arithmetic: the two-wave form (pre-drawn reset sample, reward
    behind the hand-over) writes the one-wave form's rows, boot row of the cut included, and leaves the same state behind"""
    from ml4ca_amd import _lib as L
    from ml4ca_amd.policy import policy_rollout
    torch = torch_()
    n, T = 2000 + 9, 9
    base = L.default_vessel()
    vp = np.stack([base, base * np.where(np.arange(L.NPARAM) < 4, 1.3, 1.0)]).astype(np.float32)
    for precision in ('f16', 'f32_actor', 'f32'):
        outs, states = [], []
        for form in ('two_wave', 'one_wave'):
            env, _ = H.make_pair('final_cont', n, auto_reset=True, max_ep_len=6, seed=19, reset_acts=True, current=True, current_drift=True,
                                 vessel_params=vp)
            env.set_vessel_class((torch.arange(n, device=env.device) % 2).to(torch.int32))
            env.set_current(torch.full((n,), 0.15, device=env.device), torch.full((n,), 1.0, device=env.device))
            env.reset()
            make_ac(9, 7, (80, 80, 80), seed=5, device=env.device).upload(env, precision=precision, launch_form=form)
            o1 = policy_rollout(env, T, sample=True, reset_at_end=True)
            o2 = policy_rollout(env, T, sample=True, reset_at_end=True)       # the epoch after: starts from fresh episodes
            outs.append((o1, o2))
            states.append(env.get_state() + env.get_rng_counters())
        for a, b in zip(outs[0], outs[1]):
            for k in ROWS:
                assert torch.equal(a[k], b[k]), (precision, k)
        for x, y in zip(states[0], states[1]):
            assert torch.equal(x, y)
        o1, o2 = outs[0]
        assert int(states[0][1][0].max()) == 0                            # every env starts the next epoch at step 0
        assert torch.equal(o2['obs'][0], o1['last_obs']) and bool((o1['boot'][T - 1] != 0).any())


@pytest.mark.parametrize('n', [1, 63, 64, 65, 127, 129, 257])
def test_two_wave_forms_at_ragged_batch_sizes(n):
    """batch sizes around the group (64 envs) and workgroup (128 envs) boundaries of the two-wave geometry, in every arithmetic: a group
    without envs leaves, a partial group clamps its lanes, and the rows are those of the one-wave form"""
    torch = torch_()
    for precision in ('f16', 'f32_actor', 'f32'):
        (a, b), (sa, sb) = _pair_of_launches(n, 7, precision, seed=3 + n)
        for k in ROWS:
            assert torch.equal(a[k], b[k]), (precision, n, k)
        for x, y in zip(sa, sb):
            assert torch.equal(x, y)
        assert bool(torch.isfinite(a['val']).all())


@pytest.mark.parametrize('precision,form', [('f32_actor', 'two_wave'), ('f32', 'two_wave')])
def test_two_wave_split_evaluations_equal_the_forward_kernel_at_full_size(precision, form):
    """65 536 envs, deterministic actions: every stored action and value of the two-wave launch equals what the forward kernel computes
    from the stored observation row, bit for bit - every inlined copy of the split evaluation in the network wave (first, in-loop,
    pre-reset critic).  The check that catches a register-level fault in ONE copy (round 2's asm-temporary / MFMA overlap)."""
    import ml4ca_amd
    from ml4ca_amd import DpenvError
    from ml4ca_amd.policy import policy_forward, policy_rollout
    torch = torch_()
    n, T = 65536, 12
    env = ml4ca_amd.BatchedRevoltEnv(n, auto_reset=True, max_ep_len=5, seed=11)
    ac = make_ac(9, 7, (80, 80, 80), seed=6, device=env.device)
    try:
        ac.upload(env, precision=precision, launch_form=form)
    except DpenvError as e:
        pytest.skip(str(e))
    env.reset()
    out = policy_rollout(env, T, sample=False)
    mu, v = policy_forward(env, out['obs'].reshape(T * n, 9))
    assert torch.equal(out['act'].reshape(T * n, 7), mu)
    assert torch.equal(out['val'].reshape(T * n), v)
    done = out['done']
    ended = (done != 0)
    ended[T - 1] = True
    terminal = (done & 1) != 0
    assert bool((out['boot'][~ended | terminal] == 0).all()) and bool(torch.isfinite(out['boot']).all())
    assert int((ended & ~terminal).sum()) > n


@pytest.mark.parametrize('mode,ext,hidden,activation,precision', [
    ('final_cont', True, (96, 96), 'leaky', 'f32_actor'),
    ('limited', False, (64,), 'relu', 'f32_actor'),
    ('simple', False, (33, 33, 33), 'leaky', 'f32'),
    ('full', True, (48, 48), 'relu', 'f32'),
    ('final_wrap', True, (80, 80, 80), 'leaky', 'f16'),
])
def test_two_wave_forms_for_other_shapes(mode, ext, hidden, activation, precision):
    """the shapes the two-wave kernels are templated on (5 / 6 k-steps, one to three hidden layers, every env variant's action
    width; 705 envs = the 128-env workgroup geometry with a ragged last group): rows equal to the one-wave form bit for bit"""
    from ml4ca_amd import DpenvError
    from ml4ca_amd.policy import policy_rollout
    torch = torch_()
    n, T = 700 + 5, 9
    outs = []
    for form in ('two_wave', 'one_wave'):
        env, _ = H.make_pair(mode, n, ext=ext, auto_reset=True, max_ep_len=4, seed=13)
        ac = make_ac(env.num_states, env.num_actions, hidden, seed=9, device=env.device, activation=activation)
        try:
            ac.upload(env, precision=precision, launch_form=form)
        except DpenvError as e:
            pytest.skip(str(e))
        env.reset()
        outs.append(policy_rollout(env, T, sample=True))
    for k in ROWS:
        assert torch.equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize('precision,form', [('f32_actor', 'auto'), ('f16', 'auto'), ('f32', 'one_wave')])
def test_reset_at_end_is_the_reference_epoch_boundary(precision, form):
    """dpenv_policy_rollout_io.reset_at_end (ppo.py:305-322: at t == local_steps_per_epoch - 1 EVERY env is cut and reset): rows
    0..T-1 equal the launch without it bit for bit except boot[T-1]; boot[T-1] = V(observation after step T-1) unless the env
    terminated there (then 0, ppo.py:311); afterwards every env is a fresh episode - the state equals what an explicit
    Revolt.reset of the not-yet-reset envs gives (same Philox draw: episode + 1), checked against the single-step path and the
    ORACLE's reset; the next launch starts from those states."""
    from ml4ca_amd.policy import policy_forward, policy_rollout
    torch = torch_()
    n, T = 3000 + 7, 11
    kw = dict(auto_reset=True, max_ep_len=400, seed=41)
    envA, orc = H.make_pair('final_cont', n, **kw)
    envB, _ = H.make_pair('final_cont', n, **kw)
    ac = make_ac(9, 7, (80, 80, 80), seed=3, device=envA.device)
    for e in (envA, envB):
        ac.upload(e, precision=precision, launch_form=form)
        e.reset()
    a = policy_rollout(envA, T, sample=True, reset_at_end=True)
    b = policy_rollout(envB, T, sample=True)
    for k in ('obs', 'act', 'rew', 'done', 'logp', 'val'):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(a['boot'][:T - 1], b['boot'][:T - 1])
    # the last step: B's last_obs is the observation after step T-1 (post auto-reset for envs that ended there)
    done_last = b['done'][T - 1]
    terminal = (done_last & 1) != 0
    ended = done_last != 0
    cont = ~ended                                                  # envs the epoch boundary cuts
    assert int(cont.sum()) > n // 2 and int(terminal.sum()) > 0
    _, v_last = policy_forward(envB, b['last_obs'])
    assert torch.equal(a['boot'][T - 1][cont], v_last[cont])       # V(o_T) of a running episode
    assert torch.equal(a['boot'][T - 1][cont], b['boot'][T - 1][cont])   # which is what the end of a launch leaves anyway
    assert bool((a['boot'][T - 1][terminal] == 0).all())
    assert torch.equal(a['boot'][T - 1][ended], b['boot'][T - 1][ended])
    # state after: B + an explicit reset of the envs that had NOT just been re-drawn = A
    stB, ctrB = envB.get_state()
    envB.reset(mask=cont.to(torch.uint8))
    sa, ca = envA.get_state()
    sb, cb = envB.get_state()
    assert torch.equal(sa, sb) and torch.equal(ca, cb)
    assert int(ca[0].max()) == 0 and bool((ca[1] == ctrB[1] + cont.to(ctrB.dtype)).all())
    # ... and the oracle's reset of those envs from B's pre-reset state gives the same draw (fp32, bit for bit: a sample, no arithmetic)
    ost, octr = stB.cpu().numpy().copy(), ctrB.cpu().numpy().copy()
    oobs = orc.reset(ost, octr, mask=cont.cpu().numpy().astype(np.uint8))
    assert np.array_equal(ost[0:6], sa.cpu().numpy()[0:6]) and np.array_equal(octr, ca.cpu().numpy())
    TOL.assert_close(a['last_obs'].float().cpu().numpy(), oobs, TOL.OBS_FLOOR, what='first observation of the new episodes')
    # the next launch runs from the fresh episodes
    a2 = policy_rollout(envA, 3, sample=True, reset_at_end=True)
    assert torch.equal(a2['obs'][0], a['last_obs']) and torch.equal(a2['val'][0], a['last_val'])
    # without auto_reset the flag is refused
    from ml4ca_amd import DpenvError
    envC, _ = H.make_pair('final_cont', 64, auto_reset=False)
    ac.upload(envC, precision='f16')
    envC.reset()
    with pytest.raises(DpenvError):
        policy_rollout(envC, 2, sample=True, reset_at_end=True)


@pytest.mark.parametrize('precision', ['f32_actor', 'f16'])
def test_pieces_of_one_episode_equal_one_launch(precision):
    """RolloutBuffer.collect(rows=...) (the pieces dist.EpisodeExchange posts while the next one runs): four launches of T / 4 steps
    write the rows of one launch of T steps bit for bit, except `boot` at the last row of an inner piece - which the scan does not
    read there: advantages, returns and their statistics are the same bits."""
    from ml4ca_amd import rollout as RO
    torch = torch_()
    n, T = 4096 + 64, 40
    bufs = []
    for pieces in (1, 4):
        env, _ = H.make_pair('final_cont', n, auto_reset=True, max_ep_len=25, seed=8, obs_dtype='bfloat16')
        make_ac(9, 7, (80, 80, 80), seed=4, device=env.device).upload(env, precision=precision)
        env.reset()
        buf = RO.RolloutBuffer(T, env)
        if pieces == 1:
            buf.collect(env, sample=True)
        else:
            for c in range(pieces):
                buf.collect(env, sample=True, rows=(c * T // pieces, (c + 1) * T // pieces))
        buf.finish()
        bufs.append((buf, env.get_state()))
    (b1, s1), (b4, s4) = bufs
    for k in ('obs', 'act', 'rew', 'done', 'logp', 'val', 'last_obs', 'last_val'):
        assert torch.equal(b1.blocks[k], b4.blocks[k]), k
    inner = torch.zeros(T, dtype=torch.bool, device=b1.adv.device)
    inner[[T // 4 - 1, T // 2 - 1, 3 * T // 4 - 1]] = True
    assert torch.equal(b1.blocks['boot'][~inner], b4.blocks['boot'][~inner])
    assert torch.equal(b1.adv, b4.adv) and torch.equal(b1.ret, b4.ret) and torch.equal(b1.stats, b4.stats)
    assert torch.equal(s1[0], s4[0]) and torch.equal(s1[1], s4[1])
    assert b1.blocks['obs'].dtype == torch.bfloat16
    # get() consumes the statistics once; a second get() recomputes them (three-pass form) instead of re-applying stale ones
    adv_raw = b1.adv.clone()
    _, _, adv_n, _, _ = b1.get()
    m1, s1_ = float(adv_n.mean()), float(adv_n.std())
    assert abs(m1) < 1e-4 and abs(s1_ - 1.0) < 1e-3
    b1.adv.mul_(3.0).add_(1.0)                                    # someone edits adv between two get() calls
    _, _, adv_n2, _, _ = b1.get()
    assert abs(float(adv_n2.mean())) < 1e-4 and abs(float(adv_n2.std()) - 1.0) < 1e-3
    del adv_raw


def test_checkpoint_with_draw_counters_reproduces_sampled_rollouts():
    """dpenv_get_state alone did not restore the exploration-noise / drift counters.  get_state + get_current +
    get_rng_counters + get_obs_thrust after a launch, restored into a FRESH handle (set_state, set_obs_thrust, set_current(mean) +
    set_current(present_only), set_rng_counters): the next sampled launch - in-kernel noise, drifting current, reset_acts, cut episodes - is the
    original's, bit for bit; without the counters it is not."""
    from ml4ca_amd.policy import policy_rollout
    torch = torch_()
    n, T = 1500 + 1, 9
    kw = dict(auto_reset=True, max_ep_len=7, seed=77, current=True, current_drift=True, reset_acts=True)
    mean = lambda e: (torch.full((n,), 0.2, device=e.device), torch.full((n,), 2.3, device=e.device))
    envA, _ = H.make_pair('final_cont', n, **kw)
    envA.set_current(*mean(envA))
    ac = make_ac(9, 7, (80, 80, 80), seed=1, device=envA.device)
    ac.upload(envA, precision='f32_actor')
    envA.reset()
    policy_rollout(envA, T, sample=True)
    st, ctr = envA.get_state()
    nc, dc = envA.get_rng_counters()
    vc, beta = envA.get_current()
    lag = envA.get_obs_thrust()
    assert int(nc.min()) == T and int(dc.min()) == T and not torch.equal(vc, mean(envA)[0]) and lag is not None
    want = policy_rollout(envA, T, sample=True)
    restored = {}
    for with_counters in (True, False):
        envB, _ = H.make_pair('final_cont', n, **kw)
        envB.set_current(*mean(envB))                             # the means the drift reverts to
        envB.set_current(vc, beta, present_only=True)             # the drifted values of the checkpoint
        ac.upload(envB, precision='f32_actor')
        envB.reset()
        lag0 = envB.get_obs_thrust()                              # a FULL reset writes every env's columns: the new episode's own previous thrust / 100
        stB, _ = envB.get_state()
        assert lag0 is not None and torch.equal(lag0[:, 0:3], stB[9:12].T.contiguous() * 0.01)
        envB.set_state(st, ctr)
        envB.set_obs_thrust(lag)                                  # the observation lags the stored thrust command by one step
        if with_counters:
            envB.set_rng_counters(nc, dc)
        restored[with_counters] = policy_rollout(envB, T, sample=True)
    for k in ROWS:
        assert torch.equal(restored[True][k], want[k]), k
    assert not torch.equal(restored[False]['act'], want['act'])


def test_upload_does_not_disturb_a_launch_in_flight_on_another_stream():
    """dpenv_set_policy_desc used to repack the single weight image in place.  A long launch on a side stream,
    then TWO uploads of other weights on the current stream while it runs (the second reuses the first image and must wait for
    the launch by event): the launch's rows are those of the weights it was given; a launch after the uploads uses the new ones."""
    from ml4ca_amd.policy import policy_forward, policy_rollout
    torch = torch_()
    n, T = 65536, 60
    env, _ = H.make_pair('final_cont', n, auto_reset=True, seed=5)
    acs = [make_ac(9, 7, (80, 80, 80), seed=s_, device=env.device) for s_ in (1, 2, 3)]
    acs[0].upload(env, precision='f16')
    env.reset()
    st, ctr = env.get_state()
    ref = policy_rollout(env, T, sample=False)
    torch.cuda.synchronize()
    env.set_state(st, ctr)
    side = torch.cuda.Stream(device=env.device)
    side.wait_stream(torch.cuda.current_stream(env.device))
    with torch.cuda.stream(side):
        out = policy_rollout(env, T, sample=False)             # ~0.5 ms of kernel on the side stream
    acs[1].upload(env, precision='f16')                         # current stream: other image
    acs[2].upload(env, precision='f16')                         # reuses the image `out` is reading: waits for the launch (event)
    obs = ref['obs'][0].float().contiguous()
    mu3, _ = policy_forward(env, obs)
    torch.cuda.synchronize()
    for k in ('obs', 'act', 'val', 'rew'):
        assert torch.equal(out[k], ref[k]), k
    env.set_state(st, ctr)
    acs[0].upload(env, precision='f16')
    mu1, _ = policy_forward(env, obs)
    assert torch.equal(mu1, ref['act'][0]) and not torch.equal(mu3, mu1)


def test_closed_loop_continues_from_the_observation_a_single_step_returned():
    """ADVICE r03 (low), closed in round 4: a closed-loop launch after dpenv_step / dpenv_rollout / dpenv_set_state starts from the
    observation its predecessor returned (thrust columns = the command of the step BEFORE, customEnv.py:196-205,126), not from one
    rebuilt from the state block.  Six closed-loop steps in one launch = three in a launch + one dpenv_step with the actor's own mean
    + two in a launch; and = three + a one-step dpenv_rollout + two."""
    import ml4ca_amd
    from ml4ca_amd.policy import policy_forward, policy_rollout
    torch = torch_()
    n = 1000 + 7
    envs = [ml4ca_amd.BatchedRevoltEnv(n, auto_reset=True, max_ep_len=100, seed=3) for _ in range(3)]
    ac = make_ac(9, 7, (80, 80, 80), seed=4, device=envs[0].device)
    for e in envs:
        ac.upload(e, precision='f32_actor')
        e.reset()
    whole = policy_rollout(envs[0], 6, sample=False)
    for k, e in enumerate(envs[1:]):
        a = policy_rollout(e, 3, sample=False)
        for key in ('obs', 'act', 'rew'):
            assert torch.equal(a[key], whole[key][:3]), key
        mu, _ = policy_forward(e, a['last_obs'].float().contiguous())
        assert torch.equal(mu, whole['act'][3])
        if k == 0:
            o4, r3, d3, _ = e.step(mu.contiguous())
        else:
            o_, r_, d_ = e.rollout(mu.contiguous().unsqueeze(0))
            o4, r3, d3 = o_[0], r_[0], d_[0]
        assert torch.equal(r3, whole['rew'][3]) and torch.equal(o4, whole['obs'][4])
        b = policy_rollout(e, 2, sample=False)
        for key in ('obs', 'act', 'rew', 'val', 'logp'):
            assert torch.equal(b[key], whole[key][4:6]), (k, key)
    s0, c0 = envs[0].get_state()
    for e in envs[1:]:
        s1, c1 = e.get_state()
        assert torch.equal(s0, s1) and torch.equal(c0, c1)


def test_captured_rollout_sees_every_eager_upload():
    """a rollout recorded into a HIP graph has the weight image's address baked in.  With two alternating images only
    every second eager upload reached the replays (the others ran stale weights, silently).  From the capture on, uploads go in place
    into the image the graph reads: the PPO pattern - upload each epoch, replay the rollout graph - for three consecutive uploads."""
    from ml4ca_amd.policy import policy_rollout
    torch = torch_()
    n, T = 4096, 12
    env, _ = H.make_pair('final_cont', n, auto_reset=True, seed=9)
    env2, _ = H.make_pair('final_cont', n, auto_reset=True, seed=9)
    acs = [make_ac(9, 7, (80, 80, 80), seed=s_, device=env.device) for s_ in (1, 2, 3, 4)]
    acs[0].upload(env, precision='f32_actor')
    env.reset(); env2.reset()
    st, ctr = env.get_state()
    out = policy_rollout(env, T, sample=False)                 # allocates the rows; also warms the launch path before capture
    env.set_state(st, ctr)
    side = torch.cuda.Stream(device=env.device)
    side.wait_stream(torch.cuda.current_stream(env.device))
    with torch.cuda.stream(side):
        policy_rollout(env, T, sample=False, out=out)
    torch.cuda.current_stream(env.device).wait_stream(side)
    torch.cuda.synchronize()
    env.set_state(st, ctr)                                     # as before every replay: the launch form (start from the state block) is baked in too
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        policy_rollout(env, T, sample=False, out=out)
    for k, ac in enumerate(acs[1:] + acs[:1]):                 # four uploads in a row, each followed by one replay
        ac.upload(env, precision='f32_actor')                   # eager, on the current stream: the stream the graph is replayed on
        env.set_state(st, ctr)
        g.replay()
        torch.cuda.synchronize()
        ac.upload(env2, precision='f32_actor')
        env2.set_state(st, ctr)
        want = policy_rollout(env2, T, sample=False)
        for key in ('obs', 'act', 'val', 'rew', 'logp'):
            assert torch.equal(out[key], want[key]), (k, key)
    # eager launches after the capture read the same (pinned) image and see the latest upload too
    env.set_state(st, ctr)
    eager = policy_rollout(env, T, sample=False)
    assert torch.equal(eager['act'], want['act'])


def test_upload_waits_for_readers_on_two_streams():
    """one event per image, re-recorded by whoever read last - a long rollout on stream A followed by a short
    forward on stream B left the event covering B alone, and the second-next upload repacked the image under A's kernel.  The reader on
    B now chains behind the event as it stands.  A 60-step launch on A, a forward on B, then two uploads: A's rows are those of the
    weights it was given."""
    from ml4ca_amd.policy import policy_forward, policy_rollout
    torch = torch_()
    n, T = 65536, 60
    env, _ = H.make_pair('final_cont', n, auto_reset=True, seed=5)
    acs = [make_ac(9, 7, (80, 80, 80), seed=s_, device=env.device) for s_ in (1, 2, 3)]
    acs[0].upload(env, precision='f16')
    env.reset()
    st, ctr = env.get_state()
    ref = policy_rollout(env, T, sample=False)
    obs = ref['obs'][0].float().contiguous()
    torch.cuda.synchronize()
    env.set_state(st, ctr)
    sa, sb = torch.cuda.Stream(device=env.device), torch.cuda.Stream(device=env.device)
    sa.wait_stream(torch.cuda.current_stream(env.device)); sb.wait_stream(torch.cuda.current_stream(env.device))
    with torch.cuda.stream(sa):
        out = policy_rollout(env, T, sample=False)             # long reader on A
    with torch.cuda.stream(sb):
        mu_b, _ = policy_forward(env, obs[:256].contiguous())    # short reader on B: re-records the image's event
    acs[1].upload(env, precision='f16')                         # other image
    acs[2].upload(env, precision='f16')                         # reuses the image A is still reading: must wait for A AND B
    torch.cuda.synchronize()
    for k in ('obs', 'act', 'val', 'rew'):
        assert torch.equal(out[k], ref[k]), k
    assert torch.equal(mu_b, ref['act'][0][:256])


@pytest.mark.parametrize('changes', [False, True])
def test_run_RL_policy_against_the_reference_harness_fixture(changes):
    """SURVEY 8 f-2 / VERDICT r02 item 3: tests/golden/run_rl_policy.npz is the reference's OWN run_RL_policy
    (spinup/utils/test_policy.py:97-186, imported and executed by tests/golden/gen_run_rl_policy.py) with the thesis' trained actor
    (NumPy float64 MLP from the checkpoint tensors) on the reference's RevoltFinal(testing=True) over oracle/twin_shim - per step
    ned_pos, ned_ref, action_vec, observation, reward, and EpRet / EpLen per episode, without and with test_setpoint_changes.
    evaluate.run_RL_policy (HIP env in fp32, fp32-faithful in-kernel actor) must reproduce it: EpLen exactly, everything else at the
    tolerance of an fp32 closed loop against a float64 one over 400 steps (a stabilising feedback policy: the error does not grow).
    Measured worst cases (gpurun_out/run_rl_policy_errors.json, round 3): position 1.7e-6 m, observation / reward 1.8e-6, thrust command
    1.8e-4 %, azimuth 7e-7 rad, episode return 4e-4 of ~1 000; asserted at about ten times that."""
    import json
    import os
    import ml4ca_amd
    from ml4ca_amd import evaluate as EV
    from ml4ca_amd.policy import ActorCritic
    torch = torch_()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = np.load(os.path.join(root, 'tests', 'golden', 'run_rl_policy.npz'))
    d = np.load(os.path.join(root, 'tests', 'golden', 'final_policy.npz'))
    tensors = {k.replace('.', '/'): d[k] for k in d.files if '.' in k}
    tag = 'setpoints' if changes else 'plain'
    E = len(g[tag + '_EpLen'])
    T = int(g['max_ep_len'])
    env = ml4ca_amd.BatchedRevoltEnv(E, testing=True, time_limit=False, vessel_params=g['vessel'].astype(np.float32))
    assert env.max_ep_len == T
    ac = ActorCritic.from_tensors(tensors, device=env.device).upload(env, precision='f32')
    res = EV.run_RL_policy(env, ac, num_episodes=E, test_setpoint_changes=changes)
    eplen = res['EpLen'].cpu().numpy()
    assert np.array_equal(eplen, g[tag + '_EpLen']), (eplen, g[tag + '_EpLen'])
    assert np.array_equal(g[tag + '_n_recorded'], eplen + 1)
    worst = {}
    for k in range(E):
        L = int(eplen[k]) + 1                                       # records 0 .. EpLen
        for name, key, tol in (('ned_pos', '_ned_pos', 2e-5), ('ned_ref', '_ned_ref', 1e-6), ('action_vec', '_action_vec', None),
                               ('obs', '_obs', 2e-5), ('rew', '_rew', 2e-5)):
            got = res[name][:L, k].double().cpu().numpy()
            want = g[tag + key][k, :L]
            assert not np.isnan(want).any()
            err = np.abs(got - want)
            if name == 'action_vec':
                # thrust columns in percent (scale 100), azimuth columns in rad; an azimuth command may sit on the +-pi seam of atan2.
                # Record 0 of episode 0 is not the initial action vector in the reference: test_policy.py:125 appends the live
                # `action_vec` array itself (no copy), which :146 then keeps overwriting in place - the entry aliases whatever the
                # vector held when the run ended (or when :152 rebound the name).  Later episodes append a copy (:176).  Skipped here,
                # checked to BE that alias instead.
                lo = 1 if k == 0 else 0
                if k == 0:
                    assert not np.allclose(want[0], [0, 0, 0, np.pi / 2, 0, 0]) and np.allclose(got[0], [0, 0, 0, np.pi / 2, 0, 0], atol=1e-6)
                e_thr = err[lo:, 0:3].max()
                da = np.abs(np.angle(np.exp(1j * (got[lo:, 3:6] - want[lo:, 3:6])))).max()
                worst['thrust_pct'] = max(worst.get('thrust_pct', 0.0), float(e_thr))
                worst['azimuth_rad'] = max(worst.get('azimuth_rad', 0.0), float(da))
                assert e_thr < 2e-3 and da < 1e-5, (k, e_thr, da)
            else:
                worst[name] = max(worst.get(name, 0.0), float(err.max()))
                assert err.max() < tol, (name, k, float(err.max()), np.unravel_index(err.argmax(), err.shape))
    ret = res['EpRet'].double().cpu().numpy()
    worst['EpRet'] = float(np.abs(ret - g[tag + '_EpRet']).max())
    assert worst['EpRet'] < 5e-3, (ret, g[tag + '_EpRet'])
    out = os.path.join(root, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, 'run_rl_policy_errors.json')
    try:
        rec = json.load(open(path))
    except Exception:
        rec = {}
    rec[tag] = worst
    json.dump(rec, open(path, 'w'), indent=1)
    env.close()


def test_new_entry_points_validate_their_arguments():
    """round-3 additions to the C ABI fail loudly on misuse: NULL buffers, no policy in force, nothing to continue"""
    import ctypes as C
    import ml4ca_amd
    from ml4ca_amd import _lib
    from ml4ca_amd.policy import policy_launch_form, policy_rollout
    torch = torch_()
    env = ml4ca_amd.BatchedRevoltEnv(300, auto_reset=True, current=False)
    lib, h = env.lib, env._h
    tw, epw = C.c_int32(-1), C.c_int32(-1)
    assert lib.dpenv_get_policy_launch(h, C.byref(tw), C.byref(epw)) == _lib.EINVAL and b'set_policy' in lib.dpenv_last_error(h)
    with pytest.raises(_lib.DpenvError):
        policy_launch_form(env)
    assert lib.dpenv_get_rng_counters(h, None, None, None) == _lib.EINVAL
    assert lib.dpenv_set_rng_counters(h, None, None, None) == _lib.EINVAL
    assert lib.dpenv_set_obs_thrust(h, None, None) == _lib.EINVAL
    buf = torch.zeros((300, 4), device=env.device)
    assert lib.dpenv_get_obs_thrust(h, C.c_void_p(buf.data_ptr()), None) == _lib.EINVAL       # nothing to continue yet
    assert lib.dpenv_set_current_present(h, C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr()), None) == _lib.EINVAL   # current not enabled
    make_ac(9, 7, (80, 80, 80), device=env.device).upload(env, precision='f32_actor')
    assert policy_launch_form(env) == ('two_wave', 128)             # 300 envs: the 128-env workgroup geometry
    env.reset()
    out = policy_rollout(env, 3, sample=True)
    lag = env.get_obs_thrust()
    assert lag is not None and torch.equal(lag[:, 0:3], out['last_obs'][:, 6:9])
    nc, dc = env.get_rng_counters()
    assert int(nc.min()) == 3 and int(dc.max()) == 0
    o1, _, _, _ = env.step(out['act'][0].contiguous())
    lag1 = env.get_obs_thrust()                                      # a single step (policy in force) leaves the columns of the observation it returned
    assert lag1 is not None and torch.equal(lag1[:, 0:3], o1[:, 6:9])
    fresh = ml4ca_amd.BatchedRevoltEnv(300, auto_reset=True)
    fresh.reset()
    fresh.step(out['act'][0].contiguous())
    assert fresh.get_obs_thrust() is None                            # without a policy the step path does not write them: stale
    big = ml4ca_amd.BatchedRevoltEnv(40000)
    make_ac(9, 7, (80, 80, 80), device=big.device).upload(big, precision='f16')
    assert policy_launch_form(big) == ('two_wave', 256)
    make_ac(9, 7, (80, 80, 80), device=big.device, activation='tanh').upload(big, precision='f32')
    assert policy_launch_form(big) == ('one_wave', 256)
