"""GPU tests of the in-kernel actor-critic (dpenv_policy.hip): MFMA forward pass against an fp32 torch
reference of core.py's MLP, and the policy-in-the-loop rollout against (a) that reference applied to the rows
it wrote and (b) the single-step env kernel driven with the actions it took."""
import math

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


def torch_():
    import torch
    assert torch.cuda.is_available()
    return torch


def make_ac(obs_dim, act_dim, hidden, seed, device, scale_bias=True):
    from ml4ca_amd.policy import ActorCritic
    torch = torch_()
    ac = ActorCritic(obs_dim, act_dim, hidden, seed=seed, device=device)
    g = torch.Generator().manual_seed(seed + 100)
    # asymmetric, non-zero biases and distinct log_std so that any row/column/bias mix-up shows
    for b in ac.pi_b + ac.v_b:
        b.copy_((torch.rand(b.shape, generator=g) - 0.5).to(device) * 0.6)
    ac.log_std.copy_((torch.rand(act_dim, generator=g) - 0.8).to(device))
    return ac


@pytest.mark.parametrize('mode,ext,hidden', [
    ('final_cont', True, (80, 80, 80)),      # the shipped model shape (config.json)
    ('final_cont', True, (64,)),
    ('final_cont', True, (95, 95)),
    ('final_cont', True, (96, 96, 96)),      # the widest supported: 6 k-steps, no padding row
    ('final_cont', True, (81, 81)),          # one feature into the sixth k-step
    ('final_cont', True, (33, 33, 33, 33)),
    ('limited', False, (80, 80, 80)),
    ('full', True, (48, 48)),
    ('simple', False, (80, 80, 80)),
])
def test_policy_forward_matches_fp32_reference(mode, ext, hidden):
    from ml4ca_amd.policy import policy_forward
    torch = torch_()
    n = 1000 + 7
    env, orc = H.make_pair(mode, n, ext=ext)
    ac = make_ac(env.num_states, env.num_actions, hidden, seed=3, device=env.device).upload(env)
    g = torch.Generator(device=env.device).manual_seed(5)
    obs = torch.randn((n, env.num_states), generator=g, device=env.device) * torch.tensor(
        [3, 3, 0.3, 0.5, 0.2, 0.2, 0.5, 0.5, 0.5][:env.num_states], device=env.device)
    mu, v = policy_forward(env, obs)
    mu_ref, v_ref = ac.forward_ref(obs)
    # f16 weights and activations, f32 accumulation: ~1e-3 relative to the activation scale
    scale = float(mu_ref.abs().max()) + 1.0
    assert float((mu - mu_ref).abs().max()) < 2e-3 * scale, float((mu - mu_ref).abs().max())
    assert float((v - v_ref).abs().max()) < 2e-3 * (float(v_ref.abs().max()) + 1.0)
    # it is not accidentally close: permuting the reference's outputs breaks the match
    assert float((mu - mu_ref.roll(1, dims=1)).abs().max()) > 0.05
    # exactness probe: weights and inputs representable in f16 with small integer sums -> bit-exact
    from ml4ca_amd.policy import ActorCritic
    ac2 = ActorCritic(env.num_states, env.num_actions, hidden, seed=1, device=env.device)
    gi = torch.Generator().manual_seed(9)
    for W in ac2.pi_W + ac2.v_W:
        W.copy_((torch.randint(-2, 3, W.shape, generator=gi).float() / 8).to(env.device))
    for b in ac2.pi_b + ac2.v_b:
        b.copy_((torch.randint(-4, 5, b.shape, generator=gi).float() / 4).to(env.device))
    ac2.leak = 0.25
    ac2.upload(env)
    obs_i = (torch.randint(-4, 5, (n, env.num_states), generator=gi).float() / 4).to(env.device)
    # keep magnitudes small enough that every intermediate is exactly representable in f16
    for W in ac2.pi_W[1:] + ac2.v_W[1:]:
        W.mul_(0.25)
    ac2.upload(env)
    mu2, v2 = policy_forward(env, obs_i)
    mu2_ref, v2_ref = ac2.forward_ref(obs_i)
    assert float((mu2 - mu2_ref).abs().max()) < 2e-2 * (float(mu2_ref.abs().max()) + 1)
    assert float((v2 - v2_ref).abs().max()) < 2e-2 * (float(v2_ref.abs().max()) + 1)


@pytest.mark.parametrize('activation,hidden', [('tanh', (80, 80, 80)), ('tanh', (96, 96)), ('relu', (80, 80, 80))])
def test_other_hidden_activations(activation, hidden):
    """The reference's --activation choices besides 'leaky' (train.py:24,31): tanh (Spinning Up's default) and relu,
    forward pass against fp32 torch and a closed-loop launch replayed through single steps."""
    from ml4ca_amd.policy import ActorCritic, policy_forward, policy_rollout
    torch = torch_()
    n, T = 777, 12
    env, _ = H.make_pair('final_cont', n, auto_reset=True, max_ep_len=40)
    env2, _ = H.make_pair('final_cont', n, auto_reset=True, max_ep_len=40)
    ac = ActorCritic(9, 7, hidden, seed=11, device=env.device, activation=activation)
    g = torch.Generator().manual_seed(12)
    for b in ac.pi_b + ac.v_b:
        b.copy_((torch.rand(b.shape, generator=g) - 0.5).to(env.device) * 0.6)
    for W in ac.pi_W + ac.v_W:
        W.mul_(1.5)                                   # push pre-activations into tanh's curved range
    ac.upload(env)
    assert ac.leak == (0.0 if activation == 'relu' else 0.2)
    gd = torch.Generator(device=env.device).manual_seed(13)
    obs = torch.randn((n, 9), generator=gd, device=env.device) * torch.tensor([3, 3, 0.3, 0.5, 0.2, 0.2, 0.5, 0.5, 0.5], device=env.device)
    mu, v = policy_forward(env, obs)
    mu_ref, v_ref = ac.forward_ref(obs)
    assert float((mu - mu_ref).abs().max()) < 6e-3 * (float(mu_ref.abs().max()) + 1.0), float((mu - mu_ref).abs().max())
    assert float((v - v_ref).abs().max()) < 6e-3 * (float(v_ref.abs().max()) + 1.0)
    # the same weights with the default activation give something else
    ActorCritic.upload(_with_activation(ac, 'leaky'), env2)
    mu_l, _ = policy_forward(env2, obs)
    assert float((mu_l - mu).abs().max()) > (0.05 if activation == 'tanh' else 0.01)
    # closed loop: rows are consistent with the single-step kernel and with the fp32 networks
    ac.upload(env2)
    env.reset(); env2.reset()
    noise = torch.randn((T, n, 7), generator=gd, device=env.device)
    blk = policy_rollout(env, T, noise=noise)
    o = blk['obs'][0]
    for t in range(T):
        assert torch.equal(blk['obs'][t], o)
        mu_t, v_t = ac.forward_ref(o)
        assert float((blk['act'][t] - (mu_t + torch.exp(ac.log_std) * noise[t])).abs().max()) < 2e-2
        assert float((blk['val'][t] - v_t).abs().max()) < 6e-3 * (float(v_t.abs().max()) + 1.0)
        o, r, d, _ = env2.step(blk['act'][t].contiguous())
        assert torch.equal(r, blk['rew'][t]) and torch.equal(d, blk['done'][t])
        o = o.clone()


def _with_activation(ac, activation):
    import copy
    c = copy.copy(ac)
    c.activation = activation
    c.leak = 0.2
    return c


@pytest.mark.parametrize('mode,ext,hidden', [('limited', True, (80, 80, 80)), ('full', True, (64, 64)), ('simple', False, (80, 80, 80)),
                                             ('final_wrap', False, (48,))])
def test_policy_rollout_other_variants_replay_through_single_steps(mode, ext, hidden):
    """Every env variant / observation width / network depth: the launch's stored actions, replayed through the
    single-step kernel, reproduce rewards, done bits and next observations bit for bit; actor/critic rows match fp32."""
    from ml4ca_amd.policy import policy_rollout
    torch = torch_()
    n, T = 777, 25
    kw = dict(auto_reset=True, max_ep_len=30, seed=3)
    env, _ = H.make_pair(mode, n, ext=ext, **kw)
    env2, _ = H.make_pair(mode, n, ext=ext, **kw)
    od, ad = env.num_states, env.num_actions
    ac = make_ac(od, ad, hidden, seed=7, device=env.device).upload(env)
    noise = torch.randn((T, n, ad), device=env.device)
    env.reset()
    env2.reset()
    out = policy_rollout(env, T, noise=noise)
    mu_ref, v_ref = ac.forward_ref(out['obs'].reshape(T * n, od))
    want = mu_ref.reshape(T, n, ad) + torch.exp(ac.log_std) * noise
    assert float((out['act'] - want).abs().max()) < 6e-3 * (float(mu_ref.abs().max()) + 1.0)
    assert float((out['val'] - v_ref.reshape(T, n)).abs().max()) < 6e-3 * (float(v_ref.abs().max()) + 1.0)
    for t in range(T):
        o, r, d, _ = env2.step(out['act'][t].contiguous())
        nxt = out['obs'][t + 1] if t + 1 < T else out['last_obs']
        assert torch.equal(r, out['rew'][t]) and torch.equal(d, out['done'][t]) and torch.equal(o, nxt), t
    sa, ca = env.get_state()
    sb, cb = env2.get_state()
    assert torch.equal(sa, sb) and torch.equal(ca, cb)


@pytest.mark.parametrize('noise_on', [True, False])
def test_policy_rollout_rows_are_consistent(noise_on):
    """One launch of T steps: every stored row must satisfy the reference relations
       act = mu(obs) + exp(log_std) noise,  logp = gaussian_likelihood,  val = V(obs),
    and replaying the stored actions through the single-step kernel must give the same env trajectory
    (rewards, done bits, next observations, auto-resets, final state) bit for bit."""
    from ml4ca_amd.policy import policy_rollout
    torch = torch_()
    n, T = 2000 + 11, 45
    kw = dict(auto_reset=True, max_ep_len=40, seed=8)       # T_max = 20: every env crosses an episode boundary
    env, _ = H.make_pair('final_cont', n, **kw)
    env2, _ = H.make_pair('final_cont', n, **kw)
    ac = make_ac(9, 7, (80, 80, 80), seed=2, device=env.device).upload(env)
    for W in ac.pi_W:
        W.mul_(0.7)
    ac.upload(env)
    g = torch.Generator(device=env.device).manual_seed(1)
    noise = torch.randn((T, n, 7), generator=g, device=env.device) if noise_on else None
    env.reset()
    env2.reset()
    st0, c0 = env.get_state()
    s2, c2 = env2.get_state()
    assert torch.equal(st0, s2)
    switch = (3, 30)
    refs = torch.randn((2, 3, n), generator=g, device=env.device)
    out = policy_rollout(env, T, noise=noise, switch_steps=switch, refs=refs)
    obs, act, rew, val, logp, boot, done = (out[k] for k in ('obs', 'act', 'rew', 'val', 'logp', 'boot', 'done'))
    # (a) actor / critic relations on the stored rows
    mu_ref, v_ref = ac.forward_ref(obs.reshape(T * n, 9))
    mu_ref, v_ref = mu_ref.reshape(T, n, 7), v_ref.reshape(T, n)
    std = torch.exp(ac.log_std)
    want = mu_ref + (std * noise if noise_on else 0.0)
    sc = float(mu_ref.abs().max()) + 1.0
    assert float((act - want).abs().max()) < 6e-3 * sc
    assert float((val - v_ref).abs().max()) < 6e-3 * (float(v_ref.abs().max()) + 1.0)
    if noise_on:
        # logp is evaluated against the kernel's own mu, which act - std*noise recovers
        mu_k = act - std * noise
        lp = ac.logp_ref(act.reshape(T * n, 7), mu_k.reshape(T * n, 7)).reshape(T, n)
        assert float((logp - lp).abs().max()) < 2e-3
    else:
        const = float((-ac.log_std - 0.5 * math.log(2 * math.pi)).sum())
        assert float((logp - const).abs().max()) < 1e-5
    # (b) the env trajectory: replay the stored actions with single steps
    for t in range(T):
        nr = refs[switch.index(t)] if t in switch else None
        o, r, d, _ = env2.step(act[t].contiguous(), new_ref=nr)
        assert torch.equal(r, rew[t]), 'reward t=%d' % t
        assert torch.equal(d, done[t]), 'done t=%d' % t
        nxt = obs[t + 1] if t + 1 < T else out['last_obs']
        assert torch.equal(o, nxt), 'next obs t=%d' % t
    sa, ca = env.get_state()
    sb, cb = env2.get_state()
    assert torch.equal(sa, sb) and torch.equal(ca, cb)
    assert int(ca[1].min()) >= 2                                   # every env went through a reset in the launch
    # bootstrap values (ppo.py:311): 0 where the env terminated, V(terminal obs) where only the time limit hit,
    # V(next obs) at the end of the launch, 0 elsewhere
    term = (done & 1) != 0
    tl_only = ((done & 2) != 0) & ~term
    inner = torch.ones_like(term)
    inner[T - 1] = False
    assert float(boot[term].abs().max()) == 0.0
    assert float(boot[(done == 0) & inner].abs().max()) == 0.0
    assert int(tl_only.sum()) > 100 and float(boot[tl_only].abs().min()) > 0.0
    last_alive = done[T - 1] == 0
    assert torch.equal(boot[T - 1][last_alive], out['last_val'][last_alive])
    _, v_last = ac.forward_ref(out['last_obs'])
    assert float((out['last_val'] - v_last).abs().max()) < 6e-3 * (float(v_last.abs().max()) + 1.0)
    # val[t+1] is the critic on obs[t+1] also right after a reset
    was_reset = done[:-1] != 0
    assert float((val[1:][was_reset] - v_ref[1:][was_reset]).abs().max()) < 6e-3 * (float(v_ref.abs().max()) + 1.0)


def test_policy_rollout_feeds_gae_and_ppo_buffer():
    """Config 5 end to end on device: drifting current, in-kernel policy rollout, GAE over the block with the
    kernel's bootstrap values, advantage normalisation - against the oracle's TrajectoryBuffer restatement."""
    from ml4ca_amd import rollout
    from ml4ca_amd.policy import policy_rollout
    from oracle import oracle as O
    torch = torch_()
    n, T = 1024, 60
    env, _ = H.make_pair('final_cont', n, auto_reset=True, max_ep_len=50, seed=4, current=True, current_drift=True)
    env.set_current(torch.full((n,), 0.2, device=env.device), torch.full((n,), float(np.deg2rad(135)), device=env.device))
    ac = make_ac(9, 7, (80, 80, 80), seed=6, device=env.device).upload(env)
    noise = torch.randn((T, n, 7), device=env.device)
    env.reset()
    out = policy_rollout(env, T, noise=noise)
    adv, ret = rollout.gae(out['rew'], out['val'], end=out['done'], boot=out['boot'], gamma=0.99, lam=0.97)
    orc = O.Oracle(O.make_config(), np.float32)
    oadv, oret = orc.gae(out['rew'].cpu().numpy(), out['val'].cpu().numpy(), end=out['done'].cpu().numpy(),
                         boot=out['boot'].cpu().numpy(), gamma=0.99, lam=0.97)
    assert np.allclose(adv.cpu().numpy(), oadv, rtol=1e-5, atol=1e-5)
    assert np.allclose(ret.cpu().numpy(), oret, rtol=1e-5, atol=1e-5)
    a2, mean, std = rollout.normalize_advantages(adv.clone())
    assert abs(float(a2.mean())) < 1e-4 and abs(float(a2.std(unbiased=False)) - 1.0) < 1e-3
    assert torch.isfinite(out['obs']).all() and torch.isfinite(out['logp']).all()
    # the same through the TrajectoryBuffer mirror
    env.reset()
    buf = rollout.RolloutBuffer(T, env)
    buf.collect(env, noise=noise)
    buf.finish()
    o_, a_, adv_, ret_, lp_ = buf.get()
    assert o_.shape == (T, n, 9) and a_.shape == (T, n, 7)
    tr = buf.trajectory()
    assert sum(int(np.prod(v.shape[2:])) for v in tr.values()) == 19 and all(v.shape[:2] == (T, n) for v in tr.values())
    assert abs(float(adv_.mean())) < 1e-4 and torch.isfinite(ret_).all()


def test_policy_rollout_bf16_observation_rows():
    """Config 5 stores observations as bf16: same trajectory, stored rows = round-to-nearest-even of the f32 rows."""
    from ml4ca_amd.policy import policy_rollout
    torch = torch_()
    n, T = 1500, 20
    kw = dict(auto_reset=True, max_ep_len=30, seed=5)
    e32, _ = H.make_pair('final_cont', n, **kw)
    e16, _ = H.make_pair('final_cont', n, obs_dtype='bfloat16', **kw)
    ac = make_ac(9, 7, (80, 80, 80), seed=4, device=e32.device)
    ac.upload(e32)
    ac.upload(e16)
    noise = torch.randn((T, n, 7), device=e32.device)
    e32.reset()
    e16.reset()
    a = policy_rollout(e32, T, noise=noise)
    b = policy_rollout(e16, T, noise=noise)
    assert b['obs'].dtype == torch.bfloat16 and b['last_obs'].dtype == torch.bfloat16
    assert torch.equal(a['obs'].to(torch.bfloat16), b['obs']) and torch.equal(a['last_obs'].to(torch.bfloat16), b['last_obs'])
    for k in ('act', 'rew', 'val', 'logp', 'done', 'boot'):
        assert torch.equal(a[k], b[k]), k


def test_policy_argument_validation():
    import ml4ca_amd
    from ml4ca_amd.policy import ActorCritic, policy_forward, policy_rollout
    torch = torch_()
    env, _ = H.make_pair('final_cont', 128)
    with pytest.raises(ml4ca_amd.DpenvError):
        policy_forward(env, torch.zeros((128, 9), device=env.device))              # no policy yet
    with pytest.raises(ml4ca_amd.DpenvError):
        ActorCritic(9, 6, (80, 80), device=env.device).upload(env)                 # act_dim mismatch
    with pytest.raises(ml4ca_amd.DpenvError):
        ActorCritic(9, 7, (97, 97), device=env.device).upload(env)                 # hidden width > 96
    ActorCritic(9, 7, (80, 80, 80), device=env.device).upload(env)
    env_soa, _ = H.make_pair('final_cont', 128, layout='soa')
    ActorCritic(9, 7, (80, 80, 80), device=env.device).upload(env_soa)
    with pytest.raises(ml4ca_amd.DpenvError):
        policy_rollout(env_soa, 4)


def test_trained_reference_policy_flies_the_box_test():
    """The shipped trained actor (reference data/finalmodel/finconttothighbowder_s0, fixture final_policy.npz),
    deterministic, in the build-owned plant: forward pass against the float64 expectations, then the thesis' 4-corner
    box test (results/all_plots/box_test/plot_pos.py:55-59).  The policy was trained in Cybersea, so this is a soft
    validation of the plant: it must hold station and reach every corner."""
    import os
    from ml4ca_amd import evaluate as EV
    from ml4ca_amd.policy import ActorCritic, policy_forward, policy_rollout
    torch = torch_()
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'final_policy.npz'))
    n = 256
    env, _ = H.make_pair('final_cont', n, terminate=False, time_limit=False)
    ac = ActorCritic.from_tensors({k.replace('.', '/'): d[k] for k in d.files if '.' in k}, device=env.device).upload(env)
    obs = H.to_dev(np.tile(d['obs'].astype(np.float32), (4, 1)))
    mu, v = policy_forward(env, obs)
    mu_ref, v_ref = np.tile(d['mu'], (4, 1)), np.tile(d['v'], 4)
    assert np.abs(mu.cpu().numpy() - mu_ref).max() < 6e-3 * (np.abs(mu_ref).max() + 1)
    assert np.abs(v.cpu().numpy() - v_ref).max() < 6e-3 * (np.abs(v_ref).max() + 1)
    start = torch.zeros((3, n), device=env.device)
    env.reset(init=torch.zeros((6, n), device=env.device), new_ref=start.clone())
    steps, refs = EV.box_schedule(start)
    out = policy_rollout(env, 1250, noise=None, switch_steps=steps, refs=refs)
    e = out['obs'][:, 0, :3].cpu().numpy()
    assert np.abs(e[:50]).max() < 0.05                       # station keeping on the start setpoint for 10 s
    for t in list(steps[1:]) + [1249]:                        # just before each later switch the corner is reached
        assert np.hypot(e[t - 1, 0], e[t - 1, 1]) < 0.5 and abs(np.degrees(e[t - 1, 2])) < 2.0, (t, e[t - 1])
    iae_tot, _ = EV.iae(out['obs'])
    assert 20.0 < float(iae_tot.mean()) < 150.0
    assert float(out['rew'].mean()) > 2.0


def test_plant_tracks_the_recorded_cybersea_box_run():
    """Soft validation of the BUILD-OWNED plant (no parity claim): the trained actor, fed the recorded filtered
    setpoint series of the thesis' Cybersea box test (fixture cybersea_box_rl.npz from results/all_plots/box_test/
    bagfile__RL_*), must keep this plant's pose close to the recorded Cybersea pose.  Measured when written:
    RMS 0.17 m N, 0.23 m E, 3.6 deg yaw over 251 s."""
    import os
    from ml4ca_amd.policy import ActorCritic, policy_forward
    torch = torch_()
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    rec = np.load(os.path.join(g, 'cybersea_box_rl.npz'))
    d = np.load(os.path.join(g, 'final_policy.npz'))
    refs = np.ascontiguousarray(rec['setpoint'].T.astype(np.float32))
    T = refs.shape[1] - 1
    env, _ = H.make_pair('final_cont', 1, terminate=False, time_limit=False, wrap_mode='radians')   # ROS node wraps in rad
    ActorCritic.from_tensors({k.replace('.', '/'): d[k] for k in d.files if '.' in k}, device=env.device).upload(env)
    obs = env.reset(init=torch.zeros((6, 1), device=env.device), new_ref=H.to_dev(refs[:, :1]))
    traj = np.zeros((T, 3))
    acts = []
    for k in range(T):
        mu, _ = policy_forward(env, obs)
        acts.append(mu.clone())
        obs, _, _, _ = env.step(mu.contiguous(), new_ref=H.to_dev(refs[:, k + 1:k + 2]))
        st, _ = env.get_state()
        traj[k] = st[0:3, 0].cpu().numpy()
    dev = traj - rec['pose'][1:]
    rms = np.sqrt((dev ** 2).mean(0))
    assert rms[0] < 0.5 and rms[1] < 0.5 and np.degrees(rms[2]) < 8.0, rms
    assert np.abs(dev[:, :2]).max() < 1.5
    # the energy the actor spends on the manoeuvre in this plant against what it spent in Cybersea (the thesis' "fuel":
    # integral of the thruster power model, plot_act.py:128-135, over the RECORDED commands of the same run)
    from ml4ca_amd import evaluate as EV
    w_here = float(EV.work(EV.commanded_thrust(torch.stack(acts))).sum())
    cmd = np.load(os.path.join(g, 'cybersea_replay.npz'))['box_test_RL_n'].astype(np.float32)
    w_cyb = float(EV.work(torch.from_numpy(cmd)[:, None, :]).sum())
    assert 600.0 < w_cyb < 850.0                       # 720: the recorded run
    assert 0.85 * w_cyb < w_here < 1.15 * w_cyb, (w_here, w_cyb)      # measured 727 against 720
    # ... and the tracking error it leaves: IAE (common.py:56-74, normalised by [5 m, 5 m, 25 deg]) against the filtered
    # setpoint series, for this plant's trajectory and for the recorded one
    sp = rec['setpoint'][1:].astype(np.float64)
    iae_here = float(EV.iae(torch.from_numpy(traj - sp)[:, None, :])[0])
    iae_cyb = float(EV.iae(torch.from_numpy(rec['pose'][1:].astype(np.float64) - sp)[:, None, :])[0])
    assert 0.6 * iae_cyb < iae_here < 1.4 * iae_cyb, (iae_here, iae_cyb)
    print('work in this plant %.0f, in Cybersea %.0f; IAE vs the filtered setpoints %.2f here, %.2f in Cybersea' % (w_here, w_cyb, iae_here, iae_cyb))


def test_recorded_cybersea_commands_open_loop_through_kernel():
    """tests/golden/cybersea_replay.npz: 392 ten-second windows of seven recorded Cybersea runs, the recorded thruster
    commands applied open loop (one env per window, the fused rollout kernel, constant 0.2 m/s current where the run
    had one).  The kernel's predicted poses must agree with the float64 oracle's replay (fp32 over 50 steps) and
    reproduce its error statistics against the record (soft validation of the BUILD-OWNED plant)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'calibration'))
    import replay_cybersea as RC
    torch = torch_()
    W = RC.load_windows()
    n, T = W['eta0'].shape[0], W['act'].shape[0]
    env, _ = H.make_pair('final_cont', n, terminate=False, time_limit=False, current=True)
    env.set_current(H.to_dev(W['current'][0].astype(np.float32)), H.to_dev(W['current'][1].astype(np.float32)))
    env.reset(init=H.to_dev(np.concatenate([W['eta0'].T, W['nu0'].T], 0).astype(np.float32)),
              new_ref=H.to_dev(np.zeros((3, n), np.float32)))
    pred = np.zeros((T, n, 3))
    for t in range(T):
        env.step(H.to_dev(W['act'][t].astype(np.float32)))
        st, _ = env.get_state()
        pred[t] = st[0:3].T.cpu().numpy()
    ref = RC.replay_oracle(W)
    assert np.abs(pred[..., :2] - ref[..., :2]).max() < 2e-3 and np.abs(pred[..., 2] - ref[..., 2]).max() < 2e-3
    e, eo = RC.errors(pred, W), RC.errors(ref, W)
    for h in RC.HORIZONS:
        assert abs(e[h][0] - eo[h][0]) < 1e-3 and abs(e[h][1] - eo[h][1]) < 0.05, (h, e[h], eo[h])
    assert e[50][0] < 0.65 and e[50][1] < 17.0


def test_run_RL_policy_harness_with_the_trained_actor():
    """test_policy.py:97-186 semantics, batched: six fixed starts on the 5 m circle, deterministic trained actor,
    optional setpoint change at half time through a zero-action step."""
    import os
    import ml4ca_amd
    from ml4ca_amd import evaluate as EV
    from ml4ca_amd.policy import ActorCritic
    torch = torch_()
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'final_policy.npz'))
    tensors = {k.replace('.', '/'): d[k] for k in d.files if '.' in k}
    for changes in (False, True):
        env = ml4ca_amd.BatchedRevoltEnv(6, testing=True, time_limit=False)
        ac = ActorCritic.from_tensors(tensors, device=env.device).upload(env)
        res = EV.run_RL_policy(env, ac, num_episodes=6, test_setpoint_changes=changes)
        T = env.max_ep_len
        assert res['obs'].shape == (T + 1, 6, 9) and res['action_vec'].shape == (T + 1, 6, 6)
        pos0 = res['ned_pos'][0].cpu().numpy()
        assert np.allclose(np.hypot(pos0[:, 0], pos0[:, 1]), 5.0, atol=1e-5)          # simtools.py:91-107
        assert np.allclose(pos0[:, 2], np.radians([0, 0, -15, 15, 0, -15]), atol=1e-6)
        assert float(res['action_vec'][0, :, 3].min()) == float(res['action_vec'][0, :, 3].max())   # bow azimuth default pi/2
        assert abs(float(res['action_vec'][0, 0, 3]) - np.pi / 2) < 1e-6
        eplen = res['EpLen'].cpu().numpy()
        if not changes:
            assert (eplen == T).all(), 'the trained actor must not run any start out of bounds'
            ok = np.ones(6, bool)
        else:
            # refs[2] and refs[3] ask for a 90 deg heading step: |yaw error| > pi/4 is terminal in the final env
            # (customEnv.py:386), so - exactly as in the reference's harness - those two episodes end right after
            # the new setpoint becomes visible
            assert list(eplen) == [T, T, T // 2 + 2, T // 2 + 2, T, T], eplen
            ok = eplen == T
        # it converges on the setpoint in force: within 0.5 m / 3 deg at the end, and the return is near the thesis' (~1200-1300)
        err = (res['ned_pos'][-1] - res['ned_ref'][-1]).cpu().numpy()[ok]
        assert np.hypot(err[:, 0], err[:, 1]).max() < 0.5 and np.degrees(np.abs(err[:, 2])).max() < 3.0, err
        ret = res['EpRet'].cpu().numpy()[ok]
        assert ret.min() > (700 if changes else 900) and ret.max() < 1400, ret
        if changes:
            ref_end = res['ned_ref'][-1].cpu().numpy()
            assert np.allclose(ref_end, np.array(EV.TEST_POLICY_REFS + (EV.TEST_POLICY_REFS[0],))[:6], atol=1e-6)
            # the switching step used a zero action and restored the default action vector (test_policy.py:148-153)
            assert torch.equal(res['action_vec'][T // 2 + 1], res['action_vec'][0])
            assert np.allclose(res['ned_ref'][T // 2].cpu().numpy(), 0.0)              # visible one step late (Q4)
        env.close()


def test_two_wave_rollout_equals_single_wave_rollout_repeatedly():
    """The default launch form puts an env wave and a network wave on every SIMD (policy_rollout_ws_kernel); with
    launch_form='one_wave' one wave does both.  Same chains of arithmetic: every row must be identical bit for bit, and must
    stay so over repeated launches - built with packed fp32 the env wave once lost one term of its position sum in lanes 48-63
    now and then beside the network wave's MFMAs (hence -fno-slp-vectorize)."""
    from ml4ca_amd.policy import policy_rollout
    torch = torch_()
    n, T = 2000 + 11, 45
    kw = dict(auto_reset=True, max_ep_len=40, seed=8, current=True, current_drift=True)
    outs = {}
    for form, reps in (('one_wave', 1), ('two_wave', 25)):
        for rep in range(reps):
            env, _ = H.make_pair('final_cont', n, **kw)
            make_ac(9, 7, (80, 80, 80), seed=2, device=env.device).upload(env, launch_form=form)
            g = torch.Generator(device=env.device).manual_seed(1)
            env.set_current(torch.full((n,), 0.2, device=env.device), torch.full((n,), 2.3, device=env.device))
            env.reset()
            refs = torch.randn((2, 3, n), generator=g, device=env.device)
            noise = torch.randn((T, n, 7), generator=g, device=env.device) if rep % 2 else None
            out = policy_rollout(env, T, noise=noise, switch_steps=(3, 30), refs=refs)
            st, ctr = env.get_state()
            got = {k: v.clone() for k, v in out.items()}
            got['state'], got['ctr'] = st, ctr
            key = 'noise' if noise is not None else 'det'
            if form == 'one_wave' and key not in outs:
                outs[key] = got
                if 'noise' not in outs:          # the single-wave reference for the other input too
                    noise2 = torch.randn((T, n, 7), generator=g, device=env.device)
                    env_b, _ = H.make_pair('final_cont', n, **kw)
                    make_ac(9, 7, (80, 80, 80), seed=2, device=env_b.device).upload(env_b, launch_form=form)
                    env_b.set_current(torch.full((n,), 0.2, device=env.device), torch.full((n,), 2.3, device=env.device))
                    env_b.reset()
                    ob = policy_rollout(env_b, T, noise=noise2, switch_steps=(3, 30), refs=refs)
                    sb, cb = env_b.get_state()
                    outs['noise'] = {k: v.clone() for k, v in ob.items()}
                    outs['noise']['state'], outs['noise']['ctr'] = sb, cb
                continue
            want = outs[key]
            for k in want:
                assert torch.equal(got[k], want[k]), '%s launch, repetition %d: %s differs' % (form, rep, k)


@pytest.mark.parametrize("case", range(24))
def test_two_wave_rollout_random_configurations(case):
    """The two launch forms of dpenv_policy_rollout against each other over drawn configurations: variant, observation
    width, network shape and activation, ragged env counts (partly filled waves and pairs without envs), auto-reset
    with short episodes (previous thrust drawn at reset or not), vessel classes, current with drift, bf16 rows, setpoint
    switches, noise block / in-kernel noise / deterministic, odd T.  Every output block and the final state must be identical
    bit for bit."""
    from ml4ca_amd.policy import ActorCritic, policy_rollout
    torch = torch_()
    rng = np.random.RandomState(4200 + case)
    mode = ['full', 'simple', 'limited', 'final_wrap', 'final_cont'][rng.randint(5)]
    ext = bool(rng.randint(2)) and mode != 'simple'
    n = int(rng.choice([1, 63, 65, 200, 257, 1000, 2049]))
    T = int(rng.choice([1, 2, 7, 30, 61]))
    hidden = [(80, 80, 80), (64,), (96, 96), (33, 33, 33, 33)][rng.randint(4)]
    activation = ['leaky', 'relu', 'tanh'][rng.randint(3)]
    kw = dict(auto_reset=bool(rng.randint(2)), max_ep_len=int(rng.choice([20, 40, 800])), seed=int(rng.randint(100)),
              current=bool(rng.randint(2)), obs_dtype=['float32', 'bfloat16'][rng.randint(2)])
    kw['current_drift'] = kw['current'] and bool(rng.randint(2))
    use_noise, n_sw = bool(rng.randint(2)), int(rng.randint(3))
    switch = tuple(sorted(rng.choice(T, size=min(n_sw, T), replace=False).tolist()))
    # drawn per case as well: in-kernel exploration noise, vessel classes, previous thrust drawn at reset
    sample = (not use_noise) and bool(rng.randint(2))
    n_cls = int(rng.choice([1, 1, 3]))
    kw['reset_acts'] = bool(rng.randint(2)) and ext
    vp = None
    if n_cls > 1:
        from ml4ca_amd import _lib as L
        base = L.default_vessel()
        vp = np.stack([base * (1.0 + 0.15 * c * (np.arange(L.NPARAM) < 12)) for c in range(n_cls)]).astype(np.float32)
    res = {}
    for ws, form in (('0', 'one_wave'), ('1', 'two_wave')):
        env, _ = H.make_pair(mode, n, ext=ext, vessel_params=vp, **kw)
        ac = ActorCritic(env.num_states, env.num_actions, hidden, seed=case, device=env.device, activation=activation)
        g = torch.Generator().manual_seed(case)
        for b in ac.pi_b + ac.v_b:
            b.copy_((torch.rand(b.shape, generator=g) - 0.5).to(env.device) * 0.4)
        ac.upload(env, launch_form=form)
        gd = torch.Generator(device=env.device).manual_seed(case)
        if n_cls > 1:
            env.set_vessel_class(torch.randint(0, n_cls, (n,), generator=gd, device=env.device, dtype=torch.int32))
        if kw['current']:
            env.set_current(torch.rand(n, generator=gd, device=env.device) * 0.3, torch.rand(n, generator=gd, device=env.device) * 6.0 - 3.0)
        env.reset()
        refs = torch.randn((len(switch), 3, n), generator=gd, device=env.device) if switch else None
        noise = torch.randn((T, n, env.num_actions), generator=gd, device=env.device) if use_noise else None
        out = policy_rollout(env, T, noise=noise, switch_steps=switch, refs=refs, sample=sample)
        st, ctr = env.get_state()
        res[ws] = {k: v.clone() for k, v in out.items()}
        res[ws]['state'], res[ws]['ctr'] = st, ctr
        if kw['current']:
            res[ws]['vc'], res[ws]['beta'] = env.get_current()
    for other in ('1',):
        for k in res['0']:
            a_, b_ = res['0'][k], res[other][k]
            assert torch.equal(a_.float() if a_.dtype == torch.bfloat16 else a_, b_.float() if b_.dtype == torch.bfloat16 else b_), \
                '%s differs in form %s (mode %s ext %s n %d T %d hidden %s %s %s)' % (k, other, mode, ext, n, T, hidden, activation, kw)
