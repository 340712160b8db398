"""GPU tests of the round-2 policy path: the fp32-faithful (split-f16) network arithmetic, in-kernel exploration noise against
the oracle's stream, device-side weight packing, vessel classes and reset_acts inside the closed-loop kernels, the one-pass
GAE + statistics."""
import math

import numpy as np
import pytest

from tests import helpers as H
from tests.test_gpu_policy import make_ac, torch_

pytestmark = pytest.mark.gpu


def ref64(ac, obs):
    """float64 evaluation of core.py:29-33 with the fp32 parameters: the yardstick for the 1e-5 claims (oracle/policy_ref.py: a
    restatement, pinned since round 4 to an execution of the reference's own saved GraphDef - tests/test_policy_import_cpu.py; the
    slope is the float32 value the kernel is given)."""
    from oracle import policy_ref as PR
    torch = torch_()
    mu, v = PR.actor_critic(ac.state_dict(), obs.detach().double().cpu().numpy(), activation=ac.activation, leak=float(np.float32(ac.leak)))
    return torch.from_numpy(mu).to(obs.device), torch.from_numpy(v).to(obs.device)


def test_in_kernel_networks_against_the_reference_graph_on_the_thesis_checkpoint():
    """The thesis' trained actor-critic (tests/golden/final_policy.npz) through dpenv_policy_forward in DPENV_POLICY_F32 against what the
    reference's OWN saved graph computes for the same observations (tests/golden/final_graph_vectors.npz: saved_model.pb executed node by
    node in float64 by ml4ca_amd/tf_graph.py, tests/golden/gen_final_graph.py): mu and v within 1e-5 of the output scale; and the
    log-likelihood the rollout kernel stores for a forced action (noise block = the fixture's xi) against the graph's pi/Sum_1."""
    import os
    from ml4ca_amd.policy import ActorCritic, policy_forward, policy_rollout
    torch = torch_()
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    d = np.load(os.path.join(G, 'final_policy.npz'))
    vec = np.load(os.path.join(G, 'final_graph_vectors.npz'))
    env, _ = H.make_pair('final_cont', 64)
    ac = ActorCritic.from_tensors({k.replace('.', '/'): d[k] for k in d.files if '.' in k}, device=env.device).upload(env, precision='f32')
    obs = torch.from_numpy(vec['obs']).to(env.device)
    mu, v = policy_forward(env, obs)
    s_mu, s_v = np.abs(vec['mu_f64']).max() + 1.0, np.abs(vec['v_f64']).max() + 1.0
    e_mu, e_v = np.abs(mu.double().cpu().numpy() - vec['mu_f64']).max(), np.abs(v.double().cpu().numpy() - vec['v_f64']).max()
    assert e_mu < 1e-5 * s_mu and e_v < 1e-5 * s_v, (e_mu / s_mu, e_v / s_v)
    # one closed-loop step from a reset: act = mu(o_0) + exp(log_std) xi and logp as the graph's pi/add and pi/Sum_1 give them for o_0
    env.reset()
    noise = torch.from_numpy(vec['xi'].astype(np.float32)).to(env.device).reshape(1, 64, 7)
    out = policy_rollout(env, 1, noise=noise)
    from ml4ca_amd import tf_graph as TG
    from tests.test_policy_import_cpu import graph_fixture
    _, nodes, _ = graph_fixture()
    t = {k.replace('.', '/'): d[k] for k in d.files if '.' in k}
    o0 = out['obs'][0].cpu().numpy()
    xi32 = vec['xi'].astype(np.float32).astype(np.float64)
    pi, lp = TG.evaluate(nodes, ['pi/add', 'pi/Sum_1'], {'Placeholder': o0}, t, rng_normal=lambda shape: xi32.reshape(shape))
    assert np.abs(out['act'][0].double().cpu().numpy() - pi).max() < 1e-5 * (np.abs(pi).max() + 1.0)
    assert np.abs(out['logp'][0].double().cpu().numpy() - lp).max() < 1e-5 * (np.abs(lp).max() + 1.0)


@pytest.mark.parametrize('mode,ext,hidden,activation', [
    ('final_cont', True, (80, 80, 80), 'leaky'),      # the shipped model shape (config.json)
    ('final_cont', True, (96, 96), 'leaky'),            # 6 k-steps; three such layers would not fit the LDS in split form
    ('final_cont', True, (81, 81), 'tanh'),
    ('final_cont', True, (64,), 'relu'),
    ('limited', False, (80, 80, 80), 'leaky'),
    ('full', True, (48, 48), 'tanh'),
    ('simple', False, (33, 33, 33), 'leaky'),
])
def test_policy_forward_matches_fp32_reference(mode, ext, hidden, activation):
    """DPENV_POLICY_F32: mu and v within 1e-5 of the output scale of the reference's fp32 networks (core.py:29-33,80-107),
    measured against a float64 evaluation of the same fp32 parameters.  The f16 fast mode is ~100x further away."""
    from ml4ca_amd.policy import ActorCritic, policy_forward
    torch = torch_()
    n = 1000 + 7
    env, _ = H.make_pair(mode, n, ext=ext)
    ac = ActorCritic(env.num_states, env.num_actions, hidden, seed=3, device=env.device, activation=activation)
    g = torch.Generator().manual_seed(103)
    for b in ac.pi_b + ac.v_b:
        b.copy_((torch.rand(b.shape, generator=g) - 0.5).to(env.device) * 0.6)
    ac.log_std.copy_((torch.rand(env.num_actions, generator=g) - 0.8).to(env.device))
    gd = torch.Generator(device=env.device).manual_seed(5)
    obs = torch.randn((n, env.num_states), generator=gd, device=env.device) * torch.tensor(
        [3, 3, 0.3, 0.5, 0.2, 0.2, 0.5, 0.5, 0.5][:env.num_states], device=env.device)
    mu64, v64 = ref64(ac, obs)
    ac.upload(env, precision='f32')
    mu, v = policy_forward(env, obs)
    s_mu, s_v = float(mu64.abs().max()) + 1.0, float(v64.abs().max()) + 1.0
    e_mu, e_v = float((mu.double() - mu64).abs().max()), float((v.double() - v64).abs().max())
    assert e_mu < 1e-5 * s_mu and e_v < 1e-5 * s_v, (e_mu / s_mu, e_v / s_v)
    ac.upload(env, precision='f16')
    mu16, _ = policy_forward(env, obs)
    assert float((mu16.double() - mu64).abs().max()) > 20 * e_mu       # the fast mode is a different arithmetic


def test_exact_mode_rollout_rows_match_fp32_networks_and_single_steps():
    """One launch in DPENV_POLICY_F32: act = mu(obs) + std xi, val = V(obs) and logp = gaussian_likelihood(act | mu(obs)) (core.py:42-46)
    to 1e-5 against the float64 yardstick - so the PPO ratio exp(logp_new - logp_old) of an fp32 update starts at 1 - and the
    stored actions replayed through the single-step kernel give the same trajectory bit for bit."""
    from ml4ca_amd.policy import policy_rollout
    torch = torch_()
    n, T = 1000 + 3, 30
    kw = dict(auto_reset=True, max_ep_len=40, seed=8)
    env, _ = H.make_pair('final_cont', n, **kw)
    env2, _ = H.make_pair('final_cont', n, **kw)
    ac = make_ac(9, 7, (80, 80, 80), seed=2, device=env.device).upload(env, precision='f32')
    g = torch.Generator(device=env.device).manual_seed(1)
    noise = torch.randn((T, n, 7), generator=g, device=env.device)
    env.reset(); env2.reset()
    refs = torch.randn((1, 3, n), generator=g, device=env.device)
    out = policy_rollout(env, T, noise=noise, switch_steps=(4,), refs=refs)
    obs, act, val, logp = out['obs'], out['act'], out['val'], out['logp']
    mu64, v64 = ref64(ac, obs.reshape(T * n, 9))
    mu64, v64 = mu64.reshape(T, n, 7), v64.reshape(T, n)
    std = torch.exp(ac.log_std.double())
    sc = float(mu64.abs().max()) + 1.0
    assert float((act.double() - (mu64 + std * noise.double())).abs().max()) < 1e-5 * sc
    assert float((val.double() - v64).abs().max()) < 1e-5 * (float(v64.abs().max()) + 1.0)
    z = (act.double() - mu64) / (std + 1e-8)
    lp64 = (-0.5 * (z * z + 2 * ac.log_std.double() + math.log(2 * math.pi))).sum(-1)
    assert float((logp.double() - lp64).abs().max()) < 1e-5 * (float(lp64.abs().max()) + 1.0)
    for t in range(T):
        o, r, d, _ = env2.step(act[t].contiguous(), new_ref=refs[0] if t == 4 else None)
        nxt = obs[t + 1] if t + 1 < T else out['last_obs']
        assert torch.equal(r, out['rew'][t]) and torch.equal(d, out['done'][t]) and torch.equal(o, nxt), t
    _, vl = ref64(ac, out['last_obs'])
    assert float((out['last_val'].double() - vl).abs().max()) < 1e-5 * (float(vl.abs().max()) + 1.0)
    # bootstrap values where only the time limit cut the path: V of the terminal observation, to the same tolerance
    done, boot = out['done'], out['boot']
    tl = ((done & 2) != 0) & ((done & 1) == 0)
    assert int(tl.sum()) > 50 and float(boot[tl].abs().min()) > 0.0


@pytest.mark.parametrize('form,precision', [('two_wave', 'f16'), ('one_wave', 'f16'), ('one_wave', 'f32')])
def test_in_kernel_noise_is_the_oracle_stream_and_shard_invariant(form, precision):
    """sample=True: the kernel draws xi itself (core.py:85).  (a) act - mu(obs) = std * xi with xi = the oracle's Philox / Box-Muller
    stream keyed (seed; global env id, number of actions sampled so far); (b) a second launch continues the stream; (c) two
    shards with env_id_base offsets reproduce the single-handle trajectory bit for bit (rank-count invariance)."""
    from ml4ca_amd.policy import policy_forward, policy_rollout
    torch = torch_()
    n, T = 512 + 5, 9
    kw = dict(seed=21, auto_reset=True, max_ep_len=12)
    env, orc = H.make_pair('final_cont', n, **kw)
    ac = make_ac(9, 7, (80, 80, 80), seed=4, device=env.device).upload(env, precision=precision, launch_form=form)
    env.reset()
    outs = [{k: v.clone() for k, v in policy_rollout(env, T, sample=True).items()} for _ in range(2)]
    std = torch.exp(ac.log_std)
    tol = 2e-5 if precision == 'f32' else 2e-2
    for launch, out in enumerate(outs):
        mu, _ = policy_forward(env, out['obs'].reshape(T * n, 9))          # the kernel's own arithmetic for mu
        xi = ((out['act'] - mu.reshape(T, n, 7)) / std).cpu().numpy()
        for t in (0, 1, T - 1):
            want = orc.policy_noise(np.arange(n), launch * T + t, 7)
            assert np.abs(xi[t] - want).max() < tol + 2e-5 * np.abs(want).max(), (launch, t, np.abs(xi[t] - want).max())
        z = (out['act'] - mu.reshape(T, n, 7)) / (std + 1e-8)
        lp = (-0.5 * (z * z + 2 * ac.log_std + math.log(2 * math.pi))).sum(-1)
        assert float((out['logp'] - lp).abs().max()) < (1e-4 if precision == 'f32' else 5e-2)
    full = torch.cat([o['act'] for o in outs])
    assert 0.9 < float(((full - torch.cat([policy_forward(env, o['obs'].reshape(T * n, 9))[0].reshape(T, n, 7) for o in outs])) / std).std()) < 1.1
    # (c) shards
    n_a = 256
    parts = []
    for base, cnt in ((0, n_a), (n_a, n - n_a)):
        e, _ = H.make_pair('final_cont', cnt, env_id_base=base, **kw)
        make_ac(9, 7, (80, 80, 80), seed=4, device=e.device).upload(e, precision=precision, launch_form=form)
        e.reset()
        parts.append([{k: v.clone() for k, v in policy_rollout(e, T, sample=True).items()} for _ in range(2)])
    for launch in range(2):
        for k in ('obs', 'act', 'rew', 'val', 'logp', 'done', 'boot'):
            joined = torch.cat([parts[0][launch][k], parts[1][launch][k]], dim=1)
            assert torch.equal(joined, outs[launch][k]), (launch, k)


def test_device_pointer_upload_equals_host_upload_and_needs_no_sync():
    """dpenv_policy_desc.device_pointers: the optimiser's own CUDA tensors are packed by one kernel on the stream.  Same image as
    the host-pointer path (forward outputs identical), re-upload after an in-place parameter change takes effect in stream
    order, and the call can be captured in a graph (no host synchronisation, no allocation after the first upload)."""
    from ml4ca_amd.policy import ActorCritic, policy_forward
    torch = torch_()
    n = 777
    env, _ = H.make_pair('final_cont', n)
    env_h, _ = H.make_pair('final_cont', n)
    ac = make_ac(9, 7, (80, 80, 80), seed=9, device=env.device)
    ac_cpu = ActorCritic.from_tensors(ac.state_dict(), device='cpu')
    obs = torch.randn((n, 9), device=env.device)
    for prec in ('f16', 'f32'):
        ac.upload(env, precision=prec)                         # device pointers
        ac_cpu.upload(env_h, precision=prec)                   # host pointers
        mu_d, v_d = policy_forward(env, obs)
        mu_h, v_h = policy_forward(env_h, obs)
        assert torch.equal(mu_d, mu_h) and torch.equal(v_d, v_h), prec
    ac.upload(env, precision='f16')
    mu0, _ = policy_forward(env, obs)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ac.upload(env)                                         # warm: buffers exist
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            ac.upload(env)
    torch.cuda.current_stream().wait_stream(s)
    with torch.no_grad():
        for W in ac.pi_W:
            W.mul_(0.5)
    graph.replay()                                             # re-packs from the CURRENT parameter values
    mu1, _ = policy_forward(env, obs)
    ac.upload(env)
    mu2, _ = policy_forward(env, obs)
    assert torch.equal(mu1, mu2) and not torch.equal(mu0, mu1)


def test_launch_form_and_lds_footprint_validation():
    from ml4ca_amd.policy import ActorCritic, policy_rollout
    from ml4ca_amd._lib import DpenvError
    torch = torch_()
    env, _ = H.make_pair('final_cont', 300)
    wide = ActorCritic(9, 7, (96, 96, 96, 96), device=env.device)
    with pytest.raises(DpenvError):
        wide.upload(env, launch_form='two_wave')               # 138 KiB of networks + 50 KiB of mailboxes > 160 KiB
    wide.upload(env)                                            # auto: falls back to the one-wave form, and launches
    env.reset()
    out = policy_rollout(env, 3)
    assert bool(torch.isfinite(out['val']).all())
    with pytest.raises(DpenvError):
        wide.upload(env, precision='f32')                      # 2 x 138 KiB of split fragments do not fit at all
    # round 3: the compact weight image (28.75 KiB per network and image for 9-80-80-80) lets all four images of the split arithmetic
    # share the LDS with the two-wave form's mailboxes; tanh has no two-wave form in the split arithmetics
    ActorCritic(9, 7, (80, 80, 80), device=env.device).upload(env, precision='f32', launch_form='two_wave')
    out = policy_rollout(env, 3)
    assert bool(torch.isfinite(out['val']).all())
    with pytest.raises(DpenvError):
        ActorCritic(9, 7, (80, 80, 80), device=env.device, activation='tanh').upload(env, precision='f32', launch_form='two_wave')
    ActorCritic(9, 7, (80, 80, 80), device=env.device, activation='tanh').upload(env, precision='f32')     # auto: one wave
    ac = ActorCritic(9, 7, (80, 80, 80), leak=1.5, device=env.device)
    with pytest.raises(DpenvError):
        ac.upload(env)                                          # slope outside [0, 1]


@pytest.mark.parametrize('form,precision', [('two_wave', 'f16'), ('one_wave', 'f16'), ('one_wave', 'f32')])
def test_vessel_classes_and_reset_acts_in_the_closed_loop_kernels(form, precision):
    """Domain randomisation where PPO runs (north star: per-env mass / damping blocks): dpenv_policy_rollout with three vessel
    classes, previous thrust drawn at every in-kernel reset (customEnv.py:179-188) and a drifting current; the stored actions
    replayed through the single-step kernel (which stages the class table in LDS) give the same trajectory bit for bit, and
    the classes do differ."""
    from ml4ca_amd import _lib as L
    from ml4ca_amd.policy import policy_rollout
    torch = torch_()
    n, T = 1000 + 9, 40
    base = L.default_vessel()
    vp = np.stack([base, base * np.where(np.arange(L.NPARAM) < 4, 1.4, 1.0), base * np.where((np.arange(L.NPARAM) >= 4) & (np.arange(L.NPARAM) < 12), 0.6, 1.0)]).astype(np.float32)
    kw = dict(auto_reset=True, max_ep_len=30, seed=12, vessel_params=vp, reset_acts=True, current=True, current_drift=True)
    env, _ = H.make_pair('final_cont', n, **kw)
    env2, _ = H.make_pair('final_cont', n, **kw)
    g = torch.Generator(device=env.device).manual_seed(3)
    cls = torch.randint(0, 3, (n,), generator=g, device=env.device, dtype=torch.int32)
    for e in (env, env2):
        e.set_vessel_class(cls)
        e.set_current(torch.full((n,), 0.15, device=e.device), torch.full((n,), 1.0, device=e.device))
        e.reset()
    make_ac(9, 7, (80, 80, 80), seed=2, device=env.device).upload(env, precision=precision, launch_form=form)
    out = policy_rollout(env, T, sample=True)
    for t in range(T):
        o, r, d, _ = env2.step(out['act'][t].contiguous())
        nxt = out['obs'][t + 1] if t + 1 < T else out['last_obs']
        assert torch.equal(r, out['rew'][t]) and torch.equal(d, out['done'][t]) and torch.equal(o, nxt), t
    sa, ca = env.get_state()
    sb, cb = env2.get_state()
    assert torch.equal(sa, sb) and torch.equal(ca, cb) and int(ca[1].min()) >= 2
    # reset observations carry the drawn previous thrust
    was_reset = out['done'][:-1] != 0
    pt0 = out['obs'][1:][was_reset][:, 6:9].float()
    assert 0.08 < float(pt0.std()) < 0.12 and float(pt0.abs().max()) <= 1.0
    # the classes are really in force: same actions on a single-class env give another trajectory
    env3, _ = H.make_pair('final_cont', n, **dict(kw, vessel_params=None))
    env3.set_current(torch.full((n,), 0.15, device=env.device), torch.full((n,), 1.0, device=env.device))
    env3.reset()
    o3, r3, _, _ = env3.step(out['act'][0].contiguous())
    heavy = cls == 1
    assert torch.equal(r3[cls == 0], out['rew'][0][cls == 0]) and not torch.equal(r3[heavy], out['rew'][0][heavy])


def test_reset_acts_kernel_matches_oracle():
    """customEnv.py:179-188 inside reset_kernel and the in-step auto-reset against the oracle's restatement of the same draw:
    the oracle free-runs beside the kernel over an episode boundary (resynchronised each step so that only one step's
    arithmetic is compared)."""
    from tests.tolerances import assert_close, OBS_FLOOR
    torch = torch_()
    n = 2000 + 3
    env, orc = H.make_pair('final_cont', n, reset_acts=True, auto_reset=True, max_ep_len=20, seed=31)
    st, ctr = orc.new_state(n)
    obs_o = orc.reset(st, ctr)
    obs = env.reset()
    sd, cd = env.get_state()
    assert np.array_equal(cd.cpu().numpy(), ctr)
    assert_close(obs.cpu().numpy(), obs_o, OBS_FLOOR, what='reset obs')
    pt = sd.cpu().numpy()[9:12]
    assert np.abs(pt - st[9:12]).max() < 2e-4 and 9.0 < pt.std() < 11.0
    rng = np.random.RandomState(0)
    for t in range(11):                                        # crosses the time limit (10 agent steps): auto-reset draws again
        a = H.random_actions(rng, n, 7)
        st = sd.cpu().numpy().copy()
        ctr = cd.cpu().numpy().copy()
        o, r, d, _ = env.step(H.to_dev(a))
        oo, ro, do = orc.step(st, ctr, a)
        assert np.array_equal(d.cpu().numpy(), do), t
        assert_close(o.cpu().numpy(), oo, OBS_FLOOR, what='obs t=%d' % t)
        sd, cd = env.get_state()
        assert np.array_equal(cd.cpu().numpy(), ctr), t
    assert int(ctr[1].min()) == 2


def test_gae_one_pass_statistics_and_normalisation():
    """dpenv_gae_stats: advantages / returns equal to the oracle's fp32 scan, the (sum, sum of squares) pair equal to a float64
    reduction of the advantages and bit-identical between runs; one-pass normalisation = the reference's three-step one
    (ppo.py:99-103, mpi_tools.py:71-92); the four-columns-per-lane and the scalar form agree; two half-blocks whose statistics are
    added (the all-reduce) normalise exactly like the whole block."""
    from ml4ca_amd import rollout
    from oracle import oracle as O
    torch = torch_()
    orc = O.Oracle(O.make_config(), np.float32)
    rng = np.random.RandomState(5)
    for T, n in ((400, 4096), (37, 1000), (50, 1003), (1, 64), (9, 2)):
        rew = rng.randn(T, n).astype(np.float32)
        val = rng.randn(T, n).astype(np.float32)
        end = (rng.rand(T, n) < 0.05).astype(np.uint8)
        boot = (rng.randn(T, n) * (rng.rand(T, n) < 0.5)).astype(np.float32)
        adv_o, ret_o = orc.gae(rew, val, end=end, boot=boot)
        stats = torch.zeros(2, dtype=torch.float64, device='cuda:0')
        adv, ret = rollout.gae(H.to_dev(rew), H.to_dev(val), end=H.to_dev(end), boot=H.to_dev(boot), stats=stats)
        assert np.array_equal(adv.cpu().numpy(), adv_o) and np.array_equal(ret.cpu().numpy(), ret_o), (T, n)
        a64 = adv_o.astype(np.float64)
        s = stats.cpu().numpy()
        assert abs(s[0] - a64.sum()) <= 1e-12 * np.abs(a64).sum() and abs(s[1] - (a64 * a64).sum()) <= 1e-12 * (a64 * a64).sum()
        stats2 = torch.zeros(2, dtype=torch.float64, device='cuda:0')
        rollout.gae(H.to_dev(rew), H.to_dev(val), end=H.to_dev(end), boot=H.to_dev(boot), stats=stats2)
        assert torch.equal(stats, stats2)                                         # deterministic reduction
        # without boot: last_val at the final row, 0 at inner ends
        lv = rng.randn(n).astype(np.float32)
        adv_o2, ret_o2 = orc.gae(rew, val, end=end, last_val=lv)
        adv2, ret2 = rollout.gae(H.to_dev(rew), H.to_dev(val), end=H.to_dev(end), last_val=H.to_dev(lv))
        assert np.array_equal(adv2.cpu().numpy(), adv_o2) and np.array_equal(ret2.cpu().numpy(), ret_o2)
        if T * n > 100:
            want, ms = orc.normalize_adv(adv_o)
            one, mean, std = rollout.normalize_advantages(adv.clone(), stats=stats)
            three, mean3, std3 = rollout.normalize_advantages(adv.clone())
            assert abs(float(mean) - ms[0]) < 1e-6 and abs(float(std) - ms[1]) < 1e-5 * ms[1]
            assert np.abs(one.cpu().numpy() - want).max() < 2e-5 * np.abs(want).max()
            assert np.abs(three.cpu().numpy() - want).max() < 2e-5 * np.abs(want).max()
            t1, _, _ = rollout.normalize_advantages(adv.clone())
            assert torch.equal(three, t1)                                          # the three-pass sums are deterministic too
        if n % 2 == 0 and T * n > 100:
            # two "ranks": each scans its half of the env columns, the statistics add, everyone applies the global ones
            h = n // 2
            parts, st = [], torch.zeros(2, dtype=torch.float64, device='cuda:0')
            for sl in (slice(0, h), slice(h, n)):
                ps = torch.zeros(2, dtype=torch.float64, device='cuda:0')
                a_, _ = rollout.gae(H.to_dev(rew[:, sl]), H.to_dev(val[:, sl]), end=H.to_dev(end[:, sl]), boot=H.to_dev(boot[:, sl]), stats=ps)
                parts.append(a_)
                st += ps
            from ml4ca_amd import _lib
            import ctypes as C
            for a_ in parts:
                _lib.check(_lib.load().dpenv_adv_apply_stats(C.c_void_p(a_.data_ptr()), a_.numel(), C.c_void_p(st.data_ptr()), float(T * n),
                                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            joined = torch.cat(parts, dim=1)
            assert float((joined - one).abs().max()) <= 2e-7 * float(one.abs().max())


def test_iae_device_reduction_matches_the_reference_fixture():
    """the IAE metric as a device reduction, against the fixture produced by the reference's own IAE()
    (results/all_plots/common.py:60-74 via tools/gen_golden_iae.py) on the recorded Cybersea RL box test."""
    import os
    from ml4ca_amd import evaluate as EV
    torch = torch_()
    d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'iae.npz'))
    dev = 'cuda:0'
    integ, cum = EV.iae_series(torch.from_numpy(d['eta']).to(dev), torch.from_numpy(d['ref']).to(dev), torch.from_numpy(d['time']).to(dev),
                               norm=tuple(d['norm']))
    assert np.abs(integ.cpu().numpy() - d['integrals']).max() < 1e-11 and np.abs(cum.cpu().numpy() - d['cumsum']).max() < 1e-10
    # fp32 batched form on a [T, n, 9] observation block (what the rollout kernels write): per env equal to the fp64 series form
    T, n = 251, 33
    g = torch.Generator(device=dev).manual_seed(2)
    obs = torch.randn((T, n, 9), generator=g, device=dev)
    tot, _ = EV.iae(obs, dt=0.2)
    e = torch.stack([obs[..., 0], obs[..., 1], torch.rad2deg(obs[..., 2])], -1).double()
    _, c = EV.iae_series(e, torch.zeros_like(e), torch.arange(T, dtype=torch.float64, device=dev) * 0.2)
    assert float(((tot.double() - c[-1]).abs() / c[-1]).max()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('precision,form', [('f32', 'one_wave'), ('f16', 'two_wave'), ('f16', 'one_wave')])
def test_rollout_evaluations_equal_the_forward_kernel_at_full_size(precision, form):
    """65 536 envs, deterministic actions (act = mu): every stored action and value of a closed-loop launch equals what the forward
    kernel computes from the stored observation row, bit for bit - for every inlined copy of the network evaluation in the rollout
    kernels (the launch's first, the in-loop ones, the pre-reset critic).  This is the check that catches a register-level fault in
    ONE copy of the evaluation (round 2: asm temporaries allocated inside a running MFMA's destination, first evaluation only)."""
    import ml4ca_amd
    from ml4ca_amd.policy import ActorCritic, policy_forward, policy_rollout
    torch = torch_()
    n, T = 65536, 12
    env = ml4ca_amd.BatchedRevoltEnv(n, auto_reset=True, max_ep_len=5, seed=11)
    ac = ActorCritic(9, 7, (80, 80, 80), seed=6, device=env.device)
    ac.upload(env, precision=precision, launch_form=form)
    env.reset()
    out = policy_rollout(env, T, sample=False)
    mu, v = policy_forward(env, out['obs'].reshape(T * n, 9))
    assert torch.equal(out['act'].reshape(T * n, 7), mu)
    assert torch.equal(out['val'].reshape(T * n), v)
    # boot rows of cut episodes (max_ep_len = 5 cuts every episode that survives) are V of the pre-reset observation: finite, and
    # zero exactly where the episode terminated or did not end
    done = out['done']
    ended = (done != 0)
    ended[T - 1] = True
    terminal = (done & 1) != 0
    assert bool((out['boot'][~ended | terminal] == 0).all()) and bool(torch.isfinite(out['boot']).all())
    assert int((ended & ~terminal).sum()) > n                      # the cut path was exercised by every env at least once on average


@pytest.mark.gpu
@pytest.mark.parametrize('mode,ext,hidden,activation', [
    ('final_cont', True, (81, 81), 'tanh'),
    ('final_cont', True, (96, 96), 'leaky'),
    ('limited', False, (64,), 'relu'),
    ('simple', False, (33, 33, 33), 'leaky'),
    ('full', True, (48, 48), 'tanh'),
])
def test_exact_mode_rollout_equals_forward_kernel_for_other_shapes(mode, ext, hidden, activation):
    """DPENV_POLICY_F32 closed loop with deterministic actions for the shapes the evaluation is templated on (5 / 6 k-steps, tanh,
    one to three hidden layers, every env variant's action width): stored actions and values equal the forward kernel on the stored
    observations bit for bit, through auto-resets."""
    from ml4ca_amd.policy import ActorCritic, policy_forward, policy_rollout
    torch = torch_()
    n, T = 700 + 5, 9
    env, _ = H.make_pair(mode, n, ext=ext, auto_reset=True, max_ep_len=4, seed=13)
    ac = ActorCritic(env.num_states, env.num_actions, hidden, seed=9, device=env.device, activation=activation)
    ac.upload(env, precision='f32')
    env.reset()
    out = policy_rollout(env, T, sample=False)
    mu, v = policy_forward(env, out['obs'].reshape(T * n, env.num_states))
    assert torch.equal(out['act'].reshape(T * n, env.num_actions), mu)
    assert torch.equal(out['val'].reshape(T * n), v)
    assert bool(torch.isfinite(out['boot']).all()) and bool((out['boot'] != 0).any())


@pytest.mark.gpu
def test_f32_actor_mode_is_the_exact_actor_with_the_fast_critic():
    """DPENV_POLICY_F32_ACTOR: mu (hence action and logp) equals the DPENV_POLICY_F32 evaluation bit for bit, V equals the
    DPENV_POLICY_F16 evaluation bit for bit - in the forward kernel and in a closed-loop launch with in-kernel noise, through resets."""
    from ml4ca_amd.policy import policy_forward, policy_rollout
    torch = torch_()
    n, T = 2000 + 9, 14
    kw = dict(auto_reset=True, max_ep_len=6, seed=21)
    envs = {p: H.make_pair('final_cont', n, **kw)[0] for p in ('f16', 'f32', 'f32_actor')}
    ac = make_ac(9, 7, (80, 80, 80), seed=5, device=envs['f16'].device)
    for p, e in envs.items():
        ac.upload(e, precision=p, launch_form='one_wave' if p == 'f16' else 'auto')
        e.reset()
    gd = torch.Generator(device=envs['f16'].device).manual_seed(2)
    obs = torch.randn((n, 9), generator=gd, device=envs['f16'].device)
    fw = {p: policy_forward(e, obs) for p, e in envs.items()}
    assert torch.equal(fw['f32_actor'][0], fw['f32'][0]) and torch.equal(fw['f32_actor'][1], fw['f16'][1])
    assert not torch.equal(fw['f32'][1], fw['f16'][1])
    out = policy_rollout(envs['f32_actor'], T, sample=True)
    ref = policy_rollout(envs['f32'], T, sample=True)
    # same seed, same exact actor, same noise stream -> the same trajectory as the all-exact mode; only the value rows differ
    for k in ('obs', 'act', 'rew', 'done', 'logp'):
        assert torch.equal(out[k], ref[k]), k
    v16 = policy_forward(envs['f16'], out['obs'].reshape(T * n, 9))[1]
    assert torch.equal(out['val'].reshape(T * n), v16)
    assert not torch.equal(out['val'], ref['val'])
    assert float((out['val'] - ref['val']).abs().max()) < 5e-3 * (float(ref['val'].abs().max()) + 1.0)
