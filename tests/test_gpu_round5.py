"""Round-5 GPU tests (-m gpu) of the review items that are not about vessel blocks: the layout of a policy image pinned by a captured
graph, the binding's per-call plumbing (the eager loop of spinup/algos/tf1/ppo/ppo.py:291-293), the launch form the
library reports, the single-env adapter's argument check."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


def torch_():
    import torch
    assert torch.cuda.is_available(), 'gpu tests need an MI355X'
    return torch


def make_ac(*a, **kw):
    from ml4ca_amd.policy import ActorCritic
    return ActorCritic(*a, **kw)


def _capture(env, T, out):
    from ml4ca_amd.policy import policy_rollout
    torch = torch_()
    side = torch.cuda.Stream(device=env.device)
    side.wait_stream(torch.cuda.current_stream(env.device))
    with torch.cuda.stream(side):
        policy_rollout(env, T, sample=False, out=out)
    torch.cuda.current_stream(env.device).wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        policy_rollout(env, T, sample=False, out=out)
    return g


def test_pinned_policy_image_refuses_another_layout_until_released():
    """A rollout recorded into a HIP graph holds the weight image's address AND its layout / launch form by value.  An in-place upload
    with another precision, activation, leak or hidden shape that still fits the buffer used to be accepted and replayed with the old
    layout (wrong weights, no error): now DPENV_EINVAL, until dpenv_release_policy_graphs says the graphs are gone.  Uploads of the SAME
    layout keep reaching the replays - also after an upload recorded into a second graph (which used to re-pin the other image)."""
    from ml4ca_amd import _lib
    from ml4ca_amd.policy import policy_rollout, release_policy_graphs
    torch = torch_()
    n, T = 2048, 8
    env, _ = H.make_pair('final_cont', n, auto_reset=True, seed=9)
    env2, _ = H.make_pair('final_cont', n, auto_reset=True, seed=9)
    a1, a2, a3 = (make_ac(9, 7, (80, 80, 80), seed=s_, device=env.device) for s_ in (1, 2, 3))
    a1.upload(env, precision='f32')                          # the largest image first: everything below fits the buffer
    a1.upload(env, precision='f16')
    env.reset(); env2.reset()
    st, ctr = env.get_state()
    out = policy_rollout(env, T, sample=False)
    env.set_state(st, ctr)
    g = _capture(env, T, out)
    for bad in (dict(precision='f32'), dict(precision='f32_actor'), dict(precision='f16', launch_form='one_wave')):
        with pytest.raises(_lib.DpenvError, match='captured graph'):
            a2.upload(env, **bad)
    with pytest.raises(_lib.DpenvError, match='captured graph'):
        make_ac(9, 7, (80, 80), seed=5, device=env.device).upload(env, precision='f16')          # fewer layers
    with pytest.raises(_lib.DpenvError, match='captured graph'):
        make_ac(9, 7, (80, 80, 80), seed=5, leak=0.1, device=env.device).upload(env, precision='f16')   # another slope
    # the refused uploads changed nothing: the graph still runs a1; a same-layout upload reaches it
    for ac in (a1, a2, a3):
        if ac is not a1:
            ac.upload(env, precision='f16')
        env.set_state(st, ctr)
        g.replay()
        torch.cuda.synchronize()
        ac.upload(env2, precision='f16')
        env2.set_state(st, ctr)
        want = policy_rollout(env2, T, sample=False)
        assert torch.equal(out['act'], want['act']) and torch.equal(out['val'], want['val'])
    # a SECOND graph with an upload recorded inside it (device pointers): it must write the pinned image, not re-pin the other one
    out2 = {k: v.clone() for k, v in out.items()}
    side = torch.cuda.Stream(device=env.device)
    side.wait_stream(torch.cuda.current_stream(env.device))
    with torch.cuda.stream(side):
        a1.upload(env, precision='f16')
        policy_rollout(env, T, sample=False, out=out2)
    torch.cuda.current_stream(env.device).wait_stream(side)
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        a1.upload(env, precision='f16')
        policy_rollout(env, T, sample=False, out=out2)
    a2.upload(env, precision='f16')                          # eager, after the second capture: the FIRST graph must still see it
    env.set_state(st, ctr)
    g.replay()
    torch.cuda.synchronize()
    a2.upload(env2, precision='f16')
    env2.set_state(st, ctr)
    want = policy_rollout(env2, T, sample=False)
    assert torch.equal(out['act'], want['act'])
    env.set_state(st, ctr)
    g2.replay()                                              # re-packs a1 from its tensors, then rolls out
    torch.cuda.synchronize()
    a1.upload(env2, precision='f16')
    env2.set_state(st, ctr)
    want = policy_rollout(env2, T, sample=False)
    assert torch.equal(out2['act'], want['act'])
    # the graphs are gone: any layout again
    del g, g2
    release_policy_graphs(env)
    a2.upload(env, precision='f32')
    env.set_state(st, ctr)
    a2.upload(env2, precision='f32')
    env2.set_state(st, ctr)
    got, want = policy_rollout(env, T, sample=False), policy_rollout(env2, T, sample=False)
    assert torch.equal(got['act'], want['act'])


def test_launch_info_reports_what_the_library_resolved():
    from ml4ca_amd.policy import policy_launch_info, policy_launch_form
    ac = None
    for n, prec, want in ((4096, 'f16', (True, 128, 2)), (4096, 'f32', (True, 128, 3)), (4096, 'f32_actor', (True, 128, 3)),
                          (70000, 'f16', (True, 256, 2)), (70000, 'f32', (True, 256, 2))):
        env, _ = H.make_pair('final_cont', n)
        ac = make_ac(9, 7, (80, 80, 80), seed=1, device=env.device).upload(env, precision=prec)
        info = policy_launch_info(env)
        assert (info['two_wave'], info['envs_per_workgroup'], info['waves_per_64_envs']) == want and info['precision'] == prec, (n, prec, info)
        assert policy_launch_form(env) == ('two_wave', want[1])
    ac.upload(env, precision='f16', launch_form='one_wave')
    assert policy_launch_info(env)['waves_per_64_envs'] == 1


def test_step_plumbing_cache_still_checks_what_it_has_not_seen():
    """BatchedRevoltEnv.step recognises tensors it has validated by identity (a weak reference); a new tensor of the wrong shape / dtype /
    device is still refused, whatever was cached before, and rows do not depend on the cache."""
    torch = torch_()
    n = 1024
    env, _ = H.make_pair('final_cont', n)
    env2, _ = H.make_pair('final_cont', n)
    env.reset(); env2.reset()
    g = torch.Generator(device=env.device).manual_seed(0)
    acts = torch.randn((8, n, 7), generator=g, device=env.device) * 0.5
    held = [acts[k] for k in range(8)]
    for rep in range(3):
        for k in range(8):
            a = env.step(held[k])                             # cached after the first pass
            b = env2.step(acts[k].clone())                    # a fresh tensor every call
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    with pytest.raises(ValueError):
        env.step(torch.zeros((n, 6), device=env.device))
    with pytest.raises(ValueError):
        env.step(torch.zeros((n, 7), device=env.device, dtype=torch.float64))
    with pytest.raises(ValueError):
        env.step(held[0], out=(torch.zeros((n, 9), device=env.device), torch.zeros(n + 1, device=env.device), torch.zeros(n, dtype=torch.uint8, device=env.device)))
    with pytest.raises(ValueError):
        env.step(torch.zeros((n, 7)))                         # a CPU tensor
    # ids of dead tensors may be recycled: a recycled id must not vouch for another tensor
    for _ in range(200):
        t = torch.zeros((n, 7), device=env.device)
        env.step(t)
        del t
        bad = torch.zeros((n, 5), device=env.device)
        with pytest.raises(ValueError):
            env.step(bad)
        del bad


def test_single_env_adapter_rejects_a_wrong_sized_action():
    import ml4ca_amd
    env = ml4ca_amd.RevoltFinal(None, extended_state=True, cont_ang=True)
    env.reset()
    o, r, d, _ = env.step(np.zeros(7))
    assert o.shape == (9,) and isinstance(r, float) and isinstance(d, bool)
    for bad in (np.zeros(5), np.zeros(8), np.zeros((2, 7))):
        with pytest.raises(ValueError, match='7 elements'):
            env.step(bad)
    o2, _, _, _ = env.step(np.zeros((1, 7)))                  # the same 7 numbers in another shape are fine, as before
    assert o2.shape == (9,)
