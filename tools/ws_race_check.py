import sys, torch
sys.path.insert(0, '.')
from tests import helpers as H
from tests.test_gpu_policy import make_ac
from ml4ca_amd.policy import policy_rollout
n, T = 2000 + 11, 45
kw = dict(auto_reset=True, max_ep_len=40, seed=8)
ref = None
nbad = 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for rep in range(reps):
    env, _ = H.make_pair('final_cont', n, **kw)
    ac = make_ac(9, 7, (80, 80, 80), seed=2, device=env.device).upload(env)
    g = torch.Generator(device=env.device).manual_seed(1)
    env.reset()
    refs = torch.randn((2, 3, n), generator=g, device=env.device)
    out = policy_rollout(env, T, noise=None, switch_steps=(3, 30), refs=refs)
    o = {k: v.clone() for k, v in out.items()}
    if ref is None:
        ref = o; continue
    if not torch.equal(o['rew'], ref['rew']):
        nbad += 1
        d = (o['rew'] != ref['rew']).nonzero()
        t0, e0 = int(d[0, 0]), int(d[0, 1])
        envs = sorted(set(d[d[:, 0] == t0][:, 1].tolist()))
        print('rep', rep, 'rew first differs at t', t0, 'envs', envs, 'lanes', sorted(set(e % 64 for e in envs)))
        print('  done[t0-1] of env', int(ref['done'][t0 - 1, e0]), 'obs[t0] equal', bool(torch.equal(o['obs'][t0], ref['obs'][t0])), 'act[t0] equal', bool(torch.equal(o['act'][t0], ref['act'][t0])))
        if t0 + 1 < T:
            print('  obs[t0+1] bad', o['obs'][t0 + 1, e0].tolist()); print('  obs[t0+1] ref', ref['obs'][t0 + 1, e0].tolist())
        print('  rew bad', float(o['rew'][t0, e0]), 'ref', float(ref['rew'][t0, e0]))
print('bad runs', nbad, 'of', reps - 1)
