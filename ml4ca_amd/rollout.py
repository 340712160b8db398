"""On-device rollout storage, GAE-lambda and advantage normalisation.

Batched mirror of the reference's ``TrajectoryBuffer`` (src/rl/windows_workspace/spinup/algos/tf1/ppo/ppo.py:21-105,
``discount_cumsum`` core.py:48-63, ``mpi_statistics_scalar`` spinup/utils/mpi_tools.py:71-92): the reference keeps
one trajectory in host NumPy arrays; here the env kernel writes observation / reward / done rows of a
[T, n_envs, .] block directly (BatchedRevoltEnv.step(out=...)), GAE is one reverse-scan kernel with a lane per
env column, and the advantage statistics are two device reductions with an optional all-reduce in between
(the two MPI all-reduces of mpi_statistics_scalar).
"""
import ctypes as C

from . import _lib


def _torch():
    import torch
    return torch


def _s(t):
    return C.c_void_p(_torch().cuda.current_stream(t.device).cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _req(t, shape, dtype, what):
    if t is None:
        return
    if tuple(t.shape) != tuple(shape) or t.dtype != dtype or not t.is_contiguous() or not t.is_cuda:
        raise ValueError('%s must be a contiguous cuda %s tensor of shape %s' % (what, dtype, tuple(shape)))


def gae(rew, val, end=None, boot=None, last_val=None, gamma=0.99, lam=0.97, out=None):
    """TrajectoryBuffer.finish_path (ppo.py:65-91) for every env column of a [T, n] rollout.

    A path ends after step t of env i where end[t, i] != 0, and always after T-1.  The value appended at a path
    end (ppo.py:82-83) is boot[t, i] if boot is given, else 0 at inner ends and last_val[i] (default 0) at T-1.
    Returns (adv, ret), both [T, n] float32."""
    torch = _torch()
    lib = _lib.load()
    T, n = rew.shape
    _req(rew, (T, n), torch.float32, 'rew')
    _req(val, (T, n), torch.float32, 'val')
    _req(end, (T, n), torch.uint8, 'end')
    _req(boot, (T, n), torch.float32, 'boot')
    _req(last_val, (n,), torch.float32, 'last_val')
    adv, ret = out if out is not None else (torch.empty_like(rew), torch.empty_like(rew))
    _req(adv, (T, n), torch.float32, 'adv')
    _req(ret, (T, n), torch.float32, 'ret')
    with torch.cuda.device(rew.device):
        _lib.check(lib.dpenv_gae(_p(rew), _p(val), _p(end), _p(boot), _p(last_val), T, n, gamma, lam, _p(adv), _p(ret),
                                 _s(rew)))
    return adv, ret


def normalize_advantages(adv, group=None):
    """TrajectoryBuffer.get's normalisation (ppo.py:99-103): adv <- (adv - mean) / (std + 1e-8) with the
    global mean / population std over all ranks (mpi_tools.py:71-92).  In place; returns (adv, mean, std)."""
    torch = _torch()
    lib = _lib.load()
    import torch.distributed as dist
    assert adv.dtype == torch.float32 and adv.is_contiguous() and adv.is_cuda
    count = adv.numel()
    acc = torch.zeros(4, dtype=torch.float32, device=adv.device)   # sum, count, sumsq, -
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    with torch.cuda.device(adv.device):
        _lib.check(lib.dpenv_adv_sum(_p(adv), count, _p(acc[0:1]), _s(adv)))
        acc[1] = float(count)
        if multi:
            dist.all_reduce(acc[0:2], group=group)          # global_sum, global_n  (mpi_tools.py:83)
        mean = (acc[0:1] / acc[1:2]).contiguous()
        _lib.check(lib.dpenv_adv_sumsq(_p(adv), count, _p(mean), _p(acc[2:3]), _s(adv)))
        if multi:
            dist.all_reduce(acc[2:3], group=group)          # global_sum_sq       (mpi_tools.py:86)
        std = torch.sqrt(acc[2:3] / acc[1:2]).contiguous()
        _lib.check(lib.dpenv_adv_apply(_p(adv), count, _p(mean), _p(std), _s(adv)))
    return adv, mean[0], std[0]


class RolloutBuffer(object):
    """The reference's TrajectoryBuffer (ppo.py:21-105) for N envs at once: a [T, n_envs] block in HBM holding
    obs 9 + act 7 + rew + val + logp = 19 floats per env-step (ppo.py:40-46) plus done bits, bootstrap values,
    advantages and returns.  ``collect`` fills it with ONE launch (dpenv_policy_rollout: the loop ppo.py:289-322),
    ``finish`` is finish_path for every path of every env, ``get`` is TrajectoryBuffer.get."""

    def __init__(self, T, env, gamma=0.99, lam=0.97):
        torch = _torch()
        n, od, ad, dev = env.n_envs, env.num_states, env.num_actions, env.device
        self.T, self.n, self.gamma, self.lam = int(T), n, gamma, lam
        f32 = torch.float32
        self.blocks = dict(obs=torch.zeros((T, n, od), dtype=env.obs_torch_dtype, device=dev), act=torch.zeros((T, n, ad), dtype=f32, device=dev),
                           rew=torch.zeros((T, n), dtype=f32, device=dev), val=torch.zeros((T, n), dtype=f32, device=dev),
                           logp=torch.zeros((T, n), dtype=f32, device=dev), boot=torch.zeros((T, n), dtype=f32, device=dev),
                           done=torch.zeros((T, n), dtype=torch.uint8, device=dev),
                           last_obs=torch.zeros((n, od), dtype=env.obs_torch_dtype, device=dev), last_val=torch.zeros(n, dtype=f32, device=dev))
        self.adv = torch.zeros((T, n), dtype=f32, device=dev)
        self.ret = torch.zeros((T, n), dtype=f32, device=dev)

    def collect(self, env, noise=None, switch_steps=(), refs=None):
        """Run T policy-in-the-loop steps from the env's current state into the block (the env must have a policy
        uploaded, policy.ActorCritic.upload)."""
        from .policy import policy_rollout
        return policy_rollout(env, self.T, noise=noise, switch_steps=switch_steps, refs=refs, out=self.blocks)

    def finish(self):
        """GAE-lambda advantages and rewards-to-go for every path in the block (ppo.py:65-91); paths end where done != 0
        and at the end of the block, bootstrapped with the values the rollout kernel left in ``boot`` (ppo.py:311)."""
        b = self.blocks
        return gae(b['rew'], b['val'], end=b['done'], boot=b['boot'], gamma=self.gamma, lam=self.lam, out=(self.adv, self.ret))

    def get(self, group=None):
        """ppo.py:93-105: obs, act, normalised adv, ret, logp."""
        normalize_advantages(self.adv, group=group)
        b = self.blocks
        return b['obs'], b['act'], self.adv, self.ret, b['logp']

    def packed(self):
        """[T, n, 19] float32 = obs 9 | act 7 | rew | val | logp: the block the episode-boundary all-gather moves."""
        torch = _torch()
        b = self.blocks
        return torch.cat([b['obs'].float(), b['act'], b['rew'][..., None], b['val'][..., None], b['logp'][..., None]], dim=-1)
