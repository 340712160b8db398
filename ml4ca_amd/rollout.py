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
    """[T, n_envs] trajectory block in HBM: obs 9 + act 7 + rew + val + logp = 19 floats per env-step
    (ppo.py:40-46) plus done bits, advantages and returns."""

    def __init__(self, T, n_envs, obs_dim, act_dim, device, gamma=0.99, lam=0.97, obs_dtype=None):
        torch = _torch()
        self.T, self.n = int(T), int(n_envs)
        self.gamma, self.lam = gamma, lam
        f32 = torch.float32
        self.obs = torch.zeros((T, n_envs, obs_dim), dtype=obs_dtype or f32, device=device)
        self.act = torch.zeros((T, n_envs, act_dim), dtype=f32, device=device)
        self.rew = torch.zeros((T, n_envs), dtype=f32, device=device)
        self.val = torch.zeros((T, n_envs), dtype=f32, device=device)
        self.logp = torch.zeros((T, n_envs), dtype=f32, device=device)
        self.done = torch.zeros((T, n_envs), dtype=torch.uint8, device=device)
        self.boot = torch.zeros((T, n_envs), dtype=f32, device=device)
        self.adv = torch.zeros((T, n_envs), dtype=f32, device=device)
        self.ret = torch.zeros((T, n_envs), dtype=f32, device=device)
        self.ptr = 0

    def step_outputs(self, t):
        """(next_obs_row, reward_row, done_row) views for BatchedRevoltEnv.step(out=...): the observation that
        step t produces is the policy input of step t+1, so it lands in obs[t+1] (the caller puts the reset
        observation in obs[0] and keeps the last one for the bootstrap value)."""
        return self.rew[t], self.done[t]

    def finish(self, last_val=None):
        """GAE over the whole block (finish_path for every path of every env)."""
        return gae(self.rew, self.val, end=self.done, boot=self.boot, last_val=None if last_val is None else last_val,
                   gamma=self.gamma, lam=self.lam, out=(self.adv, self.ret)) if last_val is None else \
            self._finish_with_last(last_val)

    def _finish_with_last(self, last_val):
        self.boot[self.T - 1] = last_val
        return gae(self.rew, self.val, end=self.done, boot=self.boot, gamma=self.gamma, lam=self.lam,
                   out=(self.adv, self.ret))

    def get(self, group=None):
        """ppo.py:93-105: obs, act, normalised adv, ret, logp."""
        normalize_advantages(self.adv, group=group)
        return self.obs, self.act, self.adv, self.ret, self.logp

    def packed(self):
        """[T, n, 19] float32 = obs 9 | act 7 | rew | val | logp: the block the episode-boundary all-gather moves."""
        torch = _torch()
        return torch.cat([self.obs.float(), self.act, self.rew[..., None], self.val[..., None], self.logp[..., None]], dim=-1)
