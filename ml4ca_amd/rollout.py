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


_ws_cache = {}


def _workspace(dev, n):
    """Per-device scratch of dpenv_gae_stats (per-workgroup partial sums), grown on demand and reused."""
    torch = _torch()
    need = int(_lib.load().dpenv_gae_workspace_bytes(int(n)))
    key = (dev.type, dev.index)
    w = _ws_cache.get(key)
    if w is None or w.numel() * 8 < need:
        w = torch.empty((need + 7) // 8, dtype=torch.float64, device=dev)
        _ws_cache[key] = w
    return w


def gae(rew, val, end=None, boot=None, last_val=None, gamma=0.99, lam=0.97, out=None, stats=None, workspace=None):
    """TrajectoryBuffer.finish_path (ppo.py:65-91) for every env column of a [T, n] rollout.

    A path ends after step t of env i where end[t, i] != 0, and always after T-1.  The value appended at a path
    end (ppo.py:82-83) is boot[t, i] if boot is given, else 0 at inner ends and last_val[i] (default 0) at T-1.
    Returns (adv, ret), both [T, n] float32.  stats: optional float64[2] cuda tensor that receives (sum adv, sum adv^2) from the
    same pass (deterministic; feed it to normalize_advantages).  workspace: float64 scratch of dpenv_gae_workspace_bytes(n) bytes for
    the per-workgroup partial sums; default one per device, which two streams of one device must not share - a RolloutBuffer
    brings its own."""
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
    _req(stats, (2,), torch.float64, 'stats')
    with torch.cuda.device(rew.device):
        ws = (workspace if workspace is not None else _workspace(rew.device, n)) if stats is not None else None
        _lib.check(lib.dpenv_gae_stats(_p(rew), _p(val), _p(end), _p(boot), _p(last_val), T, n, gamma, lam, _p(adv), _p(ret),
                                       _p(ws), _p(stats), _s(rew)))
    return adv, ret


def combine_stats(stats, count, group=None):
    """The all-reduce hook of the one-pass normalisation: (sum adv, sum adv^2) float64[2] of this rank and its sample count
    -> the same over all ranks (mpi_statistics_scalar's two all-reduces, mpi_tools.py:83-87, as ONE 24-byte SUM all-reduce).
    Pure torch + torch.distributed: runs on any backend (the gloo CPU test calls it).  Returns (stats_global, total_count)."""
    torch = _torch()
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if not multi:
        return stats, float(count)
    buf = torch.empty(3, dtype=torch.float64, device=stats.device)
    buf[0:2] = stats
    buf[2] = float(count)
    dist.all_reduce(buf, group=group)
    return buf[0:2], buf[2]          # total_count stays a device scalar on GPUs: no host sync


def normalize_advantages(adv, group=None, stats=None):
    """TrajectoryBuffer.get's normalisation (ppo.py:99-103): adv <- (adv - mean) / (std + 1e-8) with the
    global mean / population std over all ranks (mpi_tools.py:71-92).  In place; returns (adv, mean, std).

    stats = the float64[2] tensor gae(..., stats=...) filled: one all-reduce, one apply pass, nothing read back to the host.
    Without it the reference's own order runs: sum -> all-reduce -> mean -> sum of squared deviations -> all-reduce -> std ->
    apply (three passes over adv; deterministic reductions)."""
    torch = _torch()
    lib = _lib.load()
    import torch.distributed as dist
    assert adv.dtype == torch.float32 and adv.is_contiguous() and adv.is_cuda
    count = adv.numel()
    if stats is not None:
        _req(stats, (2,), torch.float64, 'stats')
        g, total = combine_stats(stats, count, group=group)
        with torch.cuda.device(adv.device):
            if isinstance(total, float):
                _lib.check(lib.dpenv_adv_apply_stats(_p(adv), count, _p(g), total, _s(adv)))
                mean = g[0] / total
                std = torch.sqrt(torch.clamp(g[1] / total - mean * mean, min=0.0))
            else:
                # the count is a device scalar after the all-reduce: fold it into the statistics instead of reading it back
                gn = (g / total).contiguous()                                  # (mean, E[adv^2])
                _lib.check(lib.dpenv_adv_apply_stats(_p(adv), count, _p(gn), 1.0, _s(adv)))
                mean = gn[0]
                std = torch.sqrt(torch.clamp(gn[1] - mean * mean, min=0.0))
        return adv, mean.float(), std.float()
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


TRAJ_FIELDS = ('obs', 'act', 'rew', 'val', 'logp')      # the 9 + 7 + 1 + 1 + 1 = 19 floats per env-step of ppo.py:40-46


class RolloutBuffer(object):
    """The reference's TrajectoryBuffer (ppo.py:21-105) for N envs at once: a [T, n_envs] block in HBM holding
    obs 9 + act 7 + rew + val + logp = 19 floats per env-step (ppo.py:40-46) plus done bits, bootstrap values,
    advantages and returns.  ``collect`` fills it with ONE launch (dpenv_policy_rollout: the loop ppo.py:289-322),
    ``finish`` is finish_path for every path of every env (and leaves the advantage statistics behind), ``get`` is
    TrajectoryBuffer.get."""

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
        self.stats = torch.zeros(2, dtype=torch.float64, device=dev)
        self._stats_fresh = False       # True between finish() and the get() that consumes the statistics
        need = int(_lib.load().dpenv_gae_workspace_bytes(int(n)))
        self._ws = torch.empty((need + 7) // 8, dtype=torch.float64, device=dev)     # this buffer's own scratch (any stream)

    def collect(self, env, noise=None, switch_steps=(), refs=None, sample=None, rows=None, reset_at_end=False):
        """Run T policy-in-the-loop steps from the env's current state into the block (the env must have a policy
        uploaded, policy.ActorCritic.upload).  sample: see policy.policy_rollout.
        rows=(t0, t1): fill only rows [t0, t1) with ONE launch of t1 - t0 steps (the pieces of a pipelined episode exchange,
        dist.EpisodeExchange); the pieces of one episode give the same rows as one launch of T steps, bit for bit, except `boot` at
        the last row of an inner piece (V of the next observation instead of 0) - which the scan only reads where a path ends
        (ppo.py:311), i.e. at done rows and at T-1, so adv / ret are the same bits too.
        reset_at_end: the reference's epoch boundary (ppo.py:305-322) - after the LAST step of the block every env is cut
        (boot = V(last obs), ppo.py:311) and re-drawn, whether or not its episode had ended."""
        from .policy import policy_rollout
        self._stats_fresh = False
        if rows is None:
            return policy_rollout(env, self.T, noise=noise, switch_steps=switch_steps, refs=refs, out=self.blocks, sample=sample,
                                  reset_at_end=reset_at_end)
        t0, t1 = rows
        assert 0 <= t0 < t1 <= self.T
        b = self.blocks
        part = {k: (b[k] if k in ('last_obs', 'last_val') else b[k][t0:t1]) for k in b}
        sel = [j for j, st in enumerate(switch_steps) if t0 <= st < t1]
        policy_rollout(env, t1 - t0, noise=None if noise is None else noise[t0:t1], switch_steps=tuple(switch_steps[j] - t0 for j in sel),
                       refs=None if not sel else refs[sel[0]:sel[-1] + 1].contiguous(), out=part, sample=sample,
                       reset_at_end=reset_at_end and t1 == self.T)
        return b

    def finish(self):
        """GAE-lambda advantages and rewards-to-go for every path in the block (ppo.py:65-91); paths end where done != 0
        and at the end of the block, bootstrapped with the values the rollout kernel left in ``boot`` (ppo.py:311).  The same
        pass leaves (sum adv, sum adv^2) in ``self.stats`` for ``get``."""
        b = self.blocks
        res = gae(b['rew'], b['val'], end=b['done'], boot=b['boot'], gamma=self.gamma, lam=self.lam, out=(self.adv, self.ret),
                  stats=self.stats, workspace=self._ws)
        self._stats_fresh = True
        return res

    def get(self, group=None):
        """ppo.py:93-105: obs, act, normalised adv, ret, logp.  The statistics finish() left behind are used ONCE: a second get()
        (or a get() after adv was edited through another finish-less path) recomputes them with the three-pass form, like the
        reference's get() which recomputes mean / std every time it is called (ppo.py:99-101)."""
        if self._stats_fresh:
            normalize_advantages(self.adv, group=group, stats=self.stats)
            self._stats_fresh = False
        else:
            normalize_advantages(self.adv, group=group)
        b = self.blocks
        return b['obs'], b['act'], self.adv, self.ret, b['logp']

    def exchange_blocks(self):
        """What an update on ANOTHER rank needs of this rank's episode (ppo.py:93-105): obs | act | logp (complete as the steps
        are) and adv | ret (after finish() / get(): GAE and the normalisation are local, only the 24-byte statistics cross).
        58 B per env-step with bf16 observation rows (config 5's obs_dtype), 76 B with f32 rows; views, no copy.
        rew, val, boot and done stay home."""
        b = self.blocks
        return {'obs': b['obs'], 'act': b['act'], 'logp': b['logp'], 'adv': self.adv, 'ret': self.ret}

    def trajectory(self):
        """The five blocks of ppo.py:40-46 as they lie in HBM: {'obs': [T, n, 9], 'act': [T, n, 7], 'rew' / 'val' / 'logp': [T, n]}
        (views, no copy).  dist.gather_rollout moves them between ranks in place - there is no packed [T, n, 19] staging copy."""
        return {k: self.blocks[k] for k in TRAJ_FIELDS}
