"""Multi-GPU plumbing: one process per GPU, envs sharded with no per-step communication.

The reference's only parallelism is data-parallel env ownership per MPI rank (specific/trainer.py:61-75,
ppo.py:226) with gradient all-reduce / parameter broadcast per optimiser step (spinup/utils/mpi_tf.py).
Here envs never interact, so a rank owns a contiguous block of global env ids and steps it locally; the
single exchange step is the all-gather of trajectory blocks at episode boundaries (BASELINE.json config 4),
plus the 24-byte all-reduce of the advantage statistics (rollout.combine_stats / normalize_advantages).
Backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""


def shard(total_envs, rank, world):
    """Contiguous block of global env ids owned by `rank`: returns (n_local, env_id_base).
    Remainder envs go to the first ranks.  Philox reset streams are keyed by the GLOBAL env id
    (dpenv_config.env_id_base), so results do not depend on the rank count."""
    if not (0 <= rank < world) or total_envs < world:
        raise ValueError('bad shard request: total_envs=%d rank=%d world=%d' % (total_envs, rank, world))
    base, rem = divmod(total_envs, world)
    n_local = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return n_local, start


def gather_trajectories(block, group=None, out=None, async_op=False):
    """All-gather of equally shaped per-rank trajectory blocks [T, n_local, F] -> [world, T, n_local, F]
    (rank-major = global env order for contiguous shards).  One collective per episode, nothing per step.

    async_op=True returns (out, work): the collective runs on the backend's own stream, so the next rollout can be
    launched while the previous episode's block is still crossing xGMI (double-buffer ``block``; ``work.wait()``
    before reading ``out`` or overwriting ``block``)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        res = block.unsqueeze(0) if out is None else out.copy_(block.unsqueeze(0))
        return (res, None) if async_op else res
    world = dist.get_world_size(group)
    block = block.contiguous()
    if out is None:
        out = torch.empty((world,) + tuple(block.shape), dtype=block.dtype, device=block.device)
    # concatenated-along-dim-0 view: the form every backend (RCCL and gloo) accepts
    work = dist.all_gather_into_tensor(out.view((world * block.shape[0],) + tuple(block.shape[1:])), block, group=group,
                                       async_op=async_op)
    return (out, work) if async_op else out


def gather_rollout(blocks, group=None, out=None, async_op=False):
    """Episode-boundary exchange of a rollout (BASELINE.json config 4) WITHOUT a packed staging copy: every block of
    ``RolloutBuffer.trajectory()`` ([T, n_local, ...]) is all-gathered from where the rollout kernel wrote it into
    out[name] of shape [world, T, n_local, ...] (allocated when ``out`` is None; pass it back in to reuse it).  The five
    collectives are issued back to back on the backend's stream; 19 floats per env-step cross xGMI in total, as with one
    [T, n, 19] block, but nothing is concatenated first (that copy was 2 GB of extra HBM traffic at 65 536 x 400).
    Returns out, or (out, [work, ...]) with async_op=True."""
    out = {} if out is None else out
    works = []
    for name, blk in blocks.items():
        res = gather_trajectories(blk, group=group, out=out.get(name), async_op=async_op)
        if async_op:
            out[name], w = res
            works.append(w)
        else:
            out[name] = res
    return (out, works) if async_op else out


def to_global_env_order(gathered):
    """[world, T, n_local, ...] -> [T, world * n_local, ...]: env axis in global id order."""
    w, T, n = gathered.shape[:3]
    rest = tuple(gathered.shape[3:])
    return gathered.movedim(0, 1).reshape((T, w * n) + rest)


def average_gradients(params, group=None):
    """MpiAdamOptimizer.compute_gradients' all-reduce (spinup/utils/mpi_tf.py:29-62): one flat SUM all-reduce over the
    concatenated gradients, divided by the rank count.  The whole actor (14 334 floats) or critic (13 841) is a single
    57 KB bucket, so this is one latency-bound RCCL call per optimiser step."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, group=group)
    flat /= dist.get_world_size(group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def assert_params_in_step(params, group=None, what='parameters'):
    """Replicated updates (examples/train_ppo.py --exchange rollout: every rank computes the SAME update from the gathered batch, no
    gradient or parameter collective) stay replicated only while every rank's arithmetic is bit-identical.  One 16-byte all-reduce
    per epoch: the min and the max over ranks of a checksum of the parameter bits must agree; raises on the first epoch they do not
    (a GEMM algorithm choice or a kl a few ulps apart flipping the early stop on one rank would otherwise go unnoticed for the rest of
    the run)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
    bits = flat.view(torch.int32).to(torch.int64)
    w = torch.arange(1, bits.numel() + 1, device=bits.device, dtype=torch.int64)
    cs = ((bits * w).sum() % 2147483629).to(torch.float64)          # exact in float64; position-weighted so that swaps show
    t = torch.stack([cs, -cs])
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)            # max(cs), max(-cs) = -min(cs)
    if float(t[0]) != -float(t[1]):
        raise RuntimeError('%s differ between ranks (checksums %.0f .. %.0f): the replicated update has diverged; use --exchange gradients '
                           'or re-broadcast with dist.sync_params' % (what, -float(t[1]), float(t[0])))


def sync_params(params, root=0, group=None):
    """sync_all_params / sync_params (mpi_tf.py:16-26): broadcast the root's parameters, as one flat buffer."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    params = list(params)
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    dist.broadcast(flat, src=root, group=group)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p))
            off += n


def mean_across_ranks(x, group=None):
    """mpi_avg (mpi_tools.py:67-69) for a scalar tensor or float."""
    import torch
    import torch.distributed as dist
    t = x if isinstance(x, torch.Tensor) else torch.tensor(float(x))
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        t = t.clone()
        dist.all_reduce(t, group=group)
        t /= dist.get_world_size(group)
    return t


# ---- episode exchange, compact and pipelined ------------------------------------------------------------------------
# The reference never moves trajectories between ranks: every rank fills its own TrajectoryBuffer (ppo.py:226,289-322), only the
# 24 bytes of mpi_statistics_scalar (mpi_tools.py:83-87) and the gradients (mpi_tf.py:29-62) cross.  BASELINE.json config 4 asks
# for an all-gather of the trajectory buffers at episode boundaries, so that every rank can update on the global batch; what
# that update reads is obs | act | adv | ret | logp (ppo.py:93-105) - rew, val and the bootstrap values only feed the LOCAL GAE scan
# (each rank scans its own env columns; finish_path, ppo.py:65-91, never looks across envs).  So the payload is
#   obs 9 x bf16 (18 B, the rows config 5 already writes) or f32 (36 B) | act 28 | logp 4   - known as soon as a step is done
#   adv 4 | ret 4                                                                          - known after the scan
# = 58 B per env-step with bf16 rows (76 B with f32 rows) instead of 76 B + a remote GAE, and the first group can cross xGMI WHILE the episode is
# still being rolled out: EpisodeExchange gathers it chunk by chunk (rows [t0, t1) of the blocks are contiguous) on the
# backend's own stream under the next chunk's launch; only the last chunk and the 8 B of adv | ret are exposed.
STEP_FIELDS = ('obs', 'act', 'logp')      # complete when the step is
SCAN_FIELDS = ('adv', 'ret')              # complete after GAE + normalisation


class EpisodeExchange(object):
    """All-gather of one episode's update inputs, issued piecewise.

        ex = EpisodeExchange(buf.exchange_blocks(), n_chunks=4)   # {'obs','act','logp','adv','ret'}: [T, n_local, ...] each
        for c in range(4): buf.collect(env, rows=ex.rows(c)); ex.post_steps(c)       # async: crosses under the next launch
        buf.finish(); buf.get(); ex.post_scan()
        glob = ex.wait()                                          # {'obs': [C, world, T / C, n_local, 9], ...}

    Rows [t0, t1) of a [T, n, ...] block are one contiguous range, so a chunk is gathered straight from where the kernel wrote it
    with one all_gather_into_tensor into out[name][c] = [world, T / C, n_local, ...] (contiguous: no staging copy on any backend).
    The output is therefore chunk-major; an update treats the samples as a bag, so the order only has to be the same for every
    field - ``flat(name)`` gives [C * world * T / C * n_local, ...], ``episode_order(name)`` the [world, T, n_local, ...] view order
    of gather_rollout (a copy; tests use it: same values bit for bit)."""

    def __init__(self, blocks, n_chunks=1, group=None, out=None):
        import torch
        import torch.distributed as dist
        self.blocks, self.group, self.C = blocks, group, int(n_chunks)
        self.multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.world = dist.get_world_size(group) if self.multi else 1
        self.T = next(iter(blocks.values())).shape[0]
        if self.T % self.C:
            raise ValueError('n_chunks must divide T (%d %% %d)' % (self.T, self.C))
        self.Tc = self.T // self.C
        self.out = out if out is not None else {
            k: torch.empty((self.C, self.world, self.Tc) + tuple(b.shape[1:]), dtype=b.dtype, device=b.device) for k, b in blocks.items()}
        self.works = []

    def rows(self, c):
        return c * self.Tc, (c + 1) * self.Tc

    def _post(self, names, c):
        import torch.distributed as dist
        t0, t1 = self.rows(c)
        for k in names:
            src = self.blocks[k][t0:t1]
            dst = self.out[k][c]
            if not self.multi:
                dst[0].copy_(src)
                continue
            flat = dst.view((self.world * self.Tc,) + tuple(src.shape[1:]))
            self.works.append(dist.all_gather_into_tensor(flat, src, group=self.group, async_op=True))

    def post_steps(self, c):
        """chunk c of obs | act | logp: call right after the launch that wrote those rows was issued (stream-ordered on GPUs)"""
        self._post([k for k in STEP_FIELDS if k in self.blocks], c)

    def post_scan(self):
        """adv | ret of the whole episode, after RolloutBuffer.finish() / get()"""
        for c in range(self.C):
            self._post([k for k in SCAN_FIELDS if k in self.blocks], c)

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []
        return self.out

    def flat(self, name):
        o = self.out[name]
        return o.reshape((-1,) + tuple(o.shape[4:]))

    def episode_order(self, name):
        """[C, world, Tc, n, ...] -> [world, T, n, ...] (copy)"""
        o = self.out[name]
        return o.movedim(0, 1).reshape((self.world, self.T) + tuple(o.shape[3:]))

    @staticmethod
    def bytes_per_env_step(blocks):
        return sum(b[0, 0].numel() * b.element_size() for b in blocks.values())
