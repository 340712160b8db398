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
