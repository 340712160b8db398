#!/usr/bin/env python3
"""What a one-GPU box CAN rehearse of the N > 1 RCCL path: a ONE-rank `nccl` (= RCCL) process group, joined exactly as bench.py joins
its group (`pool_environment()`, `init_process_group('nccl', device_id=..., timeout=...)`, first barrier), and every collective call
form that bench.py / ml4ca_amd/dist.py / ml4ca_amd/rollout.py issue, with their dtypes and shapes at config 4's shard size: barrier,
all_reduce (float64 MAX / MIN / SUM, int64 MAX), all_gather (list form), all_gather_object, all_gather_into_tensor (f32 and bf16 blocks,
contiguous row slices, async_op + wait), broadcast.  With one rank RCCL moves nothing between devices - what this shows is that the backend
loads on this pool with the environment bench.py sets, that ProcessGroupNCCL accepts each call form, and that outputs equal inputs.
What it cannot show is anything about xGMI or IPC between processes (that needs the driver's 8-GPU node).

    python3 tools/rccl_one_rank_rehearsal.py > gpurun_out/r04_rccl_one_rank_rehearsal.txt
"""
import os
import socket
import sys
import time
from datetime import timedelta

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                    # pool_environment: the same function the bench ranks call


def main():
    env = bench.pool_environment()
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    os.environ['MASTER_PORT'] = str(port)
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    import torch
    import torch.distributed as dist
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    t0 = time.time()
    dist.init_process_group('nccl', device_id=dev, timeout=timedelta(seconds=120))
    dist.barrier()
    print('joined a 1-rank nccl group in %.1f s; environment %s' % (time.time() - t0, env))
    print('torch %s, hip %s, nccl (RCCL) %s, backend %s' % (torch.__version__, torch.version.hip, '.'.join(str(x) for x in torch.cuda.nccl.version()), dist.get_backend()))
    ok = []

    def check(name, cond):
        ok.append(bool(cond))
        print('%-78s %s' % (name, 'ok' if cond else 'FAILED'))

    # bench.py: _timed, RG agreement, per-rank times, group record
    t = torch.tensor([1.25], device=dev, dtype=torch.float64)
    lo = t.clone()
    dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    check('all_reduce float64 MAX / MIN (bench._timed)', float(t[0]) == 1.25 and float(lo[0]) == 1.25)
    rt = torch.tensor([7], device=dev, dtype=torch.int64)
    dist.all_reduce(rt, op=dist.ReduceOp.MAX)
    check('all_reduce int64 MAX (repeat count agreement)', int(rt[0]) == 7)
    lst = [torch.zeros(2, device=dev, dtype=torch.float64)]
    dist.all_gather(lst, torch.tensor([0.5, 0.25], device=dev, dtype=torch.float64))
    check('all_gather list form, float64[2] (per-rank wall / event times)', lst[0].tolist() == [0.5, 0.25])
    names = [None]
    dist.all_gather_object(names, {'rank': 0, 'pid': os.getpid()})
    check('all_gather_object (group record)', names[0]['pid'] == os.getpid())
    # rollout.combine_stats: sum, sum of squares, count in double, ONE all-reduce
    st = torch.tensor([3.0, 5.0, 11.0], device=dev, dtype=torch.float64)
    dist.all_reduce(st)
    check('all_reduce float64[3] SUM (advantage statistics)', st.tolist() == [3.0, 5.0, 11.0])
    # dist.gather_trajectories / gather_rollout: the concatenated-along-dim-0 form, at config 4's block shapes (T shortened)
    T, n = 50, 32768
    g = torch.Generator(device=dev).manual_seed(1)
    for name, shape, dt in (('obs f32 [T, n, 9]', (T, n, 9), torch.float32), ('act f32 [T, n, 7]', (T, n, 7), torch.float32),
                            ('rew f32 [T, n]', (T, n), torch.float32), ('obs bf16 [T, n, 9]', (T, n, 9), torch.bfloat16)):
        blk = torch.randn(shape, generator=g, device=dev).to(dt)
        out = torch.empty((1,) + shape, dtype=dt, device=dev)
        dist.all_gather_into_tensor(out.view((shape[0],) + shape[1:]), blk)
        check('all_gather_into_tensor %s' % name, torch.equal(out[0], blk))
        out2 = torch.zeros_like(out)
        w = dist.all_gather_into_tensor(out2.view((shape[0],) + shape[1:]), blk, async_op=True)
        w.wait()
        torch.cuda.synchronize()
        check('  ... async_op=True + wait()', torch.equal(out2[0], blk))
    # dist.EpisodeExchange posts contiguous ROW SLICES [t0, t1) of the blocks, no staging copy
    blk = torch.randn((T, n, 9), generator=g, device=dev).to(torch.bfloat16)
    rows = blk[10:20]
    out = torch.empty((1, 10, n, 9), dtype=torch.bfloat16, device=dev)
    w = dist.all_gather_into_tensor(out.view(10, n, 9), rows, async_op=True)
    w.wait(); torch.cuda.synchronize()
    check('all_gather_into_tensor of rows [10, 20) of a bf16 block (EpisodeExchange.post_steps)', rows.is_contiguous() and torch.equal(out[0], rows))
    # dist.sync_params / average_gradients
    flat = torch.randn(28175, generator=g, device=dev)
    ref = flat.clone()
    dist.broadcast(flat, src=0)
    dist.all_reduce(flat)
    check('broadcast + all_reduce of the 28 175-parameter flat buffer', torch.equal(flat, ref))
    # the library's kernels beside the group: one step, one closed-loop launch (the process holds both contexts)
    import ml4ca_amd
    from ml4ca_amd.policy import ActorCritic, policy_rollout
    e = ml4ca_amd.BatchedRevoltEnv(n, device=dev, auto_reset=True, seed=4)
    ActorCritic(9, 7, (80, 80, 80), seed=0, device=dev).upload(e, precision='f32_actor')
    e.reset()
    o = policy_rollout(e, 8, sample=True)
    outg = torch.empty((1,) + tuple(o['obs'].shape), device=dev)
    dist.all_gather_into_tensor(outg.view(o['obs'].shape), o['obs'])
    check('all_gather_into_tensor of rows a closed-loop launch just wrote', torch.equal(outg[0], o['obs']) and bool(torch.isfinite(o['logp']).all()))
    dist.barrier()
    dist.destroy_process_group()
    print('ALL OK' if all(ok) else 'FAILURES: %d' % (len(ok) - sum(ok)))
    sys.exit(0 if all(ok) else 1)


if __name__ == '__main__':
    main()
