#!/usr/bin/env python3
"""bench.py - env-steps/s of the batched ReVolt DP env.step hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per GPU, RCCL).
Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 65 536 parallel envs per
GPU, RevoltFinal / extended state / continuous-angle heads, fp32, the thesis' 4-corner box setpoint
sequence (results/all_plots/box_test/plot_pos.py:55-59: +5 m N, -5 m E, -45 deg, back S, back E at
t = 10/60/110/140/190 s = env steps 50/300/550/700/950 of 1250, dt = 0.2 s), termination off (SURVEY 8d,
config 3), synthetic Gaussian actions (std e^-0.5, core.py:83) already resident in HBM.  One "step" = one
env.step of every env = one launch of step_kernel.  Envs shard across ranks with no data-path collective
(weak scaling: 65 536 envs per GPU); the episode-boundary trajectory all-gather of config 4 is timed as a
separate, clearly labelled leg and is NOT part of `value`.

The step loop is captured into a HIP graph in chunks of 50 steps (the gcd of the box sequence's segment
lengths) so that host launch overhead does not sit between kernels.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ENVS = 65536
CHUNK = 50
BOX_SWITCH_STEPS = (50, 300, 550, 700, 950)        # plot_pos.py:59 at dt = 0.2 s
BOX_REFS = ((5.0, 0.0, 0.0), (5.0, -5.0, 0.0), (5.0, -5.0, -45.0), (0.0, -5.0, -45.0), (0.0, 0.0, 0.0))   # plot_pos.py:55-57
ALGO_BYTES_PER_ENV_STEP = 177                      # 88 B read + 89 B written
HBM_PEAK_GBPS = 8000.0                             # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1250)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--envs', type=int, default=N_ENVS, help='envs per GPU')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of HIP-graph replay')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fused', action='store_true', help='skip the fused-rollout leg')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='budget of the CPU baseline leg')
    ap.add_argument('--gather', type=int, default=-1, help='time the config-4 trajectory all-gather (default: on if gpus > 1)')
    ap.add_argument('--hold-plant', action='store_true', help='diagnostic: skip the plant sub-steps (INVALID as a result)')
    ap.add_argument('--backend', default='nccl', help="'nccl' (= RCCL; default) or 'gloo' (rehearsal only)")
    ap.add_argument('--same-device', action='store_true', help='rehearsal: every rank uses cuda:0 (needs --backend gloo)')
    ap.add_argument('--traffic-json', default=os.path.join(ROOT, 'profiles', 'traffic_latest.json'))
    return ap.parse_args()


def cpu_baseline(n_envs, budget_s):
    """The oracle (CPU port of the same step, fp32, OpenMP over envs) on the host cores of this box, on a
    bounded sample of the same workload: n_envs envs x S steps, S sized to the time budget."""
    import numpy as np
    from oracle import oracle as O
    # a GPU box gives each GPU a 16-core CPU share; more threads than that only oversubscribes
    avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    threads = O.set_threads(min(avail, 16))
    orc = O.Oracle(O.make_config(terminate=0, max_ep_len=0), np.float32)
    rng = np.random.RandomState(0)
    st, ctr = orc.new_state(n_envs)
    orc.reset(st, ctr, init=np.zeros((6, n_envs), np.float32))
    act = (rng.standard_normal((n_envs, 7)) * 0.6065).astype(np.float32)
    obs = np.zeros((n_envs, 9), np.float32)
    rew = np.zeros(n_envs, np.float32)
    done = np.zeros(n_envs, np.uint8)
    for _ in range(2):
        orc.step_into(st, ctr, act, obs, rew, done)      # warm-up, page-in, thread pool start
    steps = 0
    t0 = time.perf_counter()
    while steps < 3 or (time.perf_counter() - t0 < budget_s and steps < 100000):
        orc.step_into(st, ctr, act, obs, rew, done)
        steps += 1
    dt = time.perf_counter() - t0
    return {'value': n_envs * steps / dt, 'unit': 'env-steps/s', 'cores': threads, 'kind': 'port',
            'sample': '%d envs x %d steps of the same final/ext/cont_ang step (oracle/dpenv_oracle.c, fp32, '
                      'OpenMP over envs, %.1f s)' % (n_envs, steps, dt)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('--gpus %d needs torch.distributed.run --nproc-per-node %d' % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback for the product path)'
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)

    if not os.path.exists(os.path.join(ROOT, 'ml4ca_amd', 'lib', 'libdpenv.so')):
        # a fresh checkout (built artefacts are git-ignored): build once, rank 0 first
        if rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        if world > 1:
            dist.barrier()
    import ml4ca_amd
    n = args.envs
    env = ml4ca_amd.BatchedRevoltEnv(n, variant='final', extended_state=True, cont_ang=True, device=dev,
                                     terminate=False, time_limit=False, seed=1, env_id_base=rank * n,
                                     hold_plant=args.hold_plant)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    # synthetic inputs, resident in HBM before the timed region
    actions = torch.randn((CHUNK, n, 7), generator=g, device=dev) * 0.6065
    # testing-style start (simtools.py:81-88 radius/heading) with the setpoint at the start pose: the box is relative
    init = torch.zeros((6, n), device=dev)
    init[0:2] = (torch.rand((2, n), generator=g, device=dev) - 0.5) * 4.0
    init[2] = (torch.rand(n, generator=g, device=dev) - 0.5) * (10.0 * 3.14159265 / 180.0)
    start = init[0:3].clone()
    deg = 3.14159265358979 / 180.0
    refs = [start + torch.tensor([r[0], r[1], r[2] * deg], device=dev)[:, None] for r in BOX_REFS]
    ref_buf = start.clone().contiguous()
    env.reset(init=init, new_ref=ref_buf)
    obs = torch.empty((n, 9), device=dev)
    rew = torch.empty(n, device=dev)
    done = torch.empty(n, dtype=torch.uint8, device=dev)

    def chunk():
        # first step of a chunk (re-)applies the setpoint in force: a no-op unless the sequence switched
        env.step(actions[0], new_ref=ref_buf, out=(obs, rew, done))
        for k in range(1, CHUNK):
            env.step(actions[k], out=(obs, rew, done))

    graph = None
    if not args.no_graph:
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            chunk()                                   # warm the launch path before capture
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            chunk()

    state = {'t': 0}

    def run_steps(k):
        """advance k env steps (k multiple of CHUNK), switching setpoints on the box schedule"""
        for _ in range(k // CHUNK):
            t = state['t'] % 1250
            if t == 0:
                ref_buf.copy_(start)
            if t in BOX_SWITCH_STEPS:
                ref_buf.copy_(refs[BOX_SWITCH_STEPS.index(t)])
            if graph is not None:
                graph.replay()
            else:
                chunk()
            state['t'] += CHUNK

    K = max(CHUNK, (args.steps // CHUNK) * CHUNK)
    W = max(0, ((args.warmup + CHUNK - 1) // CHUNK) * CHUNK) if args.warmup > 0 else 0
    run_steps(W)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ev0.record()
    run_steps(K)
    ev1.record()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)
    tt = torch.tensor([wall], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    wall = float(tt[0])
    assert bool(torch.isfinite(obs).all()) and bool(torch.isfinite(rew).all()), 'non-finite outputs'

    # ---- fused-rollout leg (dpenv_rollout): same workload, CHUNK env steps per launch, state in registers -----
    fused = None
    if not args.no_fused:
        fobs = torch.empty((CHUNK, n, 9), device=dev)
        frew = torch.empty((CHUNK, n), device=dev)
        fdone = torch.empty((CHUNK, n), dtype=torch.uint8, device=dev)
        env.reset(init=init, new_ref=start.clone())
        refs1 = ref_buf.view(1, 3, n)

        def run_fused(k):
            for c in range(k // CHUNK):
                t = (c * CHUNK) % 1250
                if t == 0:
                    ref_buf.copy_(start)
                if t in BOX_SWITCH_STEPS:
                    ref_buf.copy_(refs[BOX_SWITCH_STEPS.index(t)])
                env.rollout(actions, switch_steps=(0,), refs=refs1, out=(fobs, frew, fdone))

        run_fused(max(W, CHUNK))
        fe0, fe1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        tf0 = time.perf_counter()
        fe0.record()
        run_fused(K)
        fe1.record()
        torch.cuda.synchronize(dev)
        fwall = time.perf_counter() - tf0
        fms = fe0.elapsed_time(fe1)
        assert bool(torch.isfinite(fobs).all())
        fused = {'what': 'dpenv_rollout: %d env steps per launch, state resident in registers; open-loop action block; '
                         'same workload; NOT the headline value' % CHUNK,
                 'env_steps_per_s': n * K / fwall, 'us_per_step': fwall / K * 1e6, 'launch_us_events': fms * 1e3 / (K // CHUNK),
                 'bytes_per_env_step_moved': 28 + 36 + 4 + 1,
                 'GBps_at_177B_accounting': ALGO_BYTES_PER_ENV_STEP * n * K / (fms * 1e-3) / 1e9,
                 'frac_of_8TBps_at_177B_accounting': ALGO_BYTES_PER_ENV_STEP * n * K / (fms * 1e-3) / 1e9 / HBM_PEAK_GBPS}

    # ---- closed-loop leg (dpenv_policy_rollout): actor-critic 9-80-80-80-7 / -1 evaluated in-kernel on MFMA ----
    closed = None
    if not args.no_fused:
        from ml4ca_amd.policy import ActorCritic, policy_rollout
        ac = ActorCritic(9, 7, (80, 80, 80), seed=0, device=dev).upload(env)
        noise = torch.randn((CHUNK, n, 7), generator=g, device=dev)
        env.reset(init=init, new_ref=start.clone())
        pout = policy_rollout(env, CHUNK, noise=noise)

        def run_closed(k):
            for c in range(k // CHUNK):
                policy_rollout(env, CHUNK, noise=noise, out=pout)

        run_closed(max(W, CHUNK))
        ce0, ce1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        tc0 = time.perf_counter()
        ce0.record()
        run_closed(K)
        ce1.record()
        torch.cuda.synchronize(dev)
        cwall = time.perf_counter() - tc0
        assert bool(torch.isfinite(pout['obs']).all()) and bool(torch.isfinite(pout['logp']).all())
        # reference point: the same policy as separate torch kernels (fp32) + one env.step launch per step

        def torch_loop(k):
            o = obs
            for _ in range(k):
                mu, v = ac.forward_ref(o)
                a_ = mu + torch.exp(ac.log_std) * noise[0]
                lp = ac.logp_ref(a_, mu)
                o, r_, d_, _ = env.step(a_.contiguous())
            return o

        env.reset(init=init, new_ref=start.clone(), out=obs)
        torch_loop(20)
        torch.cuda.synchronize(dev)
        tt0 = time.perf_counter()
        torch_loop(200)
        torch.cuda.synchronize(dev)
        twall = time.perf_counter() - tt0
        flops = 2 * 2 * (9 * 80 + 80 * 80 * 2 + 80 * 7) * n       # actor + critic MACs x 2, per step (critic out 1 ~ 7)
        closed = {'what': 'dpenv_policy_rollout: %d steps per launch of actor (9-80-80-80-7) -> sample -> env.step -> critic, '
                          'PPO rows (o,a,r,v,logp,done,boot) written in-kernel; f16 MFMA policy, fp32 env' % CHUNK,
                  'env_steps_per_s': n * K / cwall, 'us_per_step': cwall / K * 1e6,
                  'policy_TFLOPs': flops * K / cwall / 1e12,
                  'unfused_torch_fp32_policy_plus_step_kernel': {'env_steps_per_s': n * 200 / twall, 'us_per_step': twall / 200 * 1e6},
                  'speedup_vs_unfused': (twall / 200) / (cwall / K)}

    # ---- config-5 leg (BASELINE.json configs[4]): drifting current, bf16 observation rows, full PPO rollout block
    #      (T = 400 = one episode, auto-reset) + GAE scan + advantage normalisation, all on device ------------------
    cfg5 = None
    if not args.no_fused:
        from ml4ca_amd import rollout as RO
        env5 = ml4ca_amd.BatchedRevoltEnv(n, variant='final', extended_state=True, cont_ang=True, device=dev, auto_reset=True,
                                          seed=2, env_id_base=rank * n, obs_dtype='bfloat16', current=True, current_drift=True)
        env5.set_current(torch.full((n,), 0.2, device=dev), torch.full((n,), 135.0 * deg, device=dev))
        ac.upload(env5)
        T5 = 400
        buf5 = RO.RolloutBuffer(T5, env5)
        noise5 = torch.randn((T5, n, 7), generator=g, device=dev)
        env5.reset()

        def epoch5():
            buf5.collect(env5, noise=noise5)
            buf5.finish()
            return buf5.get()

        epoch5()
        torch.cuda.synchronize(dev)
        t50 = time.perf_counter()
        reps5 = 3
        for _ in range(reps5):
            o5, a5, adv5, ret5, lp5 = epoch5()
        torch.cuda.synchronize(dev)
        w5 = (time.perf_counter() - t50) / reps5
        assert bool(torch.isfinite(adv5).all()) and o5.dtype == torch.bfloat16
        cfg5 = {'what': 'BASELINE.json configs[4]: %d envs, Gauss-Markov current (0.2 m/s, 135 deg), bf16 obs rows, one launch of '
                        'T = 400 policy-in-the-loop steps with auto-reset + GAE(0.99, 0.97) + advantage normalisation' % n,
                'env_steps_per_s': n * T5 / w5, 'ms_per_epoch': w5 * 1e3, 'us_per_step': w5 / T5 * 1e6}
        del env5, buf5, noise5

    # ---- config-4 leg: episode-boundary all-gather of [T=400, 32768, 19] f32 trajectory blocks ------------
    gather = None
    do_gather = (args.gather == 1) or (args.gather < 0 and world > 1)
    if do_gather and world > 1:
        T, nl = 400, 32768
        traj = torch.randn((T, nl, 19), device=dev)
        out = torch.empty((world, T, nl, 19), device=dev)
        from ml4ca_amd.dist import gather_trajectories
        gather_trajectories(traj, out=out)
        torch.cuda.synchronize(dev)
        dist.barrier()
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            gather_trajectories(traj, out=out)
        torch.cuda.synchronize(dev)
        gt = torch.tensor([(time.perf_counter() - t1) / reps], device=dev, dtype=torch.float64)
        dist.all_reduce(gt, op=dist.ReduceOp.MAX)
        gsec = float(gt[0])
        shard = traj.numel() * 4
        gather = {'what': 'all_gather_into_tensor of [400, 32768, 19] f32 per rank (config 4), not in `value`',
                  'ms': gsec * 1e3, 'shard_MB': shard / 1e6,
                  'recv_GBps_per_rank': shard * (world - 1) / gsec / 1e9,
                  'env_steps_per_s_step_plus_gather': world * nl * T / (T * (wall / K) * nl / n + gsec)}
        del out, traj

    if rank == 0:
        total_envs = n * world
        per_launch_bytes = ALGO_BYTES_PER_ENV_STEP * n
        kern_s = ev_ms * 1e-3 / K                      # HIP events around the K graph-replayed launches
        achieved = per_launch_bytes / kern_s / 1e9
        traffic = None
        try:
            tj = json.load(open(args.traffic_json))
            if tj.get('n_envs') == n:
                traffic = tj.get('hbm_bytes_per_launch')
        except Exception:
            pass
        res = {
            'metric': 'env-steps/sec at 65536 parallel envs' + (' [DIAGNOSTIC hold_plant: INVALID]' if args.hold_plant else ''), 'value': total_envs * K / wall, 'unit': 'env-steps/s',
            'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': wall / K * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE.json configs[2]: %d parallel envs per GPU, final/ext/cont_ang, 4-corner box '
                                   'setpoint sequence (switch steps 50/300/550/700/950 of 1250), terminate off, fp32' % n,
                       'envs_per_gpu': n, 'total_envs': total_envs, 'integrator': 'semi-implicit Euler 20 x 10 ms',
                       'launch': 'eager' if graph is None else 'hipGraph replay, %d steps per graph' % CHUNK,
                       'sharding': 'independent env shards, no data-path collective',
                       'backend': args.backend if world > 1 else None},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic,
                         'kernel': 'dpenv::step_kernel<4,true,false>', 'algorithmic_bytes_per_launch': per_launch_bytes,
                         'avg_launch_us': kern_s * 1e6,
                         'note': 'avg_launch_us = HIP-event time over the timed region / launches (includes the ~1.5 us '
                                 'kernel-boundary gap); 177 B/env-step x %d envs' % n},
            'reference_context': {'published_derived_env_steps_per_s': 34.3,
                                  'source': 'BASELINE.md: 2.4M interactions / 69930 s, 1 env, laptop + Cybersea'},
        }
        if gather:
            res['allgather'] = gather
        if fused:
            res['fused_rollout'] = fused
        if closed:
            res['policy_rollout'] = closed
        if cfg5:
            res['config5_ppo_rollout'] = cfg5
        if not args.no_cpu_baseline and world == 1:
            res['cpu_baseline'] = cpu_baseline(n, args.cpu_seconds)
        elif not args.no_cpu_baseline:
            res['cpu_baseline'] = None
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
