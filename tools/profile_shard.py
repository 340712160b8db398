#!/usr/bin/env python3
"""Workload of the shard-size profile round (tools/profile_shard.sh): the kernels that are the DEFAULT at config 4's shard size
(32 768 envs per GPU) under the config-2 workload - termination and auto-reset ON, station keeping at the origin, training resets
(the loop they replace: spinup/algos/tf1/ppo/ppo.py:289-322) - each launched with a KNOWN number of env
steps per launch so that the profile's per-launch figures divide cleanly:

    step_kernel<4,true,false,RESETW=true>              1 step per launch   (dpenv_step, T launches in one HIP graph)
    rollout_ws_kernel<4,true,false>                     --chunk steps       (dpenv_rollout)
    policy_rollout_ws_kernel<..,ROLES=2,F16,GROUPS=2>   T steps             (dpenv_policy_rollout, f16, 128-env workgroups)
    policy_rollout_ws_kernel<..,ROLES=3,F32_ACTOR,2>    T steps             (critic wave, exact actor)
    policy_rollout_ws_kernel<..,ROLES=3,F32,2>          T steps             (critic wave, all exact)
and, with --big N (default 65 536), the 256-env-workgroup f16 form policy_rollout_ws_kernel<..,ROLES=2,F16,GROUPS=4> (the one
that carries scratch).  Prints ONE JSON line with the HIP-event time of every leg; run it plain and under rocprofv3."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=32768)
    ap.add_argument('--big', type=int, default=65536)
    ap.add_argument('--T', type=int, default=400)
    ap.add_argument('--chunk', type=int, default=50)
    ap.add_argument('--reps', type=int, default=3)
    args = ap.parse_args()
    import torch
    import ml4ca_amd
    from ml4ca_amd.policy import ActorCritic, policy_rollout, policy_launch_form
    dev = torch.device('cuda', 0)
    n, T, CH = args.envs, args.T, args.chunk
    rec = {'envs': n, 'T': T, 'chunk': CH, 'reps': args.reps, 'workload': 'config 2: final/ext/cont_ang, terminate on, auto_reset on, training resets, '
           'Gaussian actions std e^-0.5 (open-loop legs) / in-kernel exploration noise (closed-loop legs)', 'legs': {}}

    def timed(fn, reps):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) * 1e-3 / reps

    env = ml4ca_amd.BatchedRevoltEnv(n, device=dev, terminate=True, auto_reset=True, seed=4)
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    actions = torch.randn((CH, n, 7), generator=g, device=dev) * 0.6065
    obs = torch.empty((T, n, 9), device=dev)
    rew = torch.empty((T, n), device=dev)
    done = torch.empty((T, n), dtype=torch.uint8, device=dev)
    env.reset()

    def episode_steps():
        for t in range(T):
            env.step(actions[t % CH], out=(obs[t], rew[t], done[t]))

    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        episode_steps()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        episode_steps()
    s = timed(gr.replay, 2 * args.reps)
    rec['legs']['step'] = {'kernel': 'step_kernel<4,true,false,true> (reset wave)', 'steps_per_launch': 1, 'us_per_step': s / T * 1e6,
                           'resets_per_env_step': float((done != 0).float().mean())}
    del gr

    def episode_fused():
        for c in range(T // CH):
            env.rollout(actions, out=(obs[c * CH:(c + 1) * CH], rew[c * CH:(c + 1) * CH], done[c * CH:(c + 1) * CH]))

    s = timed(episode_fused, 2 * args.reps)
    rec['legs']['fused'] = {'kernel': 'rollout_ws_kernel<4,true,false>', 'steps_per_launch': CH, 'us_per_step': s / T * 1e6}
    del obs, rew, done

    ac = ActorCritic(9, 7, (80, 80, 80), seed=0, device=dev)

    def closed(e, tag, precs):
        out = None
        for prec in precs:
            ac.upload(e, precision=prec)
            out = policy_rollout(e, T, sample=True, out=out)
            s_ = timed(lambda: policy_rollout(e, T, sample=True, out=out), args.reps)
            rec['legs']['closed_%s_%s' % (tag, prec)] = {'launch_form': '%s, %d envs per workgroup' % policy_launch_form(e), 'steps_per_launch': T,
                                                          'envs': e.n_envs, 'us_per_step': s_ / T * 1e6,
                                                          'resets_per_env_step': float((out['done'] != 0).float().mean())}

    closed(env, 'shard', ('f16', 'f32_actor', 'f32'))
    del env
    # ---- per-env vessel parameter blocks (round 5), same shard size: dpenv_step with the blocks in registers / through the LDS image
    #      (terminate off: step_kernel<4,true,2|3,false>), and the randomised form on the config-2 workload (reset wave draws the hulls:
    #      step_kernel<4,true,4,true>; closed loop: the RND instantiation of the 128-env f16 form)
    import numpy as np
    base = np.asarray(ml4ca_amd.default_vessel(), np.float32)
    hulls = torch.from_numpy(np.ascontiguousarray(np.concatenate([
        base[:26, None] * (1.0 + 0.15 * np.random.RandomState(7).uniform(-1, 1, size=(26, n))), np.zeros((6, n))]).astype(np.float32))).to(dev)
    so, sr, sd = torch.empty((n, 9), device=dev), torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)

    def step_graph(e, steps=200):
        def run():
            for t in range(steps):
                e.step(actions[t % CH], out=(so, sr, sd))
        side2 = torch.cuda.Stream(device=dev)
        side2.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side2):
            run()
        torch.cuda.current_stream(dev).wait_stream(side2)
        torch.cuda.synchronize(dev)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            run()
        return timed(g2.replay, 2 * args.reps) / steps

    for tag, lds in (('registers', False), ('lds_image', True)):
        e = ml4ca_amd.BatchedRevoltEnv(n, device=dev, terminate=False, time_limit=False, seed=4, per_env_lds=lds)
        e.set_vessel_params(hulls)
        e.reset()
        rec['legs']['step_per_env_' + tag] = {'kernel': 'step_kernel<4,true,%d,false>' % (3 if lds else 2), 'steps_per_launch': 1, 'us_per_step': step_graph(e) * 1e6}
        del e
    e = ml4ca_amd.BatchedRevoltEnv(n, device=dev, terminate=True, auto_reset=True, seed=4)
    e.set_vessel_randomisation(0.15)
    e.reset()
    rec['legs']['step_randomised'] = {'kernel': 'step_kernel<4,true,4,true> (reset wave draws the hulls)', 'steps_per_launch': 1, 'us_per_step': step_graph(e) * 1e6}
    closed(e, 'shard_randomised', ('f16', 'f32_actor'))
    del e
    if args.big > 0:
        envb = ml4ca_amd.BatchedRevoltEnv(args.big, device=dev, terminate=True, auto_reset=True, seed=4)
        envb.reset()
        closed(envb, 'big', ('f16',))
    torch.cuda.synchronize(dev)
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
