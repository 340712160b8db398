#!/usr/bin/env python3
"""One GPU's envs as ONE handle against SEVERAL handles stepped as independent chains on separate streams (round 5).
A launch of dpenv_step is kernel boundary + load burst + lone-wave arithmetic + store burst with nothing overlapping, because every wave of a
launch is in the same phase.  Independent chains - envs never interact, so a trainer can shard one GPU's envs over
K handles, each with its own stream / graph - drift out of phase, and one chain's bursts run under another's arithmetic.  Prints us per step of
ALL envs and env-steps/s for one handle of N envs and for K handles of N / K, each as 50-step graphs replayed 20 times, wall clock.
    python tools/multi_handle_step.py        -> profiles/r05s_multi_handle_step.txt"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ml4ca_amd
dev = torch.device('cuda', 0)
CH = 50
def build(n, seed, stream):
    env = ml4ca_amd.BatchedRevoltEnv(n, device=dev, terminate=False, time_limit=False, seed=seed)
    g = torch.Generator(device=dev); g.manual_seed(seed)
    actions = torch.randn((CH, n, 7), generator=g, device=dev) * 0.6065
    obs = torch.empty((n, 9), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
    with torch.cuda.stream(stream):
        env.reset()
        def chunk():
            for k in range(CH): env.step(actions[k], out=(obs, rew, done))
        chunk(); torch.cuda.synchronize(dev)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=stream): chunk()
    return env, gr, (actions, obs, rew, done)
def run(parts, reps=20):
    streams = [torch.cuda.Stream(device=dev) for _ in parts]
    built = [build(n, 1 + i, st) for i, (n, st) in enumerate(zip(parts, streams))]
    torch.cuda.synchronize(dev)
    for _ in range(4):
        for (env, gr, _), st in zip(built, streams):
            with torch.cuda.stream(st): gr.replay()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        for (env, gr, _), st in zip(built, streams):
            with torch.cuda.stream(st): gr.replay()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    tot = sum(parts)
    return dt / (reps * CH) * 1e6, tot * reps * CH / dt
for rep in range(2):
    for parts in ([131072], [65536] * 2, [196608], [65536] * 3, [262144], [131072] * 2, [65536] * 4, [524288], [131072] * 4, [65536] * 8):
        us, rate = run(parts)
        print('%-28s %.3f us per step of all %d envs   %.4e env-steps/s' % ('x'.join(str(p) for p in parts), us, sum(parts), rate), flush=True)
