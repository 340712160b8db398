#!/usr/bin/env python3
"""How much of the headline's time per step is the seam between graph replays?  dpenv_step at 65 536 envs captured as graphs of C steps
(C = 10 .. 1250), replayed back to back for the same total number of steps; with and without the tiny setpoint copy bench.py used to issue
before every replay.      python tools/graph_chunk_sweep.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ml4ca_amd

n = 65536
dev = torch.device('cuda', 0)
env = ml4ca_amd.BatchedRevoltEnv(n, device=dev, terminate=False, time_limit=False, seed=1)
g = torch.Generator(device=dev); g.manual_seed(1)
actions = torch.randn((50, n, 7), generator=g, device=dev) * 0.6065
ref = torch.zeros((3, n), device=dev); ref2 = torch.zeros((3, n), device=dev)
obs = torch.empty((n, 9), device=dev); rew = torch.empty(n, device=dev); done = torch.empty(n, dtype=torch.uint8, device=dev)
env.reset()
TOTAL = 5000
for C in (10, 25, 50, 125, 250, 625, 1250):
    def chunk():
        env.step(actions[0], new_ref=ref, out=(obs, rew, done))
        for k in range(1, C):
            env.step(actions[k % 50], out=(obs, rew, done))
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        chunk()
    torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        chunk()
    for copy in (False, True):
        for _ in range(3):
            gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(TOTAL // C):
            if copy:
                ref.copy_(ref2)
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        print('graph of %4d steps, %s: %.3f us per step (events), %.3f (wall)' % (C, 'copy before every replay' if copy else 'no copy              ', e0.elapsed_time(e1) * 1e3 / TOTAL, wall * 1e6 / TOTAL), flush=True)
    del gr
