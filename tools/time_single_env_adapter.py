import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import ml4ca_amd
from ml4ca_amd.env import RevoltFinal
env = RevoltFinal(extended_state=True, cont_ang=True, testing=True)
o = env.reset()
rng = np.random.RandomState(0)
acts = rng.normal(0, 0.3, size=(3000, 7))
for k in range(200): env.step(acts[k])
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(2000): o, r, d, _ = env.step(acts[k % 3000])
dt = time.perf_counter() - t0
print('single-env adapter: %.1f us per step (%.0f steps/s); last obs %s reward %.4f' % (dt / 2000 * 1e6, 2000 / dt, np.round(o[:3], 4), r))
