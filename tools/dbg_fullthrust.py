import numpy as np, torch, sys
sys.path.insert(0, '.')
import ml4ca_amd
n = 4
env = ml4ca_amd.BatchedRevoltEnv(n, terminate=False, time_limit=False)
env.reset(init=torch.zeros((6, n), device=env.device))
a = torch.zeros((n, 7), device=env.device); a[:, 4] = 1; a[:, 6] = 1; a[:, 1:3] = 1
for t in range(500):
    o, r, d, _ = env.step(a)
    if t in (0, 1, 10, 18, 19, 20, 21, 22, 30, 100, 499):
        s, c = env.get_state()
        print(t, s[:, 0].cpu().numpy().round(5), c[:, 0].cpu().numpy(), int(d[0]))
