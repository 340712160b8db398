"""Time dpenv_gae_stats + normalisation at the config-5 shape (T = 400, n = 65 536).  Usage: python tools/gae_bench.py"""
import sys
import torch
sys.path.insert(0, '.')
from ml4ca_amd import rollout
T, n = 400, 65536
dev = 'cuda:0'
g = torch.Generator(device=dev).manual_seed(0)
rew = torch.randn((T, n), generator=g, device=dev); val = torch.randn((T, n), generator=g, device=dev)
end = (torch.rand((T, n), generator=g, device=dev) < 0.003).to(torch.uint8); boot = torch.randn((T, n), generator=g, device=dev)
adv = torch.empty_like(rew); ret = torch.empty_like(rew)
stats = torch.zeros(2, dtype=torch.float64, device=dev)
for _ in range(3):
    rollout.gae(rew, val, end=end, boot=boot, out=(adv, ret), stats=stats)
e = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
torch.cuda.synchronize()
reps = 20
keep = None
e[0].record()
for _ in range(reps):
    rollout.gae(rew, val, end=end, boot=boot, out=(adv, ret), stats=stats)
e[1].record()
keep = adv.clone()
e[2].record()
for _ in range(reps):
    adv.copy_(keep)
e[3].record()
for _ in range(reps):
    adv.copy_(keep)
    rollout.normalize_advantages(adv, stats=stats)
e[4].record()
torch.cuda.synchronize()
tg = e[0].elapsed_time(e[1]) / reps
tn = (e[3].elapsed_time(e[4]) - e[2].elapsed_time(e[3])) / reps
print('gae+stats %.1f us (%.0f GB/s at 21 B/env-step = %.2f of 8 TB/s)   normalise %.1f us (%.0f GB/s at 8 B)' % (
    tg * 1e3, 21 * T * n / tg / 1e6, 21 * T * n / tg / 1e6 / 8000, tn * 1e3, 8 * T * n / tn / 1e6))
