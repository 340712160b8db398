"""Evaluation metrics of the thesis' setpoint tests as device reductions over rollout blocks.

  * IAE  - results/all_plots/common.py:56-74 with the normalisation of box_test/plot_pos.py:174:
           integral over time of || (eta - eta_ref) / [5 m, 5 m, 25 deg] ||, trapezoidal.
  * work - results/all_plots/box_test/plot_act.py:124-135,184-211: P = sgn(n) KQ0 2 pi rho D^5 (n/100 rps_max)^3 per
           thruster, energy-equivalent W* = integral of P dt, trapezoidal.
  * BOX  - the 4-corner box setpoint sequence relative to the start pose (box_test/plot_pos.py:55-59).
Inputs are the [T, n, .] blocks that dpenv_rollout / dpenv_policy_rollout write; outputs are per-env tensors.
"""
import math

BOX_REFS = ((5.0, 0.0, 0.0), (5.0, -5.0, 0.0), (5.0, -5.0, -45.0), (0.0, -5.0, -45.0), (0.0, 0.0, 0.0))   # m, m, deg
BOX_TIMES = (10.0, 60.0, 110.0, 140.0, 190.0)                                                             # s
IAE_NORM = (5.0, 5.0, 25.0)
RPS_MAX = {'bow': 33.0, 'stern': 11.0}
DIAMETER = {'bow': 0.06, 'stern': 0.15}
KQ0 = {'bow': 0.02, 'stern': 0.036}
RHO = 1025.0


def _torch():
    import torch
    return torch


def box_schedule(start, dt=0.2):
    """(switch_steps, refs [5, 3, n]) of the box test for envs starting at start [3, n] (N, E, psi)."""
    torch = _torch()
    steps = tuple(int(round(t / dt)) for t in BOX_TIMES)
    refs = torch.stack([start + torch.tensor([r[0], r[1], math.radians(r[2])], device=start.device, dtype=start.dtype)[:, None]
                        for r in BOX_REFS])
    return steps, refs.contiguous()


def iae(obs, dt=0.2):
    """IAE per env from an observation block [T, n, >=3] (body-frame pose error; the rotation preserves the norm of
    the position error, so the NED form of common.py:56 equals the body-frame form).  Returns (total [n], cumulative [T, n])."""
    torch = _torch()
    e = obs[..., :3].float()
    err = torch.sqrt((e[..., 0] / IAE_NORM[0]) ** 2 + (e[..., 1] / IAE_NORM[1]) ** 2 +
                     (torch.rad2deg(e[..., 2]) / IAE_NORM[2]) ** 2)
    seg = 0.5 * (err[1:] + err[:-1]) * dt
    cum = torch.cat([torch.zeros_like(err[:1]), torch.cumsum(seg, dim=0)], dim=0)
    return cum[-1], cum


def thruster_power(n_pct, which):
    """plot_act.py:133-135."""
    torch = _torch()
    return torch.sign(n_pct) * KQ0[which] * 2 * math.pi * RHO * DIAMETER[which] ** 5 * (n_pct / 100.0 * RPS_MAX[which]) ** 3


def work(thrust_pct, dt=0.2):
    """W* per env and thruster from commanded thrust [T, n, 3] in percent (bow, port, star): returns [n, 3]."""
    torch = _torch()
    p = torch.stack([thruster_power(thrust_pct[..., 0], 'bow'), thruster_power(thrust_pct[..., 1], 'stern'),
                     thruster_power(thrust_pct[..., 2], 'stern')], dim=-1)
    return (0.5 * (p[1:] + p[:-1]) * dt).sum(dim=0)


def commanded_thrust(act):
    """Percent thrust commands from raw policy actions [T, n, >=3] (scale_and_clip, customEnv.py:215-225)."""
    return (act[..., :3] * 100.0).clamp(-100.0, 100.0)
