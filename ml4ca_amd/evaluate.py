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


TEST_POLICY_REFS = ((5.0, 0.0, 0.0), (0.0, -5.0, 0.0), (0.0, 0.0, math.pi / 2), (0.0, 5.0, math.pi / 2), (-5.0, 0.0, 0.0))   # test_policy.py:127


def run_RL_policy(env, ac, num_episodes=6, max_ep_len=None, test_setpoint_changes=False):
    """Batched form of the reference's evaluation harness run_RL_policy (spinup/utils/test_policy.py:97-186): episode k
    starts at fixed point k (env.reset(fixed_point=k), simtools.py:91-107) with the setpoint at the origin and runs the
    DETERMINISTIC policy (test_policy.py:90) until done or max_ep_len; with test_setpoint_changes the k-th episode gets
    refs[k] at ep_len == max_ep_len / 2 through a step with a ZERO action (test_policy.py:148-153, as the reference
    does).  The reference runs the episodes one after another on one env; here they are the envs of one batch
    (``env.n_envs`` must equal num_episodes, auto_reset off).  Returns a dict: EpRet, EpLen [episodes]; obs
    [T+1, episodes, obs_dim]; rew [T+1, episodes]; ned_pos, ned_ref [T+1, episodes, 3]; action_vec [T+1, episodes, 6]
    (thrust % x3, azimuth rad x3 via act_2_act_map_inv, test_policy.py:120-124,145-146,162); valid [T+1, episodes]."""
    torch = _torch()
    from .policy import policy_forward
    from . import reset_samplers, _lib
    n = env.n_envs
    assert n == num_episodes and not env.auto_reset, 'one env per episode, auto_reset off'
    T = int(max_ep_len or env.max_ep_len)
    dev = env.device
    init = torch.zeros((6, n), device=dev)
    for k in range(n):
        N_, E_, Y_ = reset_samplers.get_fixed_pose_on_radius(k)
        init[0, k], init[1, k], init[2, k] = N_, E_, Y_
    obs = env.reset(init=init, new_ref=torch.zeros((3, n), device=dev))
    od = env.num_states
    out_obs = torch.zeros((T + 1, n, od), device=dev)
    out_rew = torch.zeros((T + 1, n), device=dev)
    ned_pos = torch.zeros((T + 1, n, 3), device=dev)
    ned_ref = torch.zeros((T + 1, n, 3), device=dev)
    act_vec = torch.zeros((T + 1, n, 6), device=dev)
    valid = torch.zeros((T + 1, n), dtype=torch.bool, device=dev)
    action_init = torch.tensor([env.default_actions[i] for i in range(6)], dtype=torch.float32, device=dev)
    alive = torch.ones(n, dtype=torch.bool, device=dev)
    ep_ret = torch.zeros(n, device=dev)
    ep_len = torch.zeros(n, dtype=torch.int32, device=dev)
    bounds = torch.tensor(env.real_action_bounds, dtype=torch.float32, device=dev)

    def record(t, o, r, vec):
        st, _ = env.get_state()
        out_obs[t], out_rew[t], act_vec[t], valid[t] = o.float(), r, vec, alive
        ned_pos[t], ned_ref[t] = st[0:3].T, st[6:9].T

    vec = action_init.expand(n, 6).clone()
    record(0, obs, torch.zeros(n, device=dev), vec)
    refs = None
    if test_setpoint_changes:
        refs = torch.tensor([TEST_POLICY_REFS[k % len(TEST_POLICY_REFS)] for k in range(n)], dtype=torch.float32, device=dev).T.contiguous()
    for t in range(T):
        mu, _ = policy_forward(env, obs.float().contiguous() if obs.dtype != torch.float32 else obs)
        a = mu
        if env.cont_ang:                                                  # handle_continuous_angles, customEnv.py:227-235
            cmd = torch.cat([a[:, 0:3], torch.atan2(a[:, 3], a[:, 4])[:, None] / bounds[3], torch.atan2(a[:, 5], a[:, 6])[:, None] / bounds[3]], 1)
        else:
            cmd = a
        cmd = torch.clamp(cmd * bounds, -bounds, bounds)                  # scale_and_clip, customEnv.py:215-225
        vec = vec.clone()
        for i in range(cmd.shape[1]):
            vec[:, env.act_2_act_map_inv[i]] = cmd[:, i]
        if test_setpoint_changes and t == T // 2:
            obs, r, d, _ = env.step(torch.zeros_like(a), new_ref=refs)    # test_policy.py:148-153
            vec = action_init.expand(n, 6).clone()
        else:
            obs, r, d, _ = env.step(a.contiguous())
        ep_ret += torch.where(alive, r, torch.zeros_like(r))
        ep_len += alive.to(torch.int32)
        record(t + 1, obs, r, vec)
        alive = alive & ((d & (_lib.DONE_TERMINAL | _lib.DONE_FAULT)) == 0)
    return dict(EpRet=ep_ret, EpLen=ep_len, obs=out_obs, rew=out_rew, ned_pos=ned_pos, ned_ref=ned_ref, action_vec=act_vec,
                valid=valid)
