#!/usr/bin/env python3
"""PPO-clip on the batched ReVolt DP environment with the rollout entirely inside one launch per epoch.

Host-side glue around the accelerated path (the PPO update is out of the hot path's scope: SURVEY section 2 row 9):
the algorithm and hyper-parameters are the reference's (spinup/algos/tf1/ppo/ppo.py:109-347, train.py:29-55,
config.json of the shipped run): clip 0.2, pi_lr 3e-4, vf_lr 1e-3, <= 80 policy iterations with early stop at
KL > 1.5 * 0.01, 80 value iterations, gamma 0.99, lambda 0.97, hidden 3 x 80 leaky-relu, T = 400.
What differs is the rollout: N environments x T steps from ONE dpenv_policy_rollout launch (actor, sampling,
env.step, critic, trajectory rows), GAE by one scan kernel, advantage statistics by device reductions.

    python examples/train_ppo.py --envs 4096 --epochs 30
    python examples/train_ppo.py --envs 4096 --epochs 40 --randomise 0.15 --eval      domain randomisation: every episode of every env
                                                                                      runs on its own hull, +-15 % on all 26 parameters, re-drawn by the reset path
                                                                                      inside the rollout launch; --eval: the reference's evaluation harness
                                                                                      (test_policy.py:97-186) and the box test (IAE, energy) on the NOMINAL hull
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ml4ca_amd
from ml4ca_amd import dist as D
from ml4ca_amd import rollout
from ml4ca_amd.policy import ActorCritic


def evaluate_actor(ac, dev, preset, precision, seed, out=print):
    """The trained actor on the NOMINAL hull (and on a spread of hulls): the reference's run_RL_policy (spinup/utils/test_policy.py:97-186: six
    fixed starts, deterministic policy) and the thesis' 4-corner box test with its two metrics - IAE (results/all_plots/common.py:60-74) and
    the energy-equivalent work of the thruster power model (box_test/plot_act.py:128-135,184-211)."""
    from ml4ca_amd import evaluate as EV
    from ml4ca_amd.policy import policy_rollout
    nominal = ml4ca_amd.default_vessel(preset)
    env6 = ml4ca_amd.BatchedRevoltEnv(6, device=dev, auto_reset=False, testing=True, vessel_params=nominal)
    ac.upload(env6, precision=precision)
    r = EV.run_RL_policy(env6, ac)
    out('eval  run_RL_policy on the nominal hull (six fixed starts, deterministic actor): EpRet %s  EpLen %s' % (
        [round(float(x), 1) for x in r['EpRet']], [int(x) for x in r['EpLen']]))
    res = {'EpRet_mean': float(r['EpRet'].mean()), 'EpLen_mean': float(r['EpLen'].float().mean())}
    T, nb = 1250, 1024
    # (the last two rows: the thesis' current box test - 0.2 m/s towards 135 deg, results/all_plots/current_box_test/plot_pos.py:78 - and a spread
    # of currents around it: +-0.1 m/s, +-90 deg, one draw per env)
    for tag, spread, cur in (('nominal hull', 0.0, None), ('hulls +-15 %', 0.15, None), ('hulls +-30 %', 0.30, None), ('hulls +-50 %', 0.50, None),
                             ('current 0.2 m/s @ 135 deg', 0.0, (0.0, 0.0)), ('currents 0.2 +-0.1 m/s, 135 +-90 deg', 0.0, (0.1, 1.5708))):
        env = ml4ca_amd.BatchedRevoltEnv(nb, device=dev, terminate=False, time_limit=False, seed=seed + 77, vessel_params=nominal, current=cur is not None)
        if spread > 0:
            env.set_vessel_randomisation(spread, nominal=nominal)          # one draw per env at the reset below; no resets after it
        if cur is not None:
            env.set_current(torch.full((nb,), 0.2, device=dev), torch.full((nb,), 2.35619, device=dev))
            if cur[0] > 0:
                env.set_current_randomisation(cur[0], cur[1])              # one draw per env at the reset below
        ac.upload(env, precision=precision)
        start = torch.zeros((3, nb), device=dev)
        env.reset(init=torch.zeros((6, nb), device=dev), new_ref=start.clone())
        steps, refs = EV.box_schedule(start)
        o = policy_rollout(env, T, noise=None, switch_steps=steps, refs=refs)
        iae_tot, _ = EV.iae(o['obs'])
        w = EV.work(EV.commanded_thrust(o['act']))
        e = o['obs'][:, :, :3]
        k = [max(t - 1, 0) for t in list(steps)[1:] + [T - 1]]             # just before each switch: how close to the corner
        pos = torch.sqrt(e[k, :, 0] ** 2 + e[k, :, 1] ** 2)
        out('eval  box test (1250 steps, %d envs), %-13s: IAE %.2f (worst env %.2f)  work bow/port/star %s  corner error %.2f m / %.1f deg (worst %.2f m)  reward/step %.3f' % (
            nb, tag, float(iae_tot.mean()), float(iae_tot.max()), [round(float(x), 1) for x in w.mean(0)], float(pos.mean()),
            float(torch.rad2deg(e[k, :, 2].abs().mean())), float(pos.max()), float(o['rew'].mean())))
        res[tag] = {'IAE': float(iae_tot.mean()), 'IAE_worst': float(iae_tot.max()), 'work': [float(x) for x in w.mean(0)],
                    'corner_error_m': float(pos.mean()), 'reward_per_step': float(o['rew'].mean())}
        del env
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=4096)
    ap.add_argument('--epochs', type=int, default=30)
    ap.add_argument('--steps', type=int, default=400, help='rollout length per epoch = max_ep_len (train.py:70-73)')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--minibatch', type=int, default=1 << 18, help='samples per gradient step (full batch in the reference)')
    ap.add_argument('--activation', default='leaky', choices=('leaky', 'relu', 'tanh'), help='hidden activation (train.py:24,31)')
    ap.add_argument('--precision', default='f32', choices=('f16', 'f32', 'f32_actor'),
                    help="in-kernel network arithmetic: 'f32' (split-f16, within 1e-5 of the fp32 update's own evaluation: the PPO ratio starts at 1) or 'f16' (fast)")
    ap.add_argument('--exchange', default='gradients', choices=('gradients', 'rollout'),
                    help="multi-rank runs: 'gradients' = every rank updates on ITS OWN episode and the gradients are averaged, exactly the "
                         "reference (ppo.py:226, mpi_tf.py:29-62: no trajectory ever crosses); 'rollout' = BASELINE.json config 4: the ranks "
                         "all-gather the episode (dist.EpisodeExchange: obs | act | logp chunk by chunk under the next chunk's launch, adv | ret "
                         "after the local scan) and every rank runs the identical update on the global batch - no gradient all-reduce")
    ap.add_argument('--chunks', type=int, default=4, help="--exchange rollout: pieces the episode is rolled out and posted in")
    ap.add_argument('--reset-at-end', action='store_true', help='the reference\'s epoch boundary (ppo.py:305-322): every env is cut and re-drawn '
                                                                'after the last step of an epoch (matters when --steps < max_ep_len)')
    ap.add_argument('--randomise', type=float, default=0.0, help='R > 0: domain randomisation - every reset (also the ones inside the rollout launch) draws the '
                                                                  'new episode\'s hull: each of the 26 vessel parameters = nominal x (1 + R u), u ~ U[-1, 1), '
                                                                  'Philox keyed (seed; global env id, episode): independent of the rank count')
    ap.add_argument('--current', default='', help="'V,BETA_DEG': train in a current (e.g. '0.2,135': the reference's one operating point, "
                                                 'results/all_plots/current_box_test/plot_pos.py:78)')
    ap.add_argument('--randomise-current', default='', help="'RV,RB_DEG' with --current: every reset (also inside the rollout launch) draws the new episode's current, "
                                                           'V_c = max(0, V + RV u1), beta_c = BETA + RB u2 (dpenv_set_current_randomisation)')
    ap.add_argument('--preset', default='thrust_loss', choices=('no_loss', 'thrust_loss', 'dynpos_fit', 'dynpos_fit_thrust_loss'),
                    help="nominal hull (dpenv_default_vessel_ex); default (round 6): the thrust-loss preset - the steady speeds 'with thrust losses' are the velocity "
                         'bounds the reference trains with (customEnv.py:17,26), and the shared training form runs it at the cost of the default hull')
    ap.add_argument('--eval', action='store_true', help='after training: run_RL_policy + the box test (IAE, energy) on the nominal hull and on spreads of hulls')
    ap.add_argument('--eval-presets', default='', help="comma-separated presets to run --eval on (default: the training preset), e.g. 'no_loss,thrust_loss': "
                                                      'how an actor trained on one thrust regime fares on the other')
    ap.add_argument('--save', default='', help='write the trained parameters (reference variable names) to this .npz')
    ap.add_argument('--backend', default='nccl', help="'nccl' (RCCL, one GPU per rank) or 'gloo' (rehearsal)")
    ap.add_argument('--same-device', action='store_true', help='all ranks on cuda:0 (multi-rank rehearsal on a one-GPU box)')
    args = ap.parse_args()
    # one process per GPU under torch.distributed.run (backend nccl = RCCL); envs shard by global id, gradients average
    rank, world, local = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))
    dev = torch.device('cuda', 0 if args.same_device else local)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            torch.distributed.init_process_group('nccl', device_id=dev)
        else:
            torch.distributed.init_process_group(args.backend)
    torch.manual_seed(args.seed + 1000 * rank)
    env = ml4ca_amd.BatchedRevoltEnv(args.envs, auto_reset=True, seed=args.seed, device=dev, env_id_base=rank * args.envs, current=bool(args.current),
                                     vessel_params=ml4ca_amd.default_vessel(args.preset) if args.preset != 'no_loss' else None)   # final / ext / cont_ang
    if args.randomise > 0:
        env.set_vessel_randomisation(args.randomise, nominal=ml4ca_amd.default_vessel(args.preset))
    if args.current:
        import math
        v_c, b_c = (float(x) for x in args.current.split(','))
        env.set_current(torch.full((args.envs,), v_c, device=dev), torch.full((args.envs,), math.radians(b_c), device=dev))
        if args.randomise_current:
            r_v, r_b = (float(x) for x in args.randomise_current.split(','))
            env.set_current_randomisation(r_v, math.radians(r_b))
    ac = ActorCritic(9, 7, (80, 80, 80), leak=0.2, seed=args.seed, device=dev, activation=args.activation)
    D.sync_params(ac.parameters())                                               # sync_all_params, ppo.py:255
    for p in ac.parameters():
        p.requires_grad_(True)
    pi_params = ac.pi_W + ac.pi_b + [ac.log_std]
    v_params = ac.v_W + ac.v_b
    pi_opt = torch.optim.Adam(pi_params, lr=3e-4)
    v_opt = torch.optim.Adam(v_params, lr=1e-3)
    clip, target_kl, T, n = 0.2, 0.01, args.steps, args.envs
    buf = rollout.RolloutBuffer(T, env, gamma=0.99, lam=0.97)
    ac.upload(env, precision=args.precision)          # device pointers: one packing kernel, no host copy
    env.reset()
    gather = world > 1 and args.exchange == 'rollout'
    ex = D.EpisodeExchange(buf.exchange_blocks(), n_chunks=args.chunks) if gather else None
    if gather:
        torch.manual_seed(args.seed)                  # identical minibatch draws on every rank: the updates stay identical without a broadcast
    if rank == 0:
        print('epoch  mean_reward/step(max 3.5)  terminated/1k-steps  pi_iters  KL      V-loss    rollout_ms  update_s')
    for epoch in range(args.epochs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        # exploration noise is drawn inside the kernel (core.py:85), keyed by seed / global env id / draw index: no [T, n, 7]
        # noise block, and the trajectories do not depend on how many ranks share the envs
        if gather:
            # on-policy and overlapped: a piece of the episode crosses xGMI while the next piece is being rolled out; the advantages
            # are scanned and normalised locally (24 bytes of statistics cross) and follow as 8 B per env-step
            for c in range(ex.C):
                blk = buf.collect(env, sample=True, rows=ex.rows(c), reset_at_end=args.reset_at_end)
                ex.post_steps(c)
            buf.finish()
            buf.get()
            ex.post_scan()
            ex.wait()
            obs, act, adv, ret, logp_old = (ex.flat(k) for k in ('obs', 'act', 'adv', 'ret', 'logp'))
            obs = obs.float()
        else:
            blk = buf.collect(env, sample=True, reset_at_end=args.reset_at_end)
            buf.finish()
            obs, act, adv, ret, logp_old = buf.get()
        torch.cuda.synchronize()
        t_roll = time.perf_counter() - t0
        obs, act = obs.reshape(-1, 9), act.reshape(-1, 7)
        adv, ret, logp_old = adv.reshape(-1), ret.reshape(-1), logp_old.reshape(-1)
        N = obs.shape[0]
        mb = min(args.minibatch, N)
        t1 = time.perf_counter()
        kl, pi_iters = 0.0, 0
        for i in range(80):                                                     # ppo.py:265-271
            idx = torch.randint(0, N, (mb,), device=dev) if mb < N else slice(None)
            mu = ac._mlp(obs[idx], ac.pi_W, ac.pi_b)
            logp = ac.logp_ref(act[idx], mu)
            ratio = torch.exp(logp - logp_old[idx])
            a = adv[idx]
            pi_loss = -torch.min(ratio * a, torch.clamp(ratio, 1 - clip, 1 + clip) * a).mean()   # ppo.py:238-240
            kl_t = (logp_old[idx] - logp).mean().detach()
            kl = float(kl_t if gather else D.mean_across_ranks(kl_t))             # mpi_avg(kl), ppo.py:267 (identical on every rank when gathered)
            if kl > 1.5 * target_kl:                                            # ppo.py:267-270
                break
            pi_opt.zero_grad()
            pi_loss.backward()
            if not gather:
                D.average_gradients(pi_params)                                 # mpi_tf.py:59-62 (no-op on one rank)
            pi_opt.step()
            pi_iters += 1
        for i in range(80):                                                     # ppo.py:272-273
            idx = torch.randint(0, N, (mb,), device=dev) if mb < N else slice(None)
            v = ac._mlp(obs[idx], ac.v_W, ac.v_b)[:, 0]
            v_loss = ((ret[idx] - v) ** 2).mean()                               # ppo.py:241
            v_opt.zero_grad()
            v_loss.backward()
            if not gather:
                D.average_gradients(v_params)
            v_opt.step()
        with torch.no_grad():
            ac.log_std.clamp_(-4.0, 1.0)
        if gather:
            D.assert_params_in_step(ac.parameters())                            # replicated updates: no collective keeps them equal, so check
        ac.upload(env, precision=args.precision)
        torch.cuda.synchronize()
        t_upd = time.perf_counter() - t1
        done = blk['done']
        if rank == 0:
            print('%5d  %10.3f  %22.2f  %8d  %.4f  %8.1f  %9.1f  %8.2f' % (
                epoch, float(blk['rew'].mean()), 1000.0 * float((done & 1).float().mean()), pi_iters, kl,
                float(v_loss.detach()), t_roll * 1e3, t_upd))
    if rank == 0:
        print('env-steps collected: %d (%.1f M per epoch, %d rank(s))' % (args.epochs * T * n * world, T * n * world / 1e6, world))
        if args.randomise > 0:
            hp = env.get_vessel_params()[:4]
            print('hulls in force at the end: m11 %.1f .. %.1f (nominal %.1f), episodes per env so far %.1f' % (
                float(hp[0].min()), float(hp[0].max()), float(ml4ca_amd.default_vessel(args.preset)[0]), float(env.get_state()[1][1].float().mean())))
        for p in ac.parameters():
            p.requires_grad_(False)
        if args.save:
            import numpy as np
            np.savez(args.save, **{k.replace('/', '.'): v for k, v in ac.state_dict().items()})
        if args.eval:
            for pz in (args.eval_presets.split(',') if args.eval_presets else [args.preset]):
                print('eval on the %s preset (trained on %s%s)' % (pz, args.preset, ', hulls re-drawn +-%g %%' % (100 * args.randomise) if args.randomise > 0 else ''))
                evaluate_actor(ac, dev, pz, 'f32', args.seed)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
