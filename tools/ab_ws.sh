#!/bin/bash
# A/B of the closed-loop launch forms: one wave per 64 envs (DPENV_POLICY_WS=0) vs an env wave + a network wave (=1)
for ws in 0 1 0 1; do
  DPENV_POLICY_WS=$ws timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2500 --warmup 250 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('DPENV_POLICY_WS=$ws', 'step %.3f' % (d['ms_per_step']*1e3), 'fused %.3f' % d['fused_rollout']['us_per_step'], 'policy %.3f' % d['policy_rollout']['us_per_step'], 'cfg5 %.3f' % d['config5_ppo_rollout']['us_per_step'])" || exit 1
done
