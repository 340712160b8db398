#!/bin/bash
# A/B of the closed-loop launch forms: one wave per 64 envs vs an env wave + a network wave (bench.py --policy-form)
for form in one_wave two_wave one_wave two_wave; do
  timeout -k 10 300 python bench.py --side-json /tmp/dpenv_side.json --no-cpu-baseline --steps 1250 --warmup 100 --policy-form $form 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); d.update(json.load(open('/tmp/dpenv_side.json'))); print('$form', 'step %.3f' % (d['ms_per_step']*1e3), 'fused %.3f' % d['fused_rollout']['us_per_step'], 'policy f16 %.3f f32 %.3f' % (d['policy_rollout']['policy_dtype_f16']['us_per_step'], d['policy_rollout']['policy_dtype_f32']['us_per_step']), 'cfg5 %.3f' % d['config5_ppo_rollout']['us_per_step'])" || exit 1
done
