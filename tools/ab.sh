#!/bin/bash
# A/B the env kernels across library builds: tools/ab.sh build/ab/x.so build/ab/y.so ...  (run on the GPU box)
# prints us per env step for the one-launch-per-step path, the fused rollout and the closed-loop policy rollout
mkdir -p gpurun_out
for v in "" "$@"; do
  for rep in 1 2; do
    DPENV_LIB=${v:+$PWD/$v} timeout -k 10 200 python bench.py --side-json /tmp/dpenv_side.json --no-cpu-baseline --steps 2500 --warmup 250 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); d.update(json.load(open('/tmp/dpenv_side.json'))); print('%-28s' % '${v:-default}', 'step %.3f' % (d['ms_per_step']*1e3), 'evt %.3f' % d['roofline']['avg_launch_us'], 'fused %.3f' % d['fused_rollout']['us_per_step'], 'policy %.3f' % d['policy_rollout']['us_per_step'], 'cfg5 %.3f' % d['config5_ppo_rollout']['us_per_step'])" || exit 1
  done
done
