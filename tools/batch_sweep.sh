#!/bin/bash
# Batch-size sweep of the bench legs (one MI355X): bash tools/batch_sweep.sh > gpurun_out/r04_batch_sweep.txt
# us per env step of ALL N envs; frac = 177 B x N / step time / 8 TB/s.  The named workload is the 65 536 row.
cd $GRAFT_REPO_ROOT 2>/dev/null || true
echo "Batch-size sweep with the ${ROUND:-round-6} kernels (python bench.py --envs N --steps 250 --warmup 50 --no-cpu-baseline)"
for N in ${SIZES:-4096 16384 32768 65536 131072 262144 1048576}; do
  python3 bench.py --side-json /tmp/dpenv_side.json --envs $N --steps 250 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json, sys
r = json.loads(sys.stdin.read()); r.update(json.load(open('/tmp/dpenv_side.json'))); n = $N      # (round 6: the side records are a file)
p, c5 = r['policy_rollout'], r['config5_ppo_rollout']
print('envs %8d  step %7.3f us (%.3g env-steps/s, frac %.3f)  fused %7.3f us  closed f16 %7.3f us [%s]  exact actor %7.3f us [%s]  all exact %7.3f us [%s]  cfg5 f16 %.3g /s' % (
    n, r['ms_per_step'] * 1e3, r['value'], r['roofline']['frac'], r['fused_rollout']['us_per_step'],
    p['policy_dtype_f16']['us_per_step'], p['policy_dtype_f16']['launch_form'].replace(' envs per workgroup', ''),
    p['policy_dtype_f32_actor']['us_per_step'], p['policy_dtype_f32_actor']['launch_form'].replace(' envs per workgroup', ''),
    p['policy_dtype_f32']['us_per_step'], p['policy_dtype_f32']['launch_form'].replace(' envs per workgroup', ''), c5['policy_dtype_f16']['env_steps_per_s']))"
done
