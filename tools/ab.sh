#!/bin/bash
# A/B the step kernel across library variants: prints ms_per_step per variant (graph replay, 2500 steps)
mkdir -p gpurun_out
for v in "" build/variants/b64.so build/variants/b128.so build/variants/noslp.so build/variants/b64_noslp.so "$@"; do
  for rep in 1 2; do
    DPENV_LIB=${v:+$PWD/$v} timeout -k 10 120 python bench.py --no-cpu-baseline --steps 2500 --warmup 250 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${v:-default}', 'us/step %.3f'%(d['ms_per_step']*1e3), 'evt %.3f'%d['roofline']['avg_launch_us'])"
  done
done
DPENV_LIB= timeout -k 10 120 python bench.py --no-cpu-baseline --steps 2500 --warmup 250 --hold-plant 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hold_plant', 'us/step %.3f'%(d['ms_per_step']*1e3))"
