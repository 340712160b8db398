#!/bin/bash
# round-5 GPU session t: PPO (60 epochs x 4 096 envs x 400 steps) on each plant preset, each actor evaluated on BOTH presets (box test, run_RL_policy)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05t; mkdir -p $O
for pz in no_loss thrust_loss; do
  timeout -k 10 500 python3 examples/train_ppo.py --envs 4096 --epochs 60 --preset $pz --eval --eval-presets no_loss,thrust_loss > $O/ppo_$pz.log 2>&1; rc=$?; echo "ppo $pz exit $rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  grep "^eval" $O/ppo_$pz.log | cut -c1-260
done
