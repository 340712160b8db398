#!/bin/bash
# round-5 GPU session i: longer PPO runs, nominal hull against +-30 % domain randomisation, both evaluated on spreads of hulls
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05i; mkdir -p $O
timeout -k 10 500 python3 -u examples/train_ppo.py --envs 4096 --epochs 150 --eval > $O/ppo_nominal_150.log 2>&1; echo "nominal exit $?"; tail -n 6 $O/ppo_nominal_150.log
timeout -k 10 500 python3 -u examples/train_ppo.py --envs 4096 --epochs 150 --randomise 0.30 --eval > $O/ppo_rand30_150.log 2>&1; echo "rand30 exit $?"; tail -n 6 $O/ppo_rand30_150.log
