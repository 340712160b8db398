#!/bin/bash
# round-5 GPU session d: tests of the rebuilt library, which bound ends the episodes under each plant preset, PPO with and without domain randomisation
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05d; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_vessel_env.py tests/test_gpu_round5.py tests/test_gpu_c_abi.py tests/test_gpu_policy_v3.py -q -m gpu > $O/tests.txt 2>&1; echo "tests exit $?"; tail -2 $O/tests.txt
timeout -k 10 300 python3 tools/bound_shares.py 16384 1200 > $O/bound_shares.txt 2> $O/bound_shares.err; echo "bound_shares exit $?"
timeout -k 10 500 python3 examples/train_ppo.py --envs 4096 --epochs 40 --eval --save $O/actor_nominal.npz > $O/ppo_nominal.log 2>&1; echo "ppo nominal exit $?"; tail -4 $O/ppo_nominal.log
timeout -k 10 500 python3 examples/train_ppo.py --envs 4096 --epochs 40 --randomise 0.15 --eval --save $O/actor_rand15.npz > $O/ppo_rand15.log 2>&1; echo "ppo rand exit $?"; tail -4 $O/ppo_rand15.log
timeout -k 10 300 python3 bench.py --classes 3 --no-cpu-baseline > $O/bench_classes.json 2> $O/bench_classes.err; echo "bench exit $?"
