#!/bin/bash
# round-5 GPU session g: smoke(), the default bench line (now with the per-env record), the N > 1 line rehearsed with two ranks on one GPU over gloo
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05g; mkdir -p $O
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke exit $?"; tail -n 1 $O/smoke.txt
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default exit $?"
timeout -k 10 600 python3 bench.py --gpus 2 --backend gloo --same-device --steps 250 --warmup 50 > $O/rehearsal_2rank_gloo_same_device.json 2> $O/rehearsal.err; echo "rehearsal exit $?"; tail -n 3 $O/rehearsal.err
timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 examples/train_ppo.py --envs 2048 --epochs 12 --randomise 0.15 --backend gloo --same-device > $O/ppo_2rank_gloo_randomise.log 2>&1; echo "ppo 2 ranks exit $?"; tail -n 3 $O/ppo_2rank_gloo_randomise.log
timeout -k 10 400 python3 examples/train_ppo.py --envs 4096 --epochs 12 --randomise 0.15 > $O/ppo_1rank_randomise.log 2>&1; echo "ppo 1 rank exit $?"; tail -n 3 $O/ppo_1rank_randomise.log
