#!/bin/bash
# round-5 GPU session f: soaks of the final library (hand-over fences, randomised instantiations), batch sweep, config-4 record, driver-form bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05f; mkdir -p $O
timeout -k 10 900 python3 -u tools/soak_fused.py 6000 > $O/soak_fused.txt 2>&1; echo "soak_fused exit $?"
timeout -k 10 900 python3 -u tools/soak_fused.py 6000 0.15 > $O/soak_fused_randomised.txt 2>&1; echo "soak_fused randomised exit $?"
timeout -k 10 900 python3 -u tools/soak_determinism.py 1500 > $O/soak_determinism.txt 2>&1; echo "soak_determinism exit $?"
timeout -k 10 900 python3 -u tools/soak_determinism.py 1500 0.15 > $O/soak_determinism_randomised.txt 2>&1; echo "soak_determinism randomised exit $?"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err; echo "bench driver form exit $?"
timeout -k 10 400 python3 bench.py --config4 1 --no-cpu-baseline --no-fused > $O/bench_config4.json 2> $O/bench_config4.err; echo "bench config4 exit $?"
SIZES="4096 16384 32768 65536 131072 262144 1048576" timeout -k 10 600 bash tools/batch_sweep.sh > $O/batch_sweep.txt 2>&1; echo "sweep exit $?"
for f in $O/soak_*.txt; do tail -n 2 $f; done
