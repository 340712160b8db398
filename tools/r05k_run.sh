#!/bin/bash
# round-5 GPU session k: the inflow thrust loss - its tests and the per-env / randomisation tests beside it (the general per-env kernels changed),
# which bound ends the episodes under the new preset, and the default bench (thrust_loss_preset leg of the config-2 workload record).
# A step that times out ends the session (no GPU step after a hung one).
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05k; mkdir -p $O
step() { local name=$1 lim=$2; shift 2; timeout -k 10 $lim "$@"; local rc=$?; echo "$name exit $rc" >&2; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name timed out: session ends here" >&2; exit $rc; fi; return $rc; }
step tests_loss 900 python3 -m pytest tests/test_gpu_thrust_loss.py tests/test_gpu_vessel_env.py tests/test_gpu_round5.py -q -m gpu -x > $O/tests_loss.txt 2>&1 || { tail -n 30 $O/tests_loss.txt; exit 1; }
tail -n 2 $O/tests_loss.txt
for preset in no_loss thrust_loss; do step box_$preset 200 python3 tools/compare_cybersea_box.py $preset > $O/cybersea_box_$preset.txt 2>&1; head -n 4 $O/cybersea_box_$preset.txt; done
step tests_all 1100 python3 -m pytest tests -q -m gpu -x --deselect tests/test_gpu_thrust_loss.py --deselect tests/test_gpu_vessel_env.py --deselect tests/test_gpu_round5.py > $O/tests_all.txt 2>&1; tail -n 3 $O/tests_all.txt
