cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deferred_far_updates" > gpurun_out/pde_tests.log 2>&1 || { tail -30 gpurun_out/pde_tests.log; exit 1; }
tail -2 gpurun_out/pde_tests.log
bash scripts/ab.sh 16384 8 10 2 "timing=0" "pred_defer_early=5120" "pred_defer_early=6144" "pred_defer_early=7168" "pred_defer_early=7680" "pred_defer_early=8192" "pred_defer_early=8704" "pred_defer_early=9216" "pred_defer_from=11264 pred_defer_early=7680" "pred_defer_from=11264 pred_defer_early=9216" "pred_defer_from=12288 pred_defer_early=8192" > gpurun_out/pde_ab16k.txt 2>&1; cat gpurun_out/pde_ab16k.txt
timeout -k 10 120 python scripts/stamp_steps.py 16384 8 pred_defer_early=7680 > gpurun_out/pde_stamps.txt 2>&1; cat gpurun_out/pde_stamps.txt
