cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile_pipeline_tail or golden or full_size" > gpurun_out/tb_tests.log 2>&1 || { tail -30 gpurun_out/tb_tests.log; exit 1; }
tail -2 gpurun_out/tb_tests.log
bash scripts/ab.sh 4096 4 40 2 "timing=0" "tail_batch=4" "tail_batch=2" "tail_batch=8" > gpurun_out/tb_ab4k.txt 2>&1; cat gpurun_out/tb_ab4k.txt
bash scripts/ab.sh 16384 8 10 2 "timing=0" "nb_small_from=8704 tail_rows=8704 tail_batch=4" "nb_small_from=8704 tail_rows=8704 tail_batch=8" "nb_small_from=6144 tail_rows=6144 tail_batch=4" "tail_batch=4" > gpurun_out/tb_ab16k.txt 2>&1; cat gpurun_out/tb_ab16k.txt
bash scripts/ab.sh 8192 8 20 2 "timing=0" "tail_batch=4" "nb_small_from=8192 tail_rows=8192 tail_batch=4" "nb_small_from=6144 tail_rows=6144 tail_batch=4" > gpurun_out/tb_ab8k.txt 2>&1; cat gpurun_out/tb_ab8k.txt
timeout -k 10 120 python scripts/stamp_steps.py 4096 4 tail_batch=4 > gpurun_out/tb_stamps4k_f.txt 2>&1; cat gpurun_out/tb_stamps4k_f.txt
timeout -k 10 120 python scripts/stamp_steps.py 16384 8 nb_small_from=8704 tail_rows=8704 tail_batch=4 > gpurun_out/tb_stamps16k_f.txt 2>&1; cat gpurun_out/tb_stamps16k_f.txt
timeout -k 10 120 python scripts/stamp_steps.py 16384 8 > gpurun_out/tb_stamps16k_0.txt 2>&1; cat gpurun_out/tb_stamps16k_0.txt
