cd $GRAFT_REPO_ROOT
bash scripts/ab.sh 4096 4 40 1 "timing=0 fuse_predict=0" "fuse_predict=0 tail_batch=4" "fuse_predict=0 tail_stream=1" "tail_stream=1" > gpurun_out/tb_ab4k.txt 2>&1; cat gpurun_out/tb_ab4k.txt
bash scripts/ab.sh 16384 8 10 1 "timing=0 fuse_predict=0" "fuse_predict=0 nb_small_from=8704 tail_rows=8704 tail_batch=4"  "fuse_predict=0 nb_small_from=6144 tail_rows=6144 tail_batch=4" "fuse_predict=0 tail_batch=4" > gpurun_out/tb_ab16k.txt 2>&1; cat gpurun_out/tb_ab16k.txt
bash scripts/ab.sh 8192 8 20 1 "timing=0 fuse_predict=0" "fuse_predict=0 tail_batch=4" "fuse_predict=0 nb_small_from=8192 tail_rows=8192 tail_batch=4" > gpurun_out/tb_ab8k.txt 2>&1; cat gpurun_out/tb_ab8k.txt
timeout -k 10 120 python scripts/stamp_steps.py 4096 4 fuse_predict=0 tail_batch=4 > gpurun_out/tb_stamps4k.txt 2>&1; cat gpurun_out/tb_stamps4k.txt
timeout -k 10 120 python scripts/stamp_steps.py 4096 4 tail_batch=4 > gpurun_out/tb_stamps4k_f.txt 2>&1; cat gpurun_out/tb_stamps4k_f.txt
