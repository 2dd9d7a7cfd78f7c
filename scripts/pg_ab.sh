cd $GRAFT_REPO_ROOT
bash scripts/ab.sh 16384 8 10 3 "gemm_polite=1" "gemm_polite=2" > gpurun_out/pg_ab16k.txt 2>&1; cat gpurun_out/pg_ab16k.txt
bash scripts/ab.sh 8192 8 20 3 "gemm_polite=1" "gemm_polite=2" > gpurun_out/pg_ab8k.txt 2>&1; cat gpurun_out/pg_ab8k.txt
bash scripts/ab.sh 4096 4 40 3 "gemm_polite=1" "gemm_polite=2" > gpurun_out/pg_ab4k.txt 2>&1; cat gpurun_out/pg_ab4k.txt
timeout -k 10 120 python scripts/stamp_steps.py 16384 8 gemm_polite=2 > gpurun_out/pg_stamps.txt 2>&1; sed -n 1,2p gpurun_out/pg_stamps.txt
