cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6u
python scripts/ab_jit.py 12,3,3,40,35 8192 dpp0 dpp1 > gpurun_out/r6u/ab_solve_dpp.txt 2>&1
python scripts/ab_jit.py 12,3,3,50,50 4096 dpp0 dpp1 >> gpurun_out/r6u/ab_solve_dpp.txt 2>&1
grep -a "kernel ms\|rror\|disagree" gpurun_out/r6u/ab_solve_dpp.txt | cut -c1-170
python scripts/shape_sweep.py 1024 c3 12,3,3,40,35 12,3,3,50,50 12,4,4,36,36 12,2,2,70,70 12,4,4,30,20 12,4,4,30,25 > gpurun_out/r6u/sweep.txt 2>&1
python scripts/shape_sweep.py 512 all 12,2,2,70,70 12,3,3,40,35 >> gpurun_out/r6u/sweep.txt 2>&1
grep -a "nZ\|rror\|failed" gpurun_out/r6u/sweep.txt | cut -c1-150
