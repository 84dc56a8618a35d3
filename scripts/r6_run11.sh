set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6j
python scripts/shape_sweep.py 1024 c3 12,3,3,40,35 12,3,3,50,50 12,4,4,36,36 12,2,2,70,70 12,4,4,30,20 > gpurun_out/r6j/sweep.txt 2>&1
python scripts/shape_sweep.py 512 all 12,2,2,70,70 12,3,3,40,35 >> gpurun_out/r6j/sweep.txt 2>&1
grep -a "nZ\|rror\|failed" gpurun_out/r6j/sweep.txt | cut -c1-150
python scripts/ab_jit.py 12,3,3,40,35 8192 t2 > gpurun_out/r6j/ab_team_bench_sizes.txt 2>&1
python scripts/ab_jit.py 12,3,3,50,50 4096 t2 t4 >> gpurun_out/r6j/ab_team_bench_sizes.txt 2>&1
grep -a "kernel ms" gpurun_out/r6j/ab_team_bench_sizes.txt | cut -c1-150
timeout 1500 python -m pytest tests -m gpu -x -q -k "two_rows or beyond_one_row or shapes_and or team_kernel or dense_weight" 2>&1 | tail -3
