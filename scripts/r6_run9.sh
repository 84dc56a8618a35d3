set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6h
python scripts/ab_jit.py 12,3,3,40,35 8192 t1 t2 > gpurun_out/r6h/ab_team_bench_sizes.txt 2>&1
python scripts/ab_jit.py 12,3,3,50,50 4096 t1 t2 t4m0 t4m1 >> gpurun_out/r6h/ab_team_bench_sizes.txt 2>&1
grep -a "kernel ms" gpurun_out/r6h/ab_team_bench_sizes.txt | cut -c1-150
