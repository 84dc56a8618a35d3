set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6e
L=modelpredictivecontrol.jl_amd/lib/ab
python scripts/ab_lib.py $L/libmpcqp_p0.so $L/libmpcqp_p1.so $L/libmpcqp_p2.so $L/libmpcqp_p3.so $L/libmpcqp_p4.so $L/libmpcqp_p7.so $L/libmpcqp_p0.so > gpurun_out/r6e/ab_prio.txt 2>&1
grep kernel gpurun_out/r6e/ab_prio.txt | cut -c1-120
MPCQP_JIT_SELFTEST_TOL=1e30 python scripts/ab_jit.py 12,3,3,50,50 2048 base a1 a256 a512 a1024 a24 a4 a32 > gpurun_out/r6e/ablate_nz151.txt 2>&1
grep kernel gpurun_out/r6e/ablate_nz151.txt | cut -c1-140
MS_SEED=0 python scripts/ms_c3_check.py C3 8192 > gpurun_out/r6e/ms_c3_seed0.txt 2>&1
tail -8 gpurun_out/r6e/ms_c3_seed0.txt
