cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6t
MPCQP_JIT_SELFTEST_TOL=inf python scripts/ab_jit.py 12,3,3,50,50 4096 ping0 ping2000 > gpurun_out/r6t/ping.txt 2>&1
grep -a "kernel ms" gpurun_out/r6t/ping.txt | cut -c1-150
