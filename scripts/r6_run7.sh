set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6f
MPCQP_JIT_SELFTEST_TOL=inf python scripts/ab_jit.py 12,3,3,50,50 2048 m0 m1 f0 f0a512 f0a1024 f0a24 > gpurun_out/r6f/ab_nz151.txt 2>&1
grep -a "kernel ms\|rror" gpurun_out/r6f/ab_nz151.txt | cut -c1-150
