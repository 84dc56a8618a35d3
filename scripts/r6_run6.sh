set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6e
MPCQP_JIT_SELFTEST_TOL=inf python scripts/ab_jit.py 12,3,3,50,50 2048 base a1 a256 a512 a1024 a24 a4 a32 > gpurun_out/r6e/ablate_nz151.txt 2>&1
grep -a "kernel\|rror" gpurun_out/r6e/ablate_nz151.txt | cut -c1-140
