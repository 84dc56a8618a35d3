set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b
SH="12,3,3,40,35 12,3,3,50,50 12,4,4,30,20 12,4,4,36,36 12,2,2,70,70"
python scripts/shape_sweep.py 2048 c3 $SH > gpurun_out/r6b/sweep_team_auto.txt 2>&1
tail -7 gpurun_out/r6b/sweep_team_auto.txt
mkdir -p /tmp/c1 && chmod 700 /tmp/c1
MPCQP_CACHE_DIR=/tmp/c1 MPCQP_JIT_FLAGS=-DMPCQP_TEAM=1 python scripts/shape_sweep.py 2048 c3 $SH > gpurun_out/r6b/sweep_team_1.txt 2>&1
tail -7 gpurun_out/r6b/sweep_team_1.txt
mkdir -p /tmp/c2 && chmod 700 /tmp/c2
MPCQP_CACHE_DIR=/tmp/c2 MPCQP_JIT_FLAGS=-DMPCQP_TEAM=2 python scripts/shape_sweep.py 2048 c3 $SH > gpurun_out/r6b/sweep_team_2.txt 2>&1
tail -7 gpurun_out/r6b/sweep_team_2.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "two_rows or beyond_one_row or shapes_and or custom_linear or dense_weight" > gpurun_out/r6b/pytest_sel.log 2>&1
tail -5 gpurun_out/r6b/pytest_sel.log
