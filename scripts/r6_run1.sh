set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
L=modelpredictivecontrol.jl_amd/lib/ab
python scripts/ab_lib.py $L/libmpcqp_base.so $L/libmpcqp_v2.so $L/libmpcqp_v4.so $L/libmpcqp_v8.so $L/libmpcqp_base.so > gpurun_out/r6a/ab_vreg.txt 2>&1
tail -6 gpurun_out/r6a/ab_vreg.txt
python scripts/r6_stage_diag.py > gpurun_out/r6a/stage_diag.txt 2>&1
tail -12 gpurun_out/r6a/stage_diag.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "two_rows or hessian_is or step_matches or full_size or shapes_and or slack_row" > gpurun_out/r6a/pytest_sel.log 2>&1
tail -5 gpurun_out/r6a/pytest_sel.log
python bench.py > gpurun_out/r6a/bench.log 2>&1
tail -2 gpurun_out/r6a/bench.log | cut -c1-600
