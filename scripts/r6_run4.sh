set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6d
SH="12,3,3,40,35 12,3,3,50,50 12,4,4,36,36 12,2,2,70,70"
python scripts/shape_sweep.py 2048 c3 $SH > gpurun_out/r6d/sweep_team_auto.txt 2>&1
grep nZ gpurun_out/r6d/sweep_team_auto.txt | cut -c1-120
python scripts/shape_sweep.py 1024 all 12,2,2,70,70 12,3,3,40,35 >> gpurun_out/r6d/sweep_team_auto.txt 2>&1
tail -3 gpurun_out/r6d/sweep_team_auto.txt | cut -c1-120
for cfg in "12,3,3,40,35 2" "12,3,3,50,50 4"; do
  set -- $cfg
  d=/tmp/cp_$2_$(echo $1 | tr ',' '_'); mkdir -p $d; chmod 700 $d
  echo "# shape $1 team $2" >> gpurun_out/r6d/phase_profile_team.txt
  MPCQP_CACHE_DIR=$d MPCQP_JIT_FLAGS="-DMPCQP_PROFILE -DMPCQP_TEAM=$2" python scripts/phase_profile.py $1 2048 2>&1 | grep -v Warn >> gpurun_out/r6d/phase_profile_team.txt
done
cat gpurun_out/r6d/phase_profile_team.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "two_rows or beyond_one_row or shapes_and or custom_linear or dense_weight or hessian_is" > gpurun_out/r6d/pytest_sel.log 2>&1
tail -5 gpurun_out/r6d/pytest_sel.log
