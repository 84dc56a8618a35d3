set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6i
for cfg in "12,3,3,40,35 2 8192" "12,3,3,50,50 4 4096"; do
  set -- $cfg
  d=/tmp/cp_$2_$(echo $1 | tr ',' '_'); mkdir -p $d; chmod 700 $d
  echo "# shape $1 team $2" >> gpurun_out/r6i/phase_profile_team.txt
  MPCQP_CACHE_DIR=$d MPCQP_JIT_FLAGS="-DMPCQP_PROFILE -DMPCQP_TEAM=$2" python scripts/phase_profile.py $1 $3 2>&1 | grep -v Warn >> gpurun_out/r6i/phase_profile_team.txt
done
cat gpurun_out/r6i/phase_profile_team.txt
