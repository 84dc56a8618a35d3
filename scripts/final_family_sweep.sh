cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6z
python scripts/family_sweep.py 0 80 any 2 2>&1 | grep -a ' ok \|FAIL\|worst' > gpurun_out/r6z/families_any.txt
python scripts/family_sweep.py 1000 1040 small 3 2>&1 | grep -a ' ok \|FAIL\|worst' > gpurun_out/r6z/families_small.txt
python scripts/family_sweep.py 2000 2040 large 2 2>&1 | grep -a ' ok \|FAIL\|worst' > gpurun_out/r6z/families_large.txt
python scripts/family_sweep.py 0 20 any 2 MultipleShooting 2>&1 | grep -a ' ok \|FAIL\|worst' > gpurun_out/r6z/families_ms.txt
python scripts/mhe_family_sweep.py 0 40 2>&1 | tail -45 > gpurun_out/r6z/families_mhe.txt
for f in gpurun_out/r6z/*.txt; do echo "$f: $(tail -1 $f)"; done
