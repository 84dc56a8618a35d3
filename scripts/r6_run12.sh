cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6s
python scripts/family_sweep.py 3000 3020 huge 2 > gpurun_out/r6s/families_huge.txt 2>&1
tail -3 gpurun_out/r6s/families_huge.txt
python scripts/family_sweep.py 4000 4020 huge2 2 > gpurun_out/r6s/families_huge2.txt 2>&1
tail -3 gpurun_out/r6s/families_huge2.txt
