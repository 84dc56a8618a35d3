cd $GRAFT_REPO_ROOT
C="12,4,4,18,8"
rm -f modelpredictivecontrol.jl_amd/lib/spec_cache/*_4_4_16_18_8_*
python bench.py --config $C --batch 65536 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('3waves(lds 12/CU):', d['ms_per_step'], d['config']['ipm_mean_iters'])"
MPCQP_LDS_PAD=6000 python bench.py --config $C --batch 65536 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('2waves(lds pad):', d['ms_per_step'], d['config']['ipm_mean_iters'])"
