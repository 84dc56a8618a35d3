#!/bin/bash
# rocprofv3 of the MHE step kernel (bench.py --config C5): kernel stats + separate --pmc passes.
# Usage (on the GPU box, from the repo root):  scripts/profile_mhe.sh r2_mhe
set -u
TAG=${1:-mhe}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py --config C5 > $OUT/bench_line.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $REPO/bench.py --config C5 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats_run.log 2>&1
for set in "sq1:SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_FMA_F64 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "sq2:SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" \
           "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${set%%:*}; ctr=${set#*:}
  rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $OUT/pmc_$name -o pmc_$name -- python $REPO/bench.py --config C5 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_$name.log 2>&1
done
cd $REPO
python scripts/pmc_mhe_summary.py $OUT $OUT/summary
