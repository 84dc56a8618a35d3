#!/bin/bash
# One gpurun call: rocprofv3 kernel stats + separate --pmc passes (counters only with --kernel-trace, see the task notes)
# of the FULL bench command, i.e. the headline kernel AND every kernel a config.secondary record quotes (small-problem
# kernels, the nZ~ = 106 specialisation, the MultipleShooting kernel, both MHE variants); scripts/pmc_summary_all.py turns
# the output into profiles/<tag>/ (one pmc_summary per kernel, traffic.json).
# Usage (GPU box, repo root):  scripts/profile_all.sh r5a [tests]
set -u
TAG=${1:-rX}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py > $OUT/bench_line.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats_run.log 2>&1
for set in "sq1:SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "sq2:SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES" \
           "sq3:SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${set%%:*}; ctr=${set#*:}
  rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $OUT/pmc_$name -o pmc_$name -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_$name.log 2>&1
done
cd $REPO
python scripts/pmc_summary_all.py $OUT $OUT/summary > $OUT/summary.log 2>&1
if [ "${2:-}" = "tests" ]; then python -m pytest tests -m gpu -q 2>&1 | tail -5 > $OUT/pytest_gpu_tail.log; fi
