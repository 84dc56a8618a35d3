#!/bin/bash
# PMC passes of the small-problem kernel on C2 (B = 65536): instruction mix and wait fractions.  Usage (GPU box): scripts/pmc_small.sh TAG
TAG=${1:-c2pmc}; REPO=$(cd "$(dirname "$0")/.." && pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "sq1:SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "sq2:SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" \
           "sq3:SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_VALU_INT32"; do
  name=${set%%:*}; ctr=${set#*:}
  rocprofv3 --kernel-trace --output-format csv --pmc $ctr -d $OUT/pmc_$name -o pmc_$name -- python $REPO/bench.py --config C2 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $OUT/pmc_$name.log 2>&1
done
python - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = 0
for f in glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_step_small" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(tot.items()): print(f"{k:28s} {v:.4g}")
PY
