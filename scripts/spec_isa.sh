#!/bin/bash
# Developer tool: device assembly + resource line of one step-kernel specialisation.
# Usage: scripts/spec_isa.sh "4,4,16,30,10,1,141u,1" [out.s] [extra flags]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
DIMS=${1:-4,4,16,30,10,1,141u,1}; OUT=${2:-/tmp/isa/spec.s}; shift; shift
mkdir -p $(dirname $OUT)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -w --cuda-device-only -S -gline-tables-only \
  -I$ROOT/modelpredictivecontrol.jl_amd/csrc -DMPCQP_SPEC_DIMS=$DIMS "$@" $ROOT/modelpredictivecontrol.jl_amd/csrc/mpcqp_spec.hip -o $OUT
grep -E "^\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):" $OUT | tr -s ' ' | tr '\n' ' '; echo
grep -c "^\s*v_\|^\s*ds_\|^\s*s_" $OUT
