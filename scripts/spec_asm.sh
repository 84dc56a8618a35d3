#!/bin/bash
# device assembly of one specialisation: scripts/spec_asm.sh out.s "4,4,16,30,10,1,0x8Cu,1" [extra flags]
out=$1; dims=$2; shift 2
CS=/root/repo/modelpredictivecontrol.jl_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -w -I$CS -DMPCQP_SPEC_DIMS=$dims -mllvm -pragma-unroll-threshold=1048576 "$@" $CS/mpcqp_spec.hip -o $out
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):" $out | head -4 | tr '\n' ' '; echo
