#!/bin/bash
# Developer tool: lib/ab/libmpcqp_msprof.so = the library with ms_kernels.hip compiled -DMPCQP_MS_PROFILE (cycles per phase of
# the MultipleShooting step written over the first 8 entries of the X^0 output); scripts/ms_profile.py prints them.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/modelpredictivecontrol.jl_amd/csrc
OBJ=$ROOT/modelpredictivecontrol.jl_amd/lib/obj
mkdir -p $ROOT/modelpredictivecontrol.jl_amd/lib/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -w -DMPCQP_MS_PROFILE "$@" -c $CS/ms_kernels.hip -o /tmp/ms_kernels_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/modelpredictivecontrol.jl_amd/lib/ab/libmpcqp_msprof.so \
  /tmp/ms_kernels_prof.o $OBJ/mpcqp_kernels.hip.o $OBJ/mpcqp_host.hip.o $OBJ/mhe_kernels.hip.o $OBJ/small_kernels.hip.o $OBJ/mhe_host.hip.o -ldl
echo built
