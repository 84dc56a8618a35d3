#!/bin/bash
# Developer tool: build lib/ab/libmpcqp_<name>.so from the current sources with extra -D flags, recompiling only the
# LinMPC kernel translation unit (the other three objects are cached in /tmp/objs).  A/B several builds in one GPU
# call with scripts/ab_lib.py.   Usage: scripts/build_variant.sh <name> [-DFOO=1 ...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/modelpredictivecontrol.jl_amd/csrc
NAME=$1; shift
mkdir -p /tmp/objs $ROOT/modelpredictivecontrol.jl_amd/lib/ab
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -w"
for f in mpcqp_host mhe_kernels small_kernels mhe_host ms_kernels; do
  if [ ! -f /tmp/objs/$f.o ] || [ $CS/$f.hip -nt /tmp/objs/$f.o ] || [ -n "$(find $CS -name '*.h' -newer /tmp/objs/$f.o | grep -v mpcqp_bodies | head -1)" ]; then
    /opt/rocm/bin/hipcc $FL -c $CS/$f.hip -o /tmp/objs/$f.o &
  fi
done
/opt/rocm/bin/hipcc $FL "$@" -c $CS/mpcqp_kernels.hip -o /tmp/objs/mpcqp_kernels_$NAME.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/modelpredictivecontrol.jl_amd/lib/ab/libmpcqp_$NAME.so \
  /tmp/objs/mpcqp_kernels_$NAME.o /tmp/objs/mpcqp_host.o /tmp/objs/mhe_kernels.o /tmp/objs/small_kernels.o /tmp/objs/mhe_host.o /tmp/objs/ms_kernels.o -ldl
echo built $NAME
