#!/bin/bash
# Developer tool: build on-demand specialisations of one shape with several flag sets (here, cross-compiled), each into
# its own cache directory under lib/ab/jit/<name>; scripts/ab_jit.py then times them on the GPU.
# Usage: scripts/ab_jit.sh "3,3,15,40,35,1,141,1" name1:"flags" name2:"flags" ...
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/modelpredictivecontrol.jl_amd/csrc
DIMS=$1; shift
IFS=, read nu ny nxh Hp Hc neps rows dnb <<< "$DIMS"
REV=$(grep -oP '#define MPCQP_KERNEL_REV \K\d+' $CSRC/mpcqp_types.h)
CID=$(python -c "import sys; sys.path.insert(0,'$ROOT'); import __graft_entry__ as g; print('%08x' % g._compiler_id())")
WAVES=""; [ $((nu*Hc+neps)) -gt 64 ] && WAVES="-DMPCQP_STEP_WAVES=1"
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}; [ "$flags" = "$v" ] && flags=""
  dir=$ROOT/modelpredictivecontrol.jl_amd/lib/ab/jit/$name; mkdir -p $dir; chmod 700 $dir; rm -f $dir/*
  so=$dir/spec_r${REV}_c${CID}_${nu}_${ny}_${nxh}_${Hp}_${Hc}_${neps}_$(printf %x $rows)_${dnb}.so
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -w -I$CSRC \
      -DMPCQP_SPEC_DIMS=$nu,$ny,$nxh,$Hp,$Hc,$neps,${rows}u,$dnb $WAVES -mllvm -pragma-unroll-threshold=1048576 $flags $CSRC/mpcqp_spec.hip -o $so && echo "built $name: $flags" ) &
done
wait
