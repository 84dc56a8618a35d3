#!/bin/bash
# rebuild everything with absolute paths: HIP library (+ profiling build), oracle C port, CPU wave emulator
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
MPCQP_BUILD_PROF=1 python -c "import __graft_entry__ as g; g.build()"
make -s -C "$ROOT/tests/emu"
