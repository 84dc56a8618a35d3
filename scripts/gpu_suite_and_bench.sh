#!/bin/bash
# One gpurun call: the whole GPU suite, the smoke test of __graft_entry__ and the bench line.  Usage: scripts/gpu_suite_and_bench.sh <tag>
cd ${GRAFT_REPO_ROOT:-.}
TAG=${1:-rX}
mkdir -p gpurun_out/$TAG
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/$TAG/pytest_gpu.log 2>&1
tail -4 gpurun_out/$TAG/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/$TAG/smoke.log 2>&1
tail -2 gpurun_out/$TAG/smoke.log
python bench.py > gpurun_out/$TAG/bench.log 2>&1
tail -1 gpurun_out/$TAG/bench.log > gpurun_out/$TAG/bench_line.json
cut -c1-300 gpurun_out/$TAG/bench_line.json
