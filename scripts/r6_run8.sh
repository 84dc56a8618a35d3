set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6g
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6g/pytest_gpu.log 2>&1
tail -15 gpurun_out/r6g/pytest_gpu.log
python bench.py > gpurun_out/r6g/bench.log 2>&1
tail -1 gpurun_out/r6g/bench.log > gpurun_out/r6g/bench_line.json
python - <<'PY'
import json
r = json.load(open('gpurun_out/r6g/bench_line.json'))
print(r["value"], r["ms_per_step"], r["roofline"]["frac"])
for s in r["config"]["secondary"]:
    print(s.get("workload"), s.get("batch"), s.get("value"), s.get("kernel_ms"), (s.get("roofline") or {}).get("frac"))
PY
