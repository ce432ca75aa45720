#!/bin/bash
# final sanity of the tree: the whole GPU suite, smoke, the default bench line
set -o pipefail
mkdir -p gpurun_out/r02n
python -m pytest tests -q -m gpu > gpurun_out/r02n/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r02n/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02n/smoke.log 2>&1
echo "smoke rc=$?"; tail -2 gpurun_out/r02n/smoke.log
python bench.py > gpurun_out/r02n/bench.json 2> gpurun_out/r02n/bench.err
echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02n/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["build_id"], d["roofline"]["frac"], d["roofline"]["pmc_source"][:50], d["train_step"]["backward_ms"], d["train_step"]["ms_per_step"])
PY
