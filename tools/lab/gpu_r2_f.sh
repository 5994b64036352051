#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall > $O/bench_r29.json 2> $O/bench_r29.err
ZKMI_R29=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall > $O/bench_r32.json 2> $O/bench_r32.err
python - <<'PY'
import json
for t in ("r29","r32"):
    try:
        d=json.loads(open(f"gpurun_out/r2f/bench_{t}.json").read().strip().splitlines()[-1])
        print(t, d["value"], d["ms_per_step"], d.get("latency_ms_single_proof"), d["stages_ms"], d["submetrics"]["g1_msm_resident_tables_ms"])
    except Exception as e: print(t,"ERR",e, open(f"gpurun_out/r2f/bench_{t}.err").read()[-1500:])
PY
