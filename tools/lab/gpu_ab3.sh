#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab3; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or ntt or synthetic_vs_oracle or two_proofs or sharded_equals" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall > $O/$tag.json 2> $O/$tag.err; }
run batch X=1
run nobatch ZKMI_NTT_BATCH=0
run batch2 X=1
python - <<'PY'
import json
for t in ("batch","nobatch","batch2"):
    try:
        d=json.loads(open(f"gpurun_out/ab3/{t}.json").read().strip().splitlines()[-1])
        print(t, d["value"], d["ms_per_step"], d.get("latency_ms_single_proof"), {k:round(v,3) for k,v in d["stages_ms"].items() if not k.startswith("sort")})
    except Exception as e: print(t,"ERR",e, open(f"gpurun_out/ab3/{t}.err").read()[-1500:])
PY
