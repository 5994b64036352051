#!/bin/bash
# quick A/B: default library vs ZKMI_R29=0, bench only
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab; mkdir -p $O
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall > $O/a.json 2> $O/a.err
ZKMI_R29=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall > $O/b.json 2> $O/b.err
python - <<'PY'
import json
for t,n in (("a","r29"),("b","r32")):
    try:
        d=json.loads(open(f"gpurun_out/ab/{t}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d.get("latency_ms_single_proof"), {k:round(v,3) for k,v in d["stages_ms"].items() if k.startswith("accum") or k.startswith("reduce")}, d["submetrics"]["g1_msm_resident_tables_ms"])
    except Exception as e: print(n,"ERR",e, open(f"gpurun_out/ab/{t}.err").read()[-1500:])
PY
