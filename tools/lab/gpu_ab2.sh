#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab2; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall > $O/$tag.json 2> $O/$tag.err; }
run cap128 ZKMI_AUX_RC_BLOCKS=128
run cap64 ZKMI_AUX_RC_BLOCKS=64
run cap256 ZKMI_AUX_RC_BLOCKS=256
run nocap ZKMI_AUX_RC_BLOCKS=0
python - <<'PY'
import json
for t in ("cap128","cap64","cap256","nocap"):
    try:
        d=json.loads(open(f"gpurun_out/ab2/{t}.json").read().strip().splitlines()[-1])
        print(t, d["value"], d["ms_per_step"], d.get("latency_ms_single_proof"), {k:round(v,3) for k,v in d["stages_ms"].items() if k.startswith("accum") or k.startswith("reduce")})
    except Exception as e: print(t,"ERR",e, open(f"gpurun_out/ab2/{t}.err").read()[-1500:])
PY
