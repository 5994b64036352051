#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or closed_form or valid_key or resident_tables or sharded or two_proofs or soak or synthetic" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall > $O/a.json 2> $O/a.err
ZKMI_R29_G2=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall > $O/b.json 2> $O/b.err
ZKMI_R29=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall > $O/c.json 2> $O/c.err
python - <<'PY'
import json
for t,n in (("a","r29 g1+g2"),("b","r29 g1 only"),("c","r32")):
    try:
        d=json.loads(open(f"gpurun_out/r2g/{t}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d.get("latency_ms_single_proof"), {k:round(v,3) for k,v in d["stages_ms"].items() if k.startswith("accum") or k.startswith("reduce")})
    except Exception as e: print(n,"ERR",e, open(f"gpurun_out/r2g/{t}.err").read()[-1500:])
PY
