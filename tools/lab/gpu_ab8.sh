#!/bin/bash
# after the merged-reduction / squaring change of the 29-bit mixed additions: parity, then the bench (serial + pipelined)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ab8
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "not 24" 2>&1 | tail -4
for pipe in 1 2; do
  timeout 600 python bench.py --steps 16 --warmup 3 --pipeline $pipe --no-cpu-baseline --no-napi-wall > gpurun_out/ab8/bench.p$pipe.json 2> gpurun_out/ab8/bench.p$pipe.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/ab8/bench.p$pipe.json").read().strip().splitlines()[-1])
    print("pipeline=$pipe", d["value"], "proofs/s", d["ms_per_step"], "ms", {k: round(v, 2) for k, v in d["stages_ms"].items()}, d["roofline"]["int_alu"].get("valu_issue") if "int_alu" in d["roofline"] else d.get("int_alu"))
except Exception as e: print("failed", e, open("gpurun_out/ab8/bench.p$pipe.err").read()[-600:])
PY
done
ZKMI_R29=0 timeout 600 python bench.py --steps 12 --warmup 3 --pipeline 2 --no-cpu-baseline --no-napi-wall 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('32-bit limbs everywhere, pipeline=2', d['value'], d['ms_per_step'])"
