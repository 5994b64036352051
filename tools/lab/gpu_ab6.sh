#!/bin/bash
# A/B on one box: PLONK with the digit sorts of a round's MSMs on the auxiliary stream (default) vs one stream
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ab6
timeout 900 python -m pytest tests/test_gpu_plonk.py -q -x -m gpu 2>&1 | tail -4
for v in 1 0 1 0; do
  ZKMI_MULTI_OVERLAP=$v timeout 600 python bench.py --workload plonk --log-n 20 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/ab6/plonk_ov$v.json 2> gpurun_out/ab6/plonk_ov$v.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/ab6/plonk_ov$v.json").read().strip().splitlines()[-1]); print("overlap=$v", d["value"], "proofs/s", d["ms_per_step"], "ms")
except Exception as e: print("overlap=$v failed", e, open("gpurun_out/ab6/plonk_ov$v.err").read()[-800:])
PY
done
