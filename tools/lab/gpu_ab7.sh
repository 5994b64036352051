#!/bin/bash
# A/B on one box: G2 accumulation kernel modes (0 free scheduling / 1 fenced / 2 fenced + prefetch), G1 fenced / occupancy experiment
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ab7
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "golden or groth16_synthetic or valid_key or resident_tables or msm_golden" 2>&1 | tail -4
run() {
  tag=$1; shift
  for pipe in 1 2; do
    env "$@" timeout 600 python bench.py --steps 12 --warmup 3 --pipeline $pipe --no-cpu-baseline --no-napi-wall > gpurun_out/ab7/$tag.p$pipe.json 2> gpurun_out/ab7/$tag.p$pipe.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/ab7/$tag.p$pipe.json").read().strip().splitlines()[-1])
    st = d.get("stages_ms") or d.get("config", {}).get("stages_ms") or {}
    print("$tag pipeline=$pipe", d["value"], "proofs/s", d["ms_per_step"], "ms", {k: round(v, 2) for k, v in st.items()} if isinstance(st, dict) else st)
except Exception as e: print("$tag failed", e, open("gpurun_out/ab7/$tag.p$pipe.err").read()[-600:])
PY
  done
}
run base   ZKMI_G2_MODE=0 ZKMI_G1_MODE=0
run g2m1   ZKMI_G2_MODE=1 ZKMI_G1_MODE=0
run g2m2   ZKMI_G2_MODE=2 ZKMI_G1_MODE=0
run g2m2g1 ZKMI_G2_MODE=2 ZKMI_G1_MODE=1
run g2m2g2 ZKMI_G2_MODE=2 ZKMI_G1_MODE=2
run base2  ZKMI_G2_MODE=0 ZKMI_G1_MODE=0
