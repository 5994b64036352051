#!/bin/bash
# A/B on one box: physically contiguous allocations (ZKMI_CONTIG=1, default) vs plain hipMalloc; G2 kernel with the software-prefetched gather
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/ab9
run() {
  tag=$1; shift
  env "$@" timeout 600 python bench.py --steps 12 --warmup 3 --pipeline 1 --no-cpu-baseline --no-napi-wall > gpurun_out/ab9/$tag.json 2> gpurun_out/ab9/$tag.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/ab9/$tag.json").read().strip().splitlines()[-1])
    print("$tag serial", d["value"], "proofs/s", {k: round(v, 2) for k, v in d["stages_ms"].items()})
except Exception as e: print("$tag failed", e, open("gpurun_out/ab9/$tag.err").read()[-600:])
PY
}
run contig1 ZKMI_CONTIG=1
run contig0 ZKMI_CONTIG=0
run contig1b ZKMI_CONTIG=1
run contig0b ZKMI_CONTIG=0
for v in 1 0; do
  ZKMI_CONTIG=$v timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-napi-wall 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('groth16 pipelined contig=$v', d['value'], d['ms_per_step'])"
  ZKMI_CONTIG=$v timeout 600 python bench.py --workload plonk --log-n 20 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plonk contig=$v', d['value'], d['ms_per_step'])"
done
ZKMI_CONTIG=1 timeout 900 python bench.py --log-n 24 --steps 2 --warmup 1 --no-cpu-baseline --no-napi-wall 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2^24 contig=1', d['value'], d['ms_per_step'])"
