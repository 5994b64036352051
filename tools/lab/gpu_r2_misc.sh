#!/bin/bash
# round-2 misc evidence: mixed ("witness-like", SURVEY 8d second distribution) witness line, the RCCL path with one rank, bench.py --gpus 1 self-launch
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02c; mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 3 --witness mixed --no-cpu-baseline --no-napi-wall > $O/bench_mixed_witness.json 2> $O/mixed.err
ZKMI_FORCE_DIST=1 timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-napi-wall > $O/bench_force_dist.json 2> $O/dist.err
for f in bench_mixed_witness bench_force_dist; do python - "$O/$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], d.get("g1_msm_sharded"), d.get("one_proof_all_ranks") or d.get("groth16_sharded"))
except Exception as e: print(sys.argv[1], "ERR", e, open(sys.argv[1].replace('.json','.err').replace('bench_mixed_witness','mixed').replace('bench_force_dist','dist')).read()[-800:])
PY
done
tail -3 $O/dist.err
