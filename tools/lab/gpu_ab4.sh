#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab4; mkdir -p $O
run() { tag=$1; shift; env "$@" > $O/$tag.json 2> $O/$tag.err; }
run bls X=1 timeout 300 python bench.py --curve bls12381 --steps 8 --warmup 2 --no-cpu-baseline --no-napi-wall
run plonk_r29 X=1 timeout 300 python bench.py --workload plonk --log-n 20 --steps 8 --warmup 3 --no-cpu-baseline
run plonk_r32 ZKMI_R29=0 timeout 300 python bench.py --workload plonk --log-n 20 --steps 8 --warmup 3 --no-cpu-baseline
run plonk_r29b X=1 timeout 300 python bench.py --workload plonk --log-n 20 --steps 8 --warmup 3 --no-cpu-baseline
python - <<'PY'
import json
for t in ("bls","plonk_r29","plonk_r32","plonk_r29b"):
    try:
        d=json.loads(open(f"gpurun_out/ab4/{t}.json").read().strip().splitlines()[-1])
        print(t, d["value"], d["ms_per_step"])
    except Exception as e: print(t,"ERR",e, open(f"gpurun_out/ab4/{t}.err").read()[-800:])
PY
