#!/bin/bash
# r04: the multi-rank code path of bench.py on ONE GPU (ZKMI_FORCE_DIST: a 1-rank RCCL communicator): sharded MSM over resident tables, one proof
# over all ranks at 2^20 and at 2^24 (BASELINE configs[2]) with the per-rank stage timeline
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4h; mkdir -p $O
( time ZKMI_FORCE_DIST=1 timeout 1500 python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-napi-wall > $O/bench_force_dist.json 2> $O/bench_force_dist.err ) 2>&1 | tail -3
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4h/bench_force_dist.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print(json.dumps(d["g1_msm_sharded"], indent=1)[:3000])
PY
tail -5 $O/bench_force_dist.err
