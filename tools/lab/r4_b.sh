#!/bin/bash
# r04: plain-C multiply-adds (no s_nop padding between inline-asm statements): the whole GPU suite, then the three bench lines
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 400 python bench.py --curve bls12381 --steps 8 --warmup 2 --no-cpu-baseline --no-napi-wall > $O/bench_bls.json 2>/dev/null
timeout 400 python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_plonk.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench","bench_bls","bench_plonk"):
    try:
        d=json.loads(open(f"gpurun_out/r4b/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"]); print("  ", json.dumps(d.get("submetrics"))); print("  ", json.dumps(d.get("stages_ms"))); print("  ", json.dumps(d.get("accum_kernel_ms")))
        if f=="bench": print("  ", json.dumps(d.get("wall_through_napi"))); print("  ", json.dumps(d.get("int_alu"))[:600]); print("  ", json.dumps(d.get("box_calibration"))[:500])
    except Exception as e: print(f, "ERR", e)
PY
