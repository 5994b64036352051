#!/bin/bash
# r03 first GPU pass: GPU test suite on the generalised unsaturated-limb kernels (BN254 9x29, BLS12-381 14x28), benches, mul ceilings
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
timeout 300 python bench.py --curve bls12381 --steps 8 --warmup 2 --no-cpu-baseline --no-napi-wall > $O/bench_bls.json 2> $O/bench_bls.err; echo "bls rc=$?"; tail -3 $O/bench_bls.err
ZKMI_R29_BLS=0 timeout 300 python bench.py --curve bls12381 --steps 8 --warmup 2 --no-cpu-baseline --no-napi-wall > $O/bench_bls32.json 2>/dev/null
timeout 120 tools/bin/fieldbench29 > $O/fieldbench29.txt 2>&1; tail -30 $O/fieldbench29.txt
for f in bench bench_bls bench_bls32; do python - "$O/$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], d.get("stages_ms"), d.get("accum_kernel_ms"), d.get("int_alu",{}).get("mixed_additions"), d["submetrics"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
