#!/bin/bash
# r06 final: G2 layout microbench, then the whole GPU suite and the sort probe on the final library (chunk kernels on 256 blocks)
O=$GRAFT_REPO_ROOT/gpurun_out/r6final; mkdir -p $O
timeout 300 tools/bin/maddbench29_g2 > $O/maddbench29_g2.txt 2>&1; cat $O/maddbench29_g2.txt
bash tools/lab/r6_sort_probe.sh > $O/sort_probe.txt 2>&1; grep "rsort\|assign\|classify\|table msm" $O/sort_probe.txt
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 ) > $O/pytest_gpu.log; cat $O/pytest_gpu.log
for rep in 1 2; do
timeout 300 python bench.py --workload plonk --steps 32 --warmup 3 --no-napi-wall --no-cpu-baseline --no-other-configs > $O/plonk_two_$rep.json 2>/dev/null
timeout 300 python bench.py --steps 40 --warmup 3 --no-napi-wall --no-cpu-baseline --no-other-configs > $O/g16_$rep.json 2>/dev/null
done
python - <<'PY'
import json,glob
for p in sorted(glob.glob("gpurun_out/r6final/*.json")):
    try:
        j=json.loads(open(p).read().strip().splitlines()[-1]); print(p.split('/')[-1], j["value"], j.get("latency_ms_single_proof"))
    except Exception as e: print(p, "ERR", str(e)[:80])
PY
