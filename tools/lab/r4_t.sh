#!/bin/bash
# k_poly_degree by ballot: tests, FFLONK / PLONK lines
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
timeout 100 python -m pytest tests/test_gpu_plonk.py -m gpu -x -q -k "poly_degree or fflonk_golden or synthetic_fflonk or plonk_stages" 2>&1 | tail -3
timeout 60 python bench.py --workload fflonk --log-n 18 --steps 5 --warmup 1 > $O/bench_fflonk_2p18.json 2>/dev/null
timeout 60 python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_plonk_2p20_b.json 2>/dev/null
for f in bench_fflonk_2p18 bench_plonk_2p20_b; do python - "$O/$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d["value"], d["unit"], d["ms_per_step"], "ms", d.get("box_calibration",{}).get("compact_code_mask"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
