#!/bin/bash
# second pass of the r02 collection: bench lines with roofline.traffic filled from the published PMC file, rebuilt maddbench29, FFLONK A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02b; mkdir -p $O
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 3 --pipeline 1 --no-cpu-baseline --no-napi-wall > $O/bench_serial.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 3 --b-zero-every 3 --no-cpu-baseline --no-napi-wall > $O/bench_sparse_b.json 2>/dev/null
tools/bin/fieldbench29 > $O/fieldbench29.txt 2>&1; tools/bin/maddbench29 >> $O/fieldbench29.txt 2>&1
for v in 1 0 1 0; do
  ZKMI_MULTI_OVERLAP=$v timeout 300 python bench.py --workload fflonk --log-n 18 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fflonk overlap=$v', d['value'], d['ms_per_step'])"
done
for f in bench bench_serial bench_sparse_b; do python - "$O/$f.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d["value"], d["ms_per_step"], "traffic", d["roofline"]["traffic"], "frac", d["roofline"]["frac"])
PY
done
tail -5 $O/fieldbench29.txt
